// Microbenchmark: LDS atomic-add throughput on gfx950 by data type and address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
constexpr int NT=256, WIN=8192, ITERS=64, NTAP=9;

template<typename T> __device__ inline void add(T* p, T v){ atomicAdd(p, v); }

// mode: idx table gives per-(thread,iter) base address
template<typename T>
__global__ __launch_bounds__(NT) void k(const int* __restrict__ idx, T* out, int use_atomic){
  __shared__ T lds[WIN+256];
  for(int i=threadIdx.x;i<WIN+256;i+=NT) lds[i]=0;
  __syncthreads();
  const int* my = idx + (size_t)blockIdx.x*NT*ITERS;
  for(int it=0; it<ITERS; ++it){
    int a = my[it*NT+threadIdx.x];
    #pragma unroll
    for(int t=0;t<NTAP;++t){
      int o = a + (t%3) + (t/3)*64;
      if(use_atomic) add<T>(&lds[o], (T)1); else lds[o] += (T)1;
    }
  }
  __syncthreads();
  T s=0; for(int i=threadIdx.x;i<WIN+256;i+=NT) s+=lds[i];
  if(s==(T)12345) out[blockIdx.x]=s;
}
template<typename T> void run(const char* name, const std::vector<int>& h, int nblk, int use_atomic){
  int* d; T* o; CHK(hipMalloc(&d,h.size()*4)); CHK(hipMalloc(&o,nblk*sizeof(T)));
  CHK(hipMemcpy(d,h.data(),h.size()*4,hipMemcpyHostToDevice));
  hipEvent_t a,b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  k<T><<<nblk,NT>>>(d,o,use_atomic); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a)); for(int r=0;r<5;++r) k<T><<<nblk,NT>>>(d,o,use_atomic); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms,a,b)); ms/=5;
  double ops=(double)nblk*NT*ITERS*NTAP;
  printf("%-34s %8.3f ms  %8.1f G lane-ops/s  %6.2f lane-ops/clk/CU\n",name,ms,ops/ms/1e6,ops/(ms*1e-3)/256/2.4e9);
  CHK(hipFree(d)); CHK(hipFree(o));
}
int main(){
  int nblk=256*8;
  size_t n=(size_t)nblk*NT*ITERS;
  std::vector<int> distinct(n), rnd(n), rnd64(n), same(n), pairs(n), clustered(n);
  srand(1);
  for(size_t i=0;i<n;++i){
    int t=i%NT;
    distinct[i]=(t*3)%(WIN-200);            // lanes 3 apart: taps dx=0..2 never collide within a tap instruction
    rnd[i]=rand()%(WIN-200);                // random over 8K-float window
    rnd64[i]=(rand()%48)*3 + (rand()%8)*64*3;  // ~400 distinct sites
    same[i]=100;                            // all lanes same address
    pairs[i]=((t/2)*3)%(WIN-200);           // 2 lanes per address
    clustered[i]=(rand()%40)*1 + (rand()%3)*64; // ~120 sites on 3 rows (edge-like)
  }
  const char* names[]={"distinct","random(8K)","random(~400 sites)","all-same","2-lanes/addr","clustered(~120 sites)"};
  std::vector<int>* pats[]={&distinct,&rnd,&rnd64,&same,&pairs,&clustered};
  for(int p=0;p<6;++p){
    char nm[96];
    snprintf(nm,96,"f32 atomic  %s",names[p]); run<float>(nm,*pats[p],nblk,1);
    snprintf(nm,96,"u32 atomic  %s",names[p]); run<unsigned>(nm,*pats[p],nblk,1);
    snprintf(nm,96,"u64 atomic  %s",names[p]); run<unsigned long long>(nm,*pats[p],nblk,1);
    snprintf(nm,96,"f64 atomic  %s",names[p]); run<double>(nm,*pats[p],nblk,1);
    snprintf(nm,96,"f32 plain rmw %s",names[p]); run<float>(nm,*pats[p],nblk,0);
  }
  return 0;
}
