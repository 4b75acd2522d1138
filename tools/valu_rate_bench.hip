// Microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops used by the event kernels on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
constexpr int ITERS = 4096, UNROLL = 8;
// 8 independent chains per thread so that dependent-issue latency does not dominate; 8 waves per SIMD resident
#define BODY(NAME, TYPE, INIT, OP) \
__global__ __launch_bounds__(256) void NAME(TYPE* out, double seed) { \
  TYPE v[UNROLL]; for (int k = 0; k < UNROLL; ++k) v[k] = INIT; \
  for (int i = 0; i < ITERS; ++i) { _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) { OP; } } \
  TYPE s = 0; for (int k = 0; k < UNROLL; ++k) s += v[k]; if (s == (TYPE)12345.678) out[0] = s; }
BODY(k_add_f32, float, (float)(seed + k + threadIdx.x), v[k] = v[k] + 1.0001f)
BODY(k_fma_f32, float, (float)(seed + k + threadIdx.x), v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f))
BODY(k_exp_f32, float, (float)(seed * 1e-3 + k), v[k] = __builtin_amdgcn_exp2f(v[k]) * 0.5f)
BODY(k_rcp_f32, float, (float)(seed + k + 2), v[k] = __builtin_amdgcn_rcpf(v[k]) + 1.5f)
BODY(k_cvtu_f32, float, (float)(seed + k), v[k] = (float)((unsigned)(v[k] + 3.0f) & 1023u))
BODY(k_add_f64, double, (seed + k + threadIdx.x), v[k] = v[k] + 1.0001)
BODY(k_fma_f64, double, (seed + k + threadIdx.x), v[k] = __builtin_fma(v[k], 1.0001, 0.5))
BODY(k_rndne_f64, double, (seed + k + 0.3), v[k] = __builtin_rint(v[k]) + 0.7)
BODY(k_cvt_i32_f64, double, (seed + k + 0.3), v[k] = (double)((int)v[k]) + 0.7)      /* cvt_i32_f64 + cvt_f64_i32 + add */
BODY(k_cvt_f32_f64, double, (seed + k + 0.3), v[k] = (double)((float)v[k]) + 0.7)    /* cvt_f32_f64 + cvt_f64_f32 + add */
BODY(k_mul_lo_u32, unsigned, (unsigned)(seed + k + threadIdx.x), v[k] = v[k] * 2654435761u + 1u)
BODY(k_mad24, unsigned, (unsigned)(seed + k + threadIdx.x), v[k] = __umul24(v[k] & 0xfffu, 1237u) + 1u)
template <typename K, typename T> void run(const char* name, K kern, T* d, int nops) {
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  int nblk = 256 * 8;   // 8 blocks of 4 waves per CU -> 8 waves per SIMD
  kern<<<nblk, 256>>>(d, 1.0); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a)); for (int r = 0; r < 3; ++r) kern<<<nblk, 256>>>(d, 1.0); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  double wave_instr_per_simd = (double)nblk * 4 / 1024.0 * ITERS * UNROLL * nops;
  printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (%d ops per loop step, 2.4 GHz assumed)\n", name, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd, nops);
}
int main() {
  void* d; CHK(hipMalloc(&d, 1024));
  run("v_add_f32", k_add_f32, (float*)d, 1); run("v_fma_f32", k_fma_f32, (float*)d, 1);
  run("v_exp_f32(+mul)", k_exp_f32, (float*)d, 2); run("v_rcp_f32(+add)", k_rcp_f32, (float*)d, 2);
  run("cvt f32<->u32(4)", k_cvtu_f32, (float*)d, 4);
  run("v_add_f64", k_add_f64, (double*)d, 1); run("v_fma_f64", k_fma_f64, (double*)d, 1);
  run("v_rndne_f64(+add)", k_rndne_f64, (double*)d, 2); run("cvt i32<->f64(+add)", k_cvt_i32_f64, (double*)d, 3);
  run("cvt f32<->f64(+add)", k_cvt_f32_f64, (double*)d, 3);
  run("v_mul_lo_u32(+add)", k_mul_lo_u32, (unsigned*)d, 2); run("mad24(+and)", k_mad24, (unsigned*)d, 2);
  return 0;
}
