// How large may a kernel argument block be on this stack?  (theta of a 16x16 grid is 4096 bytes.)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { double v[N]; };
template <int N> __global__ void k(Big<N> a, double* out) { double s = 0; for (int i = threadIdx.x; i < N; i += 64) s += a.v[i]; atomicAdd(out, s); }
template <int N> void run() {
    double* d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
    Big<N> a; for (int i = 0; i < N; ++i) a.v[i] = 1.0;
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, a, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    double h = -1; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("N=%d (%d bytes): launch %s, sync %s, sum %g\n", N, N * 8, hipGetErrorString(e), hipGetErrorString(e2), h);
    hipFree(d);
}
int main() { run<1024>(); run<2048>(); run<4096>(); run<8000>(); return 0; }
