// What would a hipGraph buy a small-window evaluation?  Four dependent kernels, each with a ~1 KiB by-value argument block that changes
// every iteration (theta, capacities), as in a 2-DoF evaluation of a 3*10^4-event window (k_splat, k_stats_stream, k_imgrad, k_gather):
//   A) four stream launches + hipStreamSynchronize           (what the library does)
//   B) four hipGraphExecKernelNodeSetParams + hipGraphLaunch + hipStreamSynchronize
//   C) hipGraphLaunch alone with unchanged arguments          (the floor of B)
// hipcc -O2 --offload-arch=gfx950 tools/scratch/graph_launch_test.hip -o gpurun_out/graph_launch_test && gpurun_out/graph_launch_test [work]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Arg { double v[128]; };
__global__ void k(Arg a, float* p, int n, int work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = p[i] + (float)a.v[i & 127];
    for (int j = 0; j < work; ++j) s = s * 1.0001f + 0.5f;
    p[i] = s;
}
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main(int argc, char** argv) {
    const int work = argc > 1 ? atoi(argv[1]) : 2000;      // ~ device time per kernel
    const int n = 1 << 16, iters = 2000;
    float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float)));
    hipStream_t st; CK(hipStreamCreate(&st));
    Arg a{}; int nn = n, ww = work;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
    std::vector<double> tA, tB, tC, tD;
    for (int it = 0; it < iters + 100; ++it) {
        a.v[it & 127] = it;
        const auto t0 = now();
        for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, st, a, p, nn, ww);
        CK(hipStreamSynchronize(st));
        if (it >= 100) tA.push_back(us(now() - t0));
    }
    // one kernel alone: the device time of a link of the chain (launch + sync included)
    for (int it = 0; it < iters; ++it) {
        const auto t0 = now();
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, st, a, p, nn, ww);
        CK(hipStreamSynchronize(st));
        tD.push_back(us(now() - t0));
    }
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    hipGraphNode_t nodes[4];
    void* args[4] = {&a, &p, &nn, &ww};
    hipKernelNodeParams kp{};
    kp.func = (void*)k; kp.gridDim = dim3(n / 256); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
    for (int q = 0; q < 4; ++q) CK(hipGraphAddKernelNode(&nodes[q], g, q ? &nodes[q - 1] : nullptr, q ? 1 : 0, &kp));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < iters + 100; ++it) {
        a.v[it & 127] = it;
        const auto t0 = now();
        for (int q = 0; q < 4; ++q) CK(hipGraphExecKernelNodeSetParams(ge, nodes[q], &kp));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        if (it >= 100) tB.push_back(us(now() - t0));
    }
    for (int it = 0; it < iters + 100; ++it) {
        const auto t0 = now();
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        if (it >= 100) tC.push_back(us(now() - t0));
    }
    printf("work %d: one kernel + sync %.1f us | A four launches + sync %.1f us | B 4 SetParams + graph launch + sync %.1f us | C graph launch + sync %.1f us\n",
           work, med(tD), med(tA), med(tB), med(tC));
    return 0;
}
