// Microbenchmark: row-wise window flushes into HBM images on gfx950, by accumulation type.
// Shape of k_splat's flush: each workgroup adds a WW x WH window (one wave per row, lanes along the row) at a random
// position into one of NIMG images of H x W pixels.  Variants: float atomic add, u32 atomic add, u64 atomic add,
// plain store (no accumulation; upper bound), u32 "add only when non-zero" with a given fill fraction.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/global_atomic_bench tools/global_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
constexpr int NT = 256, H = 260, W = 346, NIMG = 40, WW = 48, WH = 48;

struct Job { int img, ox, oy; };

template <typename T, int MODE>   // MODE 0: atomic add, 1: plain store
__global__ __launch_bounds__(NT) void k_flush(const Job* __restrict__ jobs, T* __restrict__ imgs, unsigned fill_mod) {
    const Job j = jobs[blockIdx.x];
    T* __restrict__ dst = imgs + (size_t)j.img * H * W + (size_t)j.oy * W + j.ox;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int row = wv; row < WH; row += NT / 64) {
        for (int col = lane; col < WW; col += 64) {
            const unsigned hsh = (unsigned)(row * 131 + col * 7 + blockIdx.x * 2654435761u);
            if (fill_mod && (hsh % 100u) >= fill_mod) continue;          // skip "zero" pixels
            const T v = (T)(1 + (hsh & 7));
            if (MODE == 0) atomicAdd(dst + row * W + col, v);
            else dst[row * W + col] = v;
        }
    }
}

template <typename T, int MODE> void run(const char* name, const Job* d_jobs, int njobs, unsigned fill_mod) {
    T* imgs; CHK(hipMalloc(&imgs, sizeof(T) * (size_t)NIMG * H * W));
    CHK(hipMemset(imgs, 0, sizeof(T) * (size_t)NIMG * H * W));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flush<T, MODE>), dim3(njobs), dim3(NT), 0, 0, d_jobs, imgs, fill_mod);
    CHK(hipDeviceSynchronize());
    const int reps = 10;
    CHK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_flush<T, MODE>), dim3(njobs), dim3(NT), 0, 0, d_jobs, imgs, fill_mod);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double frac = fill_mod ? fill_mod / 100.0 : 1.0;
    const double px = (double)njobs * WW * WH * frac;
    printf("%-28s fill %3.0f%%  %8.1f us   %7.1f G px/s   %7.2f TB/s of added bytes\n", name, frac * 100, ms * 1e3, px / ms / 1e6,
           px * sizeof(T) / (ms * 1e-3) / 1e12);
    CHK(hipFree(imgs));
}

int main() {
    const int njobs = 12000;          // ~ segments x R of the bench workload
    std::vector<Job> jobs(njobs);
    srand(7);
    for (auto& j : jobs) { j.img = rand() % NIMG; j.ox = rand() % (W - WW); j.oy = rand() % (H - WH); }
    Job* d; CHK(hipMalloc(&d, sizeof(Job) * njobs));
    CHK(hipMemcpy(d, jobs.data(), sizeof(Job) * njobs, hipMemcpyHostToDevice));
    for (unsigned fill : {0u, 60u, 30u}) {
        run<float, 0>("f32 atomic add", d, njobs, fill);
        run<unsigned, 0>("u32 atomic add", d, njobs, fill);
        run<unsigned long long, 0>("u64 atomic add", d, njobs, fill);
        run<double, 0>("f64 atomic add", d, njobs, fill);
        run<float, 1>("f32 plain store", d, njobs, fill);
        run<unsigned long long, 1>("u64 plain store", d, njobs, fill);
    }
    return 0;
}
