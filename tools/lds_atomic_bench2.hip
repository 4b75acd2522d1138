// Microbenchmark 2: ds_add_u32 throughput with k_splat's own shape - 3x3 taps at a row stride of ww words in a ww x wh window,
// 18 KB of LDS per workgroup (8 workgroups per CU), random tap centres - against window size and occupancy.
// hipcc -O3 --offload-arch=gfx950 -o tools/lds_atomic_bench2 tools/lds_atomic_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
constexpr int NT = 256, ITERS = 64;

__global__ __launch_bounds__(NT) void k(const int* __restrict__ idx, unsigned* out, int words, int ww, int interleave, int copies) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += NT) lds[i] = 0;
    __syncthreads();
    const int* my = idx + (size_t)blockIdx.x * NT * ITERS;
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        const int a = my[it * NT + threadIdx.x];
        unsigned* p = lds + a + ((threadIdx.x >> 6) % copies) * (words / copies);      // private copy of the window per wave group
        unsigned* p1 = p + ww; unsigned* p2 = p1 + ww;
        if (interleave) {          // some arithmetic between the atomics, like the tap products
            float f = __uint_as_float(0x3f800000u | (a & 0xffff));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f = __builtin_amdgcn_exp2f(f * 0.37f) + 1.0f;
                atomicAdd((t < 3 ? p : t < 6 ? p1 : p2) + t % 3, (unsigned)f);
            }
            acc += (unsigned)f;
        } else {
            atomicAdd(p, 1u); atomicAdd(p + 1, 1u); atomicAdd(p + 2, 1u);
            atomicAdd(p1, 1u); atomicAdd(p1 + 1, 1u); atomicAdd(p1 + 2, 1u);
            atomicAdd(p2, 1u); atomicAdd(p2 + 1, 1u); atomicAdd(p2 + 2, 1u);
        }
    }
    __syncthreads();
    unsigned s = acc;
    for (int i = threadIdx.x; i < words; i += NT) s += lds[i];
    if (s == 12345u) out[blockIdx.x] = s;
}

int main() {
    const int nblk = 256 * 16;
    const size_t n = (size_t)nblk * NT * ITERS;
    int* d; unsigned* o;
    CHK(hipMalloc(&d, n * 4)); CHK(hipMalloc(&o, nblk * 4));
    std::vector<int> h(n);
    struct Case { const char* name; int ww, wh, lds_words, sites; int interleave; int copies; int pattern = 0; };   // pattern 1 / 2: see below
    const Case cases[] = {
        {"46x46 window, 18 KB LDS (8 WG/CU), uniform centres", 46, 46, 4608, 0, 0, 1},
        {"46x46 window, 36 KB LDS (4 WG/CU), uniform centres", 46, 46, 9216, 0, 0, 1},
        {"46x46 window, 72 KB LDS (2 WG/CU), uniform centres", 46, 46, 18432, 0, 0, 1},
        {"64x64 window, 18 KB LDS, uniform centres", 64, 64, 4608, 0, 0, 1},
        {"90x90 window, 36 KB LDS, uniform centres", 90, 90, 9216, 0, 0, 1},
        {"46x46 window, 18 KB LDS, centres on 300 sites (edges)", 46, 46, 4608, 300, 0, 1},
        {"46x46 window, 18 KB LDS, uniform centres, math between the atomics", 46, 46, 4608, 0, 1, 1},
        {"46x46 window, 300 sites, 2 copies by wave parity (36 KB LDS)", 46, 46, 9216, 300, 0, 2},
        {"46x46 window, 300 sites, 4 copies, one per wave (72 KB LDS)", 46, 46, 18432, 300, 0, 4},
        {"46x46 window, 300 sites, 36 KB LDS, 1 copy (occupancy control)", 46, 46, 9216, 300, 0, 1},
        {"46x46 window, 100 sites, 18 KB LDS", 46, 46, 4608, 100, 0, 1},
        {"46x46 window, 100 sites, 2 copies (36 KB LDS)", 46, 46, 9216, 100, 0, 2},
        {"46x46 window, 1000 sites, 18 KB LDS", 46, 46, 4608, 1000, 0, 1},
        // round 3: would a row pitch of 64 words with 32 DISTINCT columns per half-wave (bank = column) pay?  (pattern 1; 2 = the control:
        // the same pitch, random columns; 3 = distinct columns with a +-1 jitter on 30 % of the lanes, as rounding produces)
        {"64-word pitch, 36 rows, 32 distinct columns per half-wave", 64, 36, 4608, 0, 0, 1, 1},
        {"64-word pitch, 36 rows, random columns of 32 (control)", 64, 36, 4608, 0, 0, 1, 2},
        {"64-word pitch, 36 rows, distinct columns, +-1 jitter on 30 %", 64, 36, 4608, 0, 0, 1, 3},
        {"64-word pitch, distinct columns, math between the atomics", 64, 36, 4608, 0, 1, 1, 1},
        {"64-word pitch, random columns, math between the atomics", 64, 36, 4608, 0, 1, 1, 2},
    };
    for (const Case& c : cases) {
        srand(3);
        std::vector<int> sites(c.sites > 0 ? c.sites : 1);
        for (int& s : sites) s = (rand() % (c.wh - 2)) * c.ww + rand() % (c.ww - 2);
        for (size_t i = 0; i < n; ++i)
            h[i] = c.sites > 0 ? sites[rand() % c.sites] : (rand() % (c.wh - 2)) * c.ww + rand() % (c.ww - 2);
        if (c.pattern) {
            int perm[32];
            for (size_t i = 0; i < n; i += 32) {            // 32 consecutive lanes = one half-wave of one trip
                for (int k = 0; k < 32; ++k) perm[k] = k;
                for (int k = 31; k > 0; --k) { const int j = rand() % (k + 1); const int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
                for (int k = 0; k < 32 && i + k < n; ++k) {
                    int x = 8 + (c.pattern == 2 ? rand() % 32 : perm[k]);              // columns 8..39 of the 64-word rows
                    if (c.pattern == 3 && rand() % 10 < 3) x += (rand() & 1) ? 1 : -1;
                    h[i + k] = (rand() % (c.wh - 2)) * c.ww + x;
                }
            }
        }
        CHK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        const size_t lds_bytes = (size_t)c.lds_words * 4;
        hipLaunchKernelGGL(k, dim3(nblk), dim3(NT), lds_bytes, 0, d, o, c.lds_words, c.ww, c.interleave, c.copies); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(nblk), dim3(NT), lds_bytes, 0, d, o, c.lds_words, c.ww, c.interleave, c.copies);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        const double ops = (double)nblk * NT * ITERS * 9;
        printf("%-72s %7.3f ms  %6.2f lane-atomics/clk/CU\n", c.name, ms, ops / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
