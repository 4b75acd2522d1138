// eincm_edges.hip.h — SURVEY row f-4: the step that produces `edges` for eincm_set_windows, and the tiled objectives.
//
//   inverse exponential distance transform   src/utils/img_utils.py:229-233 (scipy EDT) and :236-410 (RTEF_IEDT, Meijster)
//   Gaussian smoothing                        src/utils/img_utils.py:210-220 (cv.GaussianBlur on a float64 image)
//   tiled ("adaptive") objectives             src/eincm/objectives/contrast_objectives.py:42-87,
//                                             src/eincm/objectives/correlation_objectives.py:28-130, src/utils/img_utils.py:105-120
//
// The distance transform is integer work and exact: phase 1 walks columns (thread per column, coalesced), phase 2 takes,
// per pixel, the minimum over the row of g(y,x')^2 + (x-x')^2, searching outwards from x and stopping once k^2 can no
// longer beat the best candidate (so the work per pixel is the distance itself, not the row length).  The squared
// distance image is what both reference flavours compute (the transform is unique); the float stage is three fp64 ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eincm_kernels.hip.h"

namespace eincm {

constexpr uint32_t EDT_INF = 32768u;      // "no edge pixel in this column"; sensor sides are < 32768 (int16 coordinates)
constexpr int BLUR_MAX_TAPS = 257;

// grid (ceil(W/NT), n).  g (n,H,W): vertical distance to the nearest edge pixel of the same column, EDT_INF if none.
__global__ __launch_bounds__(NT) void k_edt_cols(int H, int W, const uint8_t* __restrict__ edge, uint32_t* __restrict__ g,
                                                  uint32_t* __restrict__ n_edge)
{
    const int x = blockIdx.x * NT + threadIdx.x;
    const size_t base = (size_t)blockIdx.y * H * W + x;
    uint32_t d = EDT_INF, cnt = 0;
    const int Hc = (x < W) ? H : 0;                    // lanes beyond the row stay for the wave reduction below
    for (int y = 0; y < Hc; ++y) {
        const bool e = edge[base + (size_t)y * W] != 0;
        cnt += e;
        d = e ? 0u : min(d + 1u, EDT_INF);
        g[base + (size_t)y * W] = d;
    }
    d = EDT_INF;
    for (int y = Hc - 1; y >= 0; --y) {
        const uint32_t down = g[base + (size_t)y * W];
        d = (down == 0u) ? 0u : min(d + 1u, EDT_INF);
        g[base + (size_t)y * W] = min(down, d);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off, 64);   // one atomic per wave, not per thread
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&n_edge[blockIdx.y], cnt);
}

// grid (ceil(W/NT), H, n).  sq (n,H,W) squared Euclidean distance to the nearest edge pixel; maxsq (n).
__global__ __launch_bounds__(NT) void k_edt_rows(int H, int W, const uint32_t* __restrict__ g, int32_t* __restrict__ sq,
                                                  uint32_t* __restrict__ maxsq)
{
    const int x = blockIdx.x * NT + threadIdx.x;
    const uint32_t* __restrict__ row = g + ((size_t)blockIdx.z * H + blockIdx.y) * W;
    uint32_t best = 0u;
    if (x < W) {
        const uint32_t g0 = row[x];
        best = g0 * g0;
        for (uint32_t k = 1; k * k < best; ++k) {
            const int xl = x - (int)k, xr = x + (int)k;
            if (xl < 0 && xr >= W) break;
            const uint32_t gl = (xl >= 0) ? row[xl] : EDT_INF, gr = (xr < W) ? row[xr] : EDT_INF;
            const uint32_t gm = min(gl, gr);
            best = min(best, gm * gm + k * k);
        }
        sq[((size_t)blockIdx.z * H + blockIdx.y) * W + x] = (int32_t)best;
    }
    // one atomic per workgroup, and only when it can raise the maximum (same-address atomics serialise at L2)
    __shared__ uint32_t wmax[NWAVE];
    uint32_t m = best;
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NWAVE; ++i) m = max(m, wmax[i]);
        if (m > __builtin_nontemporal_load(&maxsq[blockIdx.z])) atomicMax(&maxsq[blockIdx.z], m);
    }
}

__device__ __forceinline__ double edt_formulation(double d, int formulation, double alpha, double d_sat) {
    switch (formulation) {
        case 1: return d;                                   // 'linear'
        case 2: return fmin(d, d_sat);                      // 'linear-bound'      img_utils.py:378
        case 3: return log(d + 1.0);                        // 'logarithmic'       img_utils.py:380
        default: return 1.0 - exp(-d / alpha);              // 'exponential'       img_utils.py:232,382
    }
}

// out = 1 - (f - min f) / (max f - min f + eps); min f = f(0) = 0 because every image has an edge pixel (img_utils.py:233,388-391,408)
__global__ __launch_bounds__(NT) void k_edt_finish(int64_t npix, const int32_t* __restrict__ sq, const uint32_t* __restrict__ maxsq,
                                                    int formulation, double alpha, double d_sat, double* __restrict__ out)
{
    const int img = blockIdx.y;
    const double fmax = edt_formulation(sqrt((double)maxsq[img]), formulation, alpha, d_sat);
    const double den = fmax - 0.0 + EPSN;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < npix; i += (int64_t)gridDim.x * NT) {
        const double f = edt_formulation(sqrt((double)sq[(size_t)img * npix + i]), formulation, alpha, d_sat);
        out[(size_t)img * npix + i] = 1.0 - (f - 0.0) / den;
    }
}

// BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
__device__ __forceinline__ int reflect101(int i, int n) {
    i = (i < 0) ? -i : i;
    return (i >= n) ? 2 * n - 2 - i : i;
}

// one pass of the separable filter along x (axis = 0) or y (axis = 1); grid (ceil(W/NT), H, n); kern has 2*radius+1 taps.
__global__ __launch_bounds__(NT) void k_blur(int H, int W, int axis, int radius, const double* __restrict__ kern,
                                              const double* __restrict__ src, double* __restrict__ dst)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const double* __restrict__ S = src + (size_t)blockIdx.z * H * W;
    double s = kern[radius] * S[(size_t)y * W + x];
    for (int j = 1; j <= radius; ++j) {
        double a, b;
        if (axis == 0) { a = S[(size_t)y * W + reflect101(x - j, W)]; b = S[(size_t)y * W + reflect101(x + j, W)]; }
        else           { a = S[(size_t)reflect101(y - j, H) * W + x]; b = S[(size_t)reflect101(y + j, H) * W + x]; }
        s += kern[radius + j] * (a + b);
    }
    dst[((size_t)blockIdx.z * H + y) * W + x] = s;
}

// ------------------------------------------------------------------------------------------------
// Tiled objectives on the images of the last evaluation.  grid (n_tiles, R, B); out (B,R,n_tiles,3):
//   [0] mean(gx^2+gy^2) of the RAW IWE tile, convolved on its own (zero padded at the tile border)
//   [1] population variance of the raw IWE tile      [2] MSE(edge tile, min-max-normalised IWE tile)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_tiled(Geom g, int th, int tw, int ntx, const float* __restrict__ iwe,
                                               const float* __restrict__ edges, const StatPart* __restrict__ parts,
                                               double* __restrict__ out)
{
    __shared__ double scratch[NWAVE];
    const int tile = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    const int y0 = (tile / ntx) * th, x0 = (tile % ntx) * tw;
    const float* __restrict__ I = iwe + ((size_t)b * g.R + r) * g.H * g.W;
    const float* __restrict__ E = edges + ((size_t)b * g.R + r) * g.H * g.W;
    const ImgScal sc = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
    const auto at = [&](int ly, int lx) -> double {
        return (ly >= 0 && ly < th && lx >= 0 && lx < tw) ? (double)I[(size_t)(y0 + ly) * g.W + x0 + lx] : 0.0;
    };
    const int n = th * tw;
    double s = 0.0;
    for (int p = threadIdx.x; p < n; p += NT) s += at(p / tw, p % tw);
    s = block_sum(s, scratch);
    __shared__ double mean_sh;
    if (threadIdx.x == 0) mean_sh = s / n;
    __syncthreads();
    const double mean = mean_sh;
    double sG2 = 0.0, sVar = 0.0, sMse = 0.0;
    for (int p = threadIdx.x; p < n; p += NT) {
        const int ly = p / tw, lx = p % tw;
        double gx, gy;
        scharr_at(at, ly, lx, gx, gy);
        sG2 += gx * gx + gy * gy;
        const double v = at(ly, lx);
        sVar += (v - mean) * (v - mean);
        const double d = (double)E[(size_t)(y0 + ly) * g.W + x0 + lx] - (v - sc.m) / sc.D;
        sMse += d * d;
    }
    sG2 = block_sum(sG2, scratch); sVar = block_sum(sVar, scratch); sMse = block_sum(sMse, scratch);
    if (threadIdx.x == 0) {
        double* o = out + (((size_t)b * g.R + r) * gridDim.x + tile) * 3;
        o[0] = sG2 / n; o[1] = sVar / n; o[2] = sMse / n;
    }
}

// Untiled siblings on (edge E, normalised IWE n).  grid (nblk, R, B); out (B,R,nblk,3): every workgroup's partial of sum (E-n)^2,
// sum E*n, sum |Scharr(E+n)|^2 (whole image, zero padded), stored in its own slot; the host adds the slots in index order (no float
// atomics anywhere: an fp64 atomicAdd is a compare-and-swap loop here and its sum depends on the arrival order).
__global__ __launch_bounds__(NT) void k_pair_objectives(Geom g, const float* __restrict__ iwe, const float* __restrict__ edges,
                                                         const StatPart* __restrict__ parts, double* __restrict__ out)
{
    __shared__ double scratch[NWAVE];
    const int r = blockIdx.y, b = blockIdx.z;
    const float* __restrict__ I = iwe + ((size_t)b * g.R + r) * g.H * g.W;
    const float* __restrict__ E = edges + ((size_t)b * g.R + r) * g.H * g.W;
    const ImgScal sc = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
    const auto nrm = [&](int y, int x) -> double { return ((double)I[(size_t)y * g.W + x] - sc.m) / sc.D; };
    const auto joint = [&](int y, int x) -> double {
        return (y >= 0 && y < g.H && x >= 0 && x < g.W) ? (double)E[(size_t)y * g.W + x] + nrm(y, x) : 0.0;
    };
    double sse = 0.0, had = 0.0, jc = 0.0;
    const int npix = g.H * g.W;
    for (int p = blockIdx.x * NT + threadIdx.x; p < npix; p += gridDim.x * NT) {
        const int y = p / g.W, x = p % g.W;
        const double e = (double)E[p], v = nrm(y, x);
        sse += (e - v) * (e - v);
        had += e * v;
        double gx, gy;
        scharr_at(joint, y, x, gx, gy);
        jc += gx * gx + gy * gy;
    }
    sse = block_sum(sse, scratch); had = block_sum(had, scratch); jc = block_sum(jc, scratch);
    if (threadIdx.x == 0) {
        double* o = out + (((size_t)b * g.R + r) * gridDim.x + blockIdx.x) * 3;
        o[0] = sse; o[1] = had; o[2] = jc;
    }
}

}  // namespace eincm
