// eincm_kernels.hip.h — gfx950 device code of the EINCM objective-and-gradient engine.
//
// One evaluation of loss(theta) -> (value, grad) is the launch sequence
//   k_theta   theta (h,w,2) -> Theta (H,W,2) fp64 + per-tile velocity bounds      [theta_utils.py:25-35]
//   k_splat   warp + 3x3 Gaussian splat of every event at every reference time     [event_warpers.py:28-35,
//             into an LDS-resident destination window, flushed row-wise to HBM      event_utils.py:31-59]
//   k_stats   per-image min/max(+tie counts)/moments/Scharr energy partials        [img_utils.py:24-25,414-421;
//                                                                                    contrast_objectives.py:22-25]
//   k_imgrad  dL/dIWE image (contrast adjoint stencil + MSE-through-normalise)      [reverse of losses.py:61-72]
//   k_gather  backward of the splat: 9-tap gather of dL/dIWE per event -> dL/dTheta [reverse of event_utils.py:59]
//   k_tv      masked-flow total variation + its (unscaled) gradient image           [regularizers.py:14-38]
//   k_project adjoint resample dL/dTheta -> dL/dtheta                               [reverse of theta_utils.py:25-35]
//   k_final   scalar assembly of the loss, aux and the final gradient               [losses.py:176-203]
// Events are binned once per window by 32x32 SOURCE tile (time order, then re-dealt by source pixel inside blocks of 256: k_spread)
// and cut into segments; because Theta is smooth and a segment spans a known time range, the destinations of a segment fall in a
// small bounding box that lives in LDS (u32 fixed-point ds_add), so HBM sees one coalesced flush per segment and reference time
// instead of 9 scattered atomics per warped event.
//
// Every accumulation that crosses workgroups is INTEGER (fixed point): the IWE stack is summed as u64 at the fixed scale 2^30
// (k_splat -> acc, converted to the fp32 IWE by the statistics pass), dL/dTheta and dL/dtheta as i64 at a per-window scale derived
// from max|dL/dIWE| (k_gather / k_project), and the 2-DoF gradient as per-workgroup fp64 partials summed in a fixed order
// (k_final).  Integer adds commute, so results are bit-identical from run to run whatever order the hardware retires atomics in.
// Accumulators are cleared by their CONSUMER (read, then write 0), never by a separate clear pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eincm {

constexpr int TS = 32;            // source tile edge (pixels)
constexpr int NT = 256;           // threads per workgroup = 4 waves of 64
constexpr int NWAVE = NT / 64;
constexpr int WIN_CAP_DEFAULT = 2304;   // pixels of LDS for a segment's destination window: u32 chunk + f32 sum = 18 KiB -> 8 workgroups/CU
constexpr int WIN_CAP_MAX = 9216;
constexpr int NXCD = 8;           // XCDs: blocks b and b+8 share an L2 (round-robin dispatch; speed only, never correctness)
constexpr double EPSN = 2.220446049250313e-16;   // sys.float_info.epsilon (losses.py:24)
constexpr float INV_2PI = 0.15915494309189535f;
// LDS accumulation of the splat is u32 fixed point: on gfx950 ds_add_f32 retires ~1 lane per 3 clocks whatever the
// address pattern (0.33 lane-ops/clk/CU measured, tools/lds_atomic_bench.hip) while ds_add_u32 sustains 5-7.4.
// One tap is <= 1/(2*pi) = 0.1592 and a chunk holds <= MAX_CHUNK events, so with the scale 2^k of fix_shift (the largest power of two
// with count * 0.16 * 2^k <= 2^32: k = 21 at 8192 events, 22 at 4096) the integer sum of a window pixel cannot overflow.
// Resolution 2^-k (round to nearest, unbiased): 4.8e-7 absolute per tap at 8192 events.
constexpr int MAX_CHUNK = 16384;   // events per inner chunk (bounds the u32 sums; the fixed-point scale follows the count, fix_shift)
constexpr int MAX_SEG = 1 << 20;   // events per segment (one window flush per segment and reference time)
// Per-item scale 2^k, the largest power of two with count * 0.16 * 2^k <= 2^32 (k capped at 30, where the smallest
// tap 0.0137 still keeps its full fp32 mantissa): k = 23 for 2048 events, 22 for 4096, 30 for <= 25 events — sparse
// items are accumulated essentially exactly, dense ones with an absolute step (6e-8) below the fp32 ulp of their sums.
__device__ __forceinline__ int fix_shift(int count) {
    const unsigned c = (unsigned)ceilf((float)count * 0.16f);
    const int ceillog2 = (c <= 1u) ? 0 : (32 - __clz(c - 1u));
    return min(30, 32 - ceillog2);
}
// A tile's c events are cut into ceil(c/seg) segments of EQUAL length (rounded up to whole workgroup trips), not into
// seg, seg, ..., remainder: the event kernels' workgroups then finish together instead of leaving a tail of short ones.
__host__ __device__ __forceinline__ int balanced_seg_len(int c, int seg) {
    const int nseg = (c + seg - 1) / seg;
    if (nseg <= 1) return seg;
    const int len = (((c + nseg - 1) / nseg + 255) / 256) * 256;
    return len < seg ? len : seg;
}
constexpr float EXP_M05 = 0.6065306597126334f;   // exp(-1/2)

struct Geom {
    int H, W, R, B;
    int tilesX, tilesY, ntiles;
    int wincap, winmaxw;      // LDS destination-window capacity (pixels) and maximum width: the splat's segment list (items_s, "list b")
    int wincap_a, winmaxw_a;  // the same for the gather's own list (items, "list a"): its segments are longer, so they span more time and move further
    int nparts;               // StatParts per image written by the statistics kernel of this evaluation (ntiles or NSPART)
    int pstride;              // StatPart slots per image: max(ntiles, NSPART, k_imstat workgroups per image)
    int gmax_n;               // words of `gmax` per window: R * nig per-strip maxima of k_imgrad, or R bounds from k_imstat's tail
    unsigned long long wmask; // bit b: window b takes part in this evaluation (eincm_loss_grad_masked: a lockstep solver's converged windows
                              // sit out; their workgroups leave at once and their outputs are not written).  Windows >= 64 always take part.
    int igx, nig;             // k_imgrad strips per image row / per image (IG_COLS x IG_ROWS pixels each): slots of g2parts and gmax
    int pitch_aligned;        // LDS windows of the splat's event copy at a row pitch rounded up to the 32 banks (win_pitch)
};

__device__ __forceinline__ bool win_active(const Geom& g, int b) { return b >= 64 || ((g.wmask >> b) & 1ull) != 0ull; }

struct Item {                     // one segment of event work: <= seg events of one source tile of one window
    int32_t win, tile, begin, count;
    double t_lo, t_hi;            // time range of its events
};

struct StatPart {                 // per (window, ref, tile) partial of the image reductions
    double mn, mx, cmn, cmx;      // min, max and how many pixels attain them inside the tile
    double sI, sII, sEI, sG2;     // sum I, sum I^2, sum E*I, sum (gx^2+gy^2)
};

struct ImgScal {                  // per (window, ref) reduced scalars
    double m, M, D, cm, cM, sI, sII, sEI, sG2;
};
constexpr int IMGSCAL_N = 9;      // doubles per image handed to the host: m, M, D, cm, cM, sI, sII, sEI, sG2

struct WinConst {                 // theta-independent constants of a window (losses.py:54-55,66,71,80,84)
    double c0_gradmag, c0_var, d0;
    double zc[16];                // zero_corrs[r] = -MSE(E_r, n0)
    double sE[16], sEE[16];       // sum E_r, sum E_r^2
    double eabs[16];              // max |E_r| (bounds dL/dIWE: gbound_from)
    double inv_c0_gradmag, inv_c0_var, inv_zc[16];   // 1 / (c0 + eps), 1 / (zc[r] + eps): gcoef_from multiplies
    double mrw[16];               // multi-reference weights (losses.py:39-46)
    double dtmax;                 // max |t_e - tau_r| over the window's events and reference times (bounds a gradient term)
    double nev;                   // events staged in this window of this context
    double cntmax;                // most events on one source pixel (bounds a per-pixel gradient sum)
};

// The IWE accumulator holds pixel * 2^ACC_SHIFT as u64.  A segment's u32 LDS sums have scale 2^fshift with fshift <= 30 = ACC_SHIFT,
// so the flush is an exact left shift: the accumulated image is the exact sum of the per-tap fixed-point values, whatever the order.
// One tap is <= 1/(2 pi) and no event puts two taps on one pixel: pixel <= 0.16 * N, and 0.16 * 2e9 events * 2^30 < 2^63 cannot overflow.
// (A u32 accumulator at a scale safe for the worst case - 2^14 at 10^6 events, 2^11 at 10^7 - was measured first: 2.6e-4 relative
// error in the gradient at 10^7 events.  u64 atomics cost 0-5 % of k_splat, the flush hides behind the event loop.)
constexpr int ACC_SHIFT = 30;

// i64 fixed point of the gradient accumulators: value * 2^eg, magnitudes < 2^51 so that (a) the fp64 -> i64 conversion by the
// 1.5 * 2^52 magic constant is exact rounding and (b) i64 -> fp64 is exact.  bound = an upper bound of the sum of the magnitudes of
// everything that can be added into one accumulator; returns eg with bound * 2^eg < 2^50.
__device__ __forceinline__ int fix64_shift(double bound) {
    if (!(bound > 0.0) || !(bound < 1.0e300)) return 0;          // nothing to add, or NaN / Inf (the loss is NaN then anyway)
    int e;
    (void)frexp(bound, &e);                                      // bound < 2^e
    return 50 - e;
}
constexpr double FIX64_MAGIC = 6755399441055744.0;               // 1.5 * 2^52: low 32 bits of its pattern are zero
__device__ __forceinline__ long long fix64(double scaled) {      // round-to-nearest-even integer of |scaled| < 2^51
    return __double_as_longlong(scaled + FIX64_MAGIC) - __double_as_longlong(FIX64_MAGIC);
}
// The same for |scaled| < 2^62, in two 32-bit halves (gfx950 has no fp64 -> i64 conversion): used where a sum needs every bit it
// can get and the conversion is per pixel, not per event (k_project).
__device__ __forceinline__ int fix64_wide_shift(double bound) {
    if (!(bound > 0.0) || !(bound < 1.0e300)) return 0;
    int e;
    (void)frexp(bound, &e);
    return 61 - e;
}
__device__ __forceinline__ long long fix64_wide(double scaled) {
    const long long hi = fix64(scaled * (1.0 / 4294967296.0));              // |.| < 2^30
    const double rem = fma(-(double)hi, 4294967296.0, scaled);              // exact, |rem| <= 2^31
    return hi * 4294967296ll + fix64(rem);
}
// |dL/dw| of one event at one reference time is at most max|G| * sum_taps k |q| <= max|G| * 9 * 0.1592 * 1.5 = 2.15 max|G|; times
// |dt| <= dtmax.  Two scales, because max|G| sits on the few arg-min / arg-max pixels of the normalisation and is orders of
// magnitude above a typical |G|, so bits are precious:
//   per SOURCE PIXEL (LDS accumulators of k_gather, the dL/dTheta image): at most cntmax events per pixel, R reference times; < 2^50
//     (the cheap per-event conversion fix64);
//   per theta CELL (k_project's sums): all nev * R events, times <= 4 for the resampling weights (sum |A_H||A_W| of any method);
//     < 2^61 (fix64_wide): at 10^7 events a 2^50 range left 1e-6 per rounding and 2e-5 relative error in the gradient.
// Every event is rounded once at the fine pixel scale; a pixel's total is rounded once more when it enters a cell.
// max |G| of a window = max over the per-strip maxima k_imgrad stored (R * nig words; plain stores there, because 4000
// same-address atomicMax cost k_imgrad 18 us).  Called by every thread of a workgroup; result in all threads.  One __syncthreads.
__device__ __forceinline__ double gmax_of(const unsigned* __restrict__ gmax_win, int n, unsigned* lds_scratch /* NWAVE words */) {
    unsigned m = 0u;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = max(m, gmax_win[i]);       // non-negative floats order like their bit patterns
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) lds_scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    m = lds_scratch[0];
    for (int i = 1; i < nw; ++i) m = max(m, lds_scratch[i]);
    return (double)__uint_as_float(m);
}
// wide: 61 bits and the two-halves conversion per event (10 % of k_gather).  The host asks for it when a window has so few events
// that speed is irrelevant: with a handful of events the arg-max term of the normalisation can put max|G| fifteen orders of
// magnitude above the gradient that survives the cancellation (fuzz case: one event, max|G| 8.5e11, max|dL/dtheta| 0.036), and
// 50 bits under that bound leave 1e-3.
__device__ __forceinline__ int grad_shift_pixel(const WinConst& c, double gm, int R, bool wide) {
    const double bound = 2.15 * gm * c.dtmax * fmax(c.cntmax, 1.0) * (double)R;
    return wide ? fix64_wide_shift(bound) : fix64_shift(bound);
}
__device__ __forceinline__ int grad_shift(const WinConst& c, double gm, int R) {
    return fix64_wide_shift(2.15 * gm * c.dtmax * fmax(c.nev, 1.0) * (double)R * 4.0);
}

struct EvalParams {
    double alpha, beta, gamma, delta;
    int cur_pyr_lvl, contrast_kind;
    int want_div, want_tv, use_tv_grad;
    int h, w, identity;
};

constexpr int THETA_ARG_MAX = 128;    // doubles of theta that ride in the kernel arguments instead of an H2D copy
struct ThetaArg { double v[THETA_ARG_MAX]; };
constexpr int THETA_ARG_BIG = 4096;   // k_theta alone takes up to 32 KiB of theta (16x16 grids of 8 windows) in its arguments: no read of pinned host memory
struct ThetaArgBig { double v[THETA_ARG_BIG]; };
constexpr int THETA_ARG_MID = 512;    // ... and a 4 KiB form for one window's 16x16 grid: the launch copies its arguments twice on the host (k_theta<TA>)
struct ThetaArgMid { double v[THETA_ARG_MID]; };

struct OutScal {                  // per-window result block written by k_final
    double value, mean_rel_corr, mean_rel_contrast, mean_rel_div, tv, tv_scale, nonfinite, _pad;
    double corr[16], contrast_gm[16], var[16], div[16];
};

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// JAX scatter/gather index rule for frame.at[rs, cs].add(mode='drop') (event_utils.py:59): negative indices
// are normalised first (p in [-n,-1] -> p+n), anything still outside [0,n) is dropped.  Returns -1 for a drop.
__device__ __forceinline__ int wrap_drop(int p, int n) {
    p += (p < 0) ? n : 0;
    return (p >= 0 && p < n) ? p : -1;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// block-wide sum of a double (fixed order: the wave tree, then the waves in index order); result valid in thread 0.
// scratch: >= NW doubles of LDS; NW = waves in the workgroup.
template <int NW = NWAVE>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) r += scratch[i];
    }
    return r;
}

struct Window { int ox, oy, ww, wh; };

// Destination bounding box of an item at reference time tau: source tile shifted by -v*dt for
// v in the tile's velocity bounds and dt in the item's time range, +1 for the 3x3 taps, +1 for rounding.
// Clamped to WIN_CAP floats; taps that fall outside take the (rare) direct-to-HBM path, so ANY box is correct.
// Row pitch of a window in LDS: the width rounded up to the 32 banks.  The bank of a tap is then its window COLUMN (mod 32) whatever
// its row, and staging deals the events so that the 32 lanes of a half-wave come from 32 different source columns (k_spread): their
// nine taps fall on 32 different banks.  tools/lds_atomic_bench2.hip: 11.4 ds_add_u32 lane-ops per clock and CU that way, 9.8 with a
// +-1 column jitter on 30 % of the lanes, 7.2 with random columns (what a pitch equal to the width gives).
// The aligned pitch is a property of the staged batch (Geom::pitch_aligned, set_windows_impl): it pays where the windows are resident in
// numbers and a tile holds about one segment (the bench batch: both event kernels 2 % faster), and costs where tiles hold several
// segments' worth of events (480x640 with 10^7 events: k_splat 73 -> 84 us at equal LDS capacity; profiles/r03/pitch_by_shape.txt).
__device__ __forceinline__ int win_pitch(const Geom& g, int ww) { return g.pitch_aligned ? ((ww + 31) & ~31) : ww; }

// aligned: the window is stored at the bank-aligned pitch (the splat's copy of the events: k_splat and the 2-DoF gather); the theta-grid
// gather walks the pixel-sorted copy, whose half-waves hold neighbouring pixels of two or three rows: there a pitch of 64 lets the rows
// alias onto the same banks (146 -> 150 us), so it keeps pitch = width.
__device__ __forceinline__ Window item_window(const Geom& g, const Item& it, const double* __restrict__ mm4, double tau, int wincap, int winmaxw,
                                              bool aligned_list = true) {
    const bool aligned = aligned_list && g.pitch_aligned != 0;
    const int tx = it.tile % g.tilesX, ty = it.tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    const int x1 = min(x0 + TS, g.W) - 1, y1 = min(y0 + TS, g.H) - 1;
    const double dlo = it.t_lo - tau, dhi = it.t_hi - tau;
    const double LIM = 4096.0;
    double lo[2], hi[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double vmin = mm4[2 * c], vmax = mm4[2 * c + 1];
        const double a = -vmin * dlo, b = -vmin * dhi, cc = -vmax * dlo, d = -vmax * dhi;
        double mn = fmin(fmin(a, b), fmin(cc, d)), mx = fmax(fmax(a, b), fmax(cc, d));
        if (!(mn == mn) || !(mx == mx)) { mn = 0.0; mx = 0.0; }          // NaN theta: any window is correct
        lo[c] = floor(fmin(fmax(mn, -LIM), LIM));
        hi[c] = ceil(fmin(fmax(mx, -LIM), LIM));
    }
    int bx0 = x0 + (int)lo[0] - 2, bx1 = x1 + (int)hi[0] + 2;
    int by0 = y0 + (int)lo[1] - 2, by1 = y1 + (int)hi[1] + 2;
    int ww = bx1 - bx0 + 1, wh = by1 - by0 + 1;
    if ((aligned ? win_pitch(g, ww) : ww) * wh > wincap || ww > winmaxw) {          // (LDS holds pitch x wh words)
        int nww = min(ww, winmaxw);
        if (aligned && nww >= 64) nww &= ~31;        // a clamped wide window: a whole number of 32-word bank rows, so that no LDS is pitch padding
        const int nwh = min(wh, wincap / (aligned ? win_pitch(g, nww) : nww));
        bx0 = (bx0 + bx1) / 2 - nww / 2;
        by0 = (by0 + by1) / 2 - nwh / 2;
        ww = nww; wh = nwh;
    }
    Window w; w.ox = bx0; w.oy = by0; w.ww = ww; w.wh = wh;
    return w;
}

// Warp one coordinate (event_warpers.py:34-35) and split it the way events_to_pdf_frame does
// (event_utils.py:32-33): w = x - v*dt (fp64, same operations as the reference so the half-to-even rounding
// decisions agree), r = rint(w), f = w - r in [-0.5, 0.5] (exact in fp64), returned as (int r, float f).
__device__ __forceinline__ void warp_axis(int x, double v, double dt, int& ir, float& f) {
    // no FMA contraction: the reference rounds the product theta * dt before the subtraction (event_warpers.py:34-35; XLA's CPU
    // backend does not contract without fast-math, numpy cannot).  fma(-v, dt, x) differs from that in the last bit of w, which
    // decides rint() for events within 1e-16 relative of a half-integer (tests/test_gpu_parity.py::test_warp_rounds_the_product_first).
#ifndef EINCM_ABL_WARP_FMA                 // timing-only build: what the second rounding costs (profiles/r03/warp_rounding_cost.txt)
#pragma clang fp contract(off)
#endif
    const double w = (double)x - v * dt;
    const double r = rint(w);
    f = (float)(w - r);                              // garbage when the event is off-sensor, but then every tap is dropped
    ir = (int)r;                                     // v_cvt_i32_f64 saturates (NaN -> 0; a NaN theta is caught in k_final)
}
// The saturated extremes are tamed only where index arithmetic follows (the out-of-window paths): one v_med3_i32; |w| beyond any
// sensor (W, H <= 32767) drops every tap.  The in-window test `(unsigned)(ir - 1 - ox) < ww - 2` needs no clamp: |ox| < 2^16 and
// ww < 2^12, so a wrapped difference of a saturated ir lands near +-2^31, never inside the window.
__device__ __forceinline__ int clamp_far(int ir) { return min(max(ir, -(1 << 20)), 1 << 20); }

// (x-axis value, y-axis value).  A plain struct on purpose: as an ext_vector_type the pairs lower to v_pk_mul_f32 / v_pk_fma_f32, and on
// gfx950 packed fp32 issues at half the rate of scalar fp32 AND needs register-pair shuffles: building with the SLP vectoriser off
// (which packs the gather's combination the same way) took k_gather from 120 to 106 us.
struct f2v { float x, y; };

// Separable 3-tap weights of BOTH axes: k(d) = exp(-0.5 (d - f)^2) = exp(-0.5 f^2) exp(d f) exp(-0.5 d^2),
// d = -1, 0, 1 (event_utils.py:52-56), normalised to the centre tap: per axis (a, 1, b) with b = exp(f - 1/2), a = exp(-f - 1/2) = e^-1 / b;
// the common factor exp(-(fx^2 + fy^2) / 2) * scale_y (1/(2 pi), times the fixed-point scale in k_splat) rides on the y weights.
// 3 v_exp_f32 + 2 v_rcp_f32 (1 ulp each) + 10 multiply-adds (the unnormalised form took 4 + 2 + 13: k_splat is bound by VALU issue).
__device__ __forceinline__ void taps3x2(float fx, float fy, float scale_y, f2v& km, f2v& k0, f2v& kp) {
    constexpr float L2E = 1.4426950408889634f, EXP_M1 = 0.36787944117144233f;
    const float bx = __builtin_amdgcn_exp2f(fmaf(fx, L2E, -0.5f * L2E)), by = __builtin_amdgcn_exp2f(fmaf(fy, L2E, -0.5f * L2E));
    const float ax = EXP_M1 * __builtin_amdgcn_rcpf(bx), ay = EXP_M1 * __builtin_amdgcn_rcpf(by);
    // (the scale is multiplied on, not folded into the exponent: exp2 of an argument of ~21 would lose five bits of the fraction)
    const float s = __builtin_amdgcn_exp2f(fmaf(fx, fx, fy * fy) * (-0.5f * L2E)) * scale_y;
    km.x = ax; km.y = s * ay;                    // d = -1
    k0.x = 1.0f; k0.y = s;
    kp.x = bx; kp.y = s * by;                    // d = +1
}

struct EvReg { uint32_t xy; double t; };   // one event in flight through the software pipeline of the event kernels

// Where an event's velocity Theta[y,x] comes from inside the event kernels.  A per-event global gather costs ~64 L1 cycles per
// wave-instruction (64 lanes, 64 different cache lines) and was 2/3 of k_splat's time; every workgroup works on ONE source tile, so:
constexpr int THETA_CONST = 1;   // theta (1,1,2): Theta is one constant per window (read from the tile bounds, min == max)
constexpr int THETA_TILE = 2;    // otherwise: the tile's 32x32 double2 velocities are staged in LDS once per segment (16 KiB)

// round-to-nearest fixed-point conversion of a positive product < 2^31: v_mul_f32 + v_cvt_rpi_i32_f32 (floor(x + 0.5) in the converter itself;
// the compiler has no builtin for it).  Where b is the constant 1 (the centre column of the normalised taps) the multiply folds away: one
// instruction instead of the add + truncating convert of (uint32_t)(a + 0.5f) - k_splat is bound by VALU issue.
__device__ __forceinline__ uint32_t cvt_rpi(float x) { uint32_t r; asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t fix_u32(float a, float b) { return cvt_rpi(a * b); }

// The rows of A_H and the columns of A_W that one 32x32 tile needs, staged in LDS by the whole workgroup (k_theta, k_project).
// Per-pixel reads of the tap ranges and weights from global memory were chains of dependent loads: 12 us of k_theta and 18-30 us
// of k_project were that latency.  A tile touches the coarse rows [ilo, ilo + ni) and columns [jlo, jlo + nj); staged when both
// fit RS_MAXC (theta grids up to ~100 cells per axis at the usual sensors; anything else keeps the direct path).
constexpr int RS_MAXC = 24;
struct TileRange { int ilo, ni, jlo, nj; };     // the coarse cells a tile's pixels have weight on (host: ensure_resample)
struct ResampleTile {
    int ilo, ni, jlo, nj;
    bool staged;
};
struct ResampleLds {
    int2 rt[TS], ct[TS];
    double ah[TS * RS_MAXC], aw[TS * RS_MAXC];
    int rng[4];
};
// Two steps, each called by every thread of the workgroup (>= 64 threads) and each ending in a __syncthreads: the tap ranges of the
// tile's rows and columns (-> which coarse cells it touches), then the weights.  A caller issues whatever else depends only on the
// ranges (k_theta: its theta cells) between the two, so that it shares the second round trip.
__device__ __forceinline__ ResampleTile stage_resample_ranges(ResampleLds& L, int h, int w, int x0, int y0, int x1, int y1,
                                                              const int2* __restrict__ rowtap, const int2* __restrict__ coltap) {
    const int t = threadIdx.x;
    if (t < TS) L.rt[t] = (y0 + t < y1) ? rowtap[y0 + t] : make_int2(h, 0);
    else if (t < 2 * TS) L.ct[t - TS] = (x0 + t - TS < x1) ? coltap[x0 + t - TS] : make_int2(w, 0);
    __syncthreads();
    if (t < 64) {                                   // lanes 0..31: rows, 32..63: columns; min / max inside each half
        const int2 v = (t < TS) ? L.rt[t] : L.ct[t - TS];
        int lo = v.x, hi = v.y;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
        if (t == 0) { L.rng[0] = lo; L.rng[1] = hi; }
        if (t == TS) { L.rng[2] = lo; L.rng[3] = hi; }
    }
    __syncthreads();
    ResampleTile r;
    r.ilo = L.rng[0]; r.ni = max(L.rng[1] - L.rng[0], 0); r.jlo = L.rng[2]; r.nj = max(L.rng[3] - L.rng[2], 0);
    r.staged = r.ni <= RS_MAXC && r.nj <= RS_MAXC;
    return r;
}
__device__ __forceinline__ void stage_resample_weights(ResampleLds& L, const ResampleTile& r, int h, int w, int x0, int y0, int x1, int y1,
                                                       const double* __restrict__ AH, const double* __restrict__ AW) {
    if (!r.staged) return;                          // uniform
    const int t = threadIdx.x;
    for (int k = t; k < TS * r.ni; k += blockDim.x) {
        const int ly = k / r.ni, i = k - ly * r.ni;
        L.ah[k] = (y0 + ly < y1) ? AH[(size_t)(y0 + ly) * h + r.ilo + i] : 0.0;
    }
    for (int k = t; k < TS * r.nj; k += blockDim.x) {
        const int lx = k / r.nj, j = k - lx * r.nj;
        L.aw[k] = (x0 + lx < x1) ? AW[(size_t)(x0 + lx) * w + r.jlo + j] : 0.0;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// k_theta: Theta = A_H theta A_W^T per channel, and per-tile velocity bounds.
// grid (ntiles, B).  identity: theta already is (H,W,2).  Not launched for 2-DoF theta unless somebody needs the Theta image
// (k_theta_const fills the velocity bounds then).
// ------------------------------------------------------------------------------------------------
template <typename TA>                 // TA: ThetaArgBig or ThetaArgMid, the kernel-argument block theta rides in when use_arg
__global__ __launch_bounds__(NT) void k_theta(Geom g, int h, int w, int identity, int use_arg, TA targ,
        const double* __restrict__ theta,      // (B,h,w,2)
        const double* __restrict__ AH,         // (H,h)
        const double* __restrict__ AW,         // (W,w)
        const int2* __restrict__ rowtap,       // (H) [lo,hi) non-zero range of AH[y,:]
        const int2* __restrict__ coltap,       // (W)
        const TileRange* __restrict__ tilerng, // (ntiles) the coarse cells under each tile (host: ensure_resample)
        double* __restrict__ Theta,            // (B,H,W,2)
        double* __restrict__ tmm,              // (B,ntiles,4) vxmin,vxmax,vymin,vymax
        // window tables of this tile's segments (both segment lists), filled here when itembase_* != nullptr: a workgroup knows its
        // tile's velocity bounds, and the segments of a (window, tile) are contiguous from itembase[b * ntiles + tile]
        const double* __restrict__ edge_ts,
        int n_a, const Item* __restrict__ items_a, const int32_t* __restrict__ itembase_a, Window* __restrict__ wins_a,
        int n_b, const Item* __restrict__ items_b, const int32_t* __restrict__ itembase_b, Window* __restrict__ wins_b)
{
    __shared__ double red[4][NWAVE];
    __shared__ double mm4s[4];
    __shared__ double nanred[NWAVE];
    __shared__ ResampleLds RL;
    __shared__ double2 sth[RS_MAXC * RS_MAXC];      // the coarse cells under this tile
    const int tile = blockIdx.x, b = blockIdx.y;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const double* th = theta + (size_t)b * h * w * 2;
    if (!win_active(g, b)) return;
    double* Th = Theta + (size_t)b * g.H * g.W * 2;
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    bool nan = false;
    ResampleTile rs{};
    // A latency chain, not a throughput kernel (88 workgroups on a 256x336 sensor; 9.8 us of a 48 us evaluation there): every dependent
    // round trip to memory counts.  The tile's cell range comes from the host's table instead of a min / max over the tap ranges (one
    // round trip and two barriers less), and the segment ranges of the window tables below are fetched up front.
    const int idx = b * g.ntiles + tile, last = g.B * g.ntiles - 1;
    int a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (itembase_a != nullptr) {                    // uniform
        a0 = itembase_a[idx]; a1 = (idx < last) ? itembase_a[idx + 1] : n_a;
        b0 = itembase_b[idx]; b1 = (idx < last) ? itembase_b[idx + 1] : n_b;
    }
    if (!identity) {
        const int tx0 = tx * TS, ty0 = ty * TS, tx1 = min(tx * TS + TS, g.W), ty1 = min(ty * TS + TS, g.H);
        const TileRange q = tilerng[tile];
        rs.ilo = q.ilo; rs.ni = q.ni; rs.jlo = q.jlo; rs.nj = q.nj;
        rs.staged = rs.ni <= RS_MAXC && rs.nj <= RS_MAXC;
        if (rs.staged) {                            // the tap ranges of the tile's rows and columns (stage_resample_weights' barrier covers them)
            const int t = threadIdx.x;
            if (t < TS) RL.rt[t] = (ty0 + t < ty1) ? rowtap[ty0 + t] : make_int2(h, 0);
            else if (t < 2 * TS) RL.ct[t - TS] = (tx0 + t - TS < tx1) ? coltap[tx0 + t - TS] : make_int2(w, 0);
        }
        if (rs.staged) {                            // theta is read once per cell and workgroup (it may live in pinned host memory)
            for (int k = threadIdx.x; k < rs.ni * rs.nj; k += NT) {
                const int i = rs.ilo + k / rs.nj, j = rs.jlo + k % rs.nj;
                const int o = ((b * h + i) * w + j) * 2;
                sth[k] = use_arg ? make_double2(targ.v[o], targ.v[o + 1])
                                 : *reinterpret_cast<const double2*>(th + ((size_t)i * w + j) * 2);
            }
        }
        stage_resample_weights(RL, rs, h, w, tx0, ty0, tx1, ty1, AH, AW);      // its barrier covers sth as well
    }
    for (int p = threadIdx.x; p < TS * TS; p += NT) {
        const int ly = p / TS, lx = p % TS;
        const int y = ty * TS + ly, x = tx * TS + lx;
        if (y >= g.H || x >= g.W) continue;
        double vx, vy;
        if (rs.staged) {                            // the same operations in the same order as the direct forms below
            const int2 rt = RL.rt[ly], ct = RL.ct[lx];
            vx = 0.0; vy = 0.0;
            for (int i = rt.x; i < rt.y; ++i) {
                const double a = RL.ah[ly * rs.ni + (i - rs.ilo)];
                double sx = 0.0, sy = 0.0;
                for (int j = ct.x; j < ct.y; ++j) {
                    const double bw = RL.aw[lx * rs.nj + (j - rs.jlo)];
                    const double2 v = sth[(i - rs.ilo) * rs.nj + (j - rs.jlo)];
                    sx += bw * v.x; sy += bw * v.y;
                }
                vx += a * sx; vy += a * sy;
            }
        } else if (use_arg) {
            const int2 rt = rowtap[y], ct = coltap[x];
            vx = 0.0; vy = 0.0;
            for (int i = rt.x; i < rt.y; ++i) {
                const double a = AH[(size_t)y * h + i];
                double sx = 0.0, sy = 0.0;
                for (int j = ct.x; j < ct.y; ++j) {
                    const double bw = AW[(size_t)x * w + j];
                    const int o = ((b * h + i) * w + j) * 2;
                    sx += bw * targ.v[o]; sy += bw * targ.v[o + 1];
                }
                vx += a * sx; vy += a * sy;
            }
        } else if (identity) {
            const double2 v = *reinterpret_cast<const double2*>(th + ((size_t)y * g.W + x) * 2);
            vx = v.x; vy = v.y;
        } else {
            const int2 rt = rowtap[y], ct = coltap[x];
            vx = 0.0; vy = 0.0;
            for (int i = rt.x; i < rt.y; ++i) {
                const double a = AH[(size_t)y * h + i];
                double sx = 0.0, sy = 0.0;
                for (int j = ct.x; j < ct.y; ++j) {
                    const double bw = AW[(size_t)x * w + j];
                    const double2 v = *reinterpret_cast<const double2*>(th + ((size_t)i * w + j) * 2);
                    sx += bw * v.x; sy += bw * v.y;
                }
                vx += a * sx; vy += a * sy;
            }
        }
        *reinterpret_cast<double2*>(Th + ((size_t)y * g.W + x) * 2) = make_double2(vx, vy);
        nan |= !(vx == vx) || !(vy == vy);
        mnx = fmin(mnx, vx); mxx = fmax(mxx, vx); mny = fmin(mny, vy); mxy = fmax(mxy, vy);
    }
    if (nan) { mnx = mxx = mny = mxy = NAN; }   // fmin/fmax drop NaNs; keep them visible to item_window
    mnx = wave_min(mnx); mxx = wave_max(mxx); mny = wave_min(mny); mxy = wave_max(mxy);
    // NaN-ness must survive the reduction: reduce a flag alongside
    const double nanf = wave_max(nan ? 1.0 : 0.0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = mnx; red[1][wv] = mxx; red[2][wv] = mny; red[3][wv] = mxy; nanred[wv] = nanf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = red[0][0], bq = red[1][0], c = red[2][0], d = red[3][0], nf = nanred[0];
        for (int i = 1; i < NWAVE; ++i) {
            a = fmin(a, red[0][i]); bq = fmax(bq, red[1][i]); c = fmin(c, red[2][i]); d = fmax(d, red[3][i]);
            nf = fmax(nf, nanred[i]);
        }
        double* o = tmm + ((size_t)b * g.ntiles + tile) * 4;
        if (nf > 0.0) { a = bq = c = d = NAN; }
        o[0] = a; o[1] = bq; o[2] = c; o[3] = d;
        mm4s[0] = a; mm4s[1] = bq; mm4s[2] = c; mm4s[3] = d;
    }
    if (itembase_a == nullptr) return;           // uniform: the caller fills the window tables elsewhere (k_theta_const / k_windows)
    __syncthreads();
    const double mm4[4] = {mm4s[0], mm4s[1], mm4s[2], mm4s[3]};
    const int na = (a1 - a0) * g.R, nb = (b1 - b0) * g.R;
    for (int k = threadIdx.x; k < na + nb; k += NT) {
        const bool first = k < na;
        const int kk = first ? k : k - na;
        const int item = (first ? a0 : b0) + kk / g.R, r = kk % g.R;
        const Item it = (first ? items_a : items_b)[item];
        (first ? wins_a : wins_b)[(size_t)item * g.R + r] = item_window(g, it, mm4, edge_ts[it.win * g.R + r], first ? g.wincap_a : g.wincap,
                                                                          first ? g.winmaxw_a : g.winmaxw, !first);
    }
}

// The destination windows of every (segment, reference time) pair, computed once per evaluation instead of by every thread of
// every event workgroup (item_window is ~150 instructions, a tenth of a short segment's work).  Thread per (segment, r).
__device__ __forceinline__ void windows_of(const Geom& g, int idx, int n_a, const Item* __restrict__ items_a, Window* __restrict__ wins_a,
                                           int n_b, const Item* __restrict__ items_b, Window* __restrict__ wins_b,
                                           const double* __restrict__ tmm, const double* __restrict__ edge_ts, bool const_theta, const ThetaArg& targ,
                                           const double* __restrict__ theta, int use_arg) {
    const bool first = idx < n_a * g.R;
    const int k = first ? idx : idx - n_a * g.R;
    if (!first && k >= n_b * g.R) return;
    const Item it = (first ? items_a : items_b)[k / g.R];
    if (!win_active(g, it.win)) return;
    const int r = k % g.R;
    double mm4[4];
    if (const_theta) {
        const double vx = use_arg ? targ.v[2 * it.win] : theta[2 * it.win], vy = use_arg ? targ.v[2 * it.win + 1] : theta[2 * it.win + 1];
        mm4[0] = vx; mm4[1] = vx; mm4[2] = vy; mm4[3] = vy;
    } else {
        const double* mm = tmm + ((size_t)it.win * g.ntiles + it.tile) * 4;
        mm4[0] = mm[0]; mm4[1] = mm[1]; mm4[2] = mm[2]; mm4[3] = mm[3];
    }
    (first ? wins_a : wins_b)[k] = item_window(g, it, mm4, edge_ts[it.win * g.R + r], first ? g.wincap_a : g.wincap, first ? g.winmaxw_a : g.winmaxw, !first);
}

// 2-DoF theta (1,1,2): Theta is one constant per window, so the velocity bounds of every tile are that constant; no image is
// written.  Also fills the window tables of both segment lists.  grid covers max(B*ntiles, (n_a + n_b) * R) threads.
// theta rides in the kernel arguments (B*2 <= THETA_ARG_MAX) or is read from `theta`.
__global__ __launch_bounds__(NT) void k_theta_const(Geom g, int use_arg, ThetaArg targ, const double* __restrict__ theta,
                                                     double* __restrict__ tmm, const double* __restrict__ edge_ts,
                                                     int n_a, const Item* __restrict__ items_a, Window* __restrict__ wins_a,
                                                     int n_b, const Item* __restrict__ items_b, Window* __restrict__ wins_b)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i < g.B * g.ntiles && win_active(g, i / g.ntiles)) {
        const int b = i / g.ntiles;
        const double vx = use_arg ? targ.v[2 * b] : theta[2 * b], vy = use_arg ? targ.v[2 * b + 1] : theta[2 * b + 1];
        double* o = tmm + (size_t)i * 4;
        o[0] = vx; o[1] = vx; o[2] = vy; o[3] = vy;          // NaN stays NaN: item_window and k_final see it
    }
    windows_of(g, i, n_a, items_a, wins_a, n_b, items_b, wins_b, tmm, edge_ts, true, targ, theta, use_arg);
}

// Any other theta: the window tables after k_theta has written the per-tile velocity bounds.  grid covers (n_a + n_b) * R threads.
__global__ __launch_bounds__(NT) void k_windows(Geom g, const double* __restrict__ tmm, const double* __restrict__ edge_ts,
                                                 int n_a, const Item* __restrict__ items_a, Window* __restrict__ wins_a,
                                                 int n_b, const Item* __restrict__ items_b, Window* __restrict__ wins_b)
{
    ThetaArg none;
    windows_of(g, blockIdx.x * NT + threadIdx.x, n_a, items_a, wins_a, n_b, items_b, wins_b, tmm, edge_ts, false, none, nullptr, 0);
}

// Block -> (segment, reference time).  The R blocks that process one segment at the R reference times read the
// same events; blocks b, b+8, b+16, ... are dealt to the same XCD back to back, so they are made siblings and
// the 2nd..Rth read of a segment's events hits that XCD's L2 instead of HBM.  grid = ceil(n_items/8)*8*R.
// order (or nullptr = identity): the segments by decreasing length, so that the longest workgroups start first and the short ones
// fill the end of the launch (the hardware dispatches workgroups in blockIdx order); which workgroup does which segment has no
// effect on any result (integer accumulation, partials indexed by segment).
__device__ __forceinline__ bool block_to_work(int n_items, int R, const int32_t* __restrict__ order, int& item, int& r) {
    const int b = blockIdx.x;
    const int xcd = b % NXCD, slot = b / NXCD;
    const int rank = (slot / R) * NXCD + xcd;
    r = slot % R;
    if (rank >= n_items) return false;
    item = order ? order[rank] : rank;
    return true;
}

// Flat walk over a ww x wh window by the NT threads of a workgroup: thread t visits pixels t, t + NT, t + 2 NT, ... and keeps
// (row, col) by increments.  Row-wise loops (a wave per window row) left ww/64 of the lanes busy and paid the loop overhead per
// row; the window copies are 10 % of the event kernels' instructions.  ww < 2^12 and t < NT, so the float quotient is within one
// of the integer one and a single correction step makes it exact.
template <int NTH> struct WinWalkT {
    int i, row, col, q, rem, ww;
    __device__ __forceinline__ WinWalkT(int tid, int ww_) : ww(ww_) {
        const float inv = 1.0f / (float)ww_;
        row = (int)((float)tid * inv); col = tid - row * ww_;
        if (col < 0) { col += ww_; --row; } else if (col >= ww_) { col -= ww_; ++row; }
        q = (int)((float)NTH * inv); rem = NTH - q * ww_;
        if (rem < 0) { rem += ww_; --q; } else if (rem >= ww_) { rem -= ww_; ++q; }
        i = tid;
    }
    __device__ __forceinline__ void next() {
        i += NTH; col += rem; row += q;
        if (col >= ww) { col -= ww; ++row; }
    }
};
using WinWalk = WinWalkT<NT>;

// How the threads of a k_gather workgroup walk a segment of n events in the layout staging gives the gather's copy (k_segsort,
// eincm_binning.hip.h): two halves of n0 = ceil(n / 2) and n - n0 events, each stored as K coalesced steps of 256 threads, the
// last step a prefix.  A 512-thread workgroup gives each half to 256 of its threads (K0 uniform iterations); a 256-thread
// workgroup walks the halves one after the other (K0 + K1 iterations).  index() < 0: this thread has no event in iteration i.
// Consecutive events of a thread are consecutive in the segment's sort by source pixel.
template <int NTH> struct SegWalk {
    static_assert(NTH == 256 || NTH == 512, "the staged layout is dealt to groups of 256 threads");
    int n0, K0, K1, rem0, rem1, tt, half;
    __device__ __forceinline__ SegWalk(int n, int tid) {
        n0 = (n + 1) >> 1;
        const int n1 = n - n0;
        K0 = (n0 + 255) >> 8; K1 = (n1 + 255) >> 8;
        rem0 = n0 - 256 * (K0 - 1); rem1 = n1 - 256 * (K1 - 1);
        tt = tid & 255; half = tid >> 8;
    }
    __device__ __forceinline__ int iters() const { return NTH == 512 ? K0 : K0 + K1; }
    __device__ __forceinline__ int index(int i) const {
        const int c = (NTH == 512) ? half : (i >= K0 ? 1 : 0);
        const int j = (NTH == 512) ? i : (c ? i - K0 : i);
        const int K = c ? K1 : K0, rem = c ? rem1 : rem0;
        const bool ok = j < K - 1 || (j == K - 1 && tt < rem);
        return ok ? (c ? n0 : 0) + j * 256 + tt : -1;
    }
};

// ------------------------------------------------------------------------------------------------
// k_splat: the dominant kernel.  grid ceil(n_items/8)*8*R blocks (block_to_work), LDS 2*WIN_CAP*4 bytes.
// A segment is walked in chunks of <= chunk events; each chunk is accumulated in u32 fixed point (exact integer
// ds_add_u32) and committed into the segment's f32 window; the window is flushed to HBM once per segment.
// ------------------------------------------------------------------------------------------------
// MERGE = 1 is the round-3 experiment "run-merged forward accumulation" (EINCM_SPLAT_MERGE=1): the kernel
// walks the GATHER's copy of the events (sorted by source pixel inside a segment, a run per thread: SegWalk) and sums the taps of
// consecutive events of a thread that round to the same destination pixel in nine registers before the nine ds_add_u32.
template <int TM, int MULTI, int NTH, int MERGE>   // TM: the theta mode as a compile-time constant (0 = run-time argument); MULTI = 0: no segment is longer than a chunk
__global__ __launch_bounds__(NTH) void k_splat(Geom g, int n_items, int chunk, int theta_mode, int lds_multi,
        const Item* __restrict__ items,
        const uint32_t* __restrict__ ev_xy,    // x | y << 16, binned by (window, tile)
        const double* __restrict__ ev_t,
        const double* __restrict__ Theta,      // (B,H,W,2)
        const double* __restrict__ tmm,        // (B,ntiles,4)
        const double* __restrict__ edge_ts,    // (B,R)
        const Window* __restrict__ wins,       // (n_items, R) destination windows of this evaluation (k_theta / k_windows); unused for 2-DoF theta
        unsigned long long* __restrict__ acc,  // (B,R,H,W) u64 fixed point at 2^ACC_SHIFT, zero on entry (cleared by its consumer)
        const int32_t* __restrict__ order,     // (n_items) segments by decreasing length (block_to_work)
        int use_arg, const double* __restrict__ theta_c, ThetaArg targ)   // 2-DoF theta (B,2): in the kernel arguments, or behind theta_c
{
    if (TM != 0) theta_mode = TM;                 // every branch on it below folds away: 8 % on both event kernels
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsu[];
    float* ldsf = reinterpret_cast<float*>(ldsu + g.wincap);                       // present only when lds_multi
    double2* thtile = reinterpret_cast<double2*>(ldsu + (lds_multi ? 2 : 1) * g.wincap);   // present only for THETA_TILE
    int item, r;
    if (!block_to_work(n_items, g.R, order, item, r)) return;
    const Item it = items[item];
    if (!win_active(g, it.win)) return;
    const double tau = edge_ts[it.win * g.R + r];
    const int tx0 = (it.tile % g.tilesX) * TS, ty0 = (it.tile / g.tilesX) * TS;
    double2 vconst = make_double2(0.0, 0.0);
    Window wn;
    if (theta_mode == THETA_CONST) {
        // 2-DoF theta: Theta is one constant per window, so the workgroup derives its destination window itself (no k_theta_const
        // launch in front of the splat, no table): ~150 uniform instructions against a dependent kernel boundary per evaluation
        vconst = use_arg ? make_double2(targ.v[2 * it.win], targ.v[2 * it.win + 1]) : make_double2(theta_c[2 * it.win], theta_c[2 * it.win + 1]);
        const double mm4[4] = {vconst.x, vconst.x, vconst.y, vconst.y};
        wn = item_window(g, it, mm4, tau, g.wincap, g.winmaxw);
    } else {
        const double* __restrict__ ThW = Theta + (size_t)it.win * g.H * g.W * 2;
        for (int p = threadIdx.x; p < TS * TS; p += NTH) {
            const int y = ty0 + p / TS, x = tx0 + p % TS;
            thtile[p] = (y < g.H && x < g.W) ? *reinterpret_cast<const double2*>(ThW + ((size_t)y * g.W + x) * 2) : make_double2(0.0, 0.0);
        }
        wn = wins[(size_t)item * g.R + r];
    }
    const int nwin = wn.ww * wn.wh;
    const int wp = MERGE ? wn.ww : win_pitch(g, wn.ww), wp4 = wp * 4, nlds = wp * wn.wh;     // LDS row pitch and words (win_pitch; the MERGE experiment walks the
                                                                            // gather's list, whose window table is sized for pitch = width)
    const bool multi = MULTI != 0 && it.count > chunk;       // MULTI == 0: the commit logic in the loop folds away
    {   // clear the window(s), 16 B per lane
        uint4* z = reinterpret_cast<uint4*>(ldsu);
        const int nq = (nlds + 3) >> 2;
        for (int i = threadIdx.x; i < nq; i += NTH) z[i] = make_uint4(0u, 0u, 0u, 0u);
        if (multi) { uint4* zf = reinterpret_cast<uint4*>(ldsf); for (int i = threadIdx.x; i < nq; i += NTH) zf[i] = make_uint4(0u, 0u, 0u, 0u); }
    }
    __syncthreads();

    unsigned long long* __restrict__ img = acc + ((size_t)it.win * g.R + r) * g.H * g.W;
    const uint32_t* __restrict__ exy = ev_xy + it.begin;
    const double* __restrict__ et = ev_t + it.begin;
    const int n = it.count;
    const int iters = (n + NTH - 1) / NTH;            // uniform over the workgroup
    const int ipc = chunk / NTH;                     // iterations per chunk (chunk is a multiple of NTH)
    const int fshift = fix_shift(min(chunk, n));
    const float FIX_SCALE = ldexpf(1.0f, fshift), FIX_INV = ldexpf(1.0f, -fshift);

    // Software pipeline over the segment's events, unrolled x3 with renamed register sets (no rotation moves, so no forced
    // vmcnt(0)): the (xy, t) loads of event j+2 are in flight while event j is splatted.
    const int tid = threadIdx.x;
#if defined(EINCM_ABL_S_NOLOADT)                  // timing-only: no 8-byte timestamp load (what the event traffic from L2 costs)
    auto load_ev = [&](EvReg& r, int e) { if (e < n) { r.xy = exy[e]; r.t = it.t_lo + 1e-9 * (double)e; } else { r.xy = 0u; r.t = 0.0; } };
#elif defined(EINCM_ABL_S_NOLOAD)                 // timing-only: no event loads at all
    auto load_ev = [&](EvReg& r, int e) { r.xy = ((uint32_t)(ty0 + ((e * 7) & 31)) << 16) | (uint32_t)(tx0 + (e & 31)); r.t = it.t_lo + 1e-9 * (double)e; };
#else
    auto load_ev = [&](EvReg& r, int e) { if (e < n) { r.xy = exy[e]; r.t = et[e]; } else { r.xy = 0u; r.t = 0.0; } };
#endif
    const float scy = INV_2PI * FIX_SCALE;          // one fixed-point scale per segment: every chunk holds <= min(chunk, n) events
    auto splat_ev = [&](const EvReg& ev) {
        const double dt = ev.t - tau;
        const int x = ev.xy & 0xffff, y = ev.xy >> 16;
        // tiles start at multiples of TS = 32, so the in-tile index is (y & 31) * 32 + (x & 31)
        const double2 v = (theta_mode == THETA_CONST) ? vconst : thtile[((ev.xy >> 11) & (31u << 5)) | (ev.xy & 31u)];
        int irx, iry; float fx, fy;
        warp_axis(x, v.x, dt, irx, fx);
        warp_axis(y, v.y, dt, iry, fy);
        f2v km, k0, kp;                              // .x = x-axis weight, .y = y-axis weight (scaled)
#ifdef EINCM_ABL_S_NOMATH                            // timing-only: the atomics at their real addresses, no tap arithmetic
        km.x = km.y = k0.x = k0.y = kp.x = kp.y = 3.0f;
#else
        taps3x2(fx, fy, scy, km, k0, kp);
#endif
        const int lx = irx - 1 - wn.ox, ly = iry - 1 - wn.oy;        // window coords of the top-left tap
        if ((unsigned)lx < (unsigned)(wn.ww - 2) && (unsigned)ly < (unsigned)(wn.wh - 2)) {
            // byte offsets with the row pitch pre-scaled (a scalar): one v_mul_i32_i24 + one v_lshl_add_u32, then one add per further row
            uint32_t* p = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ldsu) + (__mul24(ly, wp4) + (lx << 2)));
            // 3x3 products ky[dy]*kx[dx] + 0.5 (scalar FMAs: packed ones are slower on gfx950), then truncating converts
            uint32_t* p1 = p + wp; uint32_t* p2 = p1 + wp;
#ifdef EINCM_ABL_S_NOATOMIC                          // timing-only: all the arithmetic, one plain store instead of nine atomics
            *p = fix_u32(km.y, km.x) + fix_u32(km.y, k0.x) + fix_u32(km.y, kp.x) + fix_u32(k0.y, km.x) + fix_u32(k0.y, k0.x) + fix_u32(k0.y, kp.x)
               + fix_u32(kp.y, km.x) + fix_u32(kp.y, k0.x) + fix_u32(kp.y, kp.x) + (uint32_t)(p1 - p2);
#else
            atomicAdd(p, fix_u32(km.y, km.x)); atomicAdd(p + 1, fix_u32(km.y, k0.x)); atomicAdd(p + 2, fix_u32(km.y, kp.x));
            atomicAdd(p1, fix_u32(k0.y, km.x)); atomicAdd(p1 + 1, fix_u32(k0.y, k0.x)); atomicAdd(p1 + 2, fix_u32(k0.y, kp.x));
            atomicAdd(p2, fix_u32(kp.y, km.x)); atomicAdd(p2 + 1, fix_u32(kp.y, k0.x)); atomicAdd(p2 + 2, fix_u32(kp.y, kp.x));
#endif
        } else {
            const float kx[3] = {km.x, k0.x, kp.x}, ky[3] = {km.y, k0.y, kp.y};
            const int sx = clamp_far(irx), sy = clamp_far(iry), lxs = sx - 1 - wn.ox, lys = sy - 1 - wn.oy;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cx = lxs + dx, cy = lys + dy;
                    if (cx >= 0 && cy >= 0 && cx < wn.ww && cy < wn.wh) {
                        atomicAdd(ldsu + cy * wp + cx, fix_u32(ky[dy], kx[dx]));
                    } else {
                        const int gx = wrap_drop(sx - 1 + dx, g.W), gy = wrap_drop(sy - 1 + dy, g.H);
                        // straight to HBM in the accumulator's own scale (ky carries 2^fshift; a tap * 2^30 fits 32 bits)
                        if (gx >= 0 && gy >= 0)
                            atomicAdd(img + (size_t)gy * g.W + gx, (unsigned long long)fix_u32(ky[dy] * FIX_INV * 1073741824.0f, kx[dx]));
                    }
                }
            }
        }
    };
    // one pipeline step: cur is splatted, nxt gets its (xy, t); j = iteration index of cur
    auto step = [&](EvReg& cur, EvReg& mid, EvReg& nxt, int j) {
        const int e = j * NTH + tid;
        load_ev(nxt, e + 2 * NTH);
        if (e < n) splat_ev(cur);
        if (multi && ((j + 1) % ipc == 0 || j + 1 == iters)) {     // chunk boundary (uniform): commit u32 -> f32 window
            __syncthreads();
            for (int i = tid; i < nlds; i += NTH) {
                const uint32_t u = ldsu[i];
                if (u != 0u) { ldsf[i] += (float)u * FIX_INV; ldsu[i] = 0u; }
            }
            __syncthreads();
        }
    };
    if (MERGE) {
        int cur_off = -1;                            // LDS word of the top-left tap of the run in the registers (-1: none)
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0, m8 = 0;
        auto flush9 = [&]() {
            if (cur_off < 0) return;
            uint32_t* p = ldsu + cur_off; uint32_t* p1 = p + wp; uint32_t* p2 = p1 + wp;
            atomicAdd(p, m0); atomicAdd(p + 1, m1); atomicAdd(p + 2, m2);
            atomicAdd(p1, m3); atomicAdd(p1 + 1, m4); atomicAdd(p1 + 2, m5);
            atomicAdd(p2, m6); atomicAdd(p2 + 1, m7); atomicAdd(p2 + 2, m8);
        };
        auto merge_ev = [&](const EvReg& ev) {
            const double dt = ev.t - tau;
            const int x = ev.xy & 0xffff, y = ev.xy >> 16;
            const double2 v = (theta_mode == THETA_CONST) ? vconst : thtile[((ev.xy >> 11) & (31u << 5)) | (ev.xy & 31u)];
            int irx, iry; float fx, fy;
            warp_axis(x, v.x, dt, irx, fx);
            warp_axis(y, v.y, dt, iry, fy);
            const int lx = irx - 1 - wn.ox, ly = iry - 1 - wn.oy;
            if ((unsigned)lx < (unsigned)(wn.ww - 2) && (unsigned)ly < (unsigned)(wn.wh - 2)) {
                f2v km, k0, kp;
                taps3x2(fx, fy, scy, km, k0, kp);
                const int off = __mul24(ly, wp) + lx;
                const uint32_t t0 = fix_u32(km.y, km.x), t1 = fix_u32(km.y, k0.x), t2 = fix_u32(km.y, kp.x);
                const uint32_t t3 = fix_u32(k0.y, km.x), t4 = fix_u32(k0.y, k0.x), t5 = fix_u32(k0.y, kp.x);
                const uint32_t t6 = fix_u32(kp.y, km.x), t7 = fix_u32(kp.y, k0.x), t8 = fix_u32(kp.y, kp.x);
                if (off == cur_off) {                // exact integer adds: the image is bit-identical to the unmerged one
                    m0 += t0; m1 += t1; m2 += t2; m3 += t3; m4 += t4; m5 += t5; m6 += t6; m7 += t7; m8 += t8;
                } else {
                    flush9();
                    cur_off = off;
                    m0 = t0; m1 = t1; m2 = t2; m3 = t3; m4 = t4; m5 = t5; m6 = t6; m7 = t7; m8 = t8;
                }
            } else {
                flush9();
                cur_off = -1;
                splat_ev(ev);                        // the direct path for taps outside the window
            }
        };
        const SegWalk<512> walk(n, tid);
        const int c = walk.half;
        const int K = c ? walk.K1 : walk.K0, rem = c ? walk.rem1 : walk.rem0;
        const uint32_t* __restrict__ px = exy + (c ? walk.n0 : 0) + walk.tt;
        const double* __restrict__ pt = et + (c ? walk.n0 : 0) + walk.tt;
#pragma unroll 2
        for (int j = 0; j < K - 1; ++j) { EvReg ev; ev.xy = px[j * 256]; ev.t = pt[j * 256]; merge_ev(ev); }
        if (K > 0 && walk.tt < rem) { EvReg ev; ev.xy = px[(K - 1) * 256]; ev.t = pt[(K - 1) * 256]; merge_ev(ev); }
        flush9();
    } else {
    EvReg A, B, C;
    load_ev(A, tid);
    load_ev(B, tid + NTH);
    C.xy = 0u; C.t = 0.0;
#ifndef EINCM_ABL_S_NOEVENTS           // EINCM_ABL_*: timing-only ablation builds (tools/build_variant.sh); results are wrong by design
    for (int j = 0; j < iters; j += 3) {
        step(A, B, C, j);
        if (j + 1 < iters) step(B, C, A, j + 1);
        if (j + 2 < iters) step(C, A, B, j + 2);
    }
#endif
    }
    if (!multi) __syncthreads();
#ifdef EINCM_ABL_S_NOFLUSH
    return;
#endif
    // flush (flat walk over the window: consecutive lanes, consecutive pixels of a row -> contiguous u64 atomics).  The segment's exact integer sums
    // (scale 2^fshift) are shifted to the accumulator's scale 2^ACC_SHIFT without rounding; integer adds commute, so the image
    // does not depend on the order in which the workgroups arrive.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int up = ACC_SHIFT - fshift;               // >= 0 (fix_shift caps at 30)
    auto to_acc = [&](int i) -> unsigned long long {
        if (multi) return (unsigned long long)fix64((double)ldsf[i] * 1073741824.0);     // f32 segment sums of the multi-chunk form
        return (unsigned long long)ldsu[i] << up;
    };
    if (wn.ox >= 0 && wn.oy >= 0 && wn.ox + wn.ww <= g.W && wn.oy + wn.wh <= g.H) {
        // the usual case, the window lies inside the image: no index rule per pixel
        unsigned long long* __restrict__ dst = img + (size_t)wn.oy * g.W + wn.ox;
        for (WinWalkT<NTH> w(threadIdx.x, wn.ww); w.i < nwin; w.next()) {
            const unsigned long long v = to_acc(w.row * wp + w.col);
            if (v != 0ull) atomicAdd(dst + w.row * g.W + w.col, v);
        }
        return;
    }
    for (int row = wv; row < wn.wh; row += NTH / 64) {
        const int gy = wrap_drop(wn.oy + row, g.H);
        if (gy < 0) continue;
        for (int col = lane; col < wn.ww; col += 64) {
            const unsigned long long v = to_acc(row * wp + col);
            if (v != 0ull) {
                const int gx = wrap_drop(wn.ox + col, g.W);
                if (gx >= 0) atomicAdd(img + (size_t)gy * g.W + gx, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Scharr 'same' true convolution, zero padded, difference-first (see oracle scharr_grads for why):
//   gx = 3(p[y+1,x+1]-p[y+1,x-1]) + 10(p[y,x+1]-p[y,x-1]) + 3(p[y-1,x+1]-p[y-1,x-1])
//   gy = 3(p[y+1,x+1]-p[y-1,x+1]) + 10(p[y+1,x]-p[y-1,x]) + 3(p[y+1,x-1]-p[y-1,x-1])
// T is an LDS tile accessor a(y, x) returning double.
// ------------------------------------------------------------------------------------------------
template <typename A> __device__ __forceinline__ void scharr_at(const A& a, int y, int x, double& gx, double& gy) {
    // no FMA contraction: the TV term counts gradients that are EXACTLY zero (regularizers.py:26-29); with
    // fma(3, a, round(3c)) the cancellation 3a + 3(-a) leaves the rounding error of 3a instead of 0.
#pragma clang fp contract(off)
    const double ul = a(y - 1, x - 1), uc = a(y - 1, x), ur = a(y - 1, x + 1);
    const double ml = a(y, x - 1), mr = a(y, x + 1);
    const double dl = a(y + 1, x - 1), dc = a(y + 1, x), dr = a(y + 1, x + 1);
    gx = 3.0 * (dr - dl) + 10.0 * (mr - ml) + 3.0 * (ur - ul);
    gy = 3.0 * (dr - ur) + 10.0 * (dc - uc) + 3.0 * (dl - ul);
}

// ------------------------------------------------------------------------------------------------
// k_stats: grid (ntiles, R, B).  img_of_r: stride between reference images (0 when one image serves all refs).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_stats(Geom g, const float* __restrict__ iwe, const float* __restrict__ edges,
                                               StatPart* __restrict__ parts, int with_contrast)
{
    constexpr int P = TS + 2, PP = P + 1;
    __shared__ float t[P][PP];
    __shared__ double red[NWAVE][8];
    const int tile = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    const float* __restrict__ I = iwe + ((size_t)b * g.R + r) * g.H * g.W;
    const float* __restrict__ E = edges + ((size_t)b * g.R + r) * g.H * g.W;
    if (!win_active(g, b)) return;
    // with_contrast = 0 (gradient evaluations): k_imgrad computes the Scharr images anyway and accumulates the contrast
    // energy itself, so this kernel is a pure streaming reduction (no LDS tile, no halo, no stencil).
    if (with_contrast) {
        for (int p = threadIdx.x; p < P * P; p += NT) {
            const int ly = p / P, lx = p % P;
            const int y = y0 + ly - 1, x = x0 + lx - 1;
            t[ly][lx] = (y >= 0 && y < g.H && x >= 0 && x < g.W) ? I[(size_t)y * g.W + x] : 0.0f;
        }
        __syncthreads();
    }
    auto acc = [&](int y, int x) -> double { return (double)t[y][x]; };
    double mn = INFINITY, mx = -INFINITY, cmn = 0.0, cmx = 0.0, sI = 0.0, sII = 0.0, sEI = 0.0, sG2 = 0.0;
    for (int p = threadIdx.x; p < TS * TS; p += NT) {
        const int ly = p / TS, lx = p % TS;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= g.H || x >= g.W) continue;
        const double v = with_contrast ? acc(ly + 1, lx + 1) : (double)I[(size_t)y * g.W + x];
        const double e = (double)E[(size_t)y * g.W + x];
        if (v < mn) { mn = v; cmn = 1.0; } else if (v == mn) cmn += 1.0;
        if (v > mx) { mx = v; cmx = 1.0; } else if (v == mx) cmx += 1.0;
        sI += v; sII += v * v; sEI += e * v;
        if (with_contrast) {
            double gx, gy;
            scharr_at(acc, ly + 1, lx + 1, gx, gy);
            sG2 += gx * gx + gy * gy;
        }
    }
    // (min,count) / (max,count) pairs combine associatively
    const double wmn = wave_min(mn), wmx = wave_max(mx);
    const double bmn = __shfl(wmn, 0, 64), bmx = __shfl(wmx, 0, 64);
    cmn = wave_sum(mn == bmn ? cmn : 0.0);
    cmx = wave_sum(mx == bmx ? cmx : 0.0);
    sI = wave_sum(sI); sII = wave_sum(sII); sEI = wave_sum(sEI); sG2 = wave_sum(sG2);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[wv][0] = bmn; red[wv][1] = bmx; red[wv][2] = cmn; red[wv][3] = cmx;
        red[wv][4] = sI; red[wv][5] = sII; red[wv][6] = sEI; red[wv][7] = sG2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        StatPart o;
        o.mn = red[0][0]; o.mx = red[0][1]; o.cmn = red[0][2]; o.cmx = red[0][3];
        o.sI = red[0][4]; o.sII = red[0][5]; o.sEI = red[0][6]; o.sG2 = red[0][7];
        for (int i = 1; i < NWAVE; ++i) {
            if (red[i][0] < o.mn) { o.mn = red[i][0]; o.cmn = red[i][2]; } else if (red[i][0] == o.mn) o.cmn += red[i][2];
            if (red[i][1] > o.mx) { o.mx = red[i][1]; o.cmx = red[i][3]; } else if (red[i][1] == o.mx) o.cmx += red[i][3];
            o.sI += red[i][4]; o.sII += red[i][5]; o.sEI += red[i][6]; o.sG2 += red[i][7];
        }
        parts[((size_t)b * g.R + r) * g.pstride + tile] = o;
    }
}

// k_iwe_finish: the u64 accumulator becomes the fp32 IWE stack, and is cleared for the next evaluation (consumer-clears).
// Used on the paths whose statistics kernel is the tiled k_stats (forward-only evaluations); gradient evaluations do the same
// inside k_stats_stream.  grid-stride over (B,R,H,W).
constexpr double ACC_INV = 1.0 / 1073741824.0;     // 2^-ACC_SHIFT
__global__ __launch_bounds__(NT) void k_iwe_finish(Geom g, unsigned long long* __restrict__ acc, float* __restrict__ iwe)
{
    const size_t n = (size_t)g.B * g.R * g.H * g.W;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const unsigned long long a = acc[i];
        iwe[i] = (float)((double)a * ACC_INV);
        if (a != 0ull) acc[i] = 0ull;
    }
}

// k_stats_stream: the same partials as k_stats without the contrast energy, as a pure streaming reduction.  The partials have
// no per-tile meaning (they are only ever reduced over the whole image), so NSPART fat blocks per image read the image with
// coalesced grid-stride loads and pay the fp64 cross-lane reduction once each.  grid (NSPART, R, B).
// It is also the consumer of the u64 accumulator: converts it to the fp32 IWE stack (what every later kernel reads) and clears it.
constexpr int NSPART = 32;
__global__ __launch_bounds__(NT) void k_stats_stream(Geom g, unsigned long long* __restrict__ acc, float* __restrict__ iwe,
                                                      const float* __restrict__ edges, StatPart* __restrict__ parts)
{
    __shared__ double red[NWAVE][8];
    const int part = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    const size_t n = (size_t)g.H * g.W;
    if (!win_active(g, b)) return;
    unsigned long long* __restrict__ A = acc + ((size_t)b * g.R + r) * n;
    float* __restrict__ I = iwe + ((size_t)b * g.R + r) * n;
    const float* __restrict__ E = edges + ((size_t)b * g.R + r) * n;
    double mn = INFINITY, mx = -INFINITY, cmn = 0.0, cmx = 0.0, sI = 0.0, sII = 0.0, sEI = 0.0;
    auto take = [&](float fv, float fe) {          // branch-free: (min, #ties) and (max, #ties) are tracked with selects
        const double v = (double)fv, e = (double)fe;
        cmn = (v < mn) ? 1.0 : cmn + (v == mn ? 1.0 : 0.0);
        cmx = (v > mx) ? 1.0 : cmx + (v == mx ? 1.0 : 0.0);
        mn = fmin(mn, v); mx = fmax(mx, v);
        sI += v; sII += v * v; sEI += e * v;
    };
    auto conv = [&](int i, unsigned long long a) -> float {  // exact u64 sum -> fp32 pixel (one rounding), stored for the later kernels; clear
        const float v = (float)((double)a * ACC_INV);
        I[i] = v;
        if (a != 0ull) A[i] = 0ull;
        return v;
    };
    // four independent load pairs in flight per trip: with one pair per trip the loop paid a full memory latency 11 times
    const int npx = (int)n, stride = NSPART * NT;
    int i = part * NT + (int)threadIdx.x;
    for (; i + 3 * stride < npx; i += 4 * stride) {
        const unsigned long long a0 = A[i], a1 = A[i + stride], a2 = A[i + 2 * stride], a3 = A[i + 3 * stride];
        const float e0 = E[i], e1 = E[i + stride], e2 = E[i + 2 * stride], e3 = E[i + 3 * stride];
        take(conv(i, a0), e0); take(conv(i + stride, a1), e1); take(conv(i + 2 * stride, a2), e2); take(conv(i + 3 * stride, a3), e3);
    }
    if (i < npx) {      // the last <= 3 pixels of this thread: requested together as well (one at a time they cost a memory latency each,
                        // a third of the kernel at 11 pixels per thread); same order of the sums
        const int i1 = i + stride, i2 = i + 2 * stride;
        const bool h1 = i1 < npx, h2 = i2 < npx;
        const unsigned long long a0 = A[i], a1 = h1 ? A[i1] : 0ull, a2 = h2 ? A[i2] : 0ull;
        const float e0 = E[i], e1 = h1 ? E[i1] : 0.0f, e2 = h2 ? E[i2] : 0.0f;
        take(conv(i, a0), e0);
        if (h1) take(conv(i1, a1), e1);
        if (h2) take(conv(i2, a2), e2);
    }
    const double wmn = wave_min(mn), wmx = wave_max(mx);
    const double bmn = __shfl(wmn, 0, 64), bmx = __shfl(wmx, 0, 64);
    cmn = wave_sum(mn == bmn ? cmn : 0.0);
    cmx = wave_sum(mx == bmx ? cmx : 0.0);
    sI = wave_sum(sI); sII = wave_sum(sII); sEI = wave_sum(sEI);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = bmn; red[wv][1] = bmx; red[wv][2] = cmn; red[wv][3] = cmx; red[wv][4] = sI; red[wv][5] = sII; red[wv][6] = sEI; }
    __syncthreads();
    if (threadIdx.x == 0) {
        StatPart o;
        o.mn = red[0][0]; o.mx = red[0][1]; o.cmn = red[0][2]; o.cmx = red[0][3]; o.sI = red[0][4]; o.sII = red[0][5]; o.sEI = red[0][6]; o.sG2 = 0.0;
        for (int i = 1; i < NWAVE; ++i) {
            if (red[i][0] < o.mn) { o.mn = red[i][0]; o.cmn = red[i][2]; } else if (red[i][0] == o.mn) o.cmn += red[i][2];
            if (red[i][1] > o.mx) { o.mx = red[i][1]; o.cmx = red[i][3]; } else if (red[i][1] == o.mx) o.cmx += red[i][3];
            o.sI += red[i][4]; o.sII += red[i][5]; o.sEI += red[i][6];
        }
        parts[((size_t)b * g.R + r) * g.pstride + part] = o;     // slots [0, NSPART) of the image's row of pstride = max(ntiles, NSPART) slots
    }
}

// Reduce the ntiles StatParts of image (b, r).  Must be called by a full wave (64 lanes); result in all lanes.
__device__ __forceinline__ ImgScal reduce_parts(const StatPart* __restrict__ parts, int ntiles) {
    const int lane = threadIdx.x & 63;
    double mn = INFINITY, mx = -INFINITY, cmn = 0.0, cmx = 0.0, sI = 0.0, sII = 0.0, sEI = 0.0, sG2 = 0.0;
    for (int i = lane; i < ntiles; i += 64) {
        const StatPart p = parts[i];
        if (p.mn < mn) { mn = p.mn; cmn = p.cmn; } else if (p.mn == mn) cmn += p.cmn;
        if (p.mx > mx) { mx = p.mx; cmx = p.cmx; } else if (p.mx == mx) cmx += p.cmx;
        sI += p.sI; sII += p.sII; sEI += p.sEI; sG2 += p.sG2;
    }
    ImgScal s;
    double wmn = wave_min(mn), wmx = wave_max(mx);
    s.m = __shfl(wmn, 0, 64); s.M = __shfl(wmx, 0, 64);
    s.cm = __shfl(wave_sum(mn == s.m ? cmn : 0.0), 0, 64);
    s.cM = __shfl(wave_sum(mx == s.M ? cmx : 0.0), 0, 64);
    s.sI = __shfl(wave_sum(sI), 0, 64); s.sII = __shfl(wave_sum(sII), 0, 64);
    s.sEI = __shfl(wave_sum(sEI), 0, 64); s.sG2 = __shfl(wave_sum(sG2), 0, 64);
    s.D = s.M - s.m + EPSN;                                  // img_utils.py:25
    return s;
}

// mean((E - n)^2) with n = (I - m)/D from the moments (correlation_objectives.py:25-26 on img_utils.py:24-25)
__host__ __device__ __forceinline__ double mse_from_moments(const ImgScal& s, double sE, double sEE, double HW) {
    const double a = s.m / s.D;
    return (sEE + HW * a * a + s.sII / (s.D * s.D) + 2.0 * a * sE - 2.0 * s.sEI / s.D - 2.0 * a * s.sI / s.D) / HW;
}

// ------------------------------------------------------------------------------------------------
// k_imgrad: G = dL/dIWE.  grid (ceil(nig / 4), R, B), 4 waves per workgroup, one strip of IG_COLS x IG_ROWS pixels per wave.
//   contrast (grad-mag): a_r * (2/HW) * (adj_Sx(gx) + adj_Sy(gy)),  adj_S(c) = -conv_same(c, S) for Scharr
//   contrast (variance): a_r * (2/HW) * (I - mean I)
//   correlation:         Gn/D + dm*[I==m]/#min + dM*[I==M]/#max,  Gn = b_r*(2/HW)*(E - n)
// The two stacked 3x3 stencils run as a sliding window over rows held in registers: a lane owns one image column, its
// horizontal neighbours come from the adjacent lanes by DPP wave shifts (no LDS), its vertical neighbours from the
// previous iterations.  Lanes 0,1,62,63 and rows -2,-1,+1,+2 of a strip are halo (5x5 support of the stacked stencils).
// ------------------------------------------------------------------------------------------------
constexpr int IG_ROWS = 12, IG_COLS = 60, IG_NT = 256;      // rows: 16.5 / 15.7 / 16.0 us with 16 / 12 / 8 on the 8-window batch, 8.2 / 7.2 / 6.7 on one window (k_final pays for more strips)

template <int CTRL> __device__ __forceinline__ float dpp_lane(float v) {       // zero where the source lane does not exist
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_m1(float v) { return dpp_lane<0x138>(v); }   // wave_shr:1 -> value of lane - 1
__device__ __forceinline__ float lane_p1(float v) { return dpp_lane<0x130>(v); }   // wave_shl:1 -> value of lane + 1

__global__ __launch_bounds__(IG_NT) void k_imgrad(Geom g, EvalParams ep,
        const float* __restrict__ iwe, const float* __restrict__ edges,
        const StatPart* __restrict__ parts, const WinConst* __restrict__ wc,
        const float* __restrict__ gdiv, const double* __restrict__ dgparts,    // delta != 0 only (else unused)
        double* __restrict__ g2parts,          // (B,R,nig): this strip's sum of gx^2+gy^2 (contrast energy), a by-product
        float* __restrict__ G,
        unsigned* __restrict__ gmax,           // (B,R,nig): this strip's max |G| as float bits: fixes the fixed-point scale of the
                                               // gradient accumulators (gmax_of, grad_shift)
        double* __restrict__ imgscal_out,      // (B,R,IMGSCAL_N) or nullptr: the reduced image scalars, written once per image (in pinned host
                                               // memory on the path whose scalar assembly runs on the host: host_assemble)
        int g2_per_wg)                         // g2parts holds one partial per workgroup, (B,R,gridDim.x), instead of one per strip
{
    __shared__ double sc[10];
    const bool use_div = (ep.delta != 0.0);
    const int r = blockIdx.y, b = blockIdx.z, lane = threadIdx.x & 63;
    if (!win_active(g, b)) return;
    const int strip = __builtin_amdgcn_readfirstlane(blockIdx.x * (IG_NT / 64) + (threadIdx.x >> 6));   // wave-uniform: row tests stay scalar
    const double HW = (double)g.H * (double)g.W;
    const size_t img = ((size_t)b * g.R + r) * g.H * g.W;
    const float* __restrict__ I = iwe + img;
    const float* __restrict__ E = edges + img;
    float* __restrict__ Go = G + img;
    const WinConst& c = wc[b];
    const int cx0 = (strip % g.igx) * IG_COLS, cy0 = (strip / g.igx) * IG_ROWS;
    const int x = cx0 - 2 + lane;
    // every row of the strip is requested up front from clamped (always valid) addresses, so that the loads are unconditional and
    // all in flight together (and behind the scalar prologue); padding is applied by selects afterwards
    const int xc = min(max(x, 0), g.W - 1);
    float trow[IG_ROWS + 4], erow[IG_ROWS];
#pragma unroll
    for (int k = 0; k < IG_ROWS + 4; ++k) trow[k] = I[(size_t)min(max(cy0 - 2 + k, 0), g.H - 1) * g.W + xc];
#pragma unroll
    for (int k = 0; k < IG_ROWS; ++k) erow[k] = E[(size_t)min(cy0 + k, g.H - 1) * g.W + xc];

    if (threadIdx.x < 64) {
        const ImgScal s = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
        double dAn = 0.0, dA = 0.0;
        if (use_div) {
            const double* dp = dgparts + ((size_t)b * g.R + r) * g.ntiles * 2;
            for (int i = threadIdx.x; i < g.ntiles; i += 64) { dAn += dp[2 * i]; dA += dp[2 * i + 1]; }
            dAn = __shfl(wave_sum(dAn), 0, 64); dA = __shfl(wave_sum(dA), 0, 64);
        }
        if (threadIdx.x == 0 && blockIdx.x == 0 && imgscal_out) {
            double* o = imgscal_out + ((size_t)b * g.R + r) * IMGSCAL_N;
            o[0] = s.m; o[1] = s.M; o[2] = s.D; o[3] = s.cm; o[4] = s.cM; o[5] = s.sI; o[6] = s.sII; o[7] = s.sEI;
        }
        if (threadIdx.x == 0) {
            const double c0 = (ep.contrast_kind == 1) ? c.c0_var : c.c0_gradmag;
            const double a_r = -ep.alpha * c.mrw[r] / ((double)g.R * (c0 + EPSN));
            const double b_r = -ep.beta * c.mrw[r] / ((double)g.R * (c.zc[r] + EPSN));
            const double a = s.m / s.D;
            const double S_n = s.sI / s.D - HW * a;
            const double S_En = s.sEI / s.D - a * c.sE[r];
            const double S_nn = s.sII / (s.D * s.D) - 2.0 * a * s.sI / s.D + HW * a * a;
            const double k = b_r * 2.0 / HW;
            const double e_hw = use_div ? ep.delta * c.mrw[r] / ((double)g.R * (c.d0 + EPSN) * HW) : 0.0;
            const double sGn_n = k * (S_En - S_nn) + e_hw * dAn;     // sum Gn*n
            const double sGn = k * (c.sE[r] - S_n) + e_hw * dA;      // sum Gn
            sc[8] = e_hw / s.D;                              // scale on the divergence adjoint image
            sc[0] = s.m; sc[1] = s.M; sc[2] = s.D;
            sc[3] = a_r * 2.0 / HW;                          // contrast scale
            sc[4] = k / s.D;                                 // Gn/D scale on (E - n)
            sc[5] = ((sGn_n - sGn) / s.D) / s.cm;            // dm / #argmin
            sc[6] = (-sGn_n / s.D) / s.cM;                   // dM / #argmax
            sc[7] = s.sI / HW;                               // mean I
            sc[9] = 1.0 / s.D;                               // n = (I - m) * (1/D): one fp64 division per image, not per pixel (<= 1 ulp)
        }
    }
    __syncthreads();
    const bool live = strip < g.nig;                         // wave-uniform; a dead wave (last workgroup of an image) runs along on clamped
                                                             // addresses and owns nothing, so that the workgroup can meet at the end
    const bool col_in = (x >= 0 && x < g.W);
    const bool own = live && col_in && lane >= 2 && lane < 2 + IG_COLS;
    const int yend = min(cy0 + IG_ROWS, g.H);                // own rows [cy0, yend)
    const bool gradmag = (ep.contrast_kind == 0);
    const double m = sc[0], M = sc[1], invD = sc[9], k_c = sc[3], k_n = sc[4], k_m = sc[5], k_M = sc[6], meanI = sc[7], k_d = sc[8];

    // rolling state at iteration i (image row i is loaded): tA,tB = I rows i-2,i-1; dA,dB = horizontal differences of rows i-2,i-1;
    // gyA,gyB = gy rows i-3,i-2; qA,qB = horizontal differences of gx rows i-3,i-2.  Row i-1 of (gx,gy) and row i-2 of G come out.
    float tA = 0.f, tB = 0.f, dA = 0.f, dB = 0.f, gyA = 0.f, gyB = 0.f, qA = 0.f, qB = 0.f;
    float g2f = 0.f;
    unsigned gm = 0u;                                        // max |G| as bits (non-negative floats order like their patterns; NaN > Inf)
#pragma unroll
    for (int k = 0; k < IG_ROWS + 4; ++k) {
        const int i = cy0 - 2 + k;
        const float tC = (i >= 0 && i < g.H && col_in) ? trow[k] : 0.0f;     // zero padding (losses.py Scharr 'same')
        float ax = 0.f, ay = 0.f, dC = 0.f, gy = 0.f, qC = 0.f;
        if (gradmag) {                                       // uniform: the DPP moves below always run with every lane enabled
            dC = lane_p1(tC) - lane_m1(tC);
            float gx = 3.0f * dC + 10.0f * dB + 3.0f * dA;
            const float e = tC - tA;
            gy = 3.0f * lane_p1(e) + 10.0f * e + 3.0f * lane_m1(e);
            const bool in1 = (i - 1 >= 0 && i - 1 < g.H) && col_in;          // (gx, gy) are zero outside the image
            gx = in1 ? gx : 0.0f; gy = in1 ? gy : 0.0f;
            const bool own1 = own && (i - 1 >= cy0 && i - 1 < yend);
            g2f += own1 ? (gx * gx + gy * gy) : 0.0f;
            // adj_Sx(gx) = -conv(gx, Sx), adj_Sy(gy) = -conv(gy, Sy)
            qC = lane_p1(gx) - lane_m1(gx);
            ax = 3.0f * qC + 10.0f * qB + 3.0f * qA;
            const float eg = gy - gyA;
            ay = 3.0f * lane_p1(eg) + 10.0f * eg + 3.0f * lane_m1(eg);
        }
        const int o = i - 2;
        if (o >= cy0 && o < yend && own) {
            const double v = (double)tA;
            const double dc = gradmag ? -(double)(ax + ay) : v - meanI;
            const double n = (v - m) * invD;
            const double e = (double)erow[k >= 4 ? k - 4 : 0];
            double gv = k_c * dc + k_n * (e - n);
            if (use_div) gv += k_d * (double)gdiv[img + (size_t)o * g.W + x];
            if (v == m) gv += k_m;
            if (v == M) gv += k_M;
            const float gf = (float)gv;
            Go[(size_t)o * g.W + x] = gf;
            gm = max(gm, __float_as_uint(gf) & 0x7fffffffu);
        }
        tA = tB; tB = tC; dA = dB; dB = dC; gyA = gyB; gyB = gy; qA = qB; qB = qC;
    }
    const double g2 = wave_sum((double)g2f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = max(gm, (unsigned)__shfl_down((int)gm, o, 64));
    if (lane == 0 && live) {
        if (!g2_per_wg) g2parts[((size_t)b * g.R + r) * g.nig + strip] = g2;
        gmax[((size_t)b * g.R + r) * g.nig + strip] = gm;
    }
    if (g2_per_wg) {            // host-assembled evaluations: one contrast-energy partial per workgroup (a quarter of the bytes the host reads back)
        __shared__ double g2w[IG_NT / 64];
        if (lane == 0) g2w[threadIdx.x >> 6] = live ? g2 : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = g2w[0];
            for (int i = 1; i < IG_NT / 64; ++i) t += g2w[i];
            g2parts[((size_t)b * g.R + r) * gridDim.x + blockIdx.x] = t;
        }
    }
}

// The per-image scalars G is composed from (see k_imstat), in the precision the composition runs in (fp32 like the image itself;
// the fp64 originals are the formulas of k_imgrad).  gradmag = 0: the variance contrast, dc = I - mean I.
struct GCoef { float m, M, invD, k_c, k_n, k_m, k_M, meanI; };
struct ImgCoef { GCoef q; float bound; float pad_[7]; };      // one 64-byte record per image: what the gather loads (uniformly: scalar registers)
__device__ __forceinline__ GCoef gcoef_from(const ImgScal& s, const WinConst& c, const EvalParams& ep, int r, int R, double HW) {
    // Three divisions (1/D and the two tie counts): the window constants come as reciprocals (WinConst.inv_*), everything else is
    // multiplied by the one 1/D.  A division costs ~10 instructions and ~8 live registers; a dozen of them interleaved took the
    // composing gather from 66 to 106 VGPRs (7 -> 4 waves per SIMD).  The results are rounded to fp32 anyway.
    const double iD = 1.0 / s.D;
    const double a_r = -ep.alpha * c.mrw[r] * ((ep.contrast_kind == 1) ? c.inv_c0_var : c.inv_c0_gradmag) / (double)R;
    const double k = -ep.beta * c.mrw[r] * c.inv_zc[r] * (2.0 / ((double)R * HW));
    const double a = s.m * iD;
    const double S_n = s.sI * iD - HW * a;
    const double S_En = s.sEI * iD - a * c.sE[r];
    const double S_nn = (s.sII * iD - 2.0 * a * s.sI) * iD + HW * a * a;
    const double sGn_n = k * (S_En - S_nn);                  // sum Gn*n
    const double sGn = k * (c.sE[r] - S_n);                  // sum Gn
    GCoef q;
    q.m = (float)s.m; q.M = (float)s.M;                      // exact: extrema of fp32 pixels
    q.invD = (float)iD;
    q.k_c = (float)(a_r * (2.0 / HW));
    q.k_n = (float)(k * iD);
    q.k_m = (float)(((sGn_n - sGn) * iD) / s.cm);            // dm / #argmin
    q.k_M = (float)((-sGn_n * iD) / s.cM);                   // dM / #argmax
    q.meanI = (float)(s.sI * (1.0 / HW));
    return q;
}
// An upper bound of max |G| over one image from its scalars: |dc| <= max|A| (grad-mag) or D (variance), |E - n| <= max|E| + 1.
// The fixed-point scale of the gradient accumulators needs SOME bound of what is added (grad_shift_pixel / grad_shift); the tie terms
// k_m, k_M dominate it by orders of magnitude, so this one costs at most a bit against the exact maximum k_imgrad used to measure.
__device__ __forceinline__ double gbound_from(const GCoef& q, const ImgScal& s, bool gradmag, double amax, double emax) {
    return 1.0001 * (fabs((double)q.k_c) * (gradmag ? amax : s.D) + fabs((double)q.k_n) * (emax + 1.0) + fabs((double)q.k_m) + fabs((double)q.k_M));
}
// G = dL/dIWE at one pixel from (A, E, I): 3 loads + 8 fp32 operations; every gradient evaluation of the composed path and
// eincm_get_image_grad (k_compose) use this one function, so they see the same image.
__device__ __forceinline__ float compose_G(const GCoef& q, bool gradmag, float a, float e, float v) {
    const float dc = gradmag ? a : v - q.meanI;
    const float n = (v - q.m) * q.invD;
    float gv = fmaf(q.k_c, dc, q.k_n * (e - n));
    gv += (v == q.m) ? q.k_m : 0.0f;
    gv += (v == q.M) ? q.k_M : 0.0f;
    return gv;
}

// ------------------------------------------------------------------------------------------------
// k_imstat: the image pass of a gradient evaluation in ONE kernel (round 3): consumer of the u64 accumulator (exact sum -> fp32 IWE,
// one rounding), image statistics (min / max with tie counts, sum I, sum I^2, sum E I) and the stats-INDEPENDENT part of dL/dIWE:
//   A = adj_Sx(gx) + adj_Sy(gy) = -(conv(gx, Sx) + conv(gy, Sy)),  (gx, gy) = Scharr(I)      [reverse of contrast_objectives.py:22-25]
// with the contrast energy sum(gx^2 + gy^2) and max |A| as by-products.  What depends on the statistics,
//   G = k_c A + k_n (E - n) + k_m [I == m] + k_M [I == M],   n = (I - m) / D                  [reverse of img_utils.py:24-25, losses.py:62-67]
// is linear in per-image scalars, so the gather composes G while it stages its window (compose_G) and no kernel has to wait for the
// statistics in between: splat -> imstat -> gather instead of splat -> stats -> imgrad -> gather (a dependent kernel costs 6-8 us
// here whatever its work: profiles/r03/latency_chain.md).  Same register sliding window as k_imgrad (a lane owns a column, DPP
// neighbours, no LDS).  The accumulator is NOT cleared here (a neighbour strip still reads its halo from it): the gather does that.
// grid (ceil(nig / 4), R, B); one StatPart per workgroup (its <= 4 strips combined), sG2 = contrast energy.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(IG_NT) void k_imstat(Geom g, int gradmag,
        const unsigned long long* __restrict__ acc, const float* __restrict__ edges,
        float* __restrict__ iwe, float* __restrict__ A,
        StatPart* __restrict__ parts,          // (B,R,pstride): slot blockIdx.x
        unsigned* __restrict__ amax,           // (B,R,pstride): max |A| of the workgroup's strips as float bits
        // the tail: the workgroup of an image that arrives LAST reduces the image's partials and derives what the gather composes
        // dL/dIWE with (one ImgCoef per image, instead of every gather workgroup doing that in its prologue: there it cost the gather
        // 40 VGPRs, 7 -> 4 waves per SIMD)
        unsigned* __restrict__ ticket,         // (B,R) arrival counters, zero between launches (the last arriver resets its own)
        EvalParams ep, const WinConst* __restrict__ wc,
        ImgCoef* __restrict__ coef,            // (B,R)
        unsigned* __restrict__ gbound,         // (B,R) bound of max |dL/dIWE| of the image as float bits (the gmax of a composed evaluation)
        double* __restrict__ imgscal_out)      // (B,R,IMGSCAL_N) or nullptr: the reduced image scalars for the host (pinned memory)
{
    __shared__ double red[IG_NT / 64][8];
    __shared__ unsigned redm[IG_NT / 64];
    __shared__ int s_last;
    const int r = blockIdx.y, b = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int strip = __builtin_amdgcn_readfirstlane(blockIdx.x * (IG_NT / 64) + wv);   // wave-uniform: row tests stay scalar
    if (!win_active(g, b)) return;
    const size_t img = ((size_t)b * g.R + r) * g.H * g.W;
    const unsigned long long* __restrict__ Ac = acc + img;
    const float* __restrict__ E = edges + img;
    float* __restrict__ Io = iwe + img;
    float* __restrict__ Ao = A + img;
    const bool live = strip < g.nig;                          // wave-uniform; dead waves run the loop on clamped addresses and contribute nothing
    const int sidx = live ? strip : 0;
    const int cx0 = (sidx % g.igx) * IG_COLS, cy0 = (sidx / g.igx) * IG_ROWS;
    const int x = cx0 - 2 + lane;
    const int xc = min(max(x, 0), g.W - 1);
    unsigned long long arow[IG_ROWS + 4];
    float erow[IG_ROWS];
#pragma unroll
    for (int k = 0; k < IG_ROWS + 4; ++k) arow[k] = Ac[(size_t)min(max(cy0 - 2 + k, 0), g.H - 1) * g.W + xc];
#pragma unroll
    for (int k = 0; k < IG_ROWS; ++k) erow[k] = E[(size_t)min(cy0 + k, g.H - 1) * g.W + xc];
    const bool col_in = (x >= 0 && x < g.W);
    const bool own = live && col_in && lane >= 2 && lane < 2 + IG_COLS;
    const int yend = min(cy0 + IG_ROWS, g.H);                // own rows [cy0, yend)

    float tA = 0.f, tB = 0.f, dA = 0.f, dB = 0.f, gyA = 0.f, gyB = 0.f, qA = 0.f, qB = 0.f;
    float g2f = 0.f;
    unsigned am = 0u;
    double mn = INFINITY, mx = -INFINITY, cmn = 0.0, cmx = 0.0, sI = 0.0, sII = 0.0, sEI = 0.0;
#pragma unroll
    for (int k = 0; k < IG_ROWS + 4; ++k) {
        const int i = cy0 - 2 + k;
        const float tv = (float)((double)arow[k] * ACC_INV);                  // exact u64 sum -> fp32 pixel (one rounding), as k_stats_stream
        const float tC = (i >= 0 && i < g.H && col_in) ? tv : 0.0f;           // zero padding (Scharr 'same')
        float ax = 0.f, ay = 0.f, dC = 0.f, gy = 0.f, qC = 0.f;
        if (gradmag) {                                       // uniform: the DPP moves below always run with every lane enabled
            dC = lane_p1(tC) - lane_m1(tC);
            float gx = 3.0f * dC + 10.0f * dB + 3.0f * dA;
            const float e = tC - tA;
            gy = 3.0f * lane_p1(e) + 10.0f * e + 3.0f * lane_m1(e);
            const bool in1 = (i - 1 >= 0 && i - 1 < g.H) && col_in;
            gx = in1 ? gx : 0.0f; gy = in1 ? gy : 0.0f;
            const bool own1 = own && (i - 1 >= cy0 && i - 1 < yend);
            g2f += own1 ? (gx * gx + gy * gy) : 0.0f;
            qC = lane_p1(gx) - lane_m1(gx);
            ax = 3.0f * qC + 10.0f * qB + 3.0f * qA;
            const float eg = gy - gyA;
            ay = 3.0f * lane_p1(eg) + 10.0f * eg + 3.0f * lane_m1(eg);
        }
        const int o = i - 2;
        if (o >= cy0 && o < yend && own) {
            const float af = -(ax + ay);
            Io[(size_t)o * g.W + x] = tA;
            if (gradmag) { Ao[(size_t)o * g.W + x] = af; am = max(am, __float_as_uint(af) & 0x7fffffffu); }
            const double v = (double)tA, e = (double)erow[k >= 4 ? k - 4 : 0];
            cmn = (v < mn) ? 1.0 : cmn + (v == mn ? 1.0 : 0.0);
            cmx = (v > mx) ? 1.0 : cmx + (v == mx ? 1.0 : 0.0);
            mn = fmin(mn, v); mx = fmax(mx, v);
            sI += v; sII += v * v; sEI += e * v;
        }
        tA = tB; tB = tC; dA = dB; dB = dC; gyA = gyB; gyB = gy; qA = qB; qB = qC;
    }
    // wave, then workgroup: (min, #ties) and (max, #ties) combine associatively; sums in the fixed order of the trees
    const double wmn = wave_min(mn), wmx = wave_max(mx);
    const double bmn = __shfl(wmn, 0, 64), bmx = __shfl(wmx, 0, 64);
    cmn = wave_sum(mn == bmn ? cmn : 0.0);
    cmx = wave_sum(mx == bmx ? cmx : 0.0);
    sI = wave_sum(sI); sII = wave_sum(sII); sEI = wave_sum(sEI);
    const double g2 = wave_sum((double)g2f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_down((int)am, o, 64));
    if (lane == 0) {
        red[wv][0] = bmn; red[wv][1] = bmx; red[wv][2] = cmn; red[wv][3] = cmx; red[wv][4] = sI; red[wv][5] = sII; red[wv][6] = sEI; red[wv][7] = g2;
        redm[wv] = am;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        StatPart o;
        o.mn = red[0][0]; o.mx = red[0][1]; o.cmn = red[0][2]; o.cmx = red[0][3]; o.sI = red[0][4]; o.sII = red[0][5]; o.sEI = red[0][6]; o.sG2 = red[0][7];
        unsigned m = redm[0];
        for (int i = 1; i < IG_NT / 64; ++i) {
            if (red[i][0] < o.mn) { o.mn = red[i][0]; o.cmn = red[i][2]; } else if (red[i][0] == o.mn) o.cmn += red[i][2];
            if (red[i][1] > o.mx) { o.mx = red[i][1]; o.cmx = red[i][3]; } else if (red[i][1] == o.mx) o.cmx += red[i][3];
            o.sI += red[i][4]; o.sII += red[i][5]; o.sEI += red[i][6]; o.sG2 += red[i][7];
            m = max(m, redm[i]);
        }
        // publish: write-through (agent-scope) stores by this ONE lane, drained, then the ticket (cdna_hip_programming.md Guideline 16, R1)
        const size_t slot = ((size_t)b * g.R + r) * g.pstride + blockIdx.x;
        double* po = reinterpret_cast<double*>(parts + slot);
        const double ov[8] = {o.mn, o.mx, o.cmn, o.cmx, o.sI, o.sII, o.sEI, o.sG2};
#pragma unroll
        for (int i = 0; i < 8; ++i) __hip_atomic_store(po + i, ov[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(amax + slot, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(ticket + b * g.R + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == gridDim.x - 1u) ? 1 : 0;
        if (old == gridDim.x - 1u) __hip_atomic_store(ticket + b * g.R + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;                                     // uniform
    if (threadIdx.x < 64) {                                  // one wave: acquire (this CU's L1 may hold stale lines of the partials), then reduce
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const size_t slot0 = ((size_t)b * g.R + r) * g.pstride;
        const ImgScal s = reduce_parts(parts + slot0, g.nparts);
        unsigned am2 = 0u;
        for (int i = lane; i < g.nparts; i += 64) am2 = max(am2, amax[slot0 + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am2 = max(am2, (unsigned)__shfl_xor((int)am2, o, 64));
        if (lane == 0) {
            const WinConst& c = wc[b];
            ImgCoef ic;
            ic.q = gcoef_from(s, c, ep, r, g.R, (double)g.H * (double)g.W);
            const float bound = (float)gbound_from(ic.q, s, gradmag != 0, (double)__uint_as_float(am2), c.eabs[r]);
            ic.bound = bound;
            coef[b * g.R + r] = ic;
            gbound[b * g.R + r] = __float_as_uint(bound);
            if (imgscal_out) {
                double* o = imgscal_out + ((size_t)b * g.R + r) * IMGSCAL_N;
                o[0] = s.m; o[1] = s.M; o[2] = s.D; o[3] = s.cm; o[4] = s.cM; o[5] = s.sI; o[6] = s.sII; o[7] = s.sEI; o[8] = s.sG2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_div: IWE divergence of the normalised IWE (event_collapse_objectives.py:8-20), forward only.
//   d = mean | K (*) (n (*) Sx) + K (*) (n (*) Sy) | = mean | K (*) (gx_n + gy_n) |, every stage zero padded.
// grid (ntiles, R, B); writes one partial sum per tile.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_div(Geom g, const float* __restrict__ iwe, const StatPart* __restrict__ parts,
                                             double* __restrict__ divparts)
{
    constexpr int P2 = TS + 4, P1 = TS + 2;
    __shared__ double t[P2][P2 + 1];
    __shared__ double s1[P1][P1 + 1];
    __shared__ double sc[2];
    __shared__ double scratch[NWAVE];
    const int tile = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    const float* __restrict__ I = iwe + ((size_t)b * g.R + r) * g.H * g.W;
    if (!win_active(g, b)) return;
    if (threadIdx.x < 64) {
        const ImgScal s = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
        if (threadIdx.x == 0) { sc[0] = s.m; sc[1] = s.D; }
    }
    __syncthreads();
    const double m = sc[0], D = sc[1];
    for (int p = threadIdx.x; p < P2 * P2; p += NT) {
        const int ly = p / P2, lx = p % P2;
        const int y = y0 + ly - 2, x = x0 + lx - 2;
        t[ly][lx] = (y >= 0 && y < g.H && x >= 0 && x < g.W) ? ((double)I[(size_t)y * g.W + x] - m) / D : 0.0;
    }
    __syncthreads();
    auto acc = [&](int y, int x) -> double { return t[y][x]; };
    for (int p = threadIdx.x; p < P1 * P1; p += NT) {
        const int ly = p / P1, lx = p % P1;
        const int y = y0 + ly - 1, x = x0 + lx - 1;
        double gx = 0.0, gy = 0.0;
        if (y >= 0 && y < g.H && x >= 0 && x < g.W) scharr_at(acc, ly + 1, lx + 1, gx, gy);
        s1[ly][lx] = gx + gy;
    }
    __syncthreads();
    double sum = 0.0;
    for (int p = threadIdx.x; p < TS * TS; p += NT) {
        const int ly = p / TS, lx = p % TS;
        if (y0 + ly >= g.H || x0 + lx >= g.W) continue;
        const int cy = ly + 1, cx = lx + 1;
        // K is symmetric: convolution == correlation
        const double d = (1.0 / 12.0) * (s1[cy - 1][cx - 1] + s1[cy - 1][cx + 1] + s1[cy + 1][cx - 1] + s1[cy + 1][cx + 1])
                       + (1.0 / 6.0) * (s1[cy - 1][cx] + s1[cy + 1][cx] + s1[cy][cx - 1] + s1[cy][cx + 1]);
        sum += fabs(d);
    }
    sum = block_sum(sum, scratch);
    if (threadIdx.x == 0) divparts[((size_t)b * g.R + r) * g.ntiles + tile] = sum;
}

// ------------------------------------------------------------------------------------------------
// k_divgrad: reverse of k_div for delta != 0.  d = mean|div|, div = K (*) (n (*) Sx + n (*) Sy), every stage zero padded:
//   A = d d / d n * HW = adj_Sx(t) + adj_Sy(t),  t = K (*) sign(div)  (K symmetric; sign(0) = 0; zero outside the image
//   at every stage).  Writes the unscaled image A (fp32) and per-tile partials {sum A*n, sum A}; k_imgrad applies
//   e_r/HW = delta*w_r/(R*(d0+eps)*HW) and folds the two sums into the min/max terms of the normalisation.
// grid (ntiles, R, B).  Footprint: n on tile+4, gs on tile+3, sign on tile+2, t on tile+1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_divgrad(Geom g, const float* __restrict__ iwe, const StatPart* __restrict__ parts,
                                                 float* __restrict__ gdiv, double* __restrict__ dgparts)
{
    constexpr int P4 = TS + 8, P3 = TS + 6, P2 = TS + 4, P1 = TS + 2;
    __shared__ double nn[P4][P4 + 1];
    __shared__ double gs[P3][P3 + 1];
    __shared__ float sg[P2][P2 + 1];
    __shared__ double tt[P1][P1 + 1];
    __shared__ double sc[2];
    __shared__ double scratch[NWAVE];
    const int tile = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    const float* __restrict__ I = iwe + ((size_t)b * g.R + r) * g.H * g.W;
    if (!win_active(g, b)) return;
    if (threadIdx.x < 64) {
        const ImgScal s = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
        if (threadIdx.x == 0) { sc[0] = s.m; sc[1] = s.D; }
    }
    __syncthreads();
    const double m = sc[0], D = sc[1];
    auto inimg = [&](int y, int x) -> bool { return y >= 0 && y < g.H && x >= 0 && x < g.W; };
    for (int p = threadIdx.x; p < P4 * P4; p += NT) {
        const int ly = p / P4, lx = p % P4;
        const int y = y0 + ly - 4, x = x0 + lx - 4;
        nn[ly][lx] = inimg(y, x) ? ((double)I[(size_t)y * g.W + x] - m) / D : 0.0;
    }
    __syncthreads();
    auto acc = [&](int y, int x) -> double { return nn[y][x]; };
    for (int p = threadIdx.x; p < P3 * P3; p += NT) {
        const int ly = p / P3, lx = p % P3;
        double gx = 0.0, gy = 0.0;
        if (inimg(y0 + ly - 3, x0 + lx - 3)) scharr_at(acc, ly + 1, lx + 1, gx, gy);
        gs[ly][lx] = gx + gy;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P2 * P2; p += NT) {
        const int ly = p / P2, lx = p % P2;
        float sgn = 0.0f;
        if (inimg(y0 + ly - 2, x0 + lx - 2)) {
            const int cy = ly + 1, cx = lx + 1;
            const double d = (1.0 / 12.0) * (gs[cy - 1][cx - 1] + gs[cy - 1][cx + 1] + gs[cy + 1][cx - 1] + gs[cy + 1][cx + 1])
                           + (1.0 / 6.0) * (gs[cy - 1][cx] + gs[cy + 1][cx] + gs[cy][cx - 1] + gs[cy][cx + 1]);
            sgn = (d > 0.0) ? 1.0f : ((d < 0.0) ? -1.0f : 0.0f);
        }
        sg[ly][lx] = sgn;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P1 * P1; p += NT) {
        const int ly = p / P1, lx = p % P1;
        double t = 0.0;
        if (inimg(y0 + ly - 1, x0 + lx - 1)) {
            const int cy = ly + 1, cx = lx + 1;
            t = (1.0 / 12.0) * ((double)sg[cy - 1][cx - 1] + sg[cy - 1][cx + 1] + sg[cy + 1][cx - 1] + sg[cy + 1][cx + 1])
              + (1.0 / 6.0) * ((double)sg[cy - 1][cx] + sg[cy + 1][cx] + sg[cy][cx - 1] + sg[cy][cx + 1]);
        }
        tt[ly][lx] = t;
    }
    __syncthreads();
    float* __restrict__ out = gdiv + ((size_t)b * g.R + r) * g.H * g.W;
    double sAn = 0.0, sA = 0.0;
    for (int p = threadIdx.x; p < TS * TS; p += NT) {
        const int ly = p / TS, lx = p % TS;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= g.H || x >= g.W) continue;
        const int cy = ly + 1, cx = lx + 1;
        // adj_Sx(t) + adj_Sy(t) = -(conv(t,Sx) + conv(t,Sy))
        const double ax = 3.0 * (tt[cy + 1][cx + 1] - tt[cy + 1][cx - 1]) + 10.0 * (tt[cy][cx + 1] - tt[cy][cx - 1])
                        + 3.0 * (tt[cy - 1][cx + 1] - tt[cy - 1][cx - 1]);
        const double ay = 3.0 * (tt[cy + 1][cx + 1] - tt[cy - 1][cx + 1]) + 10.0 * (tt[cy + 1][cx] - tt[cy - 1][cx])
                        + 3.0 * (tt[cy + 1][cx - 1] - tt[cy - 1][cx - 1]);
        const double A = -(ax + ay);
        out[(size_t)y * g.W + x] = (float)A;
        const double Af = (double)(float)A;              // what k_imgrad will read back
        sAn += Af * nn[ly + 4][lx + 4];
        sA += Af;
    }
    sAn = block_sum(sAn, scratch);
    sA = block_sum(sA, scratch);
    if (threadIdx.x == 0) {
        double* o = dgparts + (((size_t)b * g.R + r) * g.ntiles + tile) * 2;
        o[0] = sAn; o[1] = sA;
    }
}

// dL/dw of one event at one reference time: the event is warped like in the forward pass, its 3x3 neighbourhood of G = dL/dIWE is
// read from the LDS window `lds` (bounding box wn; taps outside it straight from the image Gi with the JAX wrap/drop rule).
template <typename GAt>      // GAt: float operator()(size_t pixel) - dL/dIWE of this (window, reference time) at a pixel of the image
__device__ __forceinline__ void event_dLdw(const Geom& g, const Window& wn, int wp /* LDS row pitch of the window */, const float* lds, const GAt& Gi,
                                           int x, int y, double2 v, double dt, float& gwx, float& gwy) {
    int irx, iry; float fx, fy;
    warp_axis(x, v.x, dt, irx, fx);
    warp_axis(y, v.y, dt, iry, fy);
    // taps k(d) = e0 * (a, 1, b) per axis with e0 = exp(-f^2/2), a = exp(-1/2 - f), b = exp(-1/2 + f): the common factor e0x e0y / (2 pi)
    // is applied once at the end, and the centre column / row needs no multiplication at all
    constexpr float L2E = 1.4426950408889634f;
    // the constant factors ride in the exponents: b = 2^(f log2e - log2e/2), a = e^-1 / b, scale = 2^(-(fx^2 + fy^2) log2e/2 - log2(2 pi))
    constexpr float EXP_M1 = 0.36787944117144233f, LOG2_INV_2PI = -2.651496129472319f;
    const float bx = __builtin_amdgcn_exp2f(fmaf(fx, L2E, -0.5f * L2E)), by = __builtin_amdgcn_exp2f(fmaf(fy, L2E, -0.5f * L2E));
    const float ax = EXP_M1 * __builtin_amdgcn_rcpf(bx), ay = EXP_M1 * __builtin_amdgcn_rcpf(by);
    const float scale = __builtin_amdgcn_exp2f(fmaf(fmaf(fx, fx, fy * fy), -0.5f * L2E, LOG2_INV_2PI));
    float gv[3][3];
    const int lx = irx - 1 - wn.ox, ly = iry - 1 - wn.oy;
    if ((unsigned)lx < (unsigned)(wn.ww - 2) && (unsigned)ly < (unsigned)(wn.wh - 2)) {
        // byte offsets with the row stride pre-scaled (a scalar): one v_mul_i32_i24 + one v_lshl_add_u32, then one add per further row
        const int ww4 = wp * 4;
        const char* pb = reinterpret_cast<const char*>(lds) + (__mul24(ly, ww4) + (lx << 2));
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float* prow = reinterpret_cast<const float*>(pb + dy * ww4);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) gv[dy][dx] = prow[dx];
        }
    } else {
        const int sx = clamp_far(irx), sy = clamp_far(iry), lxs = sx - 1 - wn.ox, lys = sy - 1 - wn.oy;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int cx = lxs + dx, cy = lys + dy;
                float val = 0.0f;
                if (cx >= 0 && cy >= 0 && cx < wn.ww && cy < wn.wh) {
                    val = lds[cy * wp + cx];
                } else {
                    const int gx = wrap_drop(sx - 1 + dx, g.W), gy = wrap_drop(sy - 1 + dy, g.H);
                    if (gx >= 0 && gy >= 0) val = Gi((size_t)gy * g.W + gx);
                }
                gv[dy][dx] = val;
            }
        }
    }
    // d k(d)/dw = k(d) ((d - 1) - f) on either axis (d = 0, 1, 2 the tap index), so with S = sum_dy sum_dx Ky[dy] Kx[dx] G[dy][dx]
    //   dL/dwx = sum_dx Kx[dx] ((dx - 1) - fx) c[dx] = (Kx[2] c[2] - Kx[0] c[0]) - fx S,   c[dx] = sum_dy Ky[dy] G[dy][dx]
    //   dL/dwy = (Ky[2] r[2] - Ky[0] r[0]) - fy S,                                         r[dy] = sum_dx Kx[dx] G[dy][dx]
    // 22 multiply-adds per event with the normalised taps above (the weighted-tap form W[d] = K[d] ((d - 1) - f) took 36, the tap-by-tap form 42)
    const float c0 = fmaf(by, gv[2][0], fmaf(ay, gv[0][0], gv[1][0]));
    const float c1 = fmaf(by, gv[2][1], fmaf(ay, gv[0][1], gv[1][1]));
    const float c2 = fmaf(by, gv[2][2], fmaf(ay, gv[0][2], gv[1][2]));
    const float u0 = ax * c0, u2 = bx * c2;
    const float S = (u0 + u2) + c1;
    gwx = scale * fmaf(-fx, S, u2 - u0);
    const float r0 = fmaf(bx, gv[0][2], fmaf(ax, gv[0][0], gv[0][1]));
    const float r2 = fmaf(bx, gv[2][2], fmaf(ax, gv[2][0], gv[2][1]));
    gwy = scale * fmaf(-fy, S, fmaf(by, r2, -(ay * r0)));}

// ------------------------------------------------------------------------------------------------
// k_gather: reverse of the splat.  grid as k_splat (block_to_work).  For every event of the segment and this reference time:
//   dL/dwx = sum_taps G[p] * k * qx,  dL/dwy likewise (q = p - w; dropped taps contribute 0, wrapped taps read
//   the wrapped pixel), then dL/dTheta[y,x,:] += -dt * (dL/dwx, dL/dwy)   (event_warpers.py:34-35).
// The G window is staged in LDS with the same bounding box as the forward.  2-DoF theta (direct11): every thread sums its events in
// fp64 in a fixed order, the workgroup reduces them in a fixed order and STORES its partial in its own slot (k_final adds the
// slots in index order).  Otherwise: per-pixel sums are accumulated in an LDS copy of the source tile as i64 fixed point
// (ds_add_u64; scale grad_shift_pixel) and flushed with i64 global atomics.  Both are bit-reproducible.
// ------------------------------------------------------------------------------------------------
// |tvg| <= 2 * 32 per pixel and component (two adjoint Scharr stencils of +-1 images); times H*W pixels, times <= 4 for the weights
__host__ __device__ __forceinline__ int tv_shift(int H, int W) {
    int e = 0;
    while ((double)(1ull << e) <= 256.0 * (double)H * (double)W) ++e;
    return 61 - e;
}
constexpr int PG_MAXC = 6;        // coarse rows / columns under one 32x32 tile that k_gather's own projection handles (16x16 theta on 260x346: 4)

// One 32x32 tile of a (H,W,2) image projected onto the theta cells, cells[i,j] += sum_{y,x} AH[y,i] AW[x,j] vals[y,x]  (the adjoint of
// theta_utils.py:25-35 restricted to the tile), by the whole workgroup.  vals: the tile as double2 in LDS, already at the cells'
// fixed-point scale, pixel (ly, lx) at (ly << 5) + (lx ^ ly) (the swizzle spreads the column walk of step A over the LDS banks);
// written by the caller, who does NOT need to synchronise before the call.  scratch: LDS for the weights and the row results
// (PROJ_SCRATCH_BYTES).  Separable and ordered: A) every (column half, tile row, coarse column) sums 16 products against A_W,
// B) every cell sums the 32 row results against A_H - fp64 in a fixed order, ONE rounding per (workgroup, cell), then an i64 atomic:
// the cell sums do not depend on the order of arrival.  tr.ni, tr.nj <= PG_MAXC.
constexpr int PROJ_SCRATCH_BYTES = 2 * TS * PG_MAXC * 8 + 2 * TS * PG_MAXC * 16;
template <int NTH>
__device__ __forceinline__ void project_tile_to_cells(const Geom& g, int h, int w, const TileRange& tr, int x0, int y0,
                                                      const double* __restrict__ AH, const double* __restrict__ AW,
                                                      const double2* vals, void* scratch, unsigned long long* __restrict__ cells) {
    const int ni = tr.ni, nj = tr.nj;
    double* ahs = reinterpret_cast<double*>(scratch);               // (32, ni)
    double* aws = ahs + TS * PG_MAXC;                                // (32, nj)
    double2* st = reinterpret_cast<double2*>(aws + TS * PG_MAXC);    // (2, 32, nj) row results of the two column halves
    for (int k = threadIdx.x; k < TS * ni; k += NTH) {
        const int ly = k / ni, i = k - ly * ni;
        ahs[k] = (y0 + ly < g.H) ? AH[(size_t)(y0 + ly) * h + tr.ilo + i] : 0.0;
    }
    for (int k = threadIdx.x; k < TS * nj; k += NTH) {
        const int lx = k / nj, j = k - lx * nj;
        aws[k] = (x0 + lx < g.W) ? AW[(size_t)(x0 + lx) * w + tr.jlo + j] : 0.0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * TS * nj; k += NTH) {
        const int q = k / (TS * nj), kk = k - q * (TS * nj);
        const int ly = kk / nj, jj = kk - ly * nj;
        double ax = 0.0, ay = 0.0;
#pragma unroll 4                                             // (fully unrolled these loops cost k_gather 90 VGPRs: 8 -> 3 waves per SIMD)
        for (int lx = 16 * q; lx < 16 * q + 16; ++lx) {
            const double wgt = aws[lx * nj + jj];
            const double2 v = vals[(ly << 5) + (lx ^ ly)];
            ax += wgt * v.x; ay += wgt * v.y;
        }
        st[k] = make_double2(ax, ay);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ni * nj; k += NTH) {
        const int ii = k / nj, jj = k - ii * nj;
        double ax = 0.0, ay = 0.0;
#pragma unroll 4
        for (int ly = 0; ly < TS; ++ly) {
            const double wgt = ahs[ly * ni + ii];
            const double2 v0 = st[ly * nj + jj], v1 = st[TS * nj + ly * nj + jj];
            ax += wgt * (v0.x + v1.x); ay += wgt * (v0.y + v1.y);
        }
        unsigned long long* o = cells + ((size_t)(tr.ilo + ii) * w + (tr.jlo + jj)) * 2;
        if (ax != 0.0) atomicAdd(o, (unsigned long long)fix64_wide(ax));
        if (ay != 0.0) atomicAdd(o + 1, (unsigned long long)fix64_wide(ay));
    }
}
template <int TM, int WIDE, int NTH, int COMPOSE, int PROJ, int ALLR = 0>      // NTH threads per workgroup: 256, or 512 where the LDS footprint allows only 3 workgroups per CU (THETA_TILE); TM: the theta mode as a compile-time constant (THETA_CONST / THETA_TILE); 0 = take the run-time argument
__global__ __launch_bounds__(NTH, (ALLR ? 6 : 1)) void k_gather(Geom g, int n_items,      // WIDE: 61-bit fixed point per event (tiny windows, see grad_shift_pixel)
        const Item* __restrict__ items, const uint32_t* __restrict__ ev_xy, const double* __restrict__ ev_t,
        const double* __restrict__ Theta, const double* __restrict__ tmm, const double* __restrict__ edge_ts,
        const float* __restrict__ G,           // (B,R,H,W) dL/dIWE written by k_imgrad (COMPOSE = 0), or the A image of k_imstat (COMPOSE = 1)
        const Window* __restrict__ wins,       // (n_items, R) windows of this evaluation
        long long* __restrict__ gTheta,        // (B,H,W,2) i64 fixed point, zero on entry (cleared by its consumer)
        int direct11, double* __restrict__ g11,                     // 2-DoF theta: (n_items, R, 2) per-workgroup partials of dL/dtheta
        const WinConst* __restrict__ wc, const unsigned* __restrict__ gmax,   // scale of the i64 accumulators (grad_shift), COMPOSE = 0
        int theta_mode, const int32_t* __restrict__ order,
        int use_arg, const double* __restrict__ theta_c, ThetaArg targ,   // 2-DoF theta (B,2): in the kernel arguments, or behind theta_c
        // COMPOSE = 1 (k_imstat in front instead of k_stats_stream + k_imgrad): G is composed from (A, E, I) while the window is staged
        int gradmag_i, const float* __restrict__ edges, const float* __restrict__ iwe,
        const ImgCoef* __restrict__ coef,      // (B,R) per-image scalars and |G| bounds from k_imstat's tail
        unsigned long long* __restrict__ acc,  // the u64 IWE accumulator: consumed by k_imstat, cleared here (a slice per workgroup)
        int list_a,                            // the segments walked are the gather's own list (window capacity wincap_a), not the splat's
        int nparts,                            // 1, 2 or 4 = gridDim.y: workgroups sharing a segment (256-thread form only)
        // PROJ = 1 (theta grids whose tiles touch <= PG_MAXC x PG_MAXC cells): the workgroup projects its tile's sums onto the theta
        // cells itself, dL/dtheta[i,j] += sum_{y,x} AH[y,i] AW[x,j] dL/dTheta[y,x] (reverse of theta_utils.py:25-35), instead of
        // flushing them into the dL/dTheta image for k_project: 2 x 32 x 32 global atomics per workgroup become <= 2 ni nj
        int h, int w, const double* __restrict__ AH, const double* __restrict__ AW, const TileRange* __restrict__ tilerng,
        long long* __restrict__ gth_main, int gth_cap,
        // tail (PROJ only, tail != 0): the workgroup of a window that finishes LAST turns the window's i64 cell sums into dL/dtheta and
        // writes it where the host reads it - what k_final did in a launch of its own (7-8 us of a 65 us evaluation)
        int tail, unsigned* __restrict__ ticket, const int32_t* __restrict__ win_item0, double* __restrict__ grad_out,
        double tv_gamma, const double* __restrict__ tvparts, long long* __restrict__ gth_tv,   // tail with the TV term (gamma != 0 at level 0)
        int all_r_unused)   // (ALLR = 1, theta grids on big batches: ONE workgroup per segment walks all R reference times - grid = segments,
                     // not segments x R -, so that the Theta tile, the accumulator clear, the projection and the ticket are paid once per segment)
{
    const int part = blockIdx.y;
    if (TM != 0) theta_mode = TM;                 // every branch on it below folds away: 8 % on both event kernels
    if (TM != 0) direct11 = (TM == THETA_CONST) ? 1 : 0;     // the host ties the two (2-DoF theta <=> per-workgroup partials)
    const int wincap = list_a ? g.wincap_a : g.wincap, winmaxw = list_a ? g.winmaxw_a : g.winmaxw;
    // LDS: [G window: wincap floats][accum: TS*TS*2 doubles unless direct11][Theta tile: TS*TS double2 if THETA_TILE]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red11[NTH / 64];
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(lds + wincap);   // i64 fixed point: ds_add_u64 (3.7 lane-ops/clk/CU; ds_add_f32: 0.33)
    double2* thtile = reinterpret_cast<double2*>(lds + wincap + (direct11 ? 0 : TS * TS * 4));
    float f11x = 0.0f, f11y = 0.0f;             // direct11: this thread's share of sum_e -dt * dL/dw
    if (COMPOSE) {
        // consumer-clears, delegated: k_imstat has read the accumulator (halos included), so every workgroup of this launch zeroes
        // an equal share of it with plain stores (fire and forget; nothing in this kernel reads it)
        const size_t total = (size_t)g.B * g.R * g.H * g.W;
        const size_t nb = (size_t)gridDim.x * gridDim.y, bid = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        const size_t lo = total * bid / nb, hi = total * (bid + 1) / nb;
        for (size_t i = lo + threadIdx.x; i < hi; i += NTH) acc[i] = 0ull;
    }
    int item, r;
    constexpr bool all_r = ALLR != 0;
    if (!block_to_work(n_items, all_r ? 1 : g.R, order, item, r)) return;
    const Item it = items[item];
    if (!win_active(g, it.win)) return;
    double tau = edge_ts[it.win * g.R + r];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool gradmag = gradmag_i != 0;
    double2 vconst = make_double2(0.0, 0.0);
    Window wn;
    if (theta_mode == THETA_CONST) {              // the same window k_splat derived for itself (see there)
        vconst = use_arg ? make_double2(targ.v[2 * it.win], targ.v[2 * it.win + 1]) : make_double2(theta_c[2 * it.win], theta_c[2 * it.win + 1]);
        const double mm4[4] = {vconst.x, vconst.x, vconst.y, vconst.y};
        wn = item_window(g, it, mm4, tau, wincap, winmaxw);
    } else {
        wn = wins[(size_t)item * g.R + r];
    }
    int wp = (theta_mode == THETA_CONST) ? win_pitch(g, wn.ww) : wn.ww;      // LDS row pitch of the G window (see item_window)
    size_t img = ((size_t)it.win * g.R + r) * g.H * g.W;
    const float* __restrict__ Gi = G + img;
    const float* __restrict__ Ei = edges + img;
    const float* __restrict__ Ii = iwe + img;
    const int tx = it.tile % g.tilesX, ty = it.tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    if (theta_mode != THETA_CONST) {
        const double* __restrict__ ThW = Theta + (size_t)it.win * g.H * g.W * 2;
        for (int p = threadIdx.x; p < TS * TS; p += NTH) {
            const int y = y0 + p / TS, x = x0 + p % TS;
            thtile[p] = (y < g.H && x < g.W) ? *reinterpret_cast<const double2*>(ThW + ((size_t)y * g.W + x) * 2) : make_double2(0.0, 0.0);
        }
    }
    if (!direct11) for (int i = threadIdx.x; i < TS * TS * 2; i += NTH) accum[i] = 0ull;
    const ImgCoef* __restrict__ cw = coef + (COMPOSE ? it.win * g.R : 0);
    GCoef q{};
    if (COMPOSE) q = cw[r].q;                      // uniform address: scalar loads, the coefficients live in SGPRs
    auto G_at = [&](size_t p) -> float {          // dL/dIWE at pixel p of this (window, reference time)
        if (COMPOSE) return compose_G(q, gradmag, gradmag ? Gi[p] : 0.0f, Ei[p], Ii[p]);
        return Gi[p];
    };
    auto G_far = [&](size_t p) -> float {         // the same for the rare taps outside the LDS window, inside the event loop
        if (!COMPOSE) return Gi[p];
        return compose_G(q, gradmag, gradmag ? Gi[p] : 0.0f, Ei[p], Ii[p]);
    };
    auto load_window = [&]() {                    // the G window of the current reference time into LDS
    if (wn.ox >= 0 && wn.oy >= 0 && wn.ox + wn.ww <= g.W && wn.oy + wn.wh <= g.H) {      // the usual case: window inside the image
        const size_t o0 = (size_t)wn.oy * g.W + wn.ox;
        for (WinWalkT<NTH> w(threadIdx.x, wn.ww); w.i < wn.ww * wn.wh; w.next()) lds[w.row * wp + w.col] = G_at(o0 + w.row * g.W + w.col);
    } else {
        for (int row = wv; row < wn.wh; row += NTH / 64) {
            const int gy = wrap_drop(wn.oy + row, g.H);
            for (int col = lane; col < wn.ww; col += 64) {
                const int gx = wrap_drop(wn.ox + col, g.W);
                lds[row * wp + col] = (gx >= 0 && gy >= 0) ? G_at((size_t)gy * g.W + gx) : 0.0f;
            }
        }
    }
    };
    load_window();
    double gscale = 0.0;                          // 2^eg of this window's gradient accumulators
    double gm_used = 0.0;                         // max |G| (or its bound) the scales derive from
    if (!direct11) {
        double gm;
        if (COMPOSE) {                            // the same R words k_project / k_final* read through gmax_of (gmax_n = R)
            float gmf = cw[0].bound;
            for (int rr = 1; rr < g.R; ++rr) gmf = fmaxf(gmf, cw[rr].bound);
            gm = (double)gmf;
        } else {
            __shared__ unsigned gms[NTH / 64];
            gm = gmax_of(gmax + (size_t)it.win * g.gmax_n, g.gmax_n, gms);
        }
        gscale = ldexp(1.0, grad_shift_pixel(wc[it.win], gm, g.R, WIDE != 0));
        gm_used = gm;
    }
    __syncthreads();

    const uint32_t* __restrict__ exy = ev_xy + it.begin;
    const double* __restrict__ et = ev_t + it.begin;
    const int n = it.count;
    const int tid = threadIdx.x;
    // Run state of a theta-grid / dense gather: consecutive events of a thread mostly share their source pixel (k_segsort), so
    // -dt dL/dw is summed in fp64 registers over a run and enters the pixel's i64 accumulator once per run (two ds_add_u64 per
    // RUN instead of per event: those atomics were what bound the round-2 kernel), and the pixel's velocity is read once per run.
    uint32_t cur_key = 0xffffffffu;
    double run_x = 0.0, run_y = 0.0;
    double2 vcur = vconst;
    auto flush_run = [&]() {
        if (cur_key == 0xffffffffu) return;
        unsigned long long* a = accum + (cur_key << 1);
        atomicAdd(a, (unsigned long long)(WIDE ? fix64_wide(run_x) : fix64(run_x)));
        atomicAdd(a + 1, (unsigned long long)(WIDE ? fix64_wide(run_y) : fix64(run_y)));
    };
    auto gather_ev = [&](const EvReg& ev) {
        const double dt = ev.t - tau;
        const int x = ev.xy & 0xffff, y = ev.xy >> 16;
        if (theta_mode != THETA_CONST) {
            const uint32_t key = ((ev.xy >> 11) & (31u << 5)) | (ev.xy & 31u);
#ifdef EINCM_ABL_G_NORUN
            vcur = thtile[key];
#else
            if (key != cur_key) {
                flush_run();
                cur_key = key; run_x = 0.0; run_y = 0.0;
                vcur = thtile[key];
            }
#endif
        }
        float gwx, gwy;
        event_dLdw(g, wn, wp, lds, G_far, x, y, vcur, dt, gwx, gwy);
        if (direct11) {          // theta (1,1,2): Theta is constant, dL/dtheta = sum over all events; no per-pixel image needed.
            // fp32 over the thread's own <= 64 terms (their rounding errors are independent across 10^6 threads and average out:
            // measured 1e-9 relative on the gradient), fp64 from there on
            const float ndt = (float)(-dt);
            f11x = fmaf(ndt, gwx, f11x); f11y = fmaf(ndt, gwy, f11y);
        } else {
            const double sdt = -dt * gscale;
            run_x = fma(sdt, (double)gwx, run_x); run_y = fma(sdt, (double)gwy, run_y);
        }
    };
    // the staged layout (SegWalk): a half of the segment is K coalesced steps of 256 threads, the last one a prefix
    const SegWalk<NTH> walk(n, tid);
    auto walk_half = [&](int c, int sub, int nsub) {      // steps [K sub / nsub, K (sub + 1) / nsub) of half c
        const int K = c ? walk.K1 : walk.K0, rem = c ? walk.rem1 : walk.rem0;
        const int j0 = K * sub / nsub, j1 = K * (sub + 1) / nsub;
        const uint32_t* __restrict__ px = exy + (c ? walk.n0 : 0) + walk.tt;
        const double* __restrict__ pt = et + (c ? walk.n0 : 0) + walk.tt;
        const int jfull = min(j1, K - 1);
#pragma unroll 2
        for (int j = j0; j < jfull; ++j) {
            EvReg ev; ev.xy = px[j * 256]; ev.t = pt[j * 256];
            gather_ev(ev);
        }
        if (j1 == K && K > 0 && j0 < K && walk.tt < rem) {
            EvReg ev; ev.xy = px[(K - 1) * 256]; ev.t = pt[(K - 1) * 256];
            gather_ev(ev);
        }
    };
    for (int rr = r; ; ) {                        // one reference time, or all of them (all_r)
#ifndef EINCM_ABL_G_NOEVENTS
    // nparts > 1 (2-DoF theta, few windows): a segment is shared by nparts workgroups (blockIdx.y), so that the segments can be long
    // (what the theta-grid gather wants from the one list both walk) and the chip still sees enough workgroups
    if (NTH == 512) walk_half(walk.half, 0, 1);
    else if (nparts == 1) { walk_half(0, 0, 1); walk_half(1, 0, 1); }
    else { const int nsub = nparts >> 1; walk_half(part / nsub, part % nsub, nsub); }
    if (!direct11) { flush_run(); cur_key = 0xffffffffu; run_x = 0.0; run_y = 0.0; }
#endif
    if (!all_r || ++rr >= g.R) break;
    __syncthreads();                              // every lane is done with the G window of the previous reference time
    tau = edge_ts[it.win * g.R + rr];
    wn = wins[(size_t)item * g.R + rr];
    wp = wn.ww;
    img = ((size_t)it.win * g.R + rr) * g.H * g.W;
    Gi = G + img; Ei = edges + img; Ii = iwe + img;
    load_window();
    __syncthreads();
    }
    if (direct11) {
        double sum11x = block_sum<NTH / 64>((double)f11x, red11);
        double sum11y = block_sum<NTH / 64>((double)f11y, red11);
        if (threadIdx.x == 0) {          // own slot, plain store, written unconditionally: nothing to clear, nothing to order
            double* dst = g11 + (((size_t)item * nparts + part) * g.R + r) * 2;
            dst[0] = sum11x; dst[1] = sum11y;
        }
        return;
    }
    __syncthreads();
#ifdef EINCM_ABL_G_NOFLUSH
    return;
#endif
    if (PROJ) {
        // The tile's sums projected onto the theta cells (project_tile_to_cells); the G window is dead by now: its LDS holds the weights.
        const TileRange tr = tilerng[it.tile];
        static_assert(PROJ_SCRATCH_BYTES <= WIN_CAP_DEFAULT * 4, "the projection's scratch fits the smallest G window");
        // the sums as doubles at the cell scale, into the (dead) Theta tile: one conversion per pixel by all threads
        const double gmd = gm_used;                                      // the same value k_final derives its scale from (a float read back)
        const double scale = ldexp(1.0, grad_shift(wc[it.win], gmd, g.R) - grad_shift_pixel(wc[it.win], gmd, g.R, WIDE != 0));
        for (int p = threadIdx.x; p < TS * TS; p += NTH) {
            const int ly = p >> 5, lx = p & 31;
            const long long vx = (long long)accum[p * 2], vy = (long long)accum[p * 2 + 1];       // exact conversions for |.| < 2^53 (always, unless WIDE)
            thtile[(ly << 5) + (lx ^ ly)] = make_double2((double)vx * scale, (double)vy * scale);
        }
        unsigned long long* __restrict__ out = reinterpret_cast<unsigned long long*>(gth_main) + (size_t)it.win * gth_cap;
        project_tile_to_cells<NTH>(g, h, w, tr, x0, y0, AH, AW, thtile, lds, out);
        if (!tail) return;
        // Every cell atomic of this workgroup has been performed (vmcnt counts them until they are) before its ticket is drawn, and
        // integer atomics of all workgroups meet at one coherence point, so the workgroup that draws the last ticket of its window
        // sees every contribution when it exchanges the cells for zero (which also leaves them clear for the next evaluation).
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n_it = ((it.win + 1 < g.B) ? win_item0[it.win + 1] : n_items) - win_item0[it.win];
            const unsigned old = __hip_atomic_fetch_add(ticket + it.win, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == (unsigned)(n_it * (all_r ? 1 : g.R)) - 1u;
            if (last) __hip_atomic_store(ticket + it.win, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
        const double inv = ldexp(1.0, -grad_shift(wc[it.win], gmd, g.R));
        const int ncell = h * w * 2;
        double stv = 0.0;                                    // gamma * 0.25 / (#non-zero TV pixels + eps) * 2^-tv_shift, as k_final forms it
        if (tv_gamma != 0.0) {
            double nz = 0.0;
            for (int i = threadIdx.x; i < g.ntiles; i += NTH) nz += tvparts[((size_t)it.win * g.ntiles + i) * 3 + 1];
            __shared__ double tvred[NTH / 64];
            nz = block_sum<NTH / 64>(nz, tvred);
            __shared__ double s_stv;
            if (threadIdx.x == 0) s_stv = tv_gamma * 0.25 / (nz + EPSN) * ldexp(1.0, -tv_shift(g.H, g.W));
            __syncthreads();
            stv = s_stv;
        }
        unsigned long long* __restrict__ out_tv = reinterpret_cast<unsigned long long*>(gth_tv) + (size_t)it.win * gth_cap;
        for (int i = threadIdx.x; i < ncell; i += NTH) {
            const long long q = (long long)atomicExch(out + i, 0ull);
            double v = (double)q * inv;
            if (tv_gamma != 0.0) v += stv * (double)(long long)atomicExch(out_tv + i, 0ull);      // k_tv's own projection (an earlier kernel)
            grad_out[(size_t)it.win * ncell + i] = v;
        }
        return;
    }
    unsigned long long* __restrict__ gT = reinterpret_cast<unsigned long long*>(gTheta) + (size_t)it.win * g.H * g.W * 2;
    const int tw = min(TS, g.W - x0), th = min(TS, g.H - y0);
    for (int i = threadIdx.x; i < TS * TS * 2; i += NTH) {
        const int c = i & 1, px = (i >> 1) % TS, py = (i >> 1) / TS;
        const unsigned long long v = accum[i];
        if (px < tw && py < th && v != 0ull) atomicAdd(gT + ((size_t)(y0 + py) * g.W + (x0 + px)) * 2 + c, v);
    }
}

// k_compose: dL/dIWE as an image, for eincm_get_image_grad after an evaluation whose gather composed it on the fly (the same
// compose_G on the same scalars).  grid (blocks, R, B), grid-stride over the pixels of image (b, r).
__global__ __launch_bounds__(NT) void k_compose(Geom g, EvalParams ep, const float* __restrict__ A, const float* __restrict__ edges,
                                                 const float* __restrict__ iwe, const StatPart* __restrict__ parts,
                                                 const WinConst* __restrict__ wc, float* __restrict__ G)
{
    __shared__ GCoef sq;
    const int r = blockIdx.y, b = blockIdx.z;
    if (!win_active(g, b)) return;
    if (threadIdx.x < 64) {
        const ImgScal s = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
        if (threadIdx.x == 0) sq = gcoef_from(s, wc[b], ep, r, g.R, (double)g.H * (double)g.W);
    }
    __syncthreads();
    const GCoef q = sq;
    const bool gradmag = (ep.contrast_kind == 0);
    const size_t n = (size_t)g.H * g.W, img = ((size_t)b * g.R + r) * n;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT)
        G[img + i] = compose_G(q, gradmag, gradmag ? A[img + i] : 0.0f, edges[img + i], iwe[img + i]);
}

// ------------------------------------------------------------------------------------------------
// k_count: integer image of the ROUNDED warped coordinates, counts[b,r,ry,rx] += 1 with the JAX index rule (the centre tap of
// events_to_pdf_frame, event_utils.py:32-33,59).  Not on the evaluation path: it exposes the fp64 warp + half-to-even
// rounding decisions as an integer image that must equal the oracle's bit for bit.  grid as k_splat (block_to_work).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_count(Geom g, int n_items, const Item* __restrict__ items, const uint32_t* __restrict__ ev_xy,
                                               const double* __restrict__ ev_t, const double* __restrict__ Theta,
                                               const double* __restrict__ edge_ts, uint32_t* __restrict__ counts)
{
    int item, r;
    if (!block_to_work(n_items, g.R, nullptr, item, r)) return;
    const Item it = items[item];
    const double tau = edge_ts[it.win * g.R + r];
    const double* __restrict__ Th = Theta + (size_t)it.win * g.H * g.W * 2;
    uint32_t* __restrict__ img = counts + ((size_t)it.win * g.R + r) * g.H * g.W;
    for (int i = threadIdx.x; i < it.count; i += NT) {
        const uint32_t xy = ev_xy[it.begin + i];
        const double dt = ev_t[it.begin + i] - tau;
        const int x = xy & 0xffff, y = xy >> 16;
        const double2 v = *reinterpret_cast<const double2*>(Th + ((size_t)y * g.W + x) * 2);
        int irx, iry; float fx, fy;
        warp_axis(x, v.x, dt, irx, fx);
        warp_axis(y, v.y, dt, iry, fy);
        const int gx = wrap_drop(clamp_far(irx), g.W), gy = wrap_drop(clamp_far(iry), g.H);
        if (gx >= 0 && gy >= 0) atomicAdd(img + (size_t)gy * g.W + gx, 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// k_warp_events: the warped coordinates themselves, warped[r, i] = x_i - Theta[y_i, x_i] * (t_i - tau_r), for ONE window's events in the
// order the caller handed them over (the raw upload of staging): the 'warped_xs' / 'warped_ys' entries of compute_loss_objectives
// (losses.py:58,90-91; event_warpers.py:34-35), which only plotters read.  Same two roundings as warp_axis.  Not on the evaluation path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_warp_events(Geom g, int win, long long n, const int16_t* __restrict__ xs, const int16_t* __restrict__ ys,
                                                     const double* __restrict__ ts, const double* __restrict__ Theta,
                                                     const double* __restrict__ edge_ts, double* __restrict__ wx, double* __restrict__ wy)
{
#pragma clang fp contract(off)
    const double* __restrict__ Th = Theta + (size_t)win * g.H * g.W * 2;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int x = xs[i], y = ys[i];
        const double t = ts[i];
        const double2 v = *reinterpret_cast<const double2*>(Th + ((size_t)y * g.W + x) * 2);
        for (int r = 0; r < g.R; ++r) {
            const double dt = t - edge_ts[win * g.R + r];
            wx[(size_t)r * n + i] = (double)x - v.x * dt;
            wy[(size_t)r * n + i] = (double)y - v.y * dt;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_mask: event-presence mask (event_utils.py:64-76 / theta_utils.py:59-71).  grid-stride over events.
// ------------------------------------------------------------------------------------------------
__global__ void k_mask(Geom g, const Item* __restrict__ items, int n_items, const uint32_t* __restrict__ ev_xy,
                       uint8_t* __restrict__ mask)
{
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const Item I = items[it];
        uint8_t* m = mask + (size_t)I.win * g.H * g.W;
        for (int i = threadIdx.x; i < I.count; i += blockDim.x) {
            const uint32_t xy = ev_xy[I.begin + i];
            m[(size_t)(xy >> 16) * g.W + (xy & 0xffff)] = 1;
        }
    }
}

// k_tile_counts: the same mask, plus the largest number of events on one source pixel (WinConst.cntmax), per (tile, window):
// LDS histogram of the tile's events.  grid (ntiles, B).  cntmax (B) zeroed beforehand (atomicMax: order-independent).
__global__ __launch_bounds__(NT) void k_tile_counts(Geom g, const int32_t* __restrict__ tilebase, const int32_t* __restrict__ tilecount,
                                                     const uint32_t* __restrict__ ev_xy, uint8_t* __restrict__ mask, unsigned* __restrict__ cntmax)
{
    __shared__ unsigned hist[TS * TS];
    __shared__ unsigned wmax[NWAVE];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int x0 = (tile % g.tilesX) * TS, y0 = (tile / g.tilesX) * TS;
    for (int i = threadIdx.x; i < TS * TS; i += NT) hist[i] = 0u;
    __syncthreads();
    const int base = tilebase[(size_t)b * g.ntiles + tile], cnt = tilecount[(size_t)b * g.ntiles + tile];
    for (int i = threadIdx.x; i < cnt; i += NT) {
        const uint32_t xy = ev_xy[base + i];
        atomicAdd(&hist[((xy >> 11) & (31u << 5)) | (xy & 31u)], 1u);
    }
    __syncthreads();
    unsigned m = 0u;
    uint8_t* mk = mask + (size_t)b * g.H * g.W;
    for (int i = threadIdx.x; i < TS * TS; i += NT) {
        const unsigned h = hist[i];
        if (h) { mk[(size_t)(y0 + (i >> 5)) * g.W + x0 + (i & 31)] = 1; m = max(m, h); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NWAVE; ++i) m = max(m, wmax[i]);
        if (m) atomicMax(cntmax + b, m);
    }
}

// ------------------------------------------------------------------------------------------------
// k_tv: total variation of the event-masked flow (regularizers.py:14-38) and its unscaled gradient image.
//   F = Theta*mask;  TV = 0.25*sum(|Fx*Sx|+|Fx*Sy|+|Fy*Sx|+|Fy*Sy|) / (#pixels with any non-zero term + eps)
//   tvg[y,x,c] = mask * sum_{S in Sx,Sy} adj_S(sign(F_c * S))         (scaled by 0.25/(count+eps) in k_final)
// grid (ntiles, B).  tvparts (B,ntiles,2) = {sum of abs, non-zero count}.
// With unmasked != 0 also accumulates |div Theta| partial (regularizers.py:41-58) into tvparts[...,2].
// ------------------------------------------------------------------------------------------------
// PROJ_TV = 1 (theta grids whose tiles touch <= PG_MAXC x PG_MAXC cells): the tile's gradient values are projected onto the theta cells
// here (gth_tv, i64 at tv_shift) instead of being written as an image for k_project: one launch and 16 B per pixel less per evaluation.
template <int PROJ_TV>
__global__ __launch_bounds__(NT) void k_tv(Geom g, const double* __restrict__ Theta, const uint8_t* __restrict__ mask,
                                            double* __restrict__ tvg, double* __restrict__ tvparts, int want_thdiv,
                                            int h, int w, const double* __restrict__ AH, const double* __restrict__ AW,
                                            const TileRange* __restrict__ tilerng, long long* __restrict__ gth_tv, int gth_cap,
                                            double* __restrict__ tvparts_host)      // or nullptr: a second copy of tvparts where the host reads it
{
    constexpr int P2 = TS + 4, P1 = TS + 2;
    __shared__ __attribute__((aligned(16))) double f[2][P2][P2 + 1];
    __shared__ __attribute__((aligned(16))) float sg[4][P1][P1 + 1];       // sign(gx_0), sign(gy_0), sign(gx_1), sign(gy_1)
    static_assert(sizeof(double) * 2 * P2 * (P2 + 1) >= sizeof(double2) * TS * TS, "f holds the tile for the projection");
    static_assert(sizeof(float) * 4 * P1 * (P1 + 1) >= PROJ_SCRATCH_BYTES, "sg holds the projection's scratch");
    __shared__ double scratch[NWAVE];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    const double* __restrict__ Th = Theta + (size_t)b * g.H * g.W * 2;
    if (!win_active(g, b)) return;
    const uint8_t* __restrict__ mk = mask + (size_t)b * g.H * g.W;
    for (int p = threadIdx.x; p < P2 * P2; p += NT) {
        const int ly = p / P2, lx = p % P2;
        const int y = y0 + ly - 2, x = x0 + lx - 2;
        double vx = 0.0, vy = 0.0;
        if (y >= 0 && y < g.H && x >= 0 && x < g.W && mk[(size_t)y * g.W + x]) {
            const double2 v = *reinterpret_cast<const double2*>(Th + ((size_t)y * g.W + x) * 2);
            vx = v.x; vy = v.y;
        }
        f[0][ly][lx] = vx; f[1][ly][lx] = vy;
    }
    __syncthreads();
    double sabs = 0.0, cnt = 0.0;
    for (int p = threadIdx.x; p < P1 * P1; p += NT) {
        const int ly = p / P1, lx = p % P1;
        const int y = y0 + ly - 1, x = x0 + lx - 1;
        const bool in = (y >= 0 && y < g.H && x >= 0 && x < g.W);
        double g4[4] = {0.0, 0.0, 0.0, 0.0};
        if (in) {
            auto a0 = [&](int yy, int xx) -> double { return f[0][yy][xx]; };
            auto a1 = [&](int yy, int xx) -> double { return f[1][yy][xx]; };
            scharr_at(a0, ly + 1, lx + 1, g4[0], g4[1]);
            scharr_at(a1, ly + 1, lx + 1, g4[2], g4[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[k][ly][lx] = (g4[k] > 0.0) ? 1.0f : ((g4[k] < 0.0) ? -1.0f : 0.0f);
        const bool own = in && ly >= 1 && ly <= TS && lx >= 1 && lx <= TS;     // pixel belongs to this tile
        if (own) {
            sabs += (fabs(g4[0]) * 0.25 + fabs(g4[1]) * 0.25) + (fabs(g4[2]) * 0.25 + fabs(g4[3]) * 0.25);
            if (fabs(g4[0]) > 0.0 || fabs(g4[1]) > 0.0 || fabs(g4[2]) > 0.0 || fabs(g4[3]) > 0.0) cnt += 1.0;
        }
    }
    __syncthreads();
    double* __restrict__ out = tvg + (size_t)b * g.H * g.W * 2;
    double2 mine[TS * TS / NT];                               // PROJ_TV: this thread's pixels, parked until every thread has read f's consumers (sg)
    const double tvscale = ldexp(1.0, tv_shift(g.H, g.W));
    for (int p = threadIdx.x, kq = 0; p < TS * TS; p += NT, ++kq) {
        const int ly = p / TS, lx = p % TS;
        const int y = y0 + ly, x = x0 + lx;
        if (PROJ_TV) mine[kq] = make_double2(0.0, 0.0);
        if (y >= g.H || x >= g.W) continue;
        double o[2] = {0.0, 0.0};
        if (mk[(size_t)y * g.W + x]) {
            const int cy = ly + 1, cx = lx + 1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float (*sx)[P1 + 1] = sg[2 * c];
                const float (*sy)[P1 + 1] = sg[2 * c + 1];
                const double ax = 3.0 * ((double)sx[cy + 1][cx + 1] - sx[cy + 1][cx - 1]) + 10.0 * ((double)sx[cy][cx + 1] - sx[cy][cx - 1])
                                + 3.0 * ((double)sx[cy - 1][cx + 1] - sx[cy - 1][cx - 1]);
                const double ay = 3.0 * ((double)sy[cy + 1][cx + 1] - sy[cy - 1][cx + 1]) + 10.0 * ((double)sy[cy + 1][cx] - sy[cy - 1][cx])
                                + 3.0 * ((double)sy[cy + 1][cx - 1] - sy[cy - 1][cx - 1]);
                o[c] = -(ax + ay);
            }
        }
        if (PROJ_TV) mine[kq] = make_double2(o[0] * tvscale, o[1] * tvscale);
        else *reinterpret_cast<double2*>(out + ((size_t)y * g.W + x) * 2) = make_double2(o[0], o[1]);
    }
    sabs = block_sum(sabs, scratch);
    cnt = block_sum(cnt, scratch);
    double* tp = tvparts + ((size_t)b * g.ntiles + tile) * 3;
    if (threadIdx.x == 0) {
        tp[0] = sabs; tp[1] = cnt; tp[2] = 0.0;
        if (tvparts_host) { double* th = tvparts_host + ((size_t)b * g.ntiles + tile) * 3; th[0] = sabs; th[1] = cnt; th[2] = 0.0; }
    }
    if (PROJ_TV) {
        __syncthreads();                                     // f and sg are dead: the tile goes into f, the projection's scratch into sg
        double2* vals = reinterpret_cast<double2*>(&f[0][0][0]);
        for (int p = threadIdx.x, kq = 0; p < TS * TS; p += NT, ++kq) {
            const int ly = p >> 5, lx = p & 31;
            vals[(ly << 5) + (lx ^ ly)] = mine[kq];
        }
        project_tile_to_cells<NT>(g, h, w, tilerng[tile], x0, y0, AH, AW, vals, &sg[0][0][0],
                                  reinterpret_cast<unsigned long long*>(gth_tv) + (size_t)b * gth_cap);
        __syncthreads();                                     // (the report below reuses f)
    }

    if (want_thdiv) {
        // per_pix_theta_divergence (regularizers.py:41-58): UNMASKED Theta
        __syncthreads();
        for (int p = threadIdx.x; p < P2 * P2; p += NT) {
            const int ly = p / P2, lx = p % P2;
            const int y = y0 + ly - 2, x = x0 + lx - 2;
            double vx = 0.0, vy = 0.0;
            if (y >= 0 && y < g.H && x >= 0 && x < g.W) {
                const double2 v = *reinterpret_cast<const double2*>(Th + ((size_t)y * g.W + x) * 2);
                vx = v.x; vy = v.y;
            }
            f[0][ly][lx] = vx; f[1][ly][lx] = vy;
        }
        __syncthreads();
        // evaluate K (*) s per tile pixel straight from f (5x5 footprint); s = sum of the four Scharr images,
        // zero outside the image (each 'same' convolution stage is zero padded).  Report-only, so recompute freely.
        double sum = 0.0;
        for (int p = threadIdx.x; p < TS * TS; p += NT) {
            const int ly = p / TS, lx = p % TS;
            const int y = y0 + ly, x = x0 + lx;
            if (y >= g.H || x >= g.W) continue;
            double d = 0.0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) continue;
                    auto a0 = [&](int q, int rr) -> double { return f[0][q][rr]; };
                    auto a1 = [&](int q, int rr) -> double { return f[1][q][rr]; };
                    double gx0, gy0, gx1, gy1;
                    scharr_at(a0, ly + 2 + dy, lx + 2 + dx, gx0, gy0);
                    scharr_at(a1, ly + 2 + dy, lx + 2 + dx, gx1, gy1);
                    const double kw = (dy != 0 && dx != 0) ? (1.0 / 12.0) : (1.0 / 6.0);
                    d += kw * (gx0 + gy0 + gx1 + gy1);
                }
            }
            sum += fabs(d);
        }
        sum = block_sum(sum, scratch);
        if (threadIdx.x == 0) tp[2] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// k_project: dL/dtheta[i,j,c] += sum_{y,x in tile} AH[y,i] AW[x,j] src[y,x,c]   (adjoint of k_theta).
// grid (ntiles, B, nsrc): z = 0 projects the i64 event gradient gTheta (and clears it: consumer-clears), z = 1 the fp64 TV
// gradient image.  All sums are i64 fixed point (event part: scale grad_shift; TV part: tv_shift), so they do not depend on the
// order of the atomics.  One cell under the tile: fp64 block sum in a fixed order, one i64 atomic.  Several cells: every pixel
// adds its <= taps x taps contributions into LDS cell accumulators (ds_add_u64), flushed with global i64 atomics.  More cells
// than PROJ_CELLS: straight to HBM.
// ------------------------------------------------------------------------------------------------
constexpr int PROJ_CELLS = 1024;
__global__ __launch_bounds__(NT) void k_project(Geom g, int h, int w, int cap, int src0, int wide,
        const double* __restrict__ AH, const double* __restrict__ AW,
        const int2* __restrict__ rowtap, const int2* __restrict__ coltap,
        long long* __restrict__ gTheta, const double* __restrict__ tvg,
        const WinConst* __restrict__ wc, const unsigned* __restrict__ gmax,
        long long* __restrict__ gth_main, long long* __restrict__ gth_tv)   // (B,cap) each, zero on entry (k_final clears)
{
    __shared__ double scratch[NWAVE];
    __shared__ ResampleLds RL;
    static_assert(PROJ_CELLS * 2 >= TS * TS * 2, "cells doubles as the tile buffer of the staged path");
    __shared__ __attribute__((aligned(16))) unsigned long long cells[PROJ_CELLS * 2 + TS * RS_MAXC * 2];   // staged path: sv (TS*TS double2) + st (TS*nj double2)
    const int tile = blockIdx.x, b = blockIdx.y, src = blockIdx.z + src0;
    if (!win_active(g, b)) return;
    const int tx = tile % g.tilesX, ty = tile / g.tilesX;
    const int x0 = tx * TS, y0 = ty * TS;
    const int x1 = min(x0 + TS, g.W), y1 = min(y0 + TS, g.H);
    // this tile's pixels, requested before the weights are staged (their latency hides behind the staging)
    long long pix[TS * TS / NT][2];
    double pixd[TS * TS / NT][2];
#pragma unroll
    for (int k = 0; k < TS * TS / NT; ++k) {
        const int p = threadIdx.x + k * NT;
        const int y = y0 + p / TS, x = x0 + p % TS;
        const bool in = (y < g.H && x < g.W);
        const size_t o = ((size_t)b * g.H * g.W + (size_t)min(y, g.H - 1) * g.W + min(x, g.W - 1)) * 2;
        pix[k][0] = 0; pix[k][1] = 0; pixd[k][0] = 0.0; pixd[k][1] = 0.0;
        if (src == 0) {
            const longlong2 v = *reinterpret_cast<const longlong2*>(gTheta + o);
            if (in) { pix[k][0] = v.x; pix[k][1] = v.y; }
        } else {
            const double2 v = *reinterpret_cast<const double2*>(tvg + o);
            if (in) { pixd[k][0] = v.x; pixd[k][1] = v.y; }
        }
    }
    __shared__ unsigned gms[NWAVE];
    const double gm = gmax_of(gmax + (size_t)b * g.gmax_n, g.gmax_n, gms);       // independent of the rest: its loads go first
    const ResampleTile rs = stage_resample_ranges(RL, h, w, x0, y0, x1, y1, rowtap, coltap);
    stage_resample_weights(RL, rs, h, w, x0, y0, x1, y1, AH, AW);
    const int ilo = rs.ilo, jlo = rs.jlo, ni = rs.ni, nj = rs.nj, ncell = ni * nj;
    unsigned long long* __restrict__ out = reinterpret_cast<unsigned long long*>(src == 0 ? gth_main : gth_tv) + (size_t)b * cap;
    // src 0: integers at the fine per-pixel scale -> the coarser cell scale (one rounding per pixel and weight).
    // src 1: fp64 TV gradient image -> fixed point at tv_shift.
    const double scale = (src == 0) ? ldexp(1.0, grad_shift(wc[b], gm, g.R) - grad_shift_pixel(wc[b], gm, g.R, wide != 0))
                                    : ldexp(1.0, tv_shift(g.H, g.W));
    if (rs.staged) {
        // Separable and ordered: A) every (pixel row, coarse column) sums its row of the tile against A_W, B) every (coarse row,
        // coarse column) sums the 32 partials against A_H - fp64, fixed order, no atomics inside the workgroup, one rounding per
        // tile and cell.  (Every pixel adding its taps x taps products into LDS cell accumulators put 1024 pixels on <= 9
        // addresses: ds_add_u64 at the same-address rate, 28 us per launch.)
        double2* sv = reinterpret_cast<double2*>(cells);                 // (TS*TS) the tile's values at the cell scale, 16 KB
        double2* st = sv + TS * TS;                                      // (TS, nj) column-reduced rows
#pragma unroll
        for (int k = 0; k < TS * TS / NT; ++k) {
            const int p = threadIdx.x + k * NT;
            const int y = y0 + p / TS, x = x0 + p % TS;
            double vx = 0.0, vy = 0.0;
            if (y < g.H && x < g.W) {
                if (src == 0) {
                    if (pix[k][0] != 0 || pix[k][1] != 0) {
                        const size_t o = ((size_t)b * g.H * g.W + (size_t)y * g.W + x) * 2;
                        *reinterpret_cast<longlong2*>(gTheta + o) = make_longlong2(0, 0);   // consumed: zero again for the next evaluation
                        vx = (double)pix[k][0] * scale; vy = (double)pix[k][1] * scale;     // exact conversion for |.| < 2^53 (always, unless `wide`)
                    }
                } else {
                    vx = pixd[k][0] * scale; vy = pixd[k][1] * scale;
                }
            }
            sv[p] = make_double2(vx, vy);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < TS * nj; k += NT) {
            const int ly = k / nj, jj = k - ly * nj;
            double ax = 0.0, ay = 0.0;
            for (int lx = 0; lx < TS; ++lx) {                            // weights outside a pixel's tap range are stored as zeros
                const double wgt = RL.aw[lx * nj + jj];
                const double2 v = sv[ly * TS + lx];
                ax += wgt * v.x; ay += wgt * v.y;
            }
            st[k] = make_double2(ax, ay);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < ncell; k += NT) {
            const int ii = k / nj, jj = k - ii * nj;
            double ax = 0.0, ay = 0.0;
            for (int ly = 0; ly < TS; ++ly) {
                const double wgt = RL.ah[ly * ni + ii];
                const double2 v = st[ly * nj + jj];
                ax += wgt * v.x; ay += wgt * v.y;
            }
            unsigned long long* o = out + ((size_t)(ilo + ii) * w + (jlo + jj)) * 2;
            if (ax != 0.0) atomicAdd(o, (unsigned long long)fix64_wide(ax));
            if (ay != 0.0) atomicAdd(o + 1, (unsigned long long)fix64_wide(ay));
        }
        return;
    }
    // theta grids too fine for the staged weights (more than RS_MAXC coarse rows or columns under one tile): per-pixel products
    const bool use_lds = (ncell > 1 && ncell <= PROJ_CELLS);
    if (use_lds) {
        for (int i = threadIdx.x; i < ncell * 2; i += NT) cells[i] = 0ull;
        __syncthreads();
    }
    double sx1 = 0.0, sy1 = 0.0;                    // single-cell path
#pragma unroll
    for (int k = 0; k < TS * TS / NT; ++k) {
        const int p = threadIdx.x + k * NT;
        const int ly = p / TS, lx = p % TS;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= g.H || x >= g.W) continue;
        const size_t o = ((size_t)b * g.H * g.W + (size_t)y * g.W + x) * 2;
        double vx, vy;
        if (src == 0) {
            const long long ix = pix[k][0], iy = pix[k][1];
            if (ix == 0 && iy == 0) continue;
            *reinterpret_cast<longlong2*>(gTheta + o) = make_longlong2(0, 0);       // consumed: zero again for the next evaluation
            vx = (double)ix * scale; vy = (double)iy * scale;       // (double)ix exact for |.| < 2^53 (always, unless `wide`)
        } else {
            vx = pixd[k][0] * scale; vy = pixd[k][1] * scale;
            if (vx == 0.0 && vy == 0.0) continue;
        }
        const int2 rt = RL.rt[ly], ct = RL.ct[lx];
        for (int i = rt.x; i < rt.y; ++i) {
            const double a = rs.staged ? RL.ah[ly * ni + (i - ilo)] : AH[(size_t)y * h + i];
            for (int j = ct.x; j < ct.y; ++j) {
                const double wt = a * (rs.staged ? RL.aw[lx * nj + (j - jlo)] : AW[(size_t)x * w + j]);
                if (ncell == 1) { sx1 += wt * vx; sy1 += wt * vy; }
                else if (use_lds) {
                    unsigned long long* c = cells + ((i - ilo) * nj + (j - jlo)) * 2;
                    atomicAdd(c, (unsigned long long)fix64_wide(wt * vx)); atomicAdd(c + 1, (unsigned long long)fix64_wide(wt * vy));
                } else {
                    atomicAdd(out + ((size_t)i * w + j) * 2, (unsigned long long)fix64_wide(wt * vx));
                    atomicAdd(out + ((size_t)i * w + j) * 2 + 1, (unsigned long long)fix64_wide(wt * vy));
                }
            }
        }
    }
    if (ncell == 1) {
        sx1 = block_sum(sx1, scratch);
        sy1 = block_sum(sy1, scratch);
        if (threadIdx.x == 0) {
            if (sx1 != 0.0) atomicAdd(out + ((size_t)ilo * w + jlo) * 2, (unsigned long long)fix64_wide(sx1));
            if (sy1 != 0.0) atomicAdd(out + ((size_t)ilo * w + jlo) * 2 + 1, (unsigned long long)fix64_wide(sy1));
        }
    } else if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < ncell * 2; i += NT) {
            const unsigned long long v = cells[i];
            if (v != 0ull) {
                const int c = i & 1, ci = (i >> 1) / nj, cj = (i >> 1) % nj;
                atomicAdd(out + ((size_t)(ilo + ci) * w + (jlo + cj)) * 2 + c, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_final: per-window scalar assembly (losses.py:176-203).  grid (B), one workgroup each.
// Also combines the coarse gradient: grad = gth_main + tv_scale * gth_tv (h*w*2 <= a few thousand values);
// the dense (identity) gradient is combined by k_final_dense.
// ------------------------------------------------------------------------------------------------
constexpr int FT = 512, FW = FT / 64;      // k_final: 8 waves, so that up to 8 reference times are reduced side by side
__global__ __launch_bounds__(FT) void k_final(Geom g, EvalParams ep,
        const StatPart* __restrict__ parts, const double* __restrict__ divparts, const double* __restrict__ tvparts,
        const double* __restrict__ tmm, const WinConst* __restrict__ wc,
        const double* __restrict__ g2parts,    // contrast energy partials from k_imgrad, or nullptr (then parts[].sG2 holds it)
        long long* __restrict__ gth_main, long long* __restrict__ gth_tv, int gth_cap,     // i64 cells (consumed and cleared here)
        const double* __restrict__ g11, const int32_t* __restrict__ win_item0, int n_items, int g11_per_item,  // 2-DoF: per-workgroup partials of the gather kernel
        const unsigned* __restrict__ gmax,
        OutScal* __restrict__ outs, double* __restrict__ grad_out, int want_grad)
{
    __shared__ double scratch[FW];
    __shared__ double sh_tvscale;
    __shared__ double sh_rel[3][16];           // per reference time: rel_contrast, rel_corr, rel_div terms
    const int b = blockIdx.x;
    const WinConst& c = wc[b];
    OutScal* __restrict__ o = outs + b;
    if (!win_active(g, b)) return;
    const double HW = (double)g.H * (double)g.W;
    // This kernel is a latency chain on 8 workgroups; everything that depends on nothing is loaded first so that the loads overlap
    // the per-reference-time reductions: the NaN scan of the velocity bounds and the first partials of the 2-DoF gradient.
    double bad = 0.0;
    for (int i = threadIdx.x; i < g.ntiles * 4; i += FT) {
        const double v = tmm[(size_t)b * g.ntiles * 4 + i];
        if (!(v == v)) bad = 1.0;
    }
    const bool grad11 = want_grad && !ep.identity && ep.h * ep.w == 1;
    double sx11 = 0.0, sy11 = 0.0;
    if (grad11) {
        // 2-DoF: add the partials of this window's k_gather workgroups in index order (fixed strided order per thread, then the
        // fixed tree of block_sum): bit-reproducible without any atomic.  Four loads in flight per trip.
        const int lo = win_item0[b], hi = (b + 1 < g.B) ? win_item0[b + 1] : n_items;
        const double2* __restrict__ q = reinterpret_cast<const double2*>(g11);
        const int kend = hi * g11_per_item;
        int k = lo * g11_per_item + threadIdx.x;
        for (; k + 3 * FT < kend; k += 4 * FT) {
            const double2 a0 = q[k], a1 = q[k + FT], a2 = q[k + 2 * FT], a3 = q[k + 3 * FT];
            sx11 += (a0.x + a1.x) + (a2.x + a3.x); sy11 += (a0.y + a1.y) + (a2.y + a3.y);
        }
        for (; k < kend; k += FT) { const double2 a = q[k]; sx11 += a.x; sy11 += a.y; }
    }
    // one wave per reference time (waves take r = wave, wave + FW, ...): no block-wide barriers inside the loop
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int r = wv; r < g.R; r += FW) {
            double dsum = 0.0, g2sum = 0.0;
            if (ep.want_div) {
                double v = 0.0;
                for (int i = lane; i < g.ntiles; i += 64) v += divparts[((size_t)b * g.R + r) * g.ntiles + i];
                dsum = __shfl(wave_sum(v), 0, 64);
            }
            if (g2parts) {
                double v = 0.0;
                for (int i = lane; i < g.nig; i += 64) v += g2parts[((size_t)b * g.R + r) * g.nig + i];
                g2sum = __shfl(wave_sum(v), 0, 64);
            }
            const ImgScal s = reduce_parts(parts + ((size_t)b * g.R + r) * g.pstride, g.nparts);
            if (lane == 0) {
                const double mse = mse_from_moments(s, c.sE[r], c.sEE[r], HW);
                const double mean = s.sI / HW;
                const double var = s.sII / HW - mean * mean;
                const double cgm = (g2parts ? g2sum : s.sG2) / HW;
                const double con = (ep.contrast_kind == 1) ? var : cgm;
                const double c0 = (ep.contrast_kind == 1) ? c.c0_var : c.c0_gradmag;
                const double dv = ep.want_div ? dsum / HW : NAN;
                o->corr[r] = -mse; o->contrast_gm[r] = cgm; o->var[r] = var; o->div[r] = dv;
                sh_rel[0][r] = c.mrw[r] * con / (c0 + EPSN);
                sh_rel[1][r] = c.mrw[r] * (-mse) / (c.zc[r] + EPSN);
                sh_rel[2][r] = ep.want_div ? c.mrw[r] * dv / (c.d0 + EPSN) : 0.0;
            }
        }
    }
    __syncthreads();
    double sum_rel_con = 0.0, sum_rel_corr = 0.0, sum_rel_div = 0.0;     // live in thread 0
    if (threadIdx.x == 0)
        for (int r = 0; r < g.R; ++r) { sum_rel_con += sh_rel[0][r]; sum_rel_corr += sh_rel[1][r]; sum_rel_div += sh_rel[2][r]; }
    double tv = 0.0, tvscale = 0.0;
    if (ep.want_tv) {
        double a = 0.0, n = 0.0;
        for (int i = threadIdx.x; i < g.ntiles; i += FT) {
            a += tvparts[((size_t)b * g.ntiles + i) * 3];
            n += tvparts[((size_t)b * g.ntiles + i) * 3 + 1];
        }
        a = block_sum<FW>(a, scratch);
        n = block_sum<FW>(n, scratch);
        if (threadIdx.x == 0) { tv = a / (n + EPSN); tvscale = ep.gamma * 0.25 / (n + EPSN); }
    }
    // a NaN anywhere in theta must surface as a NaN loss (the reference propagates it through the warp)
    bad = block_sum<FW>(bad, scratch);
    if (threadIdx.x == 0) {
        const double R = (double)g.R;
        const double mrc = sum_rel_con / R, mrr = sum_rel_corr / R;
        const double mrd = ep.want_div ? sum_rel_div / R : NAN;
        const double tv_in_loss = (ep.cur_pyr_lvl <= 0) ? tv : 0.0;
        double val = (ep.alpha * (-mrc) + ep.beta * (-mrr));
        double reg = 0.0;
        if (ep.gamma != 0.0) reg += ep.gamma * tv_in_loss;
        if (ep.delta != 0.0) reg += ep.delta * mrd;
        val += reg;
        if (bad > 0.0) val = NAN;
        o->mean_rel_contrast = mrc; o->mean_rel_corr = mrr; o->mean_rel_div = mrd;
        o->tv = (ep.cur_pyr_lvl <= 0) ? (ep.want_tv ? tv : NAN) : 0.0;
        o->value = val;
        o->tv_scale = ep.use_tv_grad ? tvscale : 0.0;
        o->nonfinite = (val - val == 0.0) ? 0.0 : 1.0;
        o->_pad = 0.0;
        sh_tvscale = ep.use_tv_grad ? tvscale : 0.0;
    }
    __syncthreads();
    if (want_grad && !ep.identity) {
        const double s = sh_tvscale * ldexp(1.0, -tv_shift(g.H, g.W));
        const int n = ep.h * ep.w * 2;
        if (n == 2) {
            double sx = sx11, sy = sy11;
            sx = block_sum<FW>(sx, scratch);
            sy = block_sum<FW>(sy, scratch);
            if (threadIdx.x == 0) {
                if (ep.use_tv_grad) {
                    sx += s * (double)gth_tv[(size_t)b * gth_cap]; sy += s * (double)gth_tv[(size_t)b * gth_cap + 1];
                    gth_tv[(size_t)b * gth_cap] = 0; gth_tv[(size_t)b * gth_cap + 1] = 0;
                }
                grad_out[(size_t)b * 2] = sx; grad_out[(size_t)b * 2 + 1] = sy;
            }
        } else {
            __shared__ unsigned gms[FW];
            const double inv = ldexp(1.0, -grad_shift(c, gmax_of(gmax + (size_t)b * g.gmax_n, g.gmax_n, gms), g.R));
            for (int i = threadIdx.x; i < n; i += FT) {
                double v = (double)gth_main[(size_t)b * gth_cap + i] * inv;
                gth_main[(size_t)b * gth_cap + i] = 0;
                if (ep.use_tv_grad) { v += s * (double)gth_tv[(size_t)b * gth_cap + i]; gth_tv[(size_t)b * gth_cap + i] = 0; }
                grad_out[(size_t)b * n + i] = v;
            }
        }
    }
}

// dense (identity resample) gradient: grad = gTheta * 2^-eg + tv_scale * tvg; consumes (clears) the i64 image.  grid-stride, grid (nblk, B)
__global__ void k_final_dense(Geom g, int use_tv, int wide, long long* __restrict__ gTheta, const double* __restrict__ tvg,
                              const WinConst* __restrict__ wc, const unsigned* __restrict__ gmax,
                              const OutScal* __restrict__ outs, double* __restrict__ grad_out)
{
    const int b = blockIdx.y;
    const size_t n = (size_t)g.H * g.W * 2;
    const double s = outs[b].tv_scale;
    if (!win_active(g, b)) return;
    __shared__ unsigned gms[NWAVE];
    const double inv = ldexp(1.0, -grad_shift_pixel(wc[b], gmax_of(gmax + (size_t)b * g.gmax_n, g.gmax_n, gms), g.R, wide != 0));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long long q = gTheta[b * n + i];
        if (q != 0) gTheta[b * n + i] = 0;
        double v = (double)q * inv;
        if (use_tv) v += s * tvg[b * n + i];
        grad_out[b * n + i] = v;
    }
}

}  // namespace eincm
