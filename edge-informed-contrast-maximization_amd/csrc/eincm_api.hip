// eincm_api.hip — host engine behind the C-ABI of include/eincm.h (libeincm_hip.so, gfx950 only).
//
// Host responsibilities (everything else runs in the kernels of eincm_kernels.hip.h):
//   - own the HBM layout of a batch of event windows (see DESIGN.md "Data layout in HBM")
//   - bin events by 32x32 source tile once per window and cut the bins into work items
//   - build the scale_and_translate weight matrices (theta_utils.py:25-35) for the current theta shape
//   - launch the evaluation sequence on one stream and hand (value, grad, aux) back as float64
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "eincm.h"
#include "eincm_kernels.hip.h"
#include "eincm_binning.hip.h"
#include "eincm_edges.hip.h"

using namespace eincm;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct eincm_ctx {
    int device = 0;
    int H = 0, W = 0, maxR = 0, maxB = 0;
    int64_t maxN = 0;
    uint32_t cflags = 0;
    int chunk = 4096;              // events per inner chunk of k_splat (u32 accumulation bound): = default splat segment, so no commit pass
    int seg = 0;                   // events per segment (0 = choose per batch); EINCM_SEG / EINCM_CHUNK override
    int seg_used = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // staged batch
    bool staged = false;
    Geom g{};
    int n_items = 0;
    int64_t n_events = 0;
    std::vector<int64_t> win_events;

    // device buffers
    uint32_t* d_xy = nullptr;      // (maxN) x | y<<16, binned by (window, tile); the splat's copy: time order inside a tile, re-dealt in blocks of 256 (k_spread)
    double* d_t = nullptr;         // (maxN)
    uint32_t* d_xy_g = nullptr;    // (maxN) the gather's copy: the same bins, every segment of d_items sorted by source pixel and dealt to its threads (k_segsort)
    double* d_t_g = nullptr;       // (maxN)
    Item* d_items = nullptr;       // (max_items) segments walked by k_gather / k_count / k_mask
    Item* d_items_s = nullptr;     // (max_items) shorter segments walked by k_splat
    int32_t* d_order = nullptr;    // (max_items) d_items by decreasing length: the order the event kernels' workgroups take them in
    int32_t* d_order_s = nullptr;  // the same for d_items_s
    std::vector<int32_t> h_order, h_order_s, h_tilecount;   // host sides of the two (kept until the upload has completed)
    std::vector<int32_t> h_win_item0;                       // host copy of d_win_item0
    Window* d_wins = nullptr;      // (max_items, maxR) destination windows of the gather segments under the current theta
    Window* d_wins_s = nullptr;    // (max_items, maxR) ... of the splat segments
    int n_items_s = 0; int seg_s = 0; int seg_s_used = 0;
    // third list: the segments the 2-DoF gather walks, on the SPLAT's copy of the events (it has no per-pixel accumulators, so the
    // time-ordered copy serves it, and it wants shorter segments than the theta-grid gather does: round-2 tuning)
    Item* d_items_2 = nullptr; int32_t* d_order_2 = nullptr; int32_t* d_win_item0_2 = nullptr;
    bool policy_evaluated = false; // an evaluation has chosen capacities since the last staging (eincm_get_launch_policy)
    int pitch_policy = 0;          // the staged batch is in the regime where the bank-aligned LDS pitch pays (set_windows_impl); eval_begin decides per evaluation
    double tspan_s = 1.0, tspan_sh = 1.0, tspan_a = 1.0, tspan_2 = 1.0;   // time span (fraction of the window) the window capacity is sized for, per segment list (span_quantile)
    Item* d_items_sh = nullptr; int32_t* d_order_sh = nullptr; int n_items_sh = 0; int seg_sh_used = 0;   // the splat's SHORT list (8192) beside a 16384-event
                                                                       // one: a 2-DoF theta too large for the long segments' windows walks it (launch_forward)
    int n_items_2 = 0; int seg_2_used = 0;
    std::vector<int32_t> h_order_2, h_win_item0_2;
    int wincap_2 = WIN_CAP_DEFAULT;
    int wincap = WIN_CAP_DEFAULT;
    bool wincap_fixed = false;     // EINCM_WINCAP pins the capacity; otherwise it is chosen per evaluation from max|theta|
    bool chunk_fixed = false;      // EINCM_CHUNK given
    int g11_per_item = 1;          // slots per segment in d_g11 written by the last gather launch
    double gather_wg_events = 4096.0;   // events a 2-DoF gather workgroup should take (set_windows: by batch size)
    // device-side staging (eincm_binning.hip.h)
    int16_t* d_raw_x = nullptr; int16_t* d_raw_y = nullptr; double* d_raw_t = nullptr;   // (maxN) events as handed over
    BinBlock* d_binblocks = nullptr; int32_t* d_win_blk = nullptr; uint32_t* d_blockhist = nullptr;
    int32_t* d_tilecount = nullptr; int32_t* d_tilebase = nullptr; int32_t* d_itembase = nullptr; int32_t* d_itembase_s = nullptr; int32_t* d_bin_misc = nullptr;
    bool itembase_valid = false;   // d_itembase / d_itembase_s hold the first segment of every (window, tile) for the staged batch
    double* d_edges_raw = nullptr; double* d_edge_moments = nullptr;
    int64_t max_binblocks = 0;
    bool host_binning = false;
    int64_t max_items = 0;
    float* d_edges = nullptr;      // (B,R,H,W)
    double* d_edge_ts = nullptr;   // (B,R)
    unsigned long long* d_acc = nullptr;   // (B,R,H,W) u64 fixed-point accumulator of the IWE stack (2^30); zero between evaluations (consumer-clears)
    float* d_iwe = nullptr;        // (B,R,H,W) the fp32 IWE stack, written by the statistics pass from d_acc
    float* d_G = nullptr;          // (B,R,H,W)
    float* d_zero_iwe = nullptr;   // (B,H,W)
    double* d_Theta = nullptr;     // (B,H,W,2)
    double* d_theta_in = nullptr;  // (B,H,W,2) capacity (coarse uses a prefix)
    long long* d_gTheta = nullptr; // (B,H,W,2) i64 fixed point, zero between evaluations (k_project / k_final_dense clear it)
    double* d_g11 = nullptr;       // (max_items, R, 2) 2-DoF theta: per-workgroup partials of dL/dtheta (k_gather -> k_final)
    int32_t* d_win_item0 = nullptr;// (B) first segment of each window in d_items
    double* d_dtmax = nullptr;     // (B) staging scratch: max |t - tau| per window
    unsigned* d_gmax = nullptr;    // (B,R,nig) per-strip max |dL/dIWE| as float bits (scale of the i64 gradient accumulators)
    unsigned* d_cntmax = nullptr;  // (B) staging scratch: most events on one source pixel
    unsigned* d_amax = nullptr;    // (B,R,pstride) max |A| per k_imstat workgroup (float bits)
    unsigned* d_gbound = nullptr;  // (B,R) bound of max |dL/dIWE| per image from k_imstat's tail (float bits): what gmax is when gmax_n == R
    unsigned* d_ticket = nullptr;  // (B,R) arrival counters of k_imstat, zero between launches
    unsigned* d_gticket = nullptr; // (B) arrival counters of the gather's tail (theta grids), zero between launches
    ImgCoef* d_coef = nullptr;     // (B,R) what the gather composes dL/dIWE with
    float* d_Gimg = nullptr;       // (B,R,H,W) dL/dIWE materialised for eincm_get_image_grad after a composed evaluation (allocated on demand)
    bool last_composed = false;    // the last gradient evaluation left A in d_G (the gather composed dL/dIWE on the fly)
    EvalParams last_ep{};
    Geom last_g{};                 // geometry of the last gradient evaluation (nparts, gmax_n)
    double* d_tvg = nullptr;       // (B,H,W,2)
    uint8_t* d_mask = nullptr;     // (B,H,W)
    double* d_tmm = nullptr;       // (B,ntiles,4)
    StatPart* d_parts = nullptr;   // (B,R,ntiles)
    double* d_divparts = nullptr;  // (B,R,ntiles)
    double* d_g2parts = nullptr;   // (B,R,nig) contrast energy partials written by k_imgrad
    float* d_gdiv = nullptr;       // (B,R,H,W) divergence adjoint image, allocated on first delta != 0 gradient
    double* d_dgparts = nullptr;   // (B,R,ntiles,2)
    double* d_tvparts = nullptr;   // (B,ntiles,3)
    WinConst* d_wc = nullptr;      // (B)
    OutScal* d_outs = nullptr;     // (B)
    long long* d_gth = nullptr;    // (2,B,maxcoarse) main | tv i64 accumulators for coarse theta, zero between evaluations (k_final clears)
    double* d_grad = nullptr;      // (B,H,W,2) capacity
    double* d_AH = nullptr; double* d_AW = nullptr;     // (H,h) (W,w) capacity H*H, W*W? -> sized on demand
    int2* d_rowtap = nullptr; int2* d_coltap = nullptr;
    TileRange* d_tilerng = nullptr;    // (ntiles) coarse cells under each tile for the current theta shape
    const double* theta_dev_in = nullptr;   // eincm_loss_grad_device: theta of the evaluation being begun lives in HBM (the caller's buffer)
    double vmax_hint = -1.0;                // ... and this bounds |theta| for the window-capacity choice (< 0: unknown, largest windows)
    double* grad_dev_out = nullptr;         // ... and the gradient goes there (device to device)
    bool device_results = false;       // eincm_set_device_results: results stay in HBM until eincm_finish_collect (event-sharded mode over RCCL)
    bool proj_in_gather = false;       // every tile touches <= PG_MAXC x PG_MAXC cells: k_gather projects its tile itself
    size_t AH_cap = 0, AW_cap = 0;
    int cur_h = -1, cur_w = -1, cur_method = -1;
    int64_t coarse_cap = 0;        // doubles per window in d_gth halves

    // scratch of the edge-smoothing / tiled-objective entry points (eincm_edges.hip.h), grown on demand
    DevBuf e_u8, e_g, e_sq, e_misc, e_a, e_b, e_kern, e_out;

    // pinned host staging
    double* h_theta = nullptr;     // (B,H,W,2) capacity
    double* h_grad = nullptr;
    OutScal* h_outs = nullptr;
    WinConst* h_wc = nullptr;
    // host-assembled evaluations (2-DoF theta, no TV / divergence / full aux): the kernels write their partials straight into these
    // and the host adds them in index order after the stream has drained (no k_final, no k_theta_const: two launches fewer per evaluation)
    double* h_g11 = nullptr;       // (max_items + NXCD, R, 2) per-workgroup partials of dL/dtheta (k_gather)
    double* h_g2 = nullptr;        // (B,R,nig) contrast energy per k_imgrad strip
    double* h_img = nullptr;       // (B,R,IMGSCAL_N) reduced image scalars (k_imgrad)
    double* h_tvparts = nullptr;   // (B,ntiles,3) k_tv's per-tile partials (host-assembled evaluations with the TV term)

    // timing.  EINCM_CF_TIMING_DOMINANT keeps a ring of event sets and reads them out later (eincm_get_timings*): asking HIP for
    // elapsed times after every evaluation cost the caller ~15 us per evaluation, which a throughput measurement should not pay.
    static constexpr int EV_RING = 64;
    hipEvent_t ev[EV_RING][EINCM_N_STAGES + 1][2] = {};
    bool ev_used[EV_RING][EINCM_N_STAGES + 1] = {};
    int ring_size = 1;             // EV_RING in the dominant mode, 1 otherwise (read out at once)
    int ring_lo = 0, ring_n = 0;   // finished evaluations whose events have not been read yet: slots ring_lo .. ring_lo + ring_n - 1
    int ring_cur = 0;              // slot of the evaluation in flight
    static constexpr int GRAD_PIECES = 4;
    hipEvent_t ev_piece[GRAD_PIECES] = {};   // dense gradients come back in pieces; the host scans piece k while piece k + 1 crosses PCIe
    int n_pieces = 0;              // pieces of the evaluation in flight (0: the gradient came with the results / in one copy)
    size_t piece_len = 0;
    int attach_stage = -1;         // EINCM_CF_TIMING: the single-kernel stage whose launch takes its events along (StageTimer)
    int time_period = 1; int64_t time_counter = 0; bool timed_now = true;   // EINCM_CF_TIMING_DOMINANT: events on every time_period-th evaluation only (eincm_set_timing_period)
    bool time_splat = true, time_gather = true;   // EINCM_CF_TIMING_DOMINANT: which event kernels carry start / stop events (eincm_set_timed_kernels)
    bool have_events = false;
    eincm_timings last_t{};
    eincm_timings sum_t{};         // running sums since the last reset (eincm_get_timings_total)
    int64_t sum_n = 0;

    // last eval bookkeeping
    bool have_eval = false;
    bool G_valid = false;          // d_G holds dL/dIWE of the last evaluation (eincm_get_count_images borrows the buffer)
    int last_nparts = 0;           // how many StatParts per image the last evaluation wrote (k_stats vs k_stats_stream)
    // an evaluation split in two halves (eval_begin ... [caller may all-reduce the IWE stack] ... eval_end)
    struct { bool active = false; bool launched = false; EvalParams ep{}; int h = 0, w = 0; bool identity = false, want_grad = false, full_aux = false, div_grad = false;
             bool pal_2 = false;                    // the 2-DoF gather's windows at the bank-aligned pitch in this evaluation
             bool splat_short = false;              // this evaluation's k_splat walks the short segment list (d_items_sh)
             bool host_asm = false;                 // scalar assembly and the 2-DoF gradient sum on the host (see h_g11)
             bool composed = false;                 // k_imstat + composing gather (host_assemble: the contrast energy rides in h_img)
             bool tv_projected = false;             // k_tv projected its gradient onto the theta cells itself (no k_project for it)
             int copy_mode = 0;                     // 1: the D2H copies of the results are still to be enqueued (device_results)
             bool use_arg = false; const double* theta_dev = nullptr; ThetaArg targ{};      // where the event kernels find a 2-DoF theta
             } pend;
    std::vector<uint8_t> theta_nan;    // (B) a NaN / Inf somewhere in window b's theta (host-assembled evaluations)
    // host-side wall time of the phases of an evaluation (eincm_get_host_profile): a few clock reads per evaluation, always on
    double hp_us[EINCM_N_HOST_PHASES] = {};
    int64_t hp_n = 0;
    bool constants_pending = false;   // staged with EINCM_SW_DEFER_CONSTANTS and not finished yet
    bool acc_dirty = false;        // a forward half was launched and its consumers were not: accumulators must be memset before reuse
    bool Theta_valid = false;      // d_Theta holds the upsampled theta of the last evaluation (2-DoF evaluations skip the image)
    std::vector<double> last_theta11;   // (B,2) theta of the last 2-DoF evaluation (to build d_Theta on demand)
};

namespace {

using hp_clock = std::chrono::steady_clock;
struct HostPhase {                 // adds the lifetime of the object to ctx->hp_us[phase]
    eincm_ctx* c; int phase; hp_clock::time_point t0;
    HostPhase(eincm_ctx* c_, int p) : c(c_), phase(p), t0(hp_clock::now()) {}
    ~HostPhase() { c->hp_us[phase] += std::chrono::duration<double, std::micro>(hp_clock::now() - t0).count(); }
};

int fail(eincm_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((c), EINCM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                       \
    } while (0)

template <typename T> hipError_t dalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)); }

// ---------------------------------------------------------------------------------------------
// jax.image.scale_and_translate per-axis weight matrix (S7; theta_utils.py:25-35), fp64 on the host.
// A is (n_out, n_in): out = A @ in.
// ---------------------------------------------------------------------------------------------
double kern_eval(int method, double x) {
    switch (method) {
        case EINCM_METHOD_BILINEAR: return std::max(0.0, 1.0 - std::fabs(x));
        case EINCM_METHOD_LANCZOS3:
        case EINCM_METHOD_LANCZOS5: {
            const double radius = (method == EINCM_METHOD_LANCZOS3) ? 3.0 : 5.0;
            if (x > radius) return 0.0;
            if (!(x > 1e-3)) return 1.0;
            const double y = radius * std::sin(M_PI * x) * std::sin(M_PI * x / radius);
            return y / (M_PI * M_PI * x * x);
        }
        case EINCM_METHOD_CUBIC: {
            if (x >= 2.0) return 0.0;
            if (x >= 1.0) return ((-0.5 * x + 2.5) * x - 4.0) * x + 2.0;
            return ((1.5 * x - 2.5) * x) * x + 1.0;
        }
    }
    return 0.0;
}

void resample_matrix(int n_in, int n_out, int method, std::vector<double>& A) {
    A.assign((size_t)n_out * n_in, 0.0);
    const double scale = (double)n_out / (double)n_in;
    const double inv_scale = 1.0 / scale;
    const double kernel_scale = std::max(inv_scale, 1.0);
    const double thresh = 1000.0 * 1.1920928955078125e-07;   // 1000 * float32 eps
    for (int o = 0; o < n_out; ++o) {
        const double sample_f = ((double)o + 0.5) * inv_scale - 0.5;
        double total = 0.0;
        for (int i = 0; i < n_in; ++i) {
            const double x = std::fabs(sample_f - (double)i) / kernel_scale;
            const double wgt = kern_eval(method, x);
            A[(size_t)o * n_in + i] = wgt;
            total += wgt;
        }
        const bool inside = (sample_f >= -0.5) && (sample_f <= (double)n_in - 0.5);
        for (int i = 0; i < n_in; ++i) {
            double& a = A[(size_t)o * n_in + i];
            a = (std::fabs(total) > thresh) ? a / (total != 0.0 ? total : 1.0) : 0.0;
            if (!inside) a = 0.0;
        }
    }
}

void multi_ref_weights(int R, double* w) {
    double s = 0.0;
    for (int r = 0; r < R; ++r) {
        // np.linspace(-1.5, 1.5, R): start + r*step with step = 3/(R-1); R == 1 -> [-1.5]
        const double x = (R > 1) ? (-1.5 + (double)r * (3.0 / (double)(R - 1))) : -1.5;
        w[r] = std::exp(-0.5 * x * x) / std::sqrt(2.0 * M_PI);
        s += w[r];
    }
    for (int r = 0; r < R; ++r) w[r] /= s;
}

void free_all(eincm_ctx* c) {
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(c->d_xy); F(c->d_t); F(c->d_xy_g); F(c->d_t_g); F(c->d_items); F(c->d_items_s); F(c->d_items_2); F(c->d_order_2); F(c->d_win_item0_2); F(c->d_items_sh); F(c->d_order_sh); F(c->d_order); F(c->d_order_s); F(c->d_wins); F(c->d_wins_s); F(c->d_raw_x); F(c->d_raw_y); F(c->d_raw_t); F(c->d_binblocks); F(c->d_win_blk);
    F(c->d_blockhist); F(c->d_tilecount); F(c->d_tilebase); F(c->d_itembase); F(c->d_itembase_s); F(c->d_bin_misc); F(c->d_edges_raw); F(c->d_edge_moments); F(c->d_edges); F(c->d_edge_ts); F(c->d_acc); F(c->d_iwe); F(c->d_G); F(c->d_zero_iwe);
    F(c->d_g11); F(c->d_win_item0); F(c->d_dtmax); F(c->d_gmax); F(c->d_cntmax); F(c->d_amax); F(c->d_gbound); F(c->d_ticket); F(c->d_gticket); F(c->d_coef); F(c->d_Gimg);
    F(c->d_Theta); F(c->d_theta_in); F(c->d_gTheta); F(c->d_tvg); F(c->d_mask); F(c->d_tmm); F(c->d_parts);
    F(c->d_divparts); F(c->d_g2parts); F(c->d_gdiv); F(c->d_dgparts); F(c->d_tvparts); F(c->d_wc); F(c->d_outs); c->d_grad = nullptr; F(c->d_gth); F(c->d_AH); F(c->d_AW);
    F(c->d_rowtap); F(c->d_coltap); F(c->d_tilerng);
    for (DevBuf* b : {&c->e_u8, &c->e_g, &c->e_sq, &c->e_misc, &c->e_a, &c->e_b, &c->e_kern, &c->e_out}) { F(b->p); b->bytes = 0; }
    auto FH = [](auto*& p) { if (p) { (void)hipHostFree(p); p = nullptr; } };
    FH(c->h_theta); FH(c->h_outs); c->h_grad = nullptr; FH(c->h_wc); FH(c->h_g11); FH(c->h_g2); FH(c->h_img); FH(c->h_tvparts);
    if (c->have_events) {
        for (int k = 0; k < eincm_ctx::EV_RING; ++k)
            for (int i = 0; i <= EINCM_N_STAGES; ++i)
                for (int e = 0; e < 2; ++e)
                    if (c->ev[k][i][e]) { (void)hipEventDestroy(c->ev[k][i][e]); c->ev[k][i][e] = nullptr; }
        for (int k = 0; k < eincm_ctx::GRAD_PIECES; ++k) if (c->ev_piece[k]) { (void)hipEventDestroy(c->ev_piece[k]); c->ev_piece[k] = nullptr; }
        c->have_events = false;
    }
    if (c->stream) { (void)hipStreamDestroy(c->stream); c->stream = nullptr; }
}

// EINCM_CF_TIMING: a stage made of ONE kernel launch (single = true) gets its start / stop events attached to that launch
// (launch_timed: the dispatch's own timestamps, no packets on the stream); any other stage is bracketed by marker events, whose
// barrier packets add a few microseconds to the interval they measure and to the evaluation.
struct StageTimer {
    eincm_ctx* c; int stage; bool on, single;
    StageTimer(eincm_ctx* c_, int s, bool single_ = false) : c(c_), stage(s), on((c_->cflags & EINCM_CF_TIMING) != 0), single(single_) {
        if (on && single) c->attach_stage = stage;
        else if (on) { (void)hipEventRecord(c->ev[c->ring_cur][stage][0], c->stream); }
    }
    ~StageTimer() {
        if (on && single) c->attach_stage = -1;
        else if (on) { (void)hipEventRecord(c->ev[c->ring_cur][stage][1], c->stream); c->ev_used[c->ring_cur][stage] = true; }
    }
};

// EINCM_CF_TIMING_DOMINANT: the two event kernels are launched with their own start / stop events (hipExtLaunchKernelGGL: the
// dispatch's completion signal carries the timestamps).  Marker events around them (hipEventRecord) cost 25 us per evaluation in
// barrier packets and lost launch overlap; attached events cost ~6 us per timed kernel.
static inline bool timing_on(const eincm_ctx* c) {
    return (c->cflags & EINCM_CF_TIMING) != 0 || ((c->cflags & EINCM_CF_TIMING_DOMINANT) != 0 && c->timed_now);
}

template <typename K, typename... Args>
void launch_timed(eincm_ctx* c, int stage, K kernel, dim3 grid, dim3 block, size_t lds, Args... args) {
    const bool attach = (c->cflags & EINCM_CF_TIMING) ? c->attach_stage == stage
                      : ((c->cflags & EINCM_CF_TIMING_DOMINANT) && c->timed_now) ? (stage == EINCM_STAGE_SPLAT ? c->time_splat : stage == EINCM_STAGE_GATHER && c->time_gather)
                      : false;
    if (attach) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, c->stream, c->ev[c->ring_cur][stage][0], c->ev[c->ring_cur][stage][1], 0,
                              args...);
        c->ev_used[c->ring_cur][stage] = true;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, c->stream, args...);
    }
}

int ensure_resample(eincm_ctx* c, int h, int w, int method) {
    if (c->cur_h == h && c->cur_w == w && c->cur_method == method) return EINCM_OK;
    const int H = c->H, W = c->W;
    std::vector<double> AH, AW;
    resample_matrix(h, H, method, AH);
    resample_matrix(w, W, method, AW);
    std::vector<int2> rt(H), ct(W);
    for (int y = 0; y < H; ++y) {
        int lo = h, hi = 0;
        for (int i = 0; i < h; ++i) if (AH[(size_t)y * h + i] != 0.0) { lo = std::min(lo, i); hi = std::max(hi, i + 1); }
        if (lo >= hi) { lo = 0; hi = 0; }
        rt[y] = make_int2(lo, hi);
    }
    for (int x = 0; x < W; ++x) {
        int lo = w, hi = 0;
        for (int j = 0; j < w; ++j) if (AW[(size_t)x * w + j] != 0.0) { lo = std::min(lo, j); hi = std::max(hi, j + 1); }
        if (lo >= hi) { lo = 0; hi = 0; }
        ct[x] = make_int2(lo, hi);
    }
    if (AH.size() > c->AH_cap) {
        if (c->d_AH) { (void)hipFree(c->d_AH); c->d_AH = nullptr; }
        HIPCHK(c, dalloc(&c->d_AH, AH.size()));
        c->AH_cap = AH.size();
    }
    if (AW.size() > c->AW_cap) {
        if (c->d_AW) { (void)hipFree(c->d_AW); c->d_AW = nullptr; }
        HIPCHK(c, dalloc(&c->d_AW, AW.size()));
        c->AW_cap = AW.size();
    }
    HIPCHK(c, hipMemcpyAsync(c->d_AH, AH.data(), AH.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_AW, AW.data(), AW.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_rowtap, rt.data(), rt.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_coltap, ct.data(), ct.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
    // the coarse cells under every 32x32 tile (what k_gather's own projection walks)
    const int tilesX = (W + TS - 1) / TS, tilesY = (H + TS - 1) / TS;
    std::vector<TileRange> tr((size_t)tilesX * tilesY);
    bool fits = true;
    for (int ty = 0; ty < tilesY; ++ty)
        for (int tx = 0; tx < tilesX; ++tx) {
            int ilo = h, ihi = 0, jlo = w, jhi = 0;
            for (int y = ty * TS; y < std::min(ty * TS + TS, H); ++y) if (rt[y].y > rt[y].x) { ilo = std::min(ilo, rt[y].x); ihi = std::max(ihi, rt[y].y); }
            for (int x = tx * TS; x < std::min(tx * TS + TS, W); ++x) if (ct[x].y > ct[x].x) { jlo = std::min(jlo, ct[x].x); jhi = std::max(jhi, ct[x].y); }
            TileRange q;
            q.ilo = ilo < ihi ? ilo : 0; q.ni = std::max(ihi - ilo, 0); q.jlo = jlo < jhi ? jlo : 0; q.nj = std::max(jhi - jlo, 0);
            fits = fits && q.ni <= PG_MAXC && q.nj <= PG_MAXC;
            tr[(size_t)ty * tilesX + tx] = q;
        }
    HIPCHK(c, hipMemcpyAsync(c->d_tilerng, tr.data(), tr.size() * sizeof(TileRange), hipMemcpyHostToDevice, c->stream));
    c->proj_in_gather = fits && !getenv("EINCM_NO_PROJ_IN_GATHER");
    HIPCHK(c, hipStreamSynchronize(c->stream));     // host vectors go out of scope
    c->cur_h = h; c->cur_w = w; c->cur_method = method;
    return EINCM_OK;
}

constexpr size_t ZERO_COPY_MAX = 65536;   // doubles of theta / gradient that cross PCIe by zero-copy access to pinned host memory (a 64-window batch at 16x16: 32768)

// The event kernels' workgroups take the segments by decreasing length (block_to_work): stable counting sort of the lengths.
void order_by_length(const std::vector<int32_t>& lens, std::vector<int32_t>& order) {
    int32_t maxlen = 0;
    for (int32_t l : lens) maxlen = std::max(maxlen, l);
    std::vector<int32_t> start((size_t)maxlen + 2, 0);
    for (int32_t l : lens) ++start[(size_t)(maxlen - l) + 1];
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    order.resize(lens.size());
    for (size_t i = 0; i < lens.size(); ++i) order[(size_t)start[(size_t)(maxlen - lens[i])]++] = (int32_t)i;
}
// segment lengths of a (window, tile) population list, in the order k_items emits the segments
void segment_lengths(const std::vector<int32_t>& tilecount, int seg, std::vector<int32_t>& lens) {
    lens.clear();
    for (int32_t cnt : tilecount) {
        const int len = balanced_seg_len(cnt, seg);
        for (int s0 = 0; s0 < cnt; s0 += len) lens.push_back(std::min(len, cnt - s0));
    }
}

// A segment list from the (window, tile) populations, the way k_items emits it (time ranges left to k_seg_minmax); win_item0: first
// segment of every window.
void host_items(const std::vector<int32_t>& tilecount, int ntiles, int seg, std::vector<Item>& items, std::vector<int32_t>& win_item0) {
    items.clear(); win_item0.clear();
    int64_t base = 0;
    for (size_t idx = 0; idx < tilecount.size(); ++idx) {
        if (idx % (size_t)ntiles == 0) win_item0.push_back((int32_t)items.size());
        const int cnt = tilecount[idx];
        const int len = balanced_seg_len(cnt, seg);
        for (int s0 = 0; s0 < cnt; s0 += len) {
            Item it;
            it.win = (int32_t)(idx / (size_t)ntiles); it.tile = (int32_t)(idx % (size_t)ntiles);
            it.begin = (int32_t)(base + s0); it.count = std::min(len, cnt - s0); it.t_lo = 0.0; it.t_hi = 0.0;
            items.push_back(it);
        }
        base += cnt;
    }
}

// blocks of the two event kernels: every (segment, reference time) pair, padded to a multiple of 8 segments (block_to_work)
unsigned event_grid(const eincm_ctx* c) { return (unsigned)(((c->n_items + NXCD - 1) / NXCD) * NXCD * c->g.R); }
unsigned splat_grid(const eincm_ctx* c) { return (unsigned)(((c->n_items_s + NXCD - 1) / NXCD) * NXCD * c->g.R); }

// Every cross-workgroup accumulator (u64 IWE stack, i64 dL/dTheta, i64 coarse cells) is zero between evaluations because its
// consumer clears it.  If a forward half was launched and never consumed (error between the two halves), clear them here.
int clear_accumulators(eincm_ctx* c) {
    const size_t img = (size_t)c->H * c->W;
    HIPCHK(c, hipMemsetAsync(c->d_acc, 0, (size_t)c->maxB * c->maxR * img * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_gTheta, 0, (size_t)c->maxB * img * 2 * sizeof(long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_gth, 0, (size_t)2 * c->maxB * c->coarse_cap * sizeof(long long), c->stream));
    c->acc_dirty = false;
    return EINCM_OK;
}

// theta -> Theta image (+ per-tile velocity bounds) for every window.  theta_dev: (B,h,w,2) on the device or in mapped host memory.
// with_windows: k_theta also fills the window tables of both segment lists (every theta but the 2-DoF one, whose k_theta_const does)
void launch_theta_image(eincm_ctx* c, int h, int w, bool identity, bool use_arg, const ThetaArgBig& targ, const double* theta_dev,
                        bool with_windows) {
    const Geom& g = c->g;
    const bool ww = with_windows && c->itembase_valid;
#define THETA_ARGS(T_) dim3(g.ntiles, g.B), dim3(NT), 0, g, h, w, identity ? 1 : 0, use_arg ? 1 : 0, T_, \
                 theta_dev, c->d_AH, c->d_AW, c->d_rowtap, c->d_coltap, c->d_tilerng, c->d_Theta, c->d_tmm, c->d_edge_ts, \
                 c->n_items, c->d_items, ww ? c->d_itembase : nullptr, c->d_wins, \
                 c->n_items_s, c->d_items_s, ww ? c->d_itembase_s : nullptr, c->d_wins_s
    // the argument block is copied by value into the launch and again into the kernarg buffer: 4 KiB where theta fits (one window at 16x16)
    if (!use_arg || (size_t)g.B * h * w * 2 <= (size_t)THETA_ARG_MID) {
        ThetaArgMid mid;
        if (use_arg) memcpy(mid.v, targ.v, (size_t)g.B * h * w * 2 * sizeof(double));
        launch_timed(c, EINCM_STAGE_THETA, k_theta<ThetaArgMid>, THETA_ARGS(mid));
    } else {
        launch_timed(c, EINCM_STAGE_THETA, k_theta<ThetaArgBig>, THETA_ARGS(targ));
    }
#undef THETA_ARGS
    c->Theta_valid = true;
}

// Launch the forward half: theta -> Theta -> u64 IWE accumulator.
// need_theta_image: somebody will read d_Theta (TV term); 2-DoF evaluations otherwise skip the image altogether.
int launch_forward(eincm_ctx* c, int h, int w, bool identity, bool need_theta_image, const double* theta_host, bool host_asm) {
    const Geom& g = c->g;
    const size_t nth = (size_t)h * w * 2;
    const bool use_arg = !c->theta_dev_in && !identity && (size_t)g.B * nth <= (size_t)THETA_ARG_MAX;
    const bool const_theta = !identity && h == 1 && w == 1;
    const double* theta_dev = c->d_theta_in;
    ThetaArg targ;
    // k_theta alone takes a larger theta in its own arguments (a 16x16 grid of one window: 4 KiB); the event kernels read the Theta image
    static const bool no_big_arg = getenv("EINCM_NO_BIG_THETA_ARG") != nullptr;
    const bool use_arg_big = !c->theta_dev_in && !identity && (size_t)g.B * nth <= (size_t)THETA_ARG_BIG && !no_big_arg;
    static thread_local ThetaArgBig targ_big;
    if (use_arg_big) memcpy(targ_big.v, theta_host, (size_t)g.B * nth * sizeof(double));
    if (c->theta_dev_in) {
        theta_dev = c->theta_dev_in;                 // device-resident theta: the kernels read the caller's buffer, nothing crosses PCIe
        if (const_theta) need_theta_image = true;    // (no host copy of theta for ensure_theta_image to rebuild the image from)
    } else if (use_arg) {
        memcpy(targ.v, theta_host, (size_t)g.B * nth * sizeof(double));     // theta rides in the kernel arguments
        if (!use_arg_big) { memcpy(c->h_theta, theta_host, (size_t)g.B * nth * sizeof(double)); theta_dev = c->h_theta; }   // (k_theta reads memory then)
    } else if (use_arg_big) {
        theta_dev = c->h_theta;                      // (not read: k_theta has theta in its arguments)
    } else if ((size_t)g.B * nth <= ZERO_COPY_MAX) {
        // medium theta (e.g. 16x16): k_theta reads it straight from the pinned, GPU-mapped staging buffer (no copy command)
        memcpy(c->h_theta, theta_host, (size_t)g.B * nth * sizeof(double));
        theta_dev = c->h_theta;
    } else {
        // dense theta (4.9 MB at 480x640): staged through pinned memory in a few pieces, so that the DMA of one piece runs while the
        // host copies the next (one memcpy + one DMA back to back: 131 + 187 us)
        StageTimer t(c, EINCM_STAGE_COPY);
        const size_t total = (size_t)g.B * nth;
        const size_t piece = std::max<size_t>((total + 5) / 6, (size_t)32768);
        for (size_t off = 0; off < total; off += piece) {
            const size_t n = std::min(piece, total - off);
            memcpy(c->h_theta + off, theta_host + off, n * sizeof(double));
            HIPCHK(c, hipMemcpyAsync(c->d_theta_in + off, c->h_theta + off, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
    }
    {
        StageTimer t(c, EINCM_STAGE_THETA, const_theta ? !need_theta_image : c->itembase_valid);       // one kernel in either case
        const int nwin_threads = (c->n_items + c->n_items_s) * g.R;
        if (const_theta) {
            if (theta_host) c->last_theta11.assign(theta_host, theta_host + (size_t)g.B * 2); else c->last_theta11.clear();
            c->Theta_valid = false;
            if (need_theta_image) launch_theta_image(c, h, w, identity, use_arg_big, targ_big, theta_dev, false);
            // the event kernels derive their windows from theta themselves; the velocity bounds (tmm) only feed k_final's NaN scan,
            // which a host-assembled evaluation does on the host
            if (!host_asm)
                launch_timed(c, EINCM_STAGE_THETA, k_theta_const, dim3((std::max(g.B * g.ntiles, nwin_threads) + NT - 1) / NT), dim3(NT), 0, g,
                                   use_arg ? 1 : 0, targ, theta_dev, c->d_tmm, c->d_edge_ts, c->n_items, c->d_items, c->d_wins,
                                   c->n_items_s, c->d_items_s, c->d_wins_s);
        } else {
            launch_theta_image(c, h, w, identity, use_arg_big, targ_big, theta_dev, true);
            if (nwin_threads > 0 && !c->itembase_valid)
                hipLaunchKernelGGL(k_windows, dim3((nwin_threads + NT - 1) / NT), dim3(NT), 0, c->stream, g, c->d_tmm, c->d_edge_ts,
                                   c->n_items, c->d_items, c->d_wins, c->n_items_s, c->d_items_s, c->d_wins_s);
        }
    }
    {
        StageTimer t(c, EINCM_STAGE_SPLAT, true);
        c->acc_dirty = true;
        if (c->n_items_s > 0) {
            const int theta_mode = const_theta ? THETA_CONST : THETA_TILE;
            const int lds_multi = (c->seg_s_used > c->chunk) ? 1 : 0;      // segments longer than a chunk need the f32 commit window
            const size_t lds_bytes = (size_t)(lds_multi ? 2 : 1) * g.wincap * sizeof(float)
                                   + (theta_mode == THETA_TILE ? TS * TS * sizeof(double2) : 0);
            const bool sshort = c->pend.splat_short && const_theta;
            const int n_sp = sshort ? c->n_items_sh : c->n_items_s;
            const Item* items_sp = sshort ? c->d_items_sh : c->d_items_s;
            const int32_t* order_sp = sshort ? c->d_order_sh : c->d_order_s;
            const unsigned grid_sp = (unsigned)(((n_sp + NXCD - 1) / NXCD) * NXCD * g.R);
#define SPLAT_ARGS(NTH) dim3(grid_sp), dim3(NTH), lds_bytes, g, n_sp, c->chunk, theta_mode, lds_multi, \
                   items_sp, c->d_xy, c->d_t, c->d_Theta, c->d_tmm, c->d_edge_ts, c->d_wins_s, c->d_acc, order_sp, \
                   use_arg ? 1 : 0, theta_dev, targ
            static const bool merge_env = getenv("EINCM_SPLAT_MERGE") != nullptr;
            if (merge_env && !lds_multi && c->seg_used <= MAX_CHUNK && c->n_items > 0) {
                // experiment: the splat walks the gather's list and copy and merges same-destination taps in registers
                Geom gs = g; gs.wincap = g.wincap_a; gs.winmaxw = g.winmaxw_a;
                const size_t lds_m = (size_t)gs.wincap * sizeof(float) + (theta_mode == THETA_TILE ? TS * TS * sizeof(double2) : 0);
#define MERGE_ARGS dim3(event_grid(c)), dim3(512), lds_m, gs, c->n_items, MAX_CHUNK, theta_mode, 0, \
                   c->d_items, c->d_xy_g, c->d_t_g, c->d_Theta, c->d_tmm, c->d_edge_ts, c->d_wins, c->d_acc, c->d_order, \
                   use_arg ? 1 : 0, theta_dev, targ
                if (theta_mode == THETA_CONST) launch_timed(c, EINCM_STAGE_SPLAT, k_splat<THETA_CONST, 0, 512, 1>, MERGE_ARGS);
                else                           launch_timed(c, EINCM_STAGE_SPLAT, k_splat<THETA_TILE, 0, 512, 1>, MERGE_ARGS);
#undef MERGE_ARGS
            } else
            if (lds_multi)                      launch_timed(c, EINCM_STAGE_SPLAT, k_splat<0, 1, NT, 0>, SPLAT_ARGS(NT));      // long segments (EINCM_SEG_SPLAT > EINCM_CHUNK)
            // 512 threads per workgroup in both compile-time modes: 93 vs 94 us on the 8-window batch, 18.8 vs 23.2 us on one window
            // (1024: 102 us; the gather is slower with 512: 93.5 vs 81.7 us)
            else if (theta_mode == THETA_CONST) launch_timed(c, EINCM_STAGE_SPLAT, k_splat<THETA_CONST, 0, 512, 0>, SPLAT_ARGS(512));
            else                                launch_timed(c, EINCM_STAGE_SPLAT, k_splat<THETA_TILE, 0, 512, 0>, SPLAT_ARGS(512));
#undef SPLAT_ARGS
        }
    }
    HIPCHK(c, hipGetLastError());
    c->pend.use_arg = use_arg; c->pend.theta_dev = theta_dev; c->pend.targ = targ;
    return EINCM_OK;
}

// Read one finished evaluation's events (its stream work has been waited for) into last_t and the running sums.
int read_event_slot(eincm_ctx* c, int k) {
    memset(&c->last_t, 0, sizeof c->last_t);
    for (int s = 0; s <= EINCM_N_STAGES; ++s) {
        if (!c->ev_used[k][s]) continue;
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[k][s][0], c->ev[k][s][1]));
        if (s < EINCM_N_STAGES) c->last_t.ms[s] = ms; else c->last_t.total_ms = ms;
    }
    for (int s = 0; s < EINCM_N_STAGES; ++s) c->sum_t.ms[s] += c->last_t.ms[s];
    c->sum_t.total_ms += c->last_t.total_ms;
    ++c->sum_n;
    return EINCM_OK;
}
int drain_event_ring(eincm_ctx* c, int keep) {          // read the oldest finished evaluations until at most `keep` are left
    while (c->ring_n > keep) {
        const int rc = read_event_slot(c, c->ring_lo);
        c->ring_lo = (c->ring_lo + 1) % c->ring_size; --c->ring_n;
        if (rc) return rc;
    }
    return EINCM_OK;
}
// The evaluation in flight has finished (stream waited for): its events join the ring; read at once unless the mode defers it.
int collect_timings(eincm_ctx* c) {
    if (!timing_on(c)) return EINCM_OK;
    ++c->ring_n;
    return (c->ring_size == 1) ? drain_event_ring(c, 0) : EINCM_OK;
}

// First half of an evaluation: theta -> Theta -> IWE stack (k_theta, k_splat).  theta_host: (B,h,w,2) doubles.
int eval_begin(eincm_ctx* c, const double* theta_host, int h, int w, const eincm_params* p, bool want_grad, const uint8_t* active = nullptr) {
    HostPhase hp(c, EINCM_HP_BEGIN);
    {   // which windows take part (eincm_loss_grad_masked); the others' workgroups leave at once, their outputs are not written
        unsigned long long m = ~0ull;
        if (active) { m = 0ull; for (int b = 0; b < c->g.B && b < 64; ++b) if (active[b]) m |= 1ull << b; }
        c->g.wmask = m;
    }
    const Geom& g = c->g;
    const bool identity = (h == g.H && w == g.W);
    const size_t img = (size_t)g.H * g.W;
    const size_t nth = (size_t)h * w * 2;
    const bool full_aux = (p->flags & EINCM_PF_FULL_AUX) != 0;
    const bool div_grad = (p->delta != 0.0 && want_grad);
    if (c->pend.active && c->pend.launched)
        return fail(c, EINCM_ERR_STATE, "an asynchronous evaluation is in flight: call eincm_loss_grad_wait first");
    c->pend.active = false; c->pend.launched = false;
    if (div_grad && !c->d_gdiv) {      // rare path (the reference keeps delta = 0, configs/main.yaml:19): allocate lazily
        HIPCHK(c, dalloc(&c->d_gdiv, (size_t)c->maxB * c->maxR * img));
        HIPCHK(c, dalloc(&c->d_dgparts, (size_t)c->maxB * c->maxR * g.ntiles * 2));
    }
    if (p->contrast_kind != EINCM_CONTRAST_GRAD_MAG && p->contrast_kind != EINCM_CONTRAST_VARIANCE)
        return fail(c, EINCM_ERR_ARG, "contrast_kind %d unknown", p->contrast_kind);
    if (!identity) {
        if ((int64_t)h * w > (int64_t)g.H * g.W)
            return fail(c, EINCM_ERR_ARG, "theta (%d,%d,2) has more cells than the %dx%d sensor has pixels: not supported", h, w, g.H, g.W);
        if ((int64_t)nth > c->coarse_cap) {          // unusual (the pyramid tops out at 16x16): grow the coarse accumulators
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_gth) { (void)hipFree(c->d_gth); c->d_gth = nullptr; }
            HIPCHK(c, dalloc(&c->d_gth, (size_t)2 * c->maxB * nth));
            HIPCHK(c, hipMemset(c->d_gth, 0, (size_t)2 * c->maxB * nth * sizeof(long long)));
            c->coarse_cap = (int64_t)nth;
        }
        int rc = ensure_resample(c, h, w, p->method);
        if (rc) return rc;
    }
    if (c->acc_dirty) { const int rcd = clear_accumulators(c); if (rcd) return rcd; }
    if (c->cflags & EINCM_CF_TIMING_DOMINANT) c->timed_now = (c->time_counter++ % c->time_period) == 0;
    const bool timing = timing_on(c);
    if (timing) {
        if (c->ring_n == c->ring_size) { const int rcr = drain_event_ring(c, c->ring_size - 1); if (rcr) return rcr; }   // ring full: read the oldest
        c->ring_cur = (c->ring_lo + c->ring_n) % c->ring_size;
        for (int s = 0; s <= EINCM_N_STAGES; ++s) c->ev_used[c->ring_cur][s] = false;
        if (c->ring_size == 1) (void)hipEventRecord(c->ev[c->ring_cur][EINCM_N_STAGES][0], c->stream);    // EINCM_CF_TIMING only
    }

    EvalParams ep{};
    ep.alpha = p->alpha; ep.beta = p->beta; ep.gamma = p->gamma; ep.delta = p->delta;
    ep.cur_pyr_lvl = p->cur_pyr_lvl; ep.contrast_kind = p->contrast_kind;
    ep.want_div = (full_aux || p->delta != 0.0) ? 1 : 0;
    ep.want_tv = ((p->cur_pyr_lvl <= 0) && (p->gamma != 0.0 || full_aux)) ? 1 : 0;
    ep.use_tv_grad = (ep.want_tv && p->gamma != 0.0 && want_grad && !(p->flags & EINCM_PF_NO_TV_GRAD)) ? 1 : 0;
    ep.h = h; ep.w = w; ep.identity = identity ? 1 : 0;

    // LDS window capacity for this evaluation: the host knows theta, hence the largest displacement a segment can see.
    // Small windows give 8 workgroups per CU; windows too small for the flow push taps onto the slow direct-to-HBM path.
    c->policy_evaluated = true;
    c->pend.splat_short = false;
    c->pend.pal_2 = c->pitch_policy >= 2;            // (a pinned capacity, EINCM_WINCAP: the pitch as staged)
    c->g.pitch_aligned = c->pitch_policy != 0 ? 1 : 0;
    if (!c->wincap_fixed) {
        double vmax = 0.0;
        const size_t nall = (size_t)g.B * nth;
        const size_t stride = nall > 8192 ? nall / 8192 : 1;            // dense theta: sample (any capacity is correct; 65536 samples cost 90 us)
        if (c->theta_dev_in) vmax = (c->vmax_hint >= 0.0 && std::isfinite(c->vmax_hint)) ? c->vmax_hint : 1e9;      // unknown: the largest windows
        else if (stride == 1) { for (size_t i = 0; i < nall; ++i) { const double a = std::fabs(theta_host[i]); vmax = std::max(vmax, a <= 1.7e308 ? a : 0.0); } }   // (vectorises)
        else for (size_t i = 0; i < nall; i += stride) { const double a = std::fabs(theta_host[i]); if (a > vmax && std::isfinite(a)) vmax = a; }
        // time span the splat's windows are sized for: what all but 3 % of the events' segments stay within (span_quantile; the mean tile
        // would size them for the dense tiles alone and send the taps of the sparse ones, whose single segment spans the whole window, to HBM)
        const double tspan = c->tspan_s;
        // LDS holds pitch x height words per window, the pitch being the width, or the width rounded up to the 32 banks (win_pitch).
        // The aligned pitch is taken where the batch is in its regime (pitch_policy) AND it does not push the window into a larger
        // capacity class: its gain is a few per cent of bank conflicts, a class costs workgroups per CU (8 windows of 10^6 events whose
        // theta needs 66-pixel windows: 96 x 66 words = the 6912 class, 5 workgroups per CU, 2-DoF gather 62 -> 68 us).
        auto lds_words = [](bool pal, double side_px) { const double sd = std::ceil(side_px); return (pal ? std::ceil(sd / 32.0) * 32.0 : sd) * sd; };
        static const int caps[] = {2304, 3072, 4608, 6912};      // 6912 keeps k_gather's LDS (window + accumulators + Theta tile) under 64 KiB
        auto cap_of = [&](double need, int floor_k) { for (int k = floor_k; k < 4; ++k) if (need <= caps[k]) return caps[k]; return caps[3]; };
        double side = TS + 4 + vmax * tspan;
        const bool two_dof = h == 1 && w == 1 && !identity;
        // a 2-DoF theta whose spread over a long splat segment outgrows the largest window: the short list (half the time span); the taps
        // of a window that is too small go to HBM one by one (117 px per window: 1220 us on the long list, 544 us on the short one)
        c->pend.splat_short = (two_dof && c->n_items_sh > 0 && lds_words(false, side) > 6912.0);
        if (c->pend.splat_short) side = TS + 4 + vmax * c->tspan_sh;
        // Where a larger window costs no residency it is taken at once (a capacity is an allocation, the windows themselves stay as small
        // as their segments need): the 2-DoF kernels hold nothing but the window in LDS, 4608 words = 18 KiB still gives the 8 workgroups
        // of 4 waves a CU can hold; the theta-grid gather carries 32 KiB beside its window and runs 3 workgroups per CU up to 5461 words.
        // The theta-grid splat (window + 16 KiB Theta tile) pays for capacity with workgroups per CU (6 / 5 / 4 / 3), so it takes what it needs.
        const int floor_k = two_dof ? 2 : 0;
        int cap = cap_of(lds_words(false, side), floor_k);
        const bool pal_s = c->pitch_policy != 0 && lds_words(true, side) <= (double)cap;
        // long splat segments (EINCM_SEG_SPLAT > EINCM_CHUNK) keep a second, f32 window: 2 * cap * 4 B + the 16 KiB Theta tile must
        // stay within the 64 KiB of dynamic LDS a launch gets without an attribute
        if (c->seg_s_used > c->chunk) cap = std::min(cap, 4608);
        c->g.wincap = cap;
        c->g.winmaxw = std::max(40, (int)std::lround(std::sqrt((double)cap * 1.4)));
        c->g.pitch_aligned = pal_s ? 1 : 0;
        // the gather's own list: longer segments see a longer time span, hence a larger displacement spread; a window too small for it
        // sends taps down the direct path, where a composed dL/dIWE costs three loads per tap
        const double side_a = TS + 4 + vmax * c->tspan_a;
        const int cap_a = cap_of(side_a * side_a, 2);          // (the theta-grid gather's windows: pitch = width)
        c->g.wincap_a = cap_a;
        c->g.winmaxw_a = std::max(40, (int)std::lround(std::sqrt((double)cap_a * 1.4)));
        // and the 2-DoF gather's list
        const double side_2 = TS + 4 + vmax * c->tspan_2;
        const int cap_2 = cap_of(lds_words(false, side_2), 2);
        // (the 2-DoF gather keeps pitch = width: at the aligned pitch it measured equal on the bench batch and 63 -> 68 us on another batch
        // of the same shape, profiles/r03/pitch_by_shape.txt; EINCM_PITCH_ALIGNED=2 aligns it too)
        c->pend.pal_2 = c->pitch_policy >= 2 && lds_words(true, side_2) <= (double)cap_2;
        c->wincap_2 = cap_2;
    }
    // 2-DoF theta with nothing but the contrast and correlation terms (every level above 0 of the reference's pyramid at its first
    // level, and the bench workload): the scalar assembly and the sum of the gather's per-workgroup partials run on the host
    static const bool no_host_asm = getenv("EINCM_NO_HOST_ASM") != nullptr;
    const bool grid_tail = !(h == 1 && w == 1) && c->proj_in_gather && c->itembase_valid && (size_t)g.B * nth <= ZERO_COPY_MAX && !c->theta_dev_in;
    // (the TV term rides along on a theta grid: k_tv projects its own gradient and the gather's tail combines it; a 2-DoF theta with
    // TV - no level of the reference's pyramid - keeps k_final)
    const bool host_asm = want_grad && !identity && (((h == 1 && w == 1) && !ep.want_tv) || grid_tail) && !ep.want_div && !full_aux && !no_host_asm &&
                          !c->device_results && !c->theta_dev_in;
    if (host_asm) {
        c->theta_nan.assign((size_t)g.B, 0);
        for (int b = 0; b < g.B; ++b) {            // branch-free (vectorises): an OR over the exponent bits, 4096 values at 16x16 x 8 windows
            const double* __restrict__ tb = theta_host + (size_t)b * nth;
            uint64_t bad = 0;
            for (size_t i = 0; i < nth; ++i) {
                uint64_t u;
                memcpy(&u, tb + i, sizeof u);
                bad |= (uint64_t)((u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
            }
            c->theta_nan[b] = bad != 0;
        }
    }
    int rc = launch_forward(c, h, w, identity, ep.want_tv != 0, theta_host, host_asm);
    if (rc) return rc;
    c->pend.active = true; c->pend.ep = ep; c->pend.h = h; c->pend.w = w; c->pend.identity = identity;
    c->pend.want_grad = want_grad; c->pend.full_aux = full_aux; c->pend.div_grad = div_grad; c->pend.host_asm = host_asm;
    return EINCM_OK;
}

// The D2H copies of an evaluation whose results k_final left in HBM (d_outs, d_grad).  Enqueued right behind the kernels, or - with
// eincm_set_device_results - only by eincm_finish_collect, after the caller has all-reduced the gradient in HBM.
int enqueue_result_copies(eincm_ctx* c) {
    const Geom& g = c->g;
    const bool want_grad = c->pend.want_grad;
    const size_t nth = (size_t)c->pend.h * c->pend.w * 2;
    c->n_pieces = 0;
    if (c->theta_dev_in) {                           // eincm_loss_grad_device: scalars to the host, the gradient device to device
        HIPCHK(c, hipMemcpyAsync(c->h_outs, c->d_outs, (size_t)g.B * sizeof(OutScal), hipMemcpyDeviceToHost, c->stream));
        if (want_grad && c->grad_dev_out)
            HIPCHK(c, hipMemcpyAsync(c->grad_dev_out, c->d_grad, (size_t)g.B * nth * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return EINCM_OK;
    }
    if (want_grad && (size_t)g.B * nth >= ((size_t)1 << 17)) {
        // a dense gradient (4.9 MB at 480x640): in pieces, an event behind each, so that eval_end_collect hands piece k over
        // (copy + finite scan on the host) while piece k + 1 is still crossing PCIe
        HIPCHK(c, hipMemcpyAsync(c->h_outs, c->d_outs, (size_t)g.B * sizeof(OutScal), hipMemcpyDeviceToHost, c->stream));
        const size_t total = (size_t)g.B * nth;
        c->piece_len = (total + eincm_ctx::GRAD_PIECES - 1) / eincm_ctx::GRAD_PIECES;
        for (int k = 0; k < eincm_ctx::GRAD_PIECES; ++k) {
            const size_t off = (size_t)k * c->piece_len;
            if (off >= total) break;
            const size_t n = std::min(c->piece_len, total - off);
            HIPCHK(c, hipMemcpyAsync(c->h_grad + off, c->d_grad + off, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->ev_piece[k], c->stream));
            c->n_pieces = k + 1;
        }
    } else if (want_grad && g.B == c->maxB) {      // outs and grad are contiguous: one copy
        HIPCHK(c, hipMemcpyAsync(c->h_outs, c->d_outs, (size_t)g.B * sizeof(OutScal) + (size_t)g.B * nth * sizeof(double),
                                 hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(c->h_outs, c->d_outs, (size_t)g.B * sizeof(OutScal), hipMemcpyDeviceToHost, c->stream));
        if (want_grad)
            HIPCHK(c, hipMemcpyAsync(c->h_grad, c->d_grad, (size_t)g.B * nth * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    return EINCM_OK;
}

// Second half, part 1: enqueue image statistics, dL/dIWE, gather, projection, scalar assembly (on whatever is in the IWE
// stack now) and the copy back to pinned memory.  Returns without synchronising.
int eval_end_launch(eincm_ctx* c) {
    if (!c->pend.active) return fail(c, EINCM_ERR_STATE, "no evaluation in flight");
    if (c->pend.launched) return EINCM_OK;
    HostPhase hp(c, EINCM_HP_LAUNCH);
    Geom g = c->g;
    const EvalParams ep = c->pend.ep;
    const int h = c->pend.h, w = c->pend.w;
    const bool identity = c->pend.identity, want_grad = c->pend.want_grad, full_aux = c->pend.full_aux, div_grad = c->pend.div_grad;
    const size_t nth = (size_t)h * w * 2;
    const bool timing = timing_on(c);
    const bool host_asm = c->pend.host_asm;
    // The image pass of a gradient evaluation.  Default: k_stats_stream -> k_imgrad -> gather.  EINCM_COMPOSE=1 selects the fused
    // form built in round 3: k_imstat (statistics + the stats-independent part of dL/dIWE in one kernel, the per-image scalars in its
    // tail) and a gather that composes dL/dIWE while staging its windows - one dependent kernel fewer, but measured no faster
    // (one window of 10^6 events: 72.4 vs 72.2 us; the 8-window batch 236 vs 231 us: the gather pays three loads per window pixel
    // and k_imstat's tail as much as the kernel boundary it saves), so it stays an option.
    // delta != 0 and forward-only evaluations always take the unfused kernels.
    static const bool compose_env = getenv("EINCM_COMPOSE") != nullptr;
    const bool compose = want_grad && !div_grad && compose_env;
    const bool g2_from_imgrad = !compose && want_grad && ep.contrast_kind == EINCM_CONTRAST_GRAD_MAG;
    const bool zero_copy_out = !c->device_results && !c->theta_dev_in && !identity && (size_t)g.B * nth <= ZERO_COPY_MAX;
    const bool stream_stats = host_asm || (g2_from_imgrad && g.ntiles >= NSPART);
    const int n_imwg = (g.nig + IG_NT / 64 - 1) / (IG_NT / 64);
    unsigned* gmax_buf = compose ? c->d_gbound : c->d_gmax;
    g.gmax_n = compose ? g.R : g.R * g.nig;
    c->pend.composed = compose;
    {
        StageTimer t(c, EINCM_STAGE_STATS, compose || stream_stats);
        // Either way the statistics pass is the consumer of the u64 accumulator: it leaves the fp32 IWE stack in d_iwe (k_imstat
        // leaves the clearing to the gather, the others clear the accumulator themselves).
        if (compose) {
            g.nparts = n_imwg;
            launch_timed(c, EINCM_STAGE_STATS, k_imstat, dim3(n_imwg, g.R, g.B), dim3(IG_NT), 0, g, ep.contrast_kind == EINCM_CONTRAST_GRAD_MAG ? 1 : 0,
                         c->d_acc, c->d_edges, c->d_iwe, c->d_G, c->d_parts, c->d_amax, c->d_ticket, ep, c->d_wc, c->d_coef, c->d_gbound,
                         host_asm ? c->h_img : nullptr);
        } else if (stream_stats) {
            // gradient evaluations with the grad-mag contrast take the contrast energy from k_imgrad (which computes the Scharr
            // images anyway), so the statistics are a pure streaming reduction with NSPART fat partials per image
            g.nparts = NSPART;
            launch_timed(c, EINCM_STAGE_STATS, k_stats_stream, dim3(NSPART, g.R, g.B), dim3(NT), 0, g, c->d_acc, c->d_iwe, c->d_edges,
                         c->d_parts);
        } else {
            g.nparts = g.ntiles;
            const size_t ntot = (size_t)g.B * g.R * g.H * g.W;
            hipLaunchKernelGGL(k_iwe_finish, dim3((unsigned)std::min<size_t>((ntot + NT - 1) / NT, 2048)), dim3(NT), 0, c->stream, g,
                               c->d_acc, c->d_iwe);
            hipLaunchKernelGGL(k_stats, dim3(g.ntiles, g.R, g.B), dim3(NT), 0, c->stream, g, c->d_iwe, c->d_edges, c->d_parts,
                               g2_from_imgrad ? 0 : 1);
        }
        c->last_nparts = g.nparts;
    }

    if (ep.want_div) {
        hipLaunchKernelGGL(k_div, dim3(g.ntiles, g.R, g.B), dim3(NT), 0, c->stream, g, c->d_iwe, c->d_parts, c->d_divparts);
    }
    if (ep.want_tv) {
        StageTimer t(c, EINCM_STAGE_TV, true);
        // theta grids coarse enough for it: k_tv projects its tile's gradient onto the theta cells itself (no image, no k_project)
        const bool tv_proj = ep.use_tv_grad && !identity && !(h == 1 && w == 1) && c->proj_in_gather;
#define TV_ARGS dim3(g.ntiles, g.B), dim3(NT), 0, g, c->d_Theta, c->d_mask, c->d_tvg, c->d_tvparts, full_aux ? 1 : 0, \
                h, w, c->d_AH, c->d_AW, c->d_tilerng, c->d_gth + (size_t)c->maxB * c->coarse_cap, (int)c->coarse_cap, host_asm ? c->h_tvparts : nullptr
        if (tv_proj) launch_timed(c, EINCM_STAGE_TV, k_tv<1>, TV_ARGS);
        else         launch_timed(c, EINCM_STAGE_TV, k_tv<0>, TV_ARGS);
#undef TV_ARGS
        c->pend.tv_projected = tv_proj;
    }
    const bool direct11 = want_grad && !identity && h == 1 && w == 1;
    const bool proj = want_grad && !identity && !direct11 && c->proj_in_gather;      // k_gather projects its tile's sums itself
    // a window with a handful of events: 61-bit fixed point in the per-pixel gradient sums (grad_shift_pixel); speed is irrelevant there
    bool wide = false;
    for (int b = 0; b < g.B; ++b) wide = wide || (c->win_events[b] * (int64_t)g.R < 4096);
    if (want_grad) {
        if (!compose) {
            StageTimer t(c, EINCM_STAGE_IMGRAD, !div_grad);
            if (div_grad)
                hipLaunchKernelGGL(k_divgrad, dim3(g.ntiles, g.R, g.B), dim3(NT), 0, c->stream, g, c->d_iwe, c->d_parts,
                                   c->d_gdiv, c->d_dgparts);
            launch_timed(c, EINCM_STAGE_IMGRAD, k_imgrad, dim3(n_imwg, g.R, g.B), dim3(IG_NT), 0, g, ep, c->d_iwe, c->d_edges,
                               c->d_parts, c->d_wc, c->d_gdiv, c->d_dgparts, host_asm ? c->h_g2 : c->d_g2parts, c->d_G, c->d_gmax,
                               host_asm ? c->h_img : nullptr, host_asm ? 1 : 0);
        }
        {
            StageTimer t(c, EINCM_STAGE_GATHER, true);
            // 44 KB of LDS (G window + i64 accumulators + Theta tile) fit 3 workgroups per CU: 512 threads each keep 24 waves there
            constexpr int NT_TILE = 512;
            if (c->n_items > 0) {
                // Theta grids / dense theta: the gather walks whichever segment list has the longer segments (its per-workgroup
                // costs - Theta tile, accumulator clear and flush - want them long even for one window, where the 2-DoF gather
                // wants 4096); nothing downstream depends on the list (the per-segment partials are a 2-DoF matter).
                // Theta grids / dense theta: the gather walks its own segment list (long segments) on its own copy of the events
                // (k_segsort); 2-DoF theta: its third list (shorter segments) on the splat's copy
                const int n_g = direct11 ? c->n_items_2 : c->n_items;
                const Item* items_g = direct11 ? c->d_items_2 : c->d_items;
                const Window* wins_g = c->d_wins;                  // (2-DoF theta derives its windows itself)
                const int32_t* order_g = direct11 ? c->d_order_2 : c->d_order;
                const uint32_t* xy_g = direct11 ? c->d_xy : c->d_xy_g;
                const double* t_g = direct11 ? c->d_t : c->d_t_g;
                Geom gg = g;
                if (direct11) { gg.pitch_aligned = c->pend.pal_2 ? 1 : 0; gg.wincap_a = c->wincap_2; gg.winmaxw_a = std::max(40, (int)std::lround(std::sqrt((double)c->wincap_2 * 1.4))); }
                // theta grids with the in-gather projection on big launches: one workgroup per segment for all reference times (k_gather, all_r)
                static const int all_r_env = getenv("EINCM_GATHER_ALL_R") ? atoi(getenv("EINCM_GATHER_ALL_R")) : -1;
                // (8 windows of 10^6 events at 16x16: 792 workgroups of 5 reference times each instead of 3960: 148 -> 135 us; one window:
                // 99 workgroups, 29 -> 84 us - so only where the segments alone fill the chip's 768 workgroup slots of this kernel)
                int all_r = (!direct11 && proj && !compose && !identity && n_g >= 700) ? 1 : 0;
                if (all_r_env >= 0) all_r = (all_r_env && !direct11 && proj && !compose && !identity) ? 1 : 0;
                const unsigned grid_g = (unsigned)(((n_g + NXCD - 1) / NXCD) * NXCD * (all_r ? 1 : g.R));
                // 2-DoF theta: workgroups per segment, so that a workgroup takes about what the round-2 tuning found best for this
                // kernel (4096 events on one window, 16384 on the 8-window batch) whatever the segment length of the list
                int nparts = 1;
                if (direct11) {
                    const double per_tile = (double)std::max<int64_t>(c->n_events, 1) / ((double)g.B * g.ntiles);
                    const double seg_eff = std::min((double)c->seg_2_used, per_tile);
                    const double target = c->gather_wg_events;
                    nparts = seg_eff >= 3.0 * target ? 4 : (seg_eff >= 1.5 * target ? 2 : 1);
                    if (const char* e = getenv("EINCM_GATHER_PARTS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) nparts = v; }
                }
#define GATHER_ARGS(NTH) dim3(grid_g, nparts), dim3(NTH), \
                    gg.wincap_a * sizeof(float) + (direct11 ? 0 : TS * TS * 2 * sizeof(double) + TS * TS * sizeof(double2)), \
                    gg, n_g, items_g, xy_g, t_g, c->d_Theta, c->d_tmm, c->d_edge_ts, c->d_G, wins_g, c->d_gTheta, \
                    direct11 ? 1 : 0, host_asm ? c->h_g11 : c->d_g11, c->d_wc, gmax_buf, direct11 ? THETA_CONST : THETA_TILE, order_g, \
                    c->pend.use_arg ? 1 : 0, c->pend.theta_dev, c->pend.targ, \
                    ep.contrast_kind == EINCM_CONTRAST_GRAD_MAG ? 1 : 0, c->d_edges, c->d_iwe, c->d_coef, c->d_acc, 1, nparts, \
                    h, w, c->d_AH, c->d_AW, c->d_tilerng, c->d_gth, (int)c->coarse_cap, \
                    (host_asm && proj) ? 1 : 0, c->d_gticket, c->d_win_item0, c->h_grad, \
                    (host_asm && proj && ep.use_tv_grad) ? ep.gamma : 0.0, c->d_tvparts, c->d_gth + (size_t)c->maxB * c->coarse_cap, all_r
#define GATHER_TILE(WIDE_, COMPOSE_, PROJ_) launch_timed(c, EINCM_STAGE_GATHER, k_gather<THETA_TILE, WIDE_, NT_TILE, COMPOSE_, PROJ_>, GATHER_ARGS(NT_TILE))
#define GATHER_ALLR(WIDE_) launch_timed(c, EINCM_STAGE_GATHER, k_gather<THETA_TILE, WIDE_, NT_TILE, 0, 1, 1>, GATHER_ARGS(NT_TILE))
                if (direct11) {
                    if (compose) launch_timed(c, EINCM_STAGE_GATHER, k_gather<THETA_CONST, 0, NT, 1, 0>, GATHER_ARGS(NT));
                    else         launch_timed(c, EINCM_STAGE_GATHER, k_gather<THETA_CONST, 0, NT, 0, 0>, GATHER_ARGS(NT));
                    c->g11_per_item = g.R * nparts;
                } else if (all_r) {
                    if (wide) GATHER_ALLR(1); else GATHER_ALLR(0);
                } else if (wide) {
                    if (compose) { if (proj) GATHER_TILE(1, 1, 1); else GATHER_TILE(1, 1, 0); }
                    else         { if (proj) GATHER_TILE(1, 0, 1); else GATHER_TILE(1, 0, 0); }
                } else {
                    if (compose) { if (proj) GATHER_TILE(0, 1, 1); else GATHER_TILE(0, 1, 0); }
                    else         { if (proj) GATHER_TILE(0, 0, 1); else GATHER_TILE(0, 0, 0); }
                }
#undef GATHER_TILE
#undef GATHER_ALLR
#undef GATHER_ARGS
            }
        }
        // accumulators: two halves (event gradient | TV gradient), each (maxB, coarse_cap) i64, zero on entry (k_final clears them).
        // 2-DoF theta: k_gather left per-workgroup partials of the event gradient for k_final; only the TV image needs projecting.
        // (a theta grid coarse enough for k_gather's own projection leaves only the TV image to k_project)
        const bool events_projected = direct11 || proj;
        const bool tv_needs_project = ep.use_tv_grad && !(ep.want_tv && c->pend.tv_projected);
        const int nsrc = (events_projected ? 0 : 1) + (tv_needs_project ? 1 : 0);
        if (!identity && nsrc > 0) {
            StageTimer t(c, EINCM_STAGE_PROJECT, true);
            launch_timed(c, EINCM_STAGE_PROJECT, k_project, dim3(g.ntiles, g.B, nsrc), dim3(NT), 0, g, h, w,
                               (int)c->coarse_cap, events_projected ? 1 : 0, wide ? 1 : 0, c->d_AH, c->d_AW, c->d_rowtap, c->d_coltap, c->d_gTheta, c->d_tvg,
                               c->d_wc, gmax_buf, c->d_gth, c->d_gth + (size_t)c->maxB * c->coarse_cap);
        }
    }
    if (!host_asm) {
        StageTimer t(c, EINCM_STAGE_FINAL, !(want_grad && identity));
        // small results (everything but a dense gradient) are written by k_final straight into pinned host memory: no D2H copy command
        launch_timed(c, EINCM_STAGE_FINAL, k_final, dim3(g.B), dim3(FT), 0, g, ep, c->d_parts, c->d_divparts, c->d_tvparts,
                           c->d_tmm, c->d_wc, g2_from_imgrad ? c->d_g2parts : nullptr, c->d_gth, c->d_gth + (size_t)c->maxB * c->coarse_cap, (int)c->coarse_cap,
                           c->d_g11, c->d_win_item0_2, c->n_items_2, c->g11_per_item, gmax_buf,
                           zero_copy_out ? c->h_outs : c->d_outs, zero_copy_out ? c->h_grad : c->d_grad, want_grad ? 1 : 0);
        if (want_grad && identity) {
            hipLaunchKernelGGL(k_final_dense, dim3(256, g.B), dim3(NT), 0, c->stream, g, ep.use_tv_grad, wide ? 1 : 0, c->d_gTheta,
                               c->d_tvg, c->d_wc, gmax_buf, c->d_outs, c->d_grad);
        }
    }
    if (want_grad) { c->last_composed = compose; c->last_ep = ep; c->last_g = g; }
    HIPCHK(c, hipGetLastError());
    c->n_pieces = 0;
    c->pend.copy_mode = (zero_copy_out || host_asm) ? 0 : 1;
    if (c->pend.copy_mode && !c->device_results) { const int rcc = enqueue_result_copies(c); if (rcc) return rcc; c->pend.copy_mode = 0; }
    if (timing && c->ring_size == 1) { (void)hipEventRecord(c->ev[c->ring_cur][EINCM_N_STAGES][1], c->stream); c->ev_used[c->ring_cur][EINCM_N_STAGES] = true; }
    c->pend.launched = true;
    c->acc_dirty = false;          // every accumulator this evaluation touched has been consumed (and cleared) by the kernels above
    return EINCM_OK;
}

// Host-assembled evaluations (pend.host_asm): what k_final does for them, in fp64 on the host from the partials the kernels wrote
// into pinned memory - the image scalars of k_imgrad (h_img), its per-strip contrast energies (h_g2) and the per-workgroup partials
// of the 2-DoF gradient (h_g11) - every sum in index order, so the result is a function of the partials alone (bit-reproducible).
// losses.py:176-203 without the TV / divergence terms (the caller routed those evaluations to k_final).
void host_assemble(eincm_ctx* c) {
    const Geom& g = c->g;
    const EvalParams& ep = c->pend.ep;
    const double HW = (double)g.H * (double)g.W, Rd = (double)g.R;
    for (int b = 0; b < g.B; ++b) {
        const WinConst& wc = c->h_wc[b];
        OutScal& o = c->h_outs[b];
        memset(&o, 0, sizeof o);
        if (b < 64 && !((g.wmask >> b) & 1ull)) continue;       // sat this evaluation out (eval_end_collect marks its outputs)
        double sum_rel_con = 0.0, sum_rel_corr = 0.0;
        for (int r = 0; r < g.R; ++r) {
            const double* q = c->h_img + ((size_t)b * g.R + r) * IMGSCAL_N;
            double zero_img[IMGSCAL_N] = {0.0, 0.0, EPSN, HW, HW, 0.0, 0.0, 0.0, 0.0};
            if (c->pend.composed && c->win_events[b] == 0) q = zero_img;      // no segment, no gather workgroup: the IWE is identically zero
            ImgScal s{};
            s.m = q[0]; s.M = q[1]; s.D = q[2]; s.cm = q[3]; s.cM = q[4]; s.sI = q[5]; s.sII = q[6]; s.sEI = q[7];
            double g2 = 0.0;
            if (c->pend.composed) {
                g2 = q[8];                           // k_imstat's per-workgroup energies, reduced with the other image scalars
            } else {
                const int nwg = (g.nig + IG_NT / 64 - 1) / (IG_NT / 64);          // one partial per k_imgrad workgroup
                const double* g2p = c->h_g2 + ((size_t)b * g.R + r) * nwg;
                for (int i = 0; i < nwg; ++i) g2 += g2p[i];
            }
            const double mse = mse_from_moments(s, wc.sE[r], wc.sEE[r], HW);
            const double mean = s.sI / HW;
            const double var = s.sII / HW - mean * mean;
            const double cgm = g2 / HW;
            const double con = (ep.contrast_kind == 1) ? var : cgm;
            const double c0 = (ep.contrast_kind == 1) ? wc.c0_var : wc.c0_gradmag;
            o.corr[r] = -mse; o.contrast_gm[r] = cgm; o.var[r] = var; o.div[r] = NAN;
            sum_rel_con += wc.mrw[r] * con / (c0 + EPSN);
            sum_rel_corr += wc.mrw[r] * (-mse) / (wc.zc[r] + EPSN);
        }
        const double mrc = sum_rel_con / Rd, mrr = sum_rel_corr / Rd;
        double val = ep.alpha * (-mrc) + ep.beta * (-mrr);
        double tv = 0.0;
        if (ep.want_tv) {                                    // regularizers.py:14-38 from k_tv's per-tile partials, losses.py:171
            double a = 0.0, nz = 0.0;
            for (int i = 0; i < g.ntiles; ++i) { a += c->h_tvparts[((size_t)b * g.ntiles + i) * 3]; nz += c->h_tvparts[((size_t)b * g.ntiles + i) * 3 + 1]; }
            tv = a / (nz + EPSN);
            if (ep.gamma != 0.0) val += ep.gamma * ((ep.cur_pyr_lvl <= 0) ? tv : 0.0);
        }
        if (c->theta_nan[b]) val = NAN;             // a NaN anywhere in theta surfaces as a NaN loss, like in the reference
        o.mean_rel_contrast = mrc; o.mean_rel_corr = mrr; o.mean_rel_div = NAN;
        o.tv = (ep.cur_pyr_lvl <= 0) ? (ep.want_tv ? tv : NAN) : 0.0;
        o.value = val; o.tv_scale = 0.0;
        o.nonfinite = std::isfinite(val) ? 0.0 : 1.0;
        if (c->pend.h == 1 && c->pend.w == 1) {              // 2-DoF: the gather's per-workgroup partials, added in index order
            const int lo = c->h_win_item0_2[b], hi = c->h_win_item0_2[b + 1];
            const double* p = c->h_g11 + (size_t)lo * c->g11_per_item * 2;
            const size_t n = (size_t)(hi - lo) * c->g11_per_item;
            // four interleaved chains per component (a fixed association, so still a function of the partials alone): one chain is
            // bound by the latency of the add, 4000 partials of the 8-window batch took 6 us
            double ax[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0};
            size_t k = 0;
            for (; k + 4 <= n; k += 4) {
                ax[0] += p[2 * k]; ay[0] += p[2 * k + 1]; ax[1] += p[2 * k + 2]; ay[1] += p[2 * k + 3];
                ax[2] += p[2 * k + 4]; ay[2] += p[2 * k + 5]; ax[3] += p[2 * k + 6]; ay[3] += p[2 * k + 7];
            }
            for (; k < n; ++k) { ax[0] += p[2 * k]; ay[0] += p[2 * k + 1]; }
            c->h_grad[(size_t)b * 2] = (ax[0] + ax[1]) + (ax[2] + ax[3]); c->h_grad[(size_t)b * 2 + 1] = (ay[0] + ay[1]) + (ay[2] + ay[3]);
        } else if (c->win_events[b] == 0) {                  // theta grid: the gather's tail wrote dL/dtheta, unless the window has no workgroup
            const size_t n = (size_t)c->pend.h * c->pend.w * 2;
            for (size_t i = 0; i < n; ++i) c->h_grad[(size_t)b * n + i] = 0.0;
        }
    }
}

// Second half, part 2: wait for the stream and hand the results over.
int eval_end_collect(eincm_ctx* c, double* value, double* grad, eincm_aux* aux) {
    if (!c->pend.active || !c->pend.launched) return fail(c, EINCM_ERR_STATE, "no evaluation in flight");
    const Geom& g = c->g;
    const bool want_grad = c->pend.want_grad;
    const size_t nth = (size_t)c->pend.h * c->pend.w * 2;
    // The stream is drained before ANY return: the kernels in flight read the pinned theta staging buffer and write the pinned
    // result block, so the context must not look idle (and accept the next theta) while they run.
    if (c->pend.copy_mode) { const int rcc = enqueue_result_copies(c); c->pend.copy_mode = 0; if (rcc) { (void)hipStreamSynchronize(c->stream); c->pend.active = false; c->pend.launched = false; return rcc; } }
    bool piece_bad = false;
    if (c->n_pieces > 0 && want_grad && grad) {
        const size_t total = (size_t)g.B * nth;
        for (int k = 0; k < c->n_pieces; ++k) {
            if (hipEventSynchronize(c->ev_piece[k]) != hipSuccess) break;      // the stream sync below reports the error
            const size_t off = (size_t)k * c->piece_len, n = std::min(c->piece_len, total - off);
            const double* __restrict__ src = c->h_grad + off;
            double* __restrict__ dst = grad + off;
            uint64_t bad = 0;
            for (size_t i = 0; i < n; ++i) {
                uint64_t u;
                memcpy(&u, src + i, sizeof u);
                dst[i] = src[i];
                bad |= (uint64_t)((u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
            }
            piece_bad = piece_bad || bad != 0;
        }
    }
    {
        HostPhase hp(c, EINCM_HP_WAIT);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->pend.active = false; c->pend.launched = false;
    ++c->hp_n;
    const bool dev_io = c->theta_dev_in != nullptr;           // the gradient went device to device (enqueue_result_copies)
    if (want_grad && !grad && !dev_io) return fail(c, EINCM_ERR_ARG, "the evaluation was begun with a gradient but grad is NULL");
    HostPhase hp(c, EINCM_HP_COLLECT);
    if (c->pend.host_asm) host_assemble(c);
    int rc = collect_timings(c);
    if (rc) return rc;
    c->have_eval = true;
    c->G_valid = want_grad;

    bool nonfinite = false;
    for (int b = 0; b < g.B; ++b) {
        if (b < 64 && !((g.wmask >> b) & 1ull)) {               // not evaluated: NaN value, zero gradient, no verdict
            if (value) value[b] = NAN;
            if (aux) { aux[b].final_loss = NAN; aux[b].mean_rel_corr = NAN; aux[b].mean_rel_contrast = NAN; aux[b].mean_rel_iwe_divergence = NAN; aux[b].theta_total_variation = NAN; }
            if (want_grad && c->n_pieces == 0 && !dev_io) for (size_t i = 0; i < nth; ++i) c->h_grad[(size_t)b * nth + i] = 0.0;
            continue;
        }
        const OutScal& o = c->h_outs[b];
        if (value) value[b] = o.value;
        if (aux) {
            aux[b].final_loss = o.value;
            aux[b].mean_rel_corr = o.mean_rel_corr;
            aux[b].mean_rel_contrast = o.mean_rel_contrast;
            aux[b].mean_rel_iwe_divergence = o.mean_rel_div;
            aux[b].theta_total_variation = o.tv;
        }
        if (o.nonfinite != 0.0) nonfinite = true;
    }
    if (piece_bad) nonfinite = true;
    if (want_grad && c->n_pieces == 0 && !dev_io) {
        // copy out and look for NaN/Inf in the same pass; an integer OR-reduction over the exponent bits vectorises, an
        // early-exit std::isfinite loop does not (0.6 ms of a 1.8 ms dense-theta evaluation at 480x640)
        const size_t n = (size_t)g.B * nth;
        const double* __restrict__ src = c->h_grad;
        uint64_t bad = 0;
        for (size_t i = 0; i < n; ++i) {
            uint64_t u;
            memcpy(&u, src + i, sizeof u);
            grad[i] = src[i];
            bad |= (uint64_t)((u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull);
        }
        if (bad) nonfinite = true;
    }
    if (nonfinite) return fail(c, EINCM_ERR_NONFINITE, "loss or gradient is not finite");
    return EINCM_OK;
}

int eval_end(eincm_ctx* c, double* value, double* grad, eincm_aux* aux) {
    if (c->pend.active && c->pend.want_grad && !grad) {
        (void)hipStreamSynchronize(c->stream);       // the forward half is in flight and reads the pinned theta buffer
        c->pend.active = false;
        return fail(c, EINCM_ERR_ARG, "the evaluation was begun with a gradient but grad is NULL");
    }
    const int rc = eval_end_launch(c);
    if (rc) { (void)hipStreamSynchronize(c->stream); c->pend.active = false; c->pend.launched = false; return rc; }
    return eval_end_collect(c, value, grad, aux);
}

// The whole evaluation.  theta_host: (B,h,w,2) doubles (already validated).
int evaluate(eincm_ctx* c, const double* theta_host, int h, int w, const eincm_params* p,
             double* value, double* grad, eincm_aux* aux, bool /*for_constants*/, const uint8_t* active = nullptr) {
    int rc = eval_begin(c, theta_host, h, w, p, grad != nullptr, active);
    if (rc) return rc;
    return eval_end(c, value, grad, aux);
}

// theta = 0 pass parameters used to obtain the window constants (every IWE_r then equals the IUE)
eincm_params zero_pass_params() {
    eincm_params p{};
    p.alpha = 1.0; p.beta = 1.0; p.cur_pyr_lvl = 1; p.method = EINCM_METHOD_BILINEAR; p.flags = EINCM_PF_FULL_AUX;
    return p;
}

// After the theta = 0 pass has been finished (eval_end): fill the window constants from its outputs.
int store_constants(eincm_ctx* c) {
    const Geom& g = c->g;
    const size_t img = (size_t)g.H * g.W;
    for (int b = 0; b < g.B; ++b) {
        WinConst& wc = c->h_wc[b];
        const OutScal& o = c->h_outs[b];
        wc.c0_gradmag = o.contrast_gm[0];
        wc.c0_var = o.var[0];
        wc.d0 = o.div[0];
        for (int r = 0; r < g.R; ++r) { wc.zc[r] = o.corr[r]; wc.inv_zc[r] = 1.0 / (wc.zc[r] + EPSN); }     // -MSE(E_r, n0)
        wc.inv_c0_gradmag = 1.0 / (wc.c0_gradmag + EPSN); wc.inv_c0_var = 1.0 / (wc.c0_var + EPSN);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_wc, c->h_wc, (size_t)g.B * sizeof(WinConst), hipMemcpyHostToDevice, c->stream));
    for (int b = 0; b < g.B; ++b)      // keep the IUE of every window (first reference image of the theta = 0 pass)
        HIPCHK(c, hipMemcpyAsync(c->d_zero_iwe + (size_t)b * img, c->d_iwe + (size_t)b * g.R * img, img * sizeof(float),
                                 hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_eval = false;
    c->constants_pending = false;
    return EINCM_OK;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int eincm_abi_version(void) { return EINCM_ABI_VERSION; }

const char* eincm_last_error(const eincm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int eincm_multi_ref_weights(int n_refs, double* w) {
    if (n_refs < 1 || n_refs > EINCM_MAX_REFS || !w) return EINCM_ERR_ARG;
    multi_ref_weights(n_refs, w);
    return EINCM_OK;
}

int eincm_resample_matrix(int n_in, int n_out, int method, double* A) {
    if (n_in < 1 || n_out < 1 || !A || method < 0 || method > EINCM_METHOD_CUBIC) return EINCM_ERR_ARG;
    std::vector<double> M;
    resample_matrix(n_in, n_out, method, M);
    memcpy(A, M.data(), M.size() * sizeof(double));
    return EINCM_OK;
}

eincm_ctx* eincm_create(int device, int H, int W, int max_refs, int max_windows, int64_t max_events_total, uint32_t flags) {
    if (H < 3 || W < 3 || H > 32767 || W > 32767 || max_refs < 1 || max_refs > EINCM_MAX_REFS || max_windows < 1 ||
        max_events_total < 1 || max_events_total > (int64_t)2000000000) {
        fail(nullptr, EINCM_ERR_ARG, "eincm_create: bad argument (H=%d W=%d max_refs=%d max_windows=%d max_events=%lld)",
             H, W, max_refs, max_windows, (long long)max_events_total);
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        fail(nullptr, EINCM_ERR_HIP, "eincm_create: no HIP device visible (%s); this engine has no CPU fallback",
             hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(nullptr, EINCM_ERR_ARG, "eincm_create: device %d out of range (0..%d)", device, ndev - 1);
        return nullptr;
    }
    eincm_ctx* c = new eincm_ctx();
    c->device = device; c->H = H; c->W = W; c->maxR = max_refs; c->maxB = max_windows; c->maxN = max_events_total;
    c->cflags = flags;
    if (const char* s = getenv("EINCM_CHUNK")) { int v = atoi(s); if (v >= NT && v <= MAX_CHUNK) { c->chunk = (v / NT) * NT; c->chunk_fixed = true; } }
    if (const char* s = getenv("EINCM_SEG")) { int v = atoi(s); if (v >= 64 && v <= MAX_SEG) c->seg = v; }
    if (const char* s = getenv("EINCM_SEG_SPLAT")) { int v = atoi(s); if (v >= 64 && v <= MAX_SEG) c->seg_s = v; }
    if (const char* s = getenv("EINCM_WINCAP")) { int v = atoi(s); if (v >= 1024 && v <= 6912) { c->wincap = (v / 4) * 4; c->wincap_fixed = true; } }
    if (c->seg_s > c->chunk && c->wincap > 4608) c->wincap = 4608;       // two LDS windows in k_splat's long-segment form
    auto bail = [&](const char* what, hipError_t err) -> eincm_ctx* {
        fail(nullptr, EINCM_ERR_HIP, "eincm_create: %s failed: %s", what, hipGetErrorString(err));
        free_all(c);
        delete c;
        return nullptr;
    };
#define TRY(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) return bail(#expr, e2_); } while (0)
    TRY(hipSetDevice(device));
    TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const int tilesX = (W + TS - 1) / TS, tilesY = (H + TS - 1) / TS, ntiles = tilesX * tilesY;
    const size_t B = max_windows, R = max_refs, img = (size_t)H * W;
    c->max_items = (int64_t)B * ntiles + max_events_total / 256 + 1;   // segments are never shorter than 256 events unless a tile is
    c->coarse_cap = 64 * 64 * 2;   // coarse theta up to 64x64 (the pyramid tops out at 16x16); grown on demand
    TRY(dalloc(&c->d_xy, (size_t)max_events_total));
    TRY(dalloc(&c->d_t, (size_t)max_events_total));
    TRY(dalloc(&c->d_xy_g, (size_t)max_events_total));
    TRY(dalloc(&c->d_t_g, (size_t)max_events_total));
    TRY(dalloc(&c->d_items, (size_t)c->max_items));
    TRY(dalloc(&c->d_items_s, (size_t)c->max_items));
    TRY(dalloc(&c->d_order, (size_t)c->max_items));
    TRY(dalloc(&c->d_order_s, (size_t)c->max_items));
    TRY(dalloc(&c->d_items_2, (size_t)c->max_items));
    TRY(dalloc(&c->d_items_sh, (size_t)c->max_items));
    TRY(dalloc(&c->d_order_sh, (size_t)c->max_items));
    TRY(dalloc(&c->d_order_2, (size_t)c->max_items));
    TRY(dalloc(&c->d_win_item0_2, B + 1));
    TRY(dalloc(&c->d_wins, (size_t)c->max_items * max_refs));
    TRY(dalloc(&c->d_wins_s, (size_t)c->max_items * max_refs));
    c->host_binning = (ntiles > BIN_MAX_TILES) || (getenv("EINCM_HOST_BINNING") != nullptr);
    TRY(dalloc(&c->d_raw_x, (size_t)max_events_total));          // kept after staging: eincm_get_warped_events walks them
    TRY(dalloc(&c->d_raw_y, (size_t)max_events_total));
    TRY(dalloc(&c->d_raw_t, (size_t)max_events_total));
    if (!c->host_binning) {
        c->max_binblocks = max_events_total / BIN_CHUNK + (int64_t)B + 1;
        TRY(dalloc(&c->d_binblocks, (size_t)c->max_binblocks));
        TRY(dalloc(&c->d_win_blk, B + 1));
        TRY(dalloc(&c->d_blockhist, (size_t)c->max_binblocks * ntiles));
        TRY(dalloc(&c->d_tilecount, B * ntiles));
        TRY(dalloc(&c->d_tilebase, B * ntiles));
        TRY(dalloc(&c->d_itembase, B * ntiles));
        TRY(dalloc(&c->d_itembase_s, B * ntiles));
        TRY(dalloc(&c->d_bin_misc, (size_t)8));
        TRY(dalloc(&c->d_edges_raw, B * R * img));
        TRY(dalloc(&c->d_edge_moments, B * R * EDGE_PARTS * EDGE_MOM));
    }
    TRY(dalloc(&c->d_edges, B * R * img));
    TRY(dalloc(&c->d_edge_ts, B * R));
    TRY(dalloc(&c->d_acc, B * R * img));
    TRY(hipMemset(c->d_acc, 0, B * R * img * sizeof(unsigned long long)));
    TRY(dalloc(&c->d_iwe, B * R * img));
    TRY(dalloc(&c->d_G, B * R * img));
    TRY(dalloc(&c->d_g11, (size_t)(c->max_items + NXCD) * R * 2 * 4));      // x4: up to four workgroups share a segment
    TRY(dalloc(&c->d_win_item0, B + 1));
    TRY(dalloc(&c->d_dtmax, B));
    TRY(dalloc(&c->d_cntmax, B));
    const size_t nig = (size_t)((W + IG_COLS - 1) / IG_COLS) * ((H + IG_ROWS - 1) / IG_ROWS);   // k_imgrad strips per image
    TRY(dalloc(&c->d_gmax, B * R * nig));
    TRY(hipMemset(c->d_gmax, 0, B * R * nig * sizeof(unsigned)));
    TRY(dalloc(&c->d_zero_iwe, B * img));
    TRY(dalloc(&c->d_Theta, B * img * 2));
    TRY(dalloc(&c->d_theta_in, B * img * 2));
    TRY(dalloc(&c->d_gTheta, B * img * 2));
    TRY(hipMemset(c->d_gTheta, 0, B * img * 2 * sizeof(long long)));
    TRY(dalloc(&c->d_tvg, B * img * 2));
    TRY(dalloc(&c->d_mask, B * img));
    TRY(dalloc(&c->d_tmm, B * ntiles * 4));
    const size_t pstride = (size_t)std::max(std::max(ntiles, NSPART), (int)((nig + IG_NT / 64 - 1) / (IG_NT / 64)));
    TRY(dalloc(&c->d_parts, B * R * pstride));
    TRY(dalloc(&c->d_amax, B * R * pstride));
    TRY(dalloc(&c->d_gbound, B * R));
    TRY(hipMemset(c->d_gbound, 0, B * R * sizeof(unsigned)));
    TRY(dalloc(&c->d_ticket, B * R));
    TRY(hipMemset(c->d_ticket, 0, B * R * sizeof(unsigned)));
    TRY(dalloc(&c->d_coef, B * R));
    TRY(dalloc(&c->d_gticket, B));
    TRY(hipMemset(c->d_gticket, 0, B * sizeof(unsigned)));
    TRY(dalloc(&c->d_divparts, B * R * ntiles));
    TRY(dalloc(&c->d_g2parts, B * R * nig));
    TRY(dalloc(&c->d_tvparts, B * ntiles * 3));
    TRY(dalloc(&c->d_wc, B));
    {   // one device block and one pinned block: [OutScal x B | grad (B,H,W,2)] -> a single D2H copy per evaluation
        char* blk = nullptr;
        TRY(hipMalloc(reinterpret_cast<void**>(&blk), B * sizeof(OutScal) + B * img * 2 * sizeof(double)));
        c->d_outs = reinterpret_cast<OutScal*>(blk);
        c->d_grad = reinterpret_cast<double*>(blk + B * sizeof(OutScal));
        char* hblk = nullptr;
        TRY(hipHostMalloc(reinterpret_cast<void**>(&hblk), B * sizeof(OutScal) + B * img * 2 * sizeof(double), hipHostMallocDefault));
        c->h_outs = reinterpret_cast<OutScal*>(hblk);
        c->h_grad = reinterpret_cast<double*>(hblk + B * sizeof(OutScal));
    }
    TRY(dalloc(&c->d_gth, 2 * B * (size_t)c->coarse_cap));
    TRY(hipMemset(c->d_gth, 0, 2 * B * (size_t)c->coarse_cap * sizeof(long long)));
    TRY(dalloc(&c->d_rowtap, (size_t)H));
    TRY(dalloc(&c->d_coltap, (size_t)W));
    TRY(dalloc(&c->d_tilerng, (size_t)ntiles));
    TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_theta), B * img * 2 * sizeof(double), hipHostMallocDefault));
    TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_wc), B * sizeof(WinConst), hipHostMallocDefault));
    TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_g11), (size_t)(c->max_items + NXCD) * R * 2 * 4 * sizeof(double), hipHostMallocDefault));
    TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_g2), B * R * nig * sizeof(double), hipHostMallocDefault));
    TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_img), B * R * IMGSCAL_N * sizeof(double), hipHostMallocDefault));
    TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_tvparts), B * ntiles * 3 * sizeof(double), hipHostMallocDefault));
    c->have_events = true;
    for (int k = 0; k < eincm_ctx::GRAD_PIECES; ++k) TRY(hipEventCreateWithFlags(&c->ev_piece[k], hipEventDisableTiming));
    c->ring_size = (flags & EINCM_CF_TIMING_DOMINANT) && !(flags & EINCM_CF_TIMING) ? eincm_ctx::EV_RING : 1;
    if (flags & (EINCM_CF_TIMING | EINCM_CF_TIMING_DOMINANT))
        for (int k = 0; k < c->ring_size; ++k)
            for (int i = 0; i <= EINCM_N_STAGES; ++i) {
                const bool needed = (flags & EINCM_CF_TIMING) || i == EINCM_STAGE_SPLAT || i == EINCM_STAGE_GATHER || i == EINCM_N_STAGES;
                if (needed) { TRY(hipEventCreate(&c->ev[k][i][0])); TRY(hipEventCreate(&c->ev[k][i][1])); }
            }
    // both event kernels need > 32 KiB... (<= 64 KiB default limit is fine on gfx950, no attribute needed)
#undef TRY
    return c;
}

void eincm_destroy(eincm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx);
    delete ctx;
}

// xs_w / ys_w / ts_w / edges_w: one pointer per window (the caller's own arrays; nothing is concatenated on the host)
// The fraction of a window's duration the segments of a list span, for the choice of the LDS window capacity (eval_begin): a tile of n events
// is cut into ceil(n / seg) segments, each spanning about 1 / that of the time.  Not the mean: sparse tiles (sensor noise between the edges)
// hold one segment that spans the WHOLE window, and their taps go to HBM one by one when the capacity follows the dense tiles (480x640
// with 10^7 events at 16x16 theta: k_gather 110 -> 90 us with the larger windows).  Returned: the span that all but 3 % of the events stay within.
static double span_quantile(const std::vector<int32_t>& tilecount, int seg, int64_t N)
{
    constexpr int K = 64;
    int64_t by_nseg[K + 1] = {0};
    for (const int32_t n : tilecount)
        if (n > 0) by_nseg[std::min<int64_t>(((int64_t)n + seg - 1) / seg, K)] += n;
    int64_t beyond = 0;
    const int64_t allow = (int64_t)(0.03 * (double)N);
    for (int k = 1; k <= K; ++k) {              // spans 1, 1/2, 1/3, ...
        beyond += by_nseg[k];
        if (beyond > allow) return 1.0 / k;
    }
    return 1.0 / K;
}

static int set_windows_impl(eincm_ctx* c, int n_windows, int n_refs, const int64_t* n_events, const int16_t* const* xs_w,
                            const int16_t* const* ys_w, const double* const* ts_w, const double* const* edges_w,
                            const double* edge_ts, uint32_t sw_flags) {
    if (c && c->pend.active && c->pend.launched)
        return fail(c, EINCM_ERR_STATE, "an asynchronous evaluation is in flight: call eincm_loss_grad_wait first");
    if (!c) return EINCM_ERR_ARG;
    if (n_windows < 1 || n_windows > c->maxB) return fail(c, EINCM_ERR_ARG, "n_windows %d outside 1..%d", n_windows, c->maxB);
    if (n_refs < 1 || n_refs > c->maxR) return fail(c, EINCM_ERR_ARG, "n_refs %d outside 1..%d", n_refs, c->maxR);
    if (!n_events || !xs_w || !ys_w || !ts_w || !edges_w || !edge_ts) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (n_windows >= 1 && n_windows <= c->maxB)
        for (int b = 0; b < n_windows; ++b)
            if (!edges_w[b] || (n_events[b] > 0 && (!xs_w[b] || !ys_w[b] || !ts_w[b]))) return fail(c, EINCM_ERR_ARG, "null pointer argument (window %d)", b);
    HIPCHK(c, hipSetDevice(c->device));
    const int H = c->H, W = c->W;
    int64_t N = 0;
    for (int b = 0; b < n_windows; ++b) {
        if (n_events[b] < 0) return fail(c, EINCM_ERR_ARG, "n_events[%d] negative", b);
        N += n_events[b];
    }
    if (N > c->maxN) return fail(c, EINCM_ERR_ARG, "total events %lld exceed capacity %lld", (long long)N, (long long)c->maxN);
    c->staged = false;
    c->Theta_valid = false;
    // Async copies below read from / write into locals of this function; whatever path leaves it (an error return included), the
    // stream is drained first (safe before only because pageable copies happen to be host-synchronous).
    struct DrainOnExit { hipStream_t s; ~DrainOnExit() { (void)hipStreamSynchronize(s); } } drain_on_exit{c->stream};
    if (c->acc_dirty) { const int rcd = clear_accumulators(c); if (rcd) return rcd; }
    Geom g{};
    g.H = H; g.W = W; g.R = n_refs; g.B = n_windows;
    g.tilesX = (W + TS - 1) / TS; g.tilesY = (H + TS - 1) / TS; g.ntiles = g.tilesX * g.tilesY;
    g.igx = (W + IG_COLS - 1) / IG_COLS; g.nig = g.igx * ((H + IG_ROWS - 1) / IG_ROWS);
    g.pstride = std::max(std::max(g.ntiles, NSPART), (g.nig + IG_NT / 64 - 1) / (IG_NT / 64));
    g.gmax_n = g.R * g.nig;
    g.wincap = c->wincap; g.winmaxw = std::max(40, (int)std::lround(std::sqrt((double)c->wincap * 1.4)));
    g.wincap_a = g.wincap; g.winmaxw_a = g.winmaxw; c->wincap_2 = g.wincap;

    // Segment lengths (events per workgroup and reference time), measured on MI355X with the longest-first order of block_to_work
    // (tools/dev_tune_seg.py, profiles/r02/segment_tuning.txt).  Per-workgroup fixed cost (window clear / flush, G-window load,
    // reductions) favours long segments, the end of the launch (the last workgroups run on a mostly idle chip) short ones; x
    // estimates the workgroups of a launch at 8192-event segments against the 2048 the chip holds at once.
    //   k_gather: x >= 4000 (8 windows x 10^6 events, one window of 10^7): 16384 (96.9 -> 91.7 us, 167 -> 140 us);
    //             x < 1000 (one 10^6-event window): 4096 (22.8 us against 25.5 with 8192 and 41 with 16384); else 8192.
    //   k_splat:  bound by LDS atomics, it gains nothing beyond 8192 (108 us at 8192 and 16384, 122 at 4096, 186 at 2048 on the
    //             8-window batch) and loses nothing with it on a single window (22.7 vs 23.5 us; theta grids 22.3 vs 24.5).
    //   theta grids / dense theta: the gather walks the list with the longer segments (one 10^6-event window at 16x16: 35.8 us
    //             with 4096, 30.3 with 8192, 28.6 with 16384; the 8-window batch 157 / 144 / 139 with 8192 / 16384 / 32768).
    const double x_wg = ((double)N / 8192.0 + 0.5 * n_windows * g.ntiles) * n_refs;
    // Round 3: the gather's list always has long segments (what its theta-grid form wants: thtile, accumulator clear and flush per
    // workgroup); its 2-DoF form shares a segment among up to four workgroups instead (gather_wg_events, k_gather's nparts).
    int seg = c->seg > 0 ? c->seg : 16384;
    c->gather_wg_events = 1e30;              // (k_gather's nparts: an experiment, EINCM_GATHER_PARTS)
    int seg_2 = (x_wg >= 4000.0 && (double)N / ((double)n_windows * g.ntiles) < 16384.0) ? 16384 : (x_wg < 1000.0 ? 4096 : 8192);   // (tiles of several segments: as seg_s below; 480x640 with 10^7 events 90 -> 82 us)       // the 2-DoF gather's own list (round-2 tuning)
    if (const char* e = getenv("EINCM_SEG_2DOF")) { const int v = atoi(e); if (v >= 64 && v <= MAX_SEG) seg_2 = v; }
    c->seg_2_used = seg_2;
    c->seg_used = seg;
    // (late round 3: k_splat is no longer bound by the LDS atomic unit, so its per-workgroup fixed work - 24 of 90 us on the 8-window
    // batch: window derivation and clear 14, flush 10 - shows: 16384-event segments there, 90.1 -> 85.2 us; 104 -> 100 us at 16x16)
    // ... but only where a tile holds about one such segment: with tiles of several segments (480x640, 10^7 events: 33 000 per tile) the long
    // segments double the duration of EVERY workgroup and the kernel ends in a tail of few resident waves (k_splat 93 -> 137 us there).
    const double per_tile = (double)N / ((double)n_windows * g.ntiles);
    int seg_s = c->seg_s > 0 ? c->seg_s : (x_wg >= 3000.0 && per_tile < 16384.0 ? 16384 : (x_wg >= 400.0 ? 8192 : 4096));     // (4 windows of 10^6 events: 52.4 -> 49.7 us; 2 windows: equal; 1: 18.8 vs 19.7 the other way)  one window at R = 1: 4096 (0.066 vs 0.075 ms per evaluation)
    // The bank-aligned LDS pitch (win_pitch) goes with the same regime - many resident windows, about one segment per tile: both event
    // kernels 2 % faster on the bench batch; everywhere else pitch = width is the faster layout (profiles/r03/pitch_by_shape.txt: one
    // window of 10^6 events 72 -> 62 us per evaluation, 2 x 3*10^6 143 -> 128, 480x640 with 5*10^6 173 -> 126, with 10^7 220 -> 184).
    g.pitch_aligned = (x_wg >= 3000.0 && per_tile < 16384.0) ? 1 : 0;
    if (const char* e = getenv("EINCM_PITCH_ALIGNED")) g.pitch_aligned = std::max(0, std::min(2, atoi(e)));      // 0: never, 1: k_splat where it costs no capacity class, 2: the 2-DoF gather too
    c->seg_s_used = seg_s;
    const bool sort_segments = getenv("EINCM_NO_SEGSORT") == nullptr;
    if (!c->chunk_fixed) c->chunk = std::max(4096, std::min(seg_s, MAX_CHUNK));     // single-chunk segments: no f32 commit pass
    const size_t img = (size_t)H * W;
    for (int b = 0; b < n_windows; ++b) {
        memset(&c->h_wc[b], 0, sizeof(WinConst));
        multi_ref_weights(n_refs, c->h_wc[b].mrw);
        for (int r = 0; r < n_refs; ++r)
            if (!std::isfinite(edge_ts[b * n_refs + r])) return fail(c, EINCM_ERR_ARG, "edge_ts[%d,%d] is not finite", b, r);
    }
    int n_items_total = 0, n_items_s_total = 0;
    c->h_win_item0.assign((size_t)n_windows + 1, 0);
    std::vector<unsigned> cntmax_h((size_t)n_windows, 0u);
    std::vector<double> dtmax_h((size_t)n_windows, 0.0);
    std::vector<int32_t> item0_h;                  // host path only
    bool edge_ts_uploaded = false;
    if (!c->host_binning) {
        // ---- device path: counting sort by (window, source tile) on the GPU (eincm_binning.hip.h) ----
        std::vector<BinBlock> blks;
        std::vector<int32_t> win_blk((size_t)n_windows + 1);
        int64_t base = 0;
        for (int b = 0; b < n_windows; ++b) {
            win_blk[b] = (int32_t)blks.size();
            for (int64_t s0 = 0; s0 < n_events[b]; s0 += BIN_CHUNK) {
                BinBlock bb; bb.win = b; bb.start = (int32_t)(base + s0); bb.count = (int32_t)std::min<int64_t>(BIN_CHUNK, n_events[b] - s0);
                bb.first_blk = win_blk[b];
                blks.push_back(bb);
            }
            base += n_events[b];
        }
        win_blk[n_windows] = (int32_t)blks.size();
        const int nblk = (int)blks.size();
        if (nblk > c->max_binblocks) return fail(c, EINCM_ERR_ARG, "internal: %d staging blocks exceed capacity", nblk);
        int32_t misc_init[4] = {0, 0, 0x7fffffff, 0x7fffffff};                      // totals[2], err[2]
        HIPCHK(c, hipMemcpyAsync(c->d_bin_misc, misc_init, sizeof misc_init, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_win_blk, win_blk.data(), win_blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        for (int b = 0; b < n_windows; ++b)
            HIPCHK(c, hipMemcpyAsync(c->d_edges_raw + (size_t)b * n_refs * img, edges_w[b], (size_t)n_refs * img * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_edges, dim3(EDGE_PARTS, n_refs, n_windows), dim3(NT), 0, c->stream, g, c->d_edges_raw, c->d_edges, c->d_edge_moments);
        if (nblk > 0) {
            HIPCHK(c, hipMemcpyAsync(c->d_binblocks, blks.data(), blks.size() * sizeof(BinBlock), hipMemcpyHostToDevice, c->stream));
            int64_t off = 0;
            for (int b = 0; b < n_windows; ++b) {
                const size_t nb = (size_t)n_events[b];
                if (nb > 0) {
                    HIPCHK(c, hipMemcpyAsync(c->d_raw_x + off, xs_w[b], nb * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->d_raw_y + off, ys_w[b], nb * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->d_raw_t + off, ts_w[b], nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
                }
                off += (int64_t)nb;
            }
            hipLaunchKernelGGL(k_bin_hist, dim3(nblk), dim3(BIN_NT), g.ntiles * sizeof(uint32_t), c->stream, g, c->d_binblocks, c->d_raw_x, c->d_raw_y,
                               c->d_raw_t, c->d_blockhist, c->d_bin_misc + 2);
        } else {
            HIPCHK(c, hipMemsetAsync(c->d_blockhist, 0, sizeof(uint32_t), c->stream));
        }
        const int M = n_windows * g.ntiles;
        hipLaunchKernelGGL(k_bin_scan, dim3((M + 255) / 256), dim3(256), 0, c->stream, g, c->d_win_blk, c->d_blockhist, c->d_tilecount);
        hipLaunchKernelGGL(k_bin_tilescan, dim3(1), dim3(1024), 0, c->stream, M, seg, c->d_tilecount, c->d_tilebase, c->d_itembase, c->d_bin_misc);
        HIPCHK(c, hipGetLastError());
        int32_t misc[4];
        std::vector<double> mom((size_t)n_windows * n_refs * EDGE_PARTS * EDGE_MOM);
        HIPCHK(c, hipMemcpyAsync(misc, c->d_bin_misc, sizeof misc, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(mom.data(), c->d_edge_moments, mom.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        c->h_tilecount.resize((size_t)M);
        HIPCHK(c, hipMemcpyAsync(c->h_tilecount.data(), c->d_tilecount, (size_t)M * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (misc[2] != 0x7fffffff) {
            const int64_t e = misc[2];
            int b = 0; int64_t off = e;
            while (b < n_windows - 1 && off >= n_events[b]) { off -= n_events[b]; ++b; }
            return fail(c, EINCM_ERR_ARG, "event %lld of window %d at (x=%d, y=%d) outside the %dx%d sensor", (long long)off, b,
                        (int)xs_w[b][off], (int)ys_w[b][off], H, W);
        }
        if (misc[3] != 0x7fffffff) {
            const int64_t e = misc[3];
            int b = 0; int64_t off = e;
            while (b < n_windows - 1 && off >= n_events[b]) { off -= n_events[b]; ++b; }
            return fail(c, EINCM_ERR_ARG, "event %lld of window %d has a non-finite timestamp", (long long)off, b);
        }
        n_items_total = misc[1];
        if (n_items_total > c->max_items) return fail(c, EINCM_ERR_ARG, "internal: %d segments exceed capacity", n_items_total);
        for (int b = 0; b < n_windows; ++b)
            for (int r = 0; r < n_refs; ++r) {          // the blocks' partials, added in index order
                double sE = 0.0, sEE = 0.0, eabs = 0.0;
                for (int k = 0; k < EDGE_PARTS; ++k) {
                    const double* m = &mom[(((size_t)b * n_refs + r) * EDGE_PARTS + k) * EDGE_MOM];
                    sE += m[0]; sEE += m[1]; eabs = std::max(eabs, m[2]);
                }
                c->h_wc[b].sE[r] = sE; c->h_wc[b].sEE[r] = sEE; c->h_wc[b].eabs[r] = eabs;
            }
        if (nblk > 0) {
            hipLaunchKernelGGL(k_bin_scatter, dim3(nblk), dim3(BIN_NT), g.ntiles * sizeof(uint32_t), c->stream, g, c->d_binblocks, c->d_raw_x, c->d_raw_y,
                               c->d_raw_t, c->d_blockhist, c->d_tilebase, c->d_xy, c->d_t);
            hipLaunchKernelGGL(k_items, dim3((M + 255) / 256), dim3(256), 0, c->stream, g, seg, c->d_tilecount, c->d_tilebase, c->d_itembase, c->d_items);
            if (n_items_total > 0) {
                // the gather's copy: every segment of its list sorted by source pixel and dealt to the threads that walk it (from the
                // binned time order, before the splat's copy is re-dealt in place)
                if (sort_segments)
                    hipLaunchKernelGGL(k_segsort, dim3(std::min(n_items_total, 8192)), dim3(SORT_NT), 0, c->stream, n_items_total, c->d_items,
                                       c->d_xy, c->d_t, c->d_xy_g, c->d_t_g);
                else {
                    HIPCHK(c, hipMemcpyAsync(c->d_xy_g, c->d_xy, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->d_t_g, c->d_t, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                }
                hipLaunchKernelGGL(k_seg_minmax, dim3(std::min(n_items_total, 4096)), dim3(NT), 0, c->stream, n_items_total, c->d_items, c->d_t_g);
            }
            if (!getenv("EINCM_NO_SPREAD"))
                hipLaunchKernelGGL(k_spread, dim3(M, SPREAD_Y), dim3(256), 0, c->stream, g, c->d_tilecount, c->d_tilebase, c->d_xy, c->d_t);
            // per window: first segment and max |t - tau| (needs the segment time ranges and the FIRST segmentation's itembase)
            HIPCHK(c, hipMemcpyAsync(c->d_edge_ts, edge_ts, (size_t)n_windows * n_refs * sizeof(double), hipMemcpyHostToDevice, c->stream));
            edge_ts_uploaded = true;
            hipLaunchKernelGGL(k_win_consts, dim3(n_windows), dim3(NT), 0, c->stream, g, n_items_total, c->d_items, c->d_itembase, c->d_edge_ts,
                               c->d_win_item0, c->d_dtmax);
            HIPCHK(c, hipMemcpyAsync(dtmax_h.data(), c->d_dtmax, (size_t)n_windows * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->h_win_item0.data(), c->d_win_item0, (size_t)n_windows * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            // second segmentation of the same binned events for k_splat
            hipLaunchKernelGGL(k_bin_tilescan, dim3(1), dim3(1024), 0, c->stream, M, seg_s, c->d_tilecount, c->d_tilebase, c->d_itembase_s, c->d_bin_misc);
            {   // its total is not read back (a synchronisation per staging): the host repeats the arithmetic on the tile populations it holds
                std::vector<int32_t> lens_s;
                segment_lengths(c->h_tilecount, seg_s, lens_s);
                n_items_s_total = (int)lens_s.size();
            }
            if (n_items_s_total > c->max_items) return fail(c, EINCM_ERR_ARG, "internal: %d splat segments exceed capacity", n_items_s_total);
            hipLaunchKernelGGL(k_items, dim3((M + 255) / 256), dim3(256), 0, c->stream, g, seg_s, c->d_tilecount, c->d_tilebase, c->d_itembase_s, c->d_items_s);
            if (n_items_s_total > 0)
                hipLaunchKernelGGL(k_seg_minmax, dim3(std::min(n_items_s_total, 4096)), dim3(NT), 0, c->stream, n_items_s_total, c->d_items_s, c->d_t);
            HIPCHK(c, hipGetLastError());
            {   // longest-first order of both segment lists, from the tile populations (the host repeats k_items' arithmetic)
                std::vector<int32_t> lens;
                segment_lengths(c->h_tilecount, seg, lens);
                if ((int)lens.size() != n_items_total) return fail(c, EINCM_ERR_ARG, "internal: segment lists disagree (%zu, %d)", lens.size(), n_items_total);
                order_by_length(lens, c->h_order);
                segment_lengths(c->h_tilecount, seg_s, lens);
                if ((int)lens.size() != n_items_s_total) return fail(c, EINCM_ERR_ARG, "internal: splat segment lists disagree (%zu, %d)", lens.size(), n_items_s_total);
                order_by_length(lens, c->h_order_s);
                if (n_items_total > 0)
                    HIPCHK(c, hipMemcpyAsync(c->d_order, c->h_order.data(), c->h_order.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                if (n_items_s_total > 0)
                    HIPCHK(c, hipMemcpyAsync(c->d_order_s, c->h_order_s.data(), c->h_order_s.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            }
        } else {
            HIPCHK(c, hipMemsetAsync(c->d_win_item0, 0, (size_t)(n_windows + 1) * sizeof(int32_t), c->stream));
        }
    } else {
    // ---- host path (sensors with more tiles than the LDS histogram holds, or EINCM_HOST_BINNING=1): stable counting sort ----
    c->h_tilecount.assign((size_t)n_windows * g.ntiles, 0);
    std::vector<uint32_t> sxy((size_t)std::max<int64_t>(N, 1));
    std::vector<double> st((size_t)std::max<int64_t>(N, 1));
    std::vector<Item> items, items_s;
    std::vector<int64_t> cnt((size_t)g.ntiles + 1);
    int64_t base = 0;
    for (int b = 0; b < n_windows; ++b) {
        const int64_t n = n_events[b];
        const int16_t* x = xs_w[b]; const int16_t* y = ys_w[b]; const double* t = ts_w[b];
        std::fill(cnt.begin(), cnt.end(), 0);
        item0_h.push_back((int32_t)items.size());
        for (int64_t i = 0; i < n; ++i) {
            for (int r = 0; r < n_refs; ++r) dtmax_h[b] = std::max(dtmax_h[b], std::fabs(t[i] - edge_ts[b * n_refs + r]));
            if (x[i] < 0 || x[i] >= W || y[i] < 0 || y[i] >= H)
                return fail(c, EINCM_ERR_ARG, "event %lld of window %d at (x=%d, y=%d) outside the %dx%d sensor",
                            (long long)i, b, (int)x[i], (int)y[i], H, W);
            if (!std::isfinite(t[i])) return fail(c, EINCM_ERR_ARG, "event %lld of window %d has a non-finite timestamp", (long long)i, b);
            ++cnt[(size_t)(y[i] / TS) * g.tilesX + (x[i] / TS) + 1];
        }
        {   // most events on one source pixel
            std::vector<uint32_t> pc((size_t)H * W, 0u);
            for (int64_t i = 0; i < n; ++i) cntmax_h[b] = std::max(cntmax_h[b], ++pc[(size_t)y[i] * W + x[i]]);
        }
        for (int k = 0; k < g.ntiles; ++k) c->h_tilecount[(size_t)b * g.ntiles + k] = (int32_t)cnt[k + 1];
        for (int k = 0; k < g.ntiles; ++k) cnt[k + 1] += cnt[k];
        std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
        for (int64_t i = 0; i < n; ++i) {
            const int tile = (y[i] / TS) * g.tilesX + (x[i] / TS);
            const int64_t d = base + pos[tile]++;
            sxy[d] = (uint32_t)(uint16_t)x[i] | ((uint32_t)(uint16_t)y[i] << 16);
            st[d] = t[i];
        }
        for (int pass = 0; pass < 2; ++pass) {
            const int sg = pass == 0 ? seg : seg_s;
            std::vector<Item>& dst = pass == 0 ? items : items_s;
            for (int k = 0; k < g.ntiles; ++k) {
                const int len = balanced_seg_len((int)(cnt[k + 1] - cnt[k]), sg);
                for (int64_t s = cnt[k]; s < cnt[k + 1]; s += len) {
                    Item it;
                    it.win = b; it.tile = k; it.begin = (int32_t)(base + s);
                    it.count = (int32_t)std::min<int64_t>(len, cnt[k + 1] - s);
                    double lo = st[it.begin], hi = st[it.begin];
                    for (int q = 1; q < it.count; ++q) { lo = std::min(lo, st[it.begin + q]); hi = std::max(hi, st[it.begin + q]); }
                    it.t_lo = lo; it.t_hi = hi;
                    dst.push_back(it);
                }
            }
        }
        base += n;
    }
    if ((int64_t)items.size() > c->max_items || (int64_t)items_s.size() > c->max_items)
        return fail(c, EINCM_ERR_ARG, "internal: %zu work items exceed capacity", std::max(items.size(), items_s.size()));
    n_items_total = (int)items.size();
    n_items_s_total = (int)items_s.size();
    std::vector<float> ef((size_t)n_windows * n_refs * img);
    for (int b = 0; b < n_windows; ++b) {
        WinConst& wc = c->h_wc[b];
        for (int r = 0; r < n_refs; ++r) {
            const double* e = edges_w[b] + (size_t)r * img;
            float* o = ef.data() + ((size_t)b * n_refs + r) * img;
            double s = 0.0, ss = 0.0, mx = 0.0;
            for (size_t i = 0; i < img; ++i) { const float f = (float)e[i]; o[i] = f; s += (double)f; ss += (double)f * (double)f; mx = std::max(mx, std::fabs((double)f)); }
            wc.sE[r] = s; wc.sEE[r] = ss; wc.eabs[r] = mx;
        }
    }
    if (N > 0) {
        HIPCHK(c, hipMemcpyAsync(c->d_xy, sxy.data(), (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_t, st.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        int64_t off = 0;                                   // the events as handed over (eincm_get_warped_events), like the device path keeps them
        for (int b = 0; b < n_windows; ++b) {
            const size_t nb = (size_t)n_events[b];
            if (nb > 0) {
                HIPCHK(c, hipMemcpyAsync(c->d_raw_x + off, xs_w[b], nb * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->d_raw_y + off, ys_w[b], nb * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->d_raw_t + off, ts_w[b], nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
            }
            off += (int64_t)nb;
        }
    }
    if (!items.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice, c->stream));
        if (sort_segments)      // the gather's copy; a permutation inside the segments: their time ranges (computed above) are unchanged
            hipLaunchKernelGGL(k_segsort, dim3((unsigned)std::min<size_t>(items.size(), 8192)), dim3(SORT_NT), 0, c->stream, (int)items.size(), c->d_items,
                               c->d_xy, c->d_t, c->d_xy_g, c->d_t_g);
        else {
            HIPCHK(c, hipMemcpyAsync(c->d_xy_g, c->d_xy, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_t_g, c->d_t, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    if (!items_s.empty())
        HIPCHK(c, hipMemcpyAsync(c->d_items_s, items_s.data(), items_s.size() * sizeof(Item), hipMemcpyHostToDevice, c->stream));
    {
        std::vector<int32_t> lens(items.size());
        for (size_t i = 0; i < items.size(); ++i) lens[i] = items[i].count;
        order_by_length(lens, c->h_order);
        lens.resize(items_s.size());
        for (size_t i = 0; i < items_s.size(); ++i) lens[i] = items_s[i].count;
        order_by_length(lens, c->h_order_s);
        if (!items.empty())
            HIPCHK(c, hipMemcpyAsync(c->d_order, c->h_order.data(), c->h_order.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        if (!items_s.empty())
            HIPCHK(c, hipMemcpyAsync(c->d_order_s, c->h_order_s.data(), c->h_order_s.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_win_item0, item0_h.data(), item0_h.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    std::copy(item0_h.begin(), item0_h.end(), c->h_win_item0.begin());
    HIPCHK(c, hipMemcpyAsync(c->d_edges, ef.data(), ef.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // host vectors go out of scope
    }
    if (!edge_ts_uploaded)
        HIPCHK(c, hipMemcpyAsync(c->d_edge_ts, edge_ts, (size_t)n_windows * n_refs * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_mask, 0, (size_t)n_windows * img, c->stream));
    std::vector<Item> items_2;
    {   // the 2-DoF gather's segment list (splat copy of the events): generated on the host from the tile populations
        host_items(c->h_tilecount, g.ntiles, seg_2, items_2, c->h_win_item0_2);
        if ((int64_t)items_2.size() > c->max_items) return fail(c, EINCM_ERR_ARG, "internal: %zu segments exceed capacity", items_2.size());
        c->h_win_item0_2.resize((size_t)n_windows + 1, (int32_t)items_2.size());
        std::vector<int32_t> lens(items_2.size());
        for (size_t i = 0; i < items_2.size(); ++i) lens[i] = items_2[i].count;
        order_by_length(lens, c->h_order_2);
        if (!items_2.empty()) {
            HIPCHK(c, hipMemcpyAsync(c->d_items_2, items_2.data(), items_2.size() * sizeof(Item), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_order_2, c->h_order_2.data(), c->h_order_2.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_seg_minmax, dim3((unsigned)std::min<size_t>(items_2.size(), 4096)), dim3(NT), 0, c->stream, (int)items_2.size(), c->d_items_2, c->d_t);
        }
        HIPCHK(c, hipMemcpyAsync(c->d_win_item0_2, c->h_win_item0_2.data(), (size_t)(n_windows + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    std::vector<Item> items_sh;
    std::vector<int32_t> order_sh;
    c->n_items_sh = 0; c->seg_sh_used = 0;
    if (seg_s > 8192) {   // the splat's short list: the same events cut into 8192-event segments, for evaluations whose theta is too large for
        std::vector<int32_t> w0;                    // the windows of the long segments (twice the time span, hence twice the spread)
        host_items(c->h_tilecount, g.ntiles, 8192, items_sh, w0);
        if ((int64_t)items_sh.size() <= c->max_items && !items_sh.empty()) {
            std::vector<int32_t> lens(items_sh.size());
            for (size_t i = 0; i < items_sh.size(); ++i) lens[i] = items_sh[i].count;
            order_by_length(lens, order_sh);
            HIPCHK(c, hipMemcpyAsync(c->d_items_sh, items_sh.data(), items_sh.size() * sizeof(Item), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_order_sh, order_sh.data(), order_sh.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_seg_minmax, dim3((unsigned)std::min<size_t>(items_sh.size(), 4096)), dim3(NT), 0, c->stream, (int)items_sh.size(), c->d_items_sh, c->d_t);
            c->n_items_sh = (int)items_sh.size(); c->seg_sh_used = 8192;
        }
    }
    if (n_items_total > 0 && !c->host_binning) {     // event mask + most events on one source pixel, per tile from its LDS histogram
        HIPCHK(c, hipMemsetAsync(c->d_cntmax, 0, (size_t)n_windows * sizeof(unsigned), c->stream));       // (before the synchronisation below: one per staging fewer)
        hipLaunchKernelGGL(k_tile_counts, dim3(g.ntiles, n_windows), dim3(NT), 0, c->stream, g, c->d_tilebase, c->d_tilecount, c->d_xy,
                           c->d_mask, c->d_cntmax);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(cntmax_h.data(), c->d_cntmax, (size_t)n_windows * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_items_2 = (int)items_2.size();
    c->tspan_s = span_quantile(c->h_tilecount, seg_s, N);
    c->tspan_sh = span_quantile(c->h_tilecount, 8192, N);
    c->tspan_a = span_quantile(c->h_tilecount, seg, N);
    c->tspan_2 = span_quantile(c->h_tilecount, seg_2, N);
    c->pitch_policy = g.pitch_aligned;
    c->policy_evaluated = false;
    c->g = g; c->n_items = n_items_total; c->n_items_s = n_items_s_total; c->n_events = N;
    c->itembase_valid = !c->host_binning && n_items_total > 0 && n_items_s_total > 0;      // both tile scans ran on the device
    c->win_events.assign(n_events, n_events + n_windows);
    if (c->n_items > 0 && c->host_binning) {
        hipLaunchKernelGGL(k_mask, dim3(std::min(c->n_items, 2048)), dim3(NT), 0, c->stream, g, c->d_items, c->n_items, c->d_xy, c->d_mask);
        HIPCHK(c, hipGetLastError());
    }

    // ---- zero-warp constants: one forward pass at theta = 0 (then IWE_r == IUE for every r) ----
    // c0, zc[r], d0 are temporarily 1 so the pass is well defined; store_constants() overwrites them.
    for (int b = 0; b < n_windows; ++b) {
        WinConst& wc = c->h_wc[b];
        wc.c0_gradmag = 1.0; wc.c0_var = 1.0; wc.d0 = 1.0;
        for (int r = 0; r < n_refs; ++r) wc.zc[r] = 1.0;
        // bounds behind the scale of the i64 gradient accumulators (grad_shift)
        wc.nev = (double)std::max<int64_t>(n_events[b], 1);
        wc.cntmax = (double)std::max(cntmax_h[b], 1u);
        wc.dtmax = dtmax_h[b];
    }
    HIPCHK(c, hipMemcpyAsync(c->d_wc, c->h_wc, (size_t)n_windows * sizeof(WinConst), hipMemcpyHostToDevice, c->stream));
    c->staged = true;
    c->have_eval = false;
    c->err.clear();
    if (sw_flags & EINCM_SW_DEFER_CONSTANTS) {       // event-sharded mode: the caller sums the shards' IUEs first
        c->constants_pending = true;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return EINCM_OK;
    }
    {
        std::vector<double> zero((size_t)n_windows * 2, 0.0);
        std::vector<double> val((size_t)n_windows);
        const eincm_params p = zero_pass_params();
        const int rc = evaluate(c, zero.data(), 1, 1, &p, val.data(), nullptr, nullptr, true);
        if (rc != EINCM_OK && rc != EINCM_ERR_NONFINITE) { c->staged = false; return rc; }
    }
    const int rc = store_constants(c);
    c->err.clear();
    return rc;
}

// The concatenated forms: per-window pointers into the caller's arrays.
static int set_windows_concat(eincm_ctx* c, int n_windows, int n_refs, const int64_t* n_events, const int16_t* xs, const int16_t* ys,
                              const double* ts, const double* edges, const double* edge_ts, uint32_t flags) {
    if (!c) return EINCM_ERR_ARG;
    if (!n_events || !xs || !ys || !ts || !edges || !edge_ts) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (n_windows < 1 || n_windows > c->maxB) return fail(c, EINCM_ERR_ARG, "n_windows %d outside 1..%d", n_windows, c->maxB);
    std::vector<const int16_t*> xw((size_t)n_windows), yw((size_t)n_windows);
    std::vector<const double*> tw((size_t)n_windows), ew((size_t)n_windows);
    int64_t base = 0;
    const size_t img = (size_t)c->H * c->W;
    for (int b = 0; b < n_windows; ++b) {
        xw[b] = xs + base; yw[b] = ys + base; tw[b] = ts + base; ew[b] = edges + (size_t)b * (size_t)std::max(n_refs, 0) * img;
        base += std::max<int64_t>(n_events[b], 0);
    }
    return set_windows_impl(c, n_windows, n_refs, n_events, xw.data(), yw.data(), tw.data(), ew.data(), edge_ts, flags);
}

int eincm_set_windows(eincm_ctx* c, int n_windows, int n_refs, const int64_t* n_events, const int16_t* xs, const int16_t* ys,
                      const double* ts, const double* edges, const double* edge_ts) {
    return set_windows_concat(c, n_windows, n_refs, n_events, xs, ys, ts, edges, edge_ts, 0u);
}

int eincm_set_windows_ex(eincm_ctx* c, int n_windows, int n_refs, const int64_t* n_events, const int16_t* xs, const int16_t* ys,
                         const double* ts, const double* edges, const double* edge_ts, uint32_t flags) {
    return set_windows_concat(c, n_windows, n_refs, n_events, xs, ys, ts, edges, edge_ts, flags);
}

int eincm_set_windows_ptrs(eincm_ctx* c, int n_windows, int n_refs, const int64_t* n_events, const int16_t* const* xs,
                           const int16_t* const* ys, const double* const* ts, const double* const* edges, const double* edge_ts,
                           uint32_t flags) {
    return set_windows_impl(c, n_windows, n_refs, n_events, xs, ys, ts, edges, edge_ts, flags);
}

// ---- event-sharded evaluation: forward half / [caller all-reduces the IWE stack] / finishing half ----
int eincm_forward_iwe(eincm_ctx* c, const double* theta, int h, int w, const eincm_params* p, int want_grad) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_forward_iwe called before eincm_set_windows");
    if (!p || h < 1 || w < 1) return fail(c, EINCM_ERR_ARG, "bad argument");
    if (p->method < 0 || p->method > EINCM_METHOD_CUBIC) return fail(c, EINCM_ERR_ARG, "method %d unknown", p->method);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> zero;
    eincm_params pz;
    if (!theta) {                                    // NULL theta = the theta = 0 pass that yields the window constants
        zero.assign((size_t)c->g.B * 2, 0.0);
        theta = zero.data(); h = 1; w = 1; pz = zero_pass_params(); p = &pz; want_grad = 0;
    } else if (c->constants_pending) {
        return fail(c, EINCM_ERR_STATE, "window constants pending: run eincm_forward_iwe(NULL theta), sum the IWE stacks, eincm_finish_constants");
    }
    int rc = eval_begin(c, theta, h, w, p, want_grad != 0);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));     // the IWE stack is complete: safe to reduce on any stream
    return EINCM_OK;
}

int eincm_finish_loss_grad(eincm_ctx* c, double* value, double* grad, eincm_aux* aux) {
    if (!c) return EINCM_ERR_ARG;
    if (!value) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (c->constants_pending) return fail(c, EINCM_ERR_STATE, "window constants pending (eincm_finish_constants)");
    HIPCHK(c, hipSetDevice(c->device));
    return eval_end(c, value, grad, aux);
}

// theta and gradient resident in HBM (a caller whose optimiser lives on the GPU): nothing but the scalars crosses PCIe.
int eincm_loss_grad_device(eincm_ctx* c, const double* theta_dev, int h, int w, const eincm_params* p, double theta_abs_max,
                           double* value, double* grad_dev, eincm_aux* aux) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_loss_grad_device called before eincm_set_windows");
    if (!theta_dev || !p || !value || h < 1 || w < 1) return fail(c, EINCM_ERR_ARG, "bad argument");
    if (p->method < 0 || p->method > EINCM_METHOD_CUBIC) return fail(c, EINCM_ERR_ARG, "method %d unknown", p->method);
    if (c->constants_pending) return fail(c, EINCM_ERR_STATE, "window constants pending (eincm_finish_constants)");
    if (c->device_results) return fail(c, EINCM_ERR_STATE, "eincm_set_device_results is on: use the split finishing half");
    HIPCHK(c, hipSetDevice(c->device));
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, theta_dev) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != c->device) {
        (void)hipGetLastError();
        return fail(c, EINCM_ERR_ARG, "theta_dev is not memory of device %d", c->device);
    }
    if (grad_dev && (hipPointerGetAttributes(&at, grad_dev) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != c->device)) {
        (void)hipGetLastError();
        return fail(c, EINCM_ERR_ARG, "grad_dev is not memory of device %d", c->device);
    }
    struct Reset { eincm_ctx* c; ~Reset() { c->theta_dev_in = nullptr; c->grad_dev_out = nullptr; c->vmax_hint = -1.0; } } reset{c};
    c->theta_dev_in = theta_dev; c->grad_dev_out = grad_dev; c->vmax_hint = theta_abs_max;
    int rc = eval_begin(c, nullptr, h, w, p, grad_dev != nullptr);
    if (rc) return rc;
    rc = eval_end_launch(c);
    if (rc) { (void)hipStreamSynchronize(c->stream); c->pend.active = false; c->pend.launched = false; return rc; }
    return eval_end_collect(c, value, nullptr, aux);
}

int eincm_set_device_results(eincm_ctx* c, int on) {
    if (!c) return EINCM_ERR_ARG;
    if (c->pend.active) return fail(c, EINCM_ERR_STATE, "an evaluation is in flight");
    c->device_results = on != 0;
    return EINCM_OK;
}

int eincm_finish_launch(eincm_ctx* c) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->device_results) return fail(c, EINCM_ERR_STATE, "eincm_finish_launch needs eincm_set_device_results(ctx, 1)");
    if (c->constants_pending) return fail(c, EINCM_ERR_STATE, "window constants pending (eincm_finish_constants)");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = eval_end_launch(c);
    if (rc) { (void)hipStreamSynchronize(c->stream); c->pend.active = false; c->pend.launched = false; return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));     // the gradient is complete in HBM: safe to reduce on any stream
    return EINCM_OK;
}

int eincm_grad_device_ptr(eincm_ctx* c, void** dptr, int64_t* n_doubles) {
    if (!c || !dptr || !n_doubles) return EINCM_ERR_ARG;
    if (!c->pend.active || !c->pend.launched || !c->pend.want_grad) return fail(c, EINCM_ERR_STATE, "no launched gradient evaluation (eincm_finish_launch)");
    *dptr = c->d_grad; *n_doubles = (int64_t)c->g.B * c->pend.h * c->pend.w * 2;
    return EINCM_OK;
}

int eincm_finish_collect(eincm_ctx* c, double* value, double* grad, eincm_aux* aux) {
    if (!c) return EINCM_ERR_ARG;
    if (!value) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (!c->pend.active || !c->pend.launched) return fail(c, EINCM_ERR_STATE, "eincm_finish_collect without eincm_finish_launch");
    HIPCHK(c, hipSetDevice(c->device));
    return eval_end_collect(c, value, grad, aux);
}

int eincm_finish_constants(eincm_ctx* c) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->constants_pending) return fail(c, EINCM_ERR_STATE, "no deferred window constants");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> val((size_t)c->g.B);
    const int rc = eval_end(c, val.data(), nullptr, nullptr);
    if (rc != EINCM_OK && rc != EINCM_ERR_NONFINITE) return rc;
    const int rc2 = store_constants(c);
    c->err.clear();
    return rc2;
}

int eincm_iwe_device_ptr(eincm_ctx* c, void** dptr, int64_t* n_words) {
    if (!c || !dptr || !n_words) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "no staged windows");
    *dptr = c->d_acc; *n_words = (int64_t)c->g.B * c->g.R * c->g.H * c->g.W;
    return EINCM_OK;
}


int eincm_mask_device_ptr(eincm_ctx* c, void** dptr, int64_t* n_bytes) {
    if (!c || !dptr || !n_bytes) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "no staged windows");
    *dptr = c->d_mask; *n_bytes = (int64_t)c->g.B * c->g.H * c->g.W;
    return EINCM_OK;
}

int eincm_loss_grad_async(eincm_ctx* c, const double* theta, int h, int w, const eincm_params* p, int want_grad) {
    return eincm_loss_grad_masked_async(c, theta, h, w, p, nullptr, want_grad);
}

int eincm_loss_grad_masked_async(eincm_ctx* c, const double* theta, int h, int w, const eincm_params* p, const uint8_t* active, int want_grad) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_loss_grad_async called before eincm_set_windows");
    if (c->constants_pending) return fail(c, EINCM_ERR_STATE, "window constants are not finished (eincm_finish_constants)");
    if (!theta || !p) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (h < 1 || w < 1) return fail(c, EINCM_ERR_ARG, "theta shape (%d,%d,2) invalid", h, w);
    if (p->method < 0 || p->method > EINCM_METHOD_CUBIC) return fail(c, EINCM_ERR_ARG, "method %d unknown", p->method);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = eval_begin(c, theta, h, w, p, want_grad != 0, active);
    if (rc) return rc;
    rc = eval_end_launch(c);
    if (rc) { (void)hipStreamSynchronize(c->stream); c->pend.active = false; c->pend.launched = false; }
    return rc;
}

int eincm_loss_grad_wait(eincm_ctx* c, double* value, double* grad, eincm_aux* aux) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->pend.active || !c->pend.launched) return fail(c, EINCM_ERR_STATE, "eincm_loss_grad_wait without eincm_loss_grad_async");
    HIPCHK(c, hipSetDevice(c->device));
    return eval_end_collect(c, value, grad, aux);      // drains the stream and clears the pending state on every path
}

int eincm_loss_grad(eincm_ctx* c, const double* theta, int h, int w, const eincm_params* p, double* value, double* grad, eincm_aux* aux) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_loss_grad called before eincm_set_windows");
    if (c->constants_pending) return fail(c, EINCM_ERR_STATE, "window constants pending (eincm_finish_constants)");
    if (!theta || !p || !value) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (h < 1 || w < 1) return fail(c, EINCM_ERR_ARG, "theta shape (%d,%d,2) invalid", h, w);
    if (p->method < 0 || p->method > EINCM_METHOD_CUBIC) return fail(c, EINCM_ERR_ARG, "method %d unknown", p->method);
    HIPCHK(c, hipSetDevice(c->device));
    return evaluate(c, theta, h, w, p, value, grad, aux, false);
}

int eincm_loss_grad_masked(eincm_ctx* c, const double* theta, int h, int w, const eincm_params* p, const uint8_t* active,
                           double* value, double* grad, eincm_aux* aux) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_loss_grad_masked called before eincm_set_windows");
    if (c->constants_pending) return fail(c, EINCM_ERR_STATE, "window constants pending (eincm_finish_constants)");
    if (!theta || !p || !value) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (h < 1 || w < 1) return fail(c, EINCM_ERR_ARG, "theta shape (%d,%d,2) invalid", h, w);
    if (p->method < 0 || p->method > EINCM_METHOD_CUBIC) return fail(c, EINCM_ERR_ARG, "method %d unknown", p->method);
    HIPCHK(c, hipSetDevice(c->device));
    return evaluate(c, theta, h, w, p, value, grad, aux, false, active);
}

int eincm_handover_loss_grad(eincm_ctx* c, const double* a, const double* prev_theta, const double* theta, int h, int w,
                             const eincm_params* p, double* value, double* dvalue_da) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_handover_loss_grad called before eincm_set_windows");
    if (!a || !prev_theta || !theta || !p || !value) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (h < 1 || w < 1) return fail(c, EINCM_ERR_ARG, "theta shape (%d,%d,2) invalid", h, w);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nth = (size_t)h * w * 2, B = c->g.B;
    std::vector<double> tho(B * nth), grad;
    for (size_t b = 0; b < B; ++b)
        for (size_t i = 0; i < nth; ++i)   // losses.py:269
            tho[b * nth + i] = a[b] * prev_theta[b * nth + i] + (1.0 - a[b]) * theta[b * nth + i];
    if (dvalue_da) grad.resize(B * nth);
    const int rc = evaluate(c, tho.data(), h, w, p, value, dvalue_da ? grad.data() : nullptr, nullptr, false);
    if (rc != EINCM_OK && rc != EINCM_ERR_NONFINITE) return rc;
    if (dvalue_da) {
        for (size_t b = 0; b < B; ++b) {
            double s = 0.0;
            for (size_t i = 0; i < nth; ++i) s += grad[b * nth + i] * (prev_theta[b * nth + i] - theta[b * nth + i]);
            dvalue_da[b] = s;
        }
    }
    return rc;
}

int eincm_objectives(eincm_ctx* c, const double* Theta, eincm_objectives_out* out) {
    if (!c) return EINCM_ERR_ARG;
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "eincm_objectives called before eincm_set_windows");
    if (!Theta || !out) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    HIPCHK(c, hipSetDevice(c->device));
    const Geom& g = c->g;
    eincm_params p{};
    p.alpha = 1.0; p.beta = 1.0; p.gamma = 0.0; p.delta = 0.0; p.cur_pyr_lvl = 0; p.method = EINCM_METHOD_BILINEAR;
    p.flags = EINCM_PF_FULL_AUX;
    std::vector<double> val((size_t)g.B);
    const int rc = evaluate(c, Theta, g.H, g.W, &p, val.data(), nullptr, nullptr, false);
    if (rc != EINCM_OK && rc != EINCM_ERR_NONFINITE) return rc;
    std::vector<double> tvp((size_t)g.B * g.ntiles * 3);
    HIPCHK(c, hipMemcpy(tvp.data(), c->d_tvparts, tvp.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double HW = (double)g.H * g.W;
    for (int b = 0; b < g.B; ++b) {
        eincm_objectives_out& o = out[b];
        memset(&o, 0, sizeof o);
        const OutScal& s = c->h_outs[b];
        const WinConst& wc = c->h_wc[b];
        o.n_refs = g.R;
        o.zero_contrast = wc.c0_gradmag; o.zero_variance = wc.c0_var; o.zero_iwe_divergence = wc.d0;
        for (int r = 0; r < g.R; ++r) {
            o.correlations[r] = s.corr[r];
            o.zero_correlations[r] = wc.zc[r];
            o.rel_correlations[r] = s.corr[r] / (wc.zc[r] + EPSN);
            o.contrasts[r] = s.contrast_gm[r];
            o.rel_contrasts[r] = s.contrast_gm[r] / (wc.c0_gradmag + EPSN);
            o.iwe_divergences[r] = s.div[r];
            o.rel_iwe_divergences[r] = s.div[r] / (wc.d0 + EPSN);
            o.variances[r] = s.var[r];
            o.flow_warp_losses[r] = s.var[r] / wc.c0_var;          // contrast_metrics.py:17
            o.multi_ref_weights[r] = wc.mrw[r];
        }
        o.theta_total_variation = s.tv;
        double td = 0.0;
        for (int k = 0; k < g.ntiles; ++k) td += tvp[((size_t)b * g.ntiles + k) * 3 + 2];
        o.theta_divergence = td / HW;
    }
    return rc;
}

static int copy_out(eincm_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return EINCM_ERR_ARG;
    if (!dst) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (!c->staged) return fail(c, EINCM_ERR_STATE, "no staged windows");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return EINCM_OK;
}

int eincm_get_iwes(eincm_ctx* c, float* iwes) {
    if (c && !c->have_eval) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    return copy_out(c, iwes, c ? c->d_iwe : nullptr, c ? (size_t)c->g.B * c->g.R * c->g.H * c->g.W * sizeof(float) : 0);
}
int eincm_get_zero_iwe(eincm_ctx* c, float* z) {
    return copy_out(c, z, c ? c->d_zero_iwe : nullptr, c ? (size_t)c->g.B * c->g.H * c->g.W * sizeof(float) : 0);
}
int eincm_get_image_grad(eincm_ctx* c, float* G) {
    if (c && !c->have_eval) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    if (c && !c->G_valid) return fail(c, EINCM_ERR_STATE, "no dL/dIWE image: the last evaluation had no gradient, or eincm_get_count_images reused the buffer");
    if (c && c->last_composed) {
        // the gather composed dL/dIWE while staging its windows; materialise the same image (same function, same scalars) on demand
        HIPCHK(c, hipSetDevice(c->device));
        const Geom& g = c->last_g;
        const size_t n = (size_t)c->maxB * c->maxR * c->H * c->W;
        if (!c->d_Gimg) HIPCHK(c, dalloc(&c->d_Gimg, n));
        hipLaunchKernelGGL(k_compose, dim3(64, g.R, g.B), dim3(NT), 0, c->stream, g, c->last_ep, c->d_G, c->d_edges, c->d_iwe, c->d_parts,
                           c->d_wc, c->d_Gimg);
        HIPCHK(c, hipGetLastError());
        return copy_out(c, G, c->d_Gimg, (size_t)g.B * g.R * g.H * g.W * sizeof(float));
    }
    return copy_out(c, G, c ? c->d_G : nullptr, c ? (size_t)c->g.B * c->g.R * c->g.H * c->g.W * sizeof(float) : 0);
}
// 2-DoF evaluations skip the Theta image; build it when somebody asks for it
static int ensure_theta_image(eincm_ctx* c) {
    if (c->Theta_valid) return EINCM_OK;
    if (c->last_theta11.size() != (size_t)c->g.B * 2) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = ensure_resample(c, 1, 1, EINCM_METHOD_BILINEAR);      // a (1,1,2) theta upsamples to a constant with every method
    if (rc) return rc;
    memcpy(c->h_theta, c->last_theta11.data(), c->last_theta11.size() * sizeof(double));
    c->g.wmask = ~0ull;                              // every window's image, whatever the last evaluation masked
    static const ThetaArgBig targ{};
    launch_theta_image(c, 1, 1, false, false, targ, c->h_theta, false);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EINCM_OK;
}

int eincm_get_scaled_theta(eincm_ctx* c, double* T) {
    if (c && !c->have_eval) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    if (c && c->staged) { const int rc = ensure_theta_image(c); if (rc) return rc; }
    return copy_out(c, T, c ? c->d_Theta : nullptr, c ? (size_t)c->g.B * c->g.H * c->g.W * 2 * sizeof(double) : 0);
}

int eincm_get_count_images(eincm_ctx* c, uint32_t* counts) {
    if (!c) return EINCM_ERR_ARG;
    if (!counts) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (!c->staged || !c->have_eval) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc = ensure_theta_image(c); if (rc) return rc; }
    const Geom& g = c->g;
    const size_t n = (size_t)g.B * g.R * g.H * g.W;
    // the dL/dIWE buffer is free between evaluations and has exactly this many 4-byte cells
    uint32_t* d = reinterpret_cast<uint32_t*>(c->d_G);
    c->G_valid = false;
    HIPCHK(c, hipMemsetAsync(d, 0, n * sizeof(uint32_t), c->stream));
    if (c->n_items > 0)
        hipLaunchKernelGGL(k_count, dim3(event_grid(c)), dim3(NT), 0, c->stream, g, c->n_items, c->d_items, c->d_xy, c->d_t, c->d_Theta,
                           c->d_edge_ts, d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(counts, d, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EINCM_OK;
}

// ---------------------------------------------------------------------------------------------
// SURVEY row f-4 (eincm_edges.hip.h)
// ---------------------------------------------------------------------------------------------
static int ensure_buf(eincm_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return EINCM_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return EINCM_OK;
}
#define ENSURE(c, buf, bytes) do { const int rc_ = ensure_buf((c), (buf), (bytes)); if (rc_ != EINCM_OK) return rc_; } while (0)

// 'warped_xs' / 'warped_ys' of compute_loss_objectives (losses.py:58,90-91) for one window under the last evaluation's theta
int eincm_get_warped_events(eincm_ctx* c, int window, double* warped_xs, double* warped_ys) {
    if (!c) return EINCM_ERR_ARG;
    if (!warped_xs || !warped_ys) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (!c->staged || !c->have_eval) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    const Geom& g = c->g;
    if (window < 0 || window >= g.B) return fail(c, EINCM_ERR_ARG, "window %d of %d", window, g.B);
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc = ensure_theta_image(c); if (rc) return rc; }
    const int64_t n = c->win_events[window];
    if (n == 0) return EINCM_OK;
    int64_t off = 0;
    for (int b = 0; b < window; ++b) off += c->win_events[b];
    const size_t bytes = (size_t)g.R * (size_t)n * sizeof(double);
    ENSURE(c, c->e_a, 2 * bytes);                              // scratch of the edge routines: free between calls
    double* dx = static_cast<double*>(c->e_a.p);
    double* dy = dx + (size_t)g.R * (size_t)n;
    const int grid = (int)std::min<int64_t>((n + NT - 1) / NT, 8192);
    hipLaunchKernelGGL(k_warp_events, dim3(grid), dim3(NT), 0, c->stream, g, window, (long long)n, c->d_raw_x + off, c->d_raw_y + off,
                       c->d_raw_t + off, c->d_Theta, c->d_edge_ts, dx, dy);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(warped_xs, dx, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(warped_ys, dy, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EINCM_OK;
}


int eincm_inv_dist_transform(eincm_ctx* c, const uint8_t* edge_img, int n, int formulation, double alpha, double d_sat,
                             double* out, int32_t* sqdist) {
    if (!c) return EINCM_ERR_ARG;
    if (!edge_img || (!out && !sqdist)) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (n < 1) return fail(c, EINCM_ERR_ARG, "n = %d images", n);
    if (formulation < EINCM_EDT_EXPONENTIAL || formulation > EINCM_EDT_LOGARITHMIC)
        return fail(c, EINCM_ERR_ARG, "unknown formulation %d", formulation);
    if (out && formulation == EINCM_EDT_EXPONENTIAL && !(alpha > 0.0)) return fail(c, EINCM_ERR_ARG, "alpha = %g must be positive", alpha);
    if (out && formulation == EINCM_EDT_LINEAR_BOUND && !(d_sat > 0.0)) return fail(c, EINCM_ERR_ARG, "d_sat = %g must be positive", d_sat);
    HIPCHK(c, hipSetDevice(c->device));
    const int H = c->H, W = c->W;
    const size_t npix = (size_t)H * W, tot = npix * n;
    ENSURE(c, c->e_u8, tot); ENSURE(c, c->e_g, tot * 4); ENSURE(c, c->e_sq, tot * 4); ENSURE(c, c->e_misc, (size_t)n * 8);
    if (out) ENSURE(c, c->e_a, tot * 8);
    uint32_t* d_misc = static_cast<uint32_t*>(c->e_misc.p);           // [n] edge pixel count | [n] max squared distance
    HIPCHK(c, hipMemcpyAsync(c->e_u8.p, edge_img, tot, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_misc, 0, (size_t)n * 8, c->stream));
    const int gx = (W + NT - 1) / NT;
    hipLaunchKernelGGL(k_edt_cols, dim3(gx, n), dim3(NT), 0, c->stream, H, W, static_cast<const uint8_t*>(c->e_u8.p),
                       static_cast<uint32_t*>(c->e_g.p), d_misc);
    hipLaunchKernelGGL(k_edt_rows, dim3(gx, H, n), dim3(NT), 0, c->stream, H, W, static_cast<const uint32_t*>(c->e_g.p),
                       static_cast<int32_t*>(c->e_sq.p), d_misc + n);
    HIPCHK(c, hipGetLastError());
    std::vector<uint32_t> misc((size_t)n * 2);
    HIPCHK(c, hipMemcpyAsync(misc.data(), d_misc, misc.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i)
        if (misc[i] == 0) return fail(c, EINCM_ERR_ARG, "edge image %d has no edge pixel: its distance transform is undefined", i);
    if (sqdist) HIPCHK(c, hipMemcpyAsync(sqdist, c->e_sq.p, tot * 4, hipMemcpyDeviceToHost, c->stream));
    if (out) {
        const int nb = (int)std::min<size_t>((npix + NT - 1) / NT, 1024);
        hipLaunchKernelGGL(k_edt_finish, dim3(nb, n), dim3(NT), 0, c->stream, (int64_t)npix, static_cast<const int32_t*>(c->e_sq.p),
                           d_misc + n, formulation, alpha, d_sat, static_cast<double*>(c->e_a.p));
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out, c->e_a.p, tot * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return EINCM_OK;
}

int eincm_gaussian_blur(eincm_ctx* c, const double* src, int n, double sigma, double* dst) {
    if (!c) return EINCM_ERR_ARG;
    if (!src || !dst) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (n < 1) return fail(c, EINCM_ERR_ARG, "n = %d images", n);
    if (!(sigma > 0.0) || !std::isfinite(sigma)) return fail(c, EINCM_ERR_ARG, "sigma = %g must be positive", sigma);
    // cv::GaussianBlur on CV_64F with ksize = Size(): ksize = cvRound(sigma*4*2 + 1) | 1; cv::getGaussianKernel (sigma > 0 branch)
    const int taps = (int)std::nearbyint(sigma * 4 * 2 + 1) | 1;
    const int radius = taps / 2;
    if (taps > BLUR_MAX_TAPS) return fail(c, EINCM_ERR_UNSUPPORTED, "sigma = %g needs %d taps (> %d)", sigma, taps, BLUR_MAX_TAPS);
    if (radius >= c->W || radius >= c->H)
        return fail(c, EINCM_ERR_ARG, "kernel radius %d does not fit the %d x %d sensor (BORDER_REFLECT_101)", radius, c->H, c->W);
    std::vector<double> k((size_t)taps);
    double sum = 0.0;
    for (int i = 0; i < taps; ++i) { const double x = i - (taps - 1) * 0.5; k[i] = std::exp(-0.5 / (sigma * sigma) * x * x); sum += k[i]; }
    for (double& v : k) v /= sum;
    HIPCHK(c, hipSetDevice(c->device));
    const int H = c->H, W = c->W;
    const size_t tot = (size_t)H * W * n;
    ENSURE(c, c->e_a, tot * 8); ENSURE(c, c->e_b, tot * 8); ENSURE(c, c->e_kern, (size_t)BLUR_MAX_TAPS * 8);
    HIPCHK(c, hipMemcpyAsync(c->e_kern.p, k.data(), k.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->e_a.p, src, tot * 8, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((W + NT - 1) / NT, H, n);
    hipLaunchKernelGGL(k_blur, grid, dim3(NT), 0, c->stream, H, W, 0, radius, static_cast<const double*>(c->e_kern.p),
                       static_cast<const double*>(c->e_a.p), static_cast<double*>(c->e_b.p));
    hipLaunchKernelGGL(k_blur, grid, dim3(NT), 0, c->stream, H, W, 1, radius, static_cast<const double*>(c->e_kern.p),
                       static_cast<const double*>(c->e_b.p), static_cast<double*>(c->e_a.p));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dst, c->e_a.p, tot * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // k (host vector) stays alive until here
    return EINCM_OK;
}

int eincm_tiled_objectives(eincm_ctx* c, int tile_h, int tile_w, eincm_tiled_out* out) {
    if (!c) return EINCM_ERR_ARG;
    if (!out) return fail(c, EINCM_ERR_ARG, "null pointer argument");
    if (!c->staged || !c->have_eval) return fail(c, EINCM_ERR_STATE, "no evaluation yet");
    Geom g = c->g;
    g.nparts = c->last_nparts;
    if (tile_h < 1 || tile_w < 1 || tile_h > g.H || tile_w > g.W)
        return fail(c, EINCM_ERR_ARG, "tile %d x %d does not fit the %d x %d sensor", tile_h, tile_w, g.H, g.W);
    HIPCHK(c, hipSetDevice(c->device));
    const int ntx = g.W / tile_w, nty = g.H / tile_h, ntl = ntx * nty;
    const int nb = std::min((g.H * g.W + NT - 1) / NT, 256);
    const size_t n_t = (size_t)g.B * g.R * ntl * 3, n_p = (size_t)g.B * g.R * nb * 3;
    ENSURE(c, c->e_out, (n_t + n_p) * 8);
    double* d_t = static_cast<double*>(c->e_out.p);
    double* d_p = d_t + n_t;
    g.wmask = ~0ull;
    hipLaunchKernelGGL(k_tiled, dim3(ntl, g.R, g.B), dim3(NT), 0, c->stream, g, tile_h, tile_w, ntx, c->d_iwe, c->d_edges, c->d_parts, d_t);
    hipLaunchKernelGGL(k_pair_objectives, dim3(nb, g.R, g.B), dim3(NT), 0, c->stream, g, c->d_iwe, c->d_edges, c->d_parts, d_p);
    HIPCHK(c, hipGetLastError());
    std::vector<double> hv(n_t + n_p);
    HIPCHK(c, hipMemcpyAsync(hv.data(), d_t, hv.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double HW = (double)g.H * g.W;
    for (int b = 0; b < g.B; ++b) {
        eincm_tiled_out& o = out[b];
        memset(&o, 0, sizeof o);
        o.n_refs = g.R; o.n_tiles = ntl;
        for (int r = 0; r < g.R; ++r) {
            const double* t = hv.data() + ((size_t)b * g.R + r) * ntl * 3;
            for (int k = 0; k < ntl; ++k) {
                o.adaptive_mean_gradient_magnitude[r] += t[k * 3];
                o.adaptive_variance[r] += t[k * 3 + 1];
                o.adaptive_mean_squared_error[r] += t[k * 3 + 2];
            }
            double q[3] = {0.0, 0.0, 0.0};                       // the workgroups' partials, added in index order
            for (int k = 0; k < nb; ++k) {
                const double* pk = hv.data() + n_t + (((size_t)b * g.R + r) * nb + k) * 3;
                q[0] += pk[0]; q[1] += pk[1]; q[2] += pk[2];
            }
            o.sum_squared_error[r] = q[0];
            o.sum_hadamard_product[r] = q[1];
            o.mean_hadamard_product[r] = q[1] / HW;
            o.joint_contrast[r] = q[2] / HW;
        }
    }
    return EINCM_OK;
}

int eincm_get_timings_total(eincm_ctx* c, eincm_timings* t, int64_t* n_evals, int reset) {
    if (!c || !t || !n_evals) return EINCM_ERR_ARG;
    if (!(c->cflags & (EINCM_CF_TIMING | EINCM_CF_TIMING_DOMINANT)))
        return fail(c, EINCM_ERR_STATE, "context was created without EINCM_CF_TIMING / EINCM_CF_TIMING_DOMINANT");
    if (c->pend.active) return fail(c, EINCM_ERR_STATE, "an evaluation is in flight");
    const int rc = drain_event_ring(c, 0);
    if (rc) return rc;
    *t = c->sum_t; *n_evals = c->sum_n;
    if (reset) { c->sum_t = eincm_timings{}; c->sum_n = 0; }
    return EINCM_OK;
}

int eincm_get_host_profile(eincm_ctx* c, double* us, int64_t* n_evals, int reset) {
    if (!c || !us || !n_evals) return EINCM_ERR_ARG;
    for (int i = 0; i < EINCM_N_HOST_PHASES; ++i) us[i] = c->hp_us[i];
    *n_evals = c->hp_n;
    if (reset) { for (double& v : c->hp_us) v = 0.0; c->hp_n = 0; }
    return EINCM_OK;
}

int eincm_get_launch_policy(eincm_ctx* c, double* out) {
    if (!c || !out) return EINCM_ERR_ARG;
    out[EINCM_LP_SEG_GATHER] = c->seg_used; out[EINCM_LP_SEG_SPLAT] = c->seg_s_used; out[EINCM_LP_SEG_GATHER_2DOF] = c->seg_2_used;
    out[EINCM_LP_SEG_SPLAT_SHORT] = c->seg_sh_used; out[EINCM_LP_PITCH_POLICY] = c->pitch_policy;
    out[EINCM_LP_SPAN_SPLAT] = c->tspan_s; out[EINCM_LP_SPAN_GATHER] = c->tspan_a; out[EINCM_LP_SPAN_GATHER_2DOF] = c->tspan_2;
    const bool evaluated = c->policy_evaluated;
    out[EINCM_LP_CAP_SPLAT] = evaluated ? c->g.wincap : 0; out[EINCM_LP_CAP_GATHER] = evaluated ? c->g.wincap_a : 0;
    out[EINCM_LP_CAP_GATHER_2DOF] = evaluated ? c->wincap_2 : 0;
    out[EINCM_LP_PITCH_ALIGNED] = evaluated ? ((c->g.pitch_aligned ? 1 : 0) | (c->pend.pal_2 ? 2 : 0)) : 0;
    out[EINCM_LP_SPLAT_SHORT] = evaluated && c->pend.splat_short ? 1 : 0;
    return EINCM_OK;
}

int eincm_set_timed_kernels(eincm_ctx* c, int splat, int gather) {
    if (!c) return EINCM_ERR_ARG;
    if (!(c->cflags & EINCM_CF_TIMING_DOMINANT)) return fail(c, EINCM_ERR_STATE, "context was created without EINCM_CF_TIMING_DOMINANT");
    if (c->pend.active) return fail(c, EINCM_ERR_STATE, "an evaluation is in flight");
    c->time_splat = splat != 0; c->time_gather = gather != 0;
    return EINCM_OK;
}

int eincm_set_timing_period(eincm_ctx* c, int period) {
    if (!c) return EINCM_ERR_ARG;
    if (!(c->cflags & EINCM_CF_TIMING_DOMINANT)) return fail(c, EINCM_ERR_STATE, "context was created without EINCM_CF_TIMING_DOMINANT");
    if (c->pend.active) return fail(c, EINCM_ERR_STATE, "an evaluation is in flight");
    if (period < 1) return fail(c, EINCM_ERR_ARG, "period %d", period);
    c->time_period = period; c->time_counter = 0;
    return EINCM_OK;
}

int eincm_get_timings(eincm_ctx* c, eincm_timings* t) {
    if (!c || !t) return EINCM_ERR_ARG;
    if (!(c->cflags & (EINCM_CF_TIMING | EINCM_CF_TIMING_DOMINANT)))
        return fail(c, EINCM_ERR_STATE, "context was created without EINCM_CF_TIMING / EINCM_CF_TIMING_DOMINANT");
    if (c->pend.active) return fail(c, EINCM_ERR_STATE, "an evaluation is in flight");
    const int rc = drain_event_ring(c, 0);
    if (rc) return rc;
    *t = c->last_t;
    return EINCM_OK;
}

}  // extern "C"
