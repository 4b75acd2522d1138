/* eincm.h — C-ABI of libeincm_hip.so: the MI355X (gfx950) EINCM objective-and-gradient engine.
 *
 * Drop-in boundary for ONE path of robotic-vision-lab/Edge-Informed-Contrast-Maximization:
 *     loss(theta, events, edge_maps) -> (value, grad)
 * i.e. eincm.losses.loss_func (src/eincm/losses.py:108-205) and the gradient the reference obtains through
 * jaxopt.ScipyMinimize(jit=True) -> jax.value_and_grad (src/eincm/solver.py:165-173).  The reference has no
 * FFI of its own (pure Python on JAX); the binding a maintainer would add is a ctypes stub, shown in
 * INTEGRATION.md.  Plain pointers and sizes only; no torch / HIP types cross this boundary.
 *
 * Conventions
 *   - all host arrays are C-order (row major), caller-owned; nothing is retained after a call returns
 *   - theta / grad / value / ts / edges / edge_ts are double (the SciPy side of the reference is float64,
 *     SURVEY 8b); x, y are int16 (the reference's event wire format, exp_mgr.py:283-284)
 *   - a context is NOT thread-safe; one context per (GPU, caller thread); every call is synchronous except the *_async forms
 *   - return value: 0 = ok, < 0 = error (see EINCM_ERR_*); eincm_last_error() gives the message
 */
#ifndef EINCM_H
#define EINCM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EINCM_ABI_VERSION 6   /* 2: eincm_iwe_device_ptr hands out the u64 fixed-point accumulator of the IWE stack; 3: eincm_set_timed_kernels, eincm_set_windows_ptrs;
                                * 4: eincm_loss_grad_masked, eincm_set_device_results / eincm_finish_launch / eincm_grad_device_ptr / eincm_finish_collect, eincm_get_host_profile;
                               * 5: eincm_get_warped_events, eincm_loss_grad_device, eincm_loss_grad_masked_async, eincm_set_timing_period;
                               * 6: eincm_get_launch_policy */

#define EINCM_OK               0
#define EINCM_ERR_ARG         -1   /* bad argument (shape, null pointer, out-of-range event coordinate) */
#define EINCM_ERR_HIP         -2   /* HIP runtime error (message holds hipGetErrorString) */
#define EINCM_ERR_STATE       -3   /* call out of order (e.g. loss_grad before set_windows) */
#define EINCM_ERR_NONFINITE   -4   /* loss or gradient is NaN/Inf (outputs are still written) */
#define EINCM_ERR_UNSUPPORTED -5   /* valid request this build does not implement (currently unused) */

/* contrast objective: 0 = mean squared Scharr gradient magnitude of the raw IWE (losses.py:70, the reference's
 * live objective); 1 = variance of the IWE (contrast_objectives.py:29-39; BASELINE config "variance-only") */
#define EINCM_CONTRAST_GRAD_MAG 0
#define EINCM_CONTRAST_VARIANCE 1

/* scale_to_sensor_size_method (theta_utils.py:25-35 -> jax.image.scale_and_translate kernels) */
#define EINCM_METHOD_BILINEAR 0    /* 'linear' / 'bilinear' / 'triangle' */
#define EINCM_METHOD_LANCZOS3 1
#define EINCM_METHOD_LANCZOS5 2
#define EINCM_METHOD_CUBIC    3    /* 'cubic' / 'bicubic' */

/* eincm_params.flags */
#define EINCM_PF_NO_TV_GRAD 2u     /* leave the TV term out of the gradient (its value stays in the loss): event-sharded
                                      evaluation adds the replicated TV gradient on one shard only */
#define EINCM_PF_FULL_AUX   1u     /* also evaluate the report-only terms: IWE divergence (losses.py:79-81),
                                      TV at any gamma (losses.py:75), FWL (losses.py:84) */

/* eincm_create flags */
#define EINCM_CF_TIMING     1u     /* every stage timed with HIP events (eincm_get_timings): attached to the launch where a stage is one
                                     * kernel, marker events otherwise and around the whole evaluation; costs ~10 % of a step */
#define EINCM_CF_TIMING_DOMINANT 2u /* the two event kernels (k_splat, k_gather) are launched with their own start / stop events
                                     * (hipExtLaunchKernelGGL: no marker packets on the stream) and the events are read when the
                                     * timings are asked for, not after every evaluation; total_ms stays 0 in this mode.
                                     * eincm_set_timed_kernels narrows it to one of the two */

typedef struct eincm_ctx eincm_ctx;

/* The keyword arguments hydra binds to loss_func (configs/theta_loss_func/default.yaml:1-9) */
typedef struct eincm_params {
    double   alpha;          /* contrast weight            losses.py:116 */
    double   beta;           /* correlation weight         losses.py:117 */
    double   gamma;          /* total-variation weight     losses.py:118 (live only if cur_pyr_lvl <= 0, :171) */
    double   delta;          /* IWE-divergence weight      losses.py:119 */
    int32_t  cur_pyr_lvl;    /* losses.py:120 */
    int32_t  method;         /* EINCM_METHOD_*             losses.py:123 */
    int32_t  contrast_kind;  /* EINCM_CONTRAST_* */
    uint32_t flags;          /* EINCM_PF_* */
} eincm_params;

/* aux_info of loss_func (losses.py:195-203) without the two arrays (scaled_theta: eincm_get_scaled_theta,
 * multi_ref_weights: eincm_multi_ref_weights) */
typedef struct eincm_aux {
    double final_loss;
    double mean_rel_corr;
    double mean_rel_contrast;
    double mean_rel_iwe_divergence;   /* NaN unless EINCM_PF_FULL_AUX or delta != 0 */
    double theta_total_variation;     /* 0 if cur_pyr_lvl > 0; NaN if cur_pyr_lvl <= 0, gamma == 0 and no FULL_AUX */
} eincm_aux;

/* per-reference-time scalars of compute_loss_objectives (losses.py:89-105); arrays are [EINCM_MAX_REFS] */
#define EINCM_MAX_REFS 16
typedef struct eincm_objectives_out {
    int32_t n_refs;
    int32_t _pad;
    double correlations[EINCM_MAX_REFS];
    double zero_correlations[EINCM_MAX_REFS];
    double rel_correlations[EINCM_MAX_REFS];
    double contrasts[EINCM_MAX_REFS];
    double zero_contrast;
    double rel_contrasts[EINCM_MAX_REFS];
    double theta_total_variation;
    double theta_divergence;
    double iwe_divergences[EINCM_MAX_REFS];
    double zero_iwe_divergence;
    double rel_iwe_divergences[EINCM_MAX_REFS];
    double flow_warp_losses[EINCM_MAX_REFS];
    double multi_ref_weights[EINCM_MAX_REFS];
    double variances[EINCM_MAX_REFS];          /* extra: var(IWE_r) (contrast_kind = variance) */
    double zero_variance;
} eincm_objectives_out;

/* device time of the last eincm_loss_grad call, per kernel, from HIP events on the engine's stream */
#define EINCM_N_STAGES 10
typedef struct eincm_timings {
    float ms[EINCM_N_STAGES];   /* index = EINCM_STAGE_* ; 0 when the stage did not run */
    float total_ms;             /* first kernel start -> results on host */
} eincm_timings;
#define EINCM_STAGE_CLEAR   0
#define EINCM_STAGE_THETA   1   /* theta upsample + per-tile velocity bounds */
#define EINCM_STAGE_SPLAT   2   /* warp + 3x3 Gaussian splat -> IWE stack (dominant) */
#define EINCM_STAGE_STATS   3   /* min/max/moments/contrast reductions */
#define EINCM_STAGE_IMGRAD  4   /* dL/dIWE image */
#define EINCM_STAGE_GATHER  5   /* backward gather -> dL/dTheta */
#define EINCM_STAGE_TV      6
#define EINCM_STAGE_PROJECT 7   /* adjoint resample dL/dTheta -> dL/dtheta */
#define EINCM_STAGE_FINAL   8
#define EINCM_STAGE_COPY    9

int         eincm_abi_version(void);
const char* eincm_last_error(const eincm_ctx* ctx);   /* ctx may be NULL: error of the last failed eincm_create */

/* One context = one GPU + one stream + capacity for a batch of up to max_windows independent event windows
 * of one sensor size (H, W), each with the same number of reference times.  Windows are what
 * MultipleLevelEINCMSolver.set_datasample holds (solver.py:185-194). */
eincm_ctx*  eincm_create(int device, int H, int W, int max_refs, int max_windows, int64_t max_events_total,
                         uint32_t flags);
void        eincm_destroy(eincm_ctx* ctx);

/* Stage a batch of windows (replaces solver.set_datasample + the theta-independent half of
 * compute_loss_objectives, losses.py:54-55,66,71,80): copies events/edges to HBM, bins events by source
 * tile, and evaluates the zero-warp constants (IUE, its contrast / variance / divergence, zero_corrs).
 *   n_events[b]            events in window b (sum <= max_events_total)
 *   xs, ys, ts             windows concatenated, sum(n_events) entries; 0 <= x < W, 0 <= y < H
 *   edges                  (n_windows, n_refs, H, W); edge_ts (n_windows, n_refs)                      */
int eincm_set_windows(eincm_ctx* ctx, int n_windows, int n_refs, const int64_t* n_events,
                      const int16_t* xs, const int16_t* ys, const double* ts,
                      const double* edges, const double* edge_ts);

/* eincm_set_windows with flags.  EINCM_SW_DEFER_CONSTANTS: stage only; the zero-warp constants are finished later by
 * eincm_forward_iwe(theta = NULL) -> [sum the IWE stacks of all shards] -> eincm_finish_constants (event-sharded mode). */
#define EINCM_SW_DEFER_CONSTANTS 1u
int eincm_set_windows_ex(eincm_ctx* ctx, int n_windows, int n_refs, const int64_t* n_events,
                         const int16_t* xs, const int16_t* ys, const double* ts,
                         const double* edges, const double* edge_ts, uint32_t flags);
/* The same with one pointer per window (xs[b], ys[b], ts[b]: n_events[b] values; edges[b]: (n_refs, H, W); edge_ts: (n_windows, n_refs)
 * contiguous): a batch is staged straight from the caller's per-window arrays, as the reference holds them (one datasample tuple per
 * window, src/eincm/solver.py:185-194) - concatenating 8 x 10^6 events on the host first cost 20 of 28 ms. */
int eincm_set_windows_ptrs(eincm_ctx* ctx, int n_windows, int n_refs, const int64_t* n_events,
                           const int16_t* const* xs, const int16_t* const* ys, const double* const* ts,
                           const double* const* edges, const double* edge_ts, uint32_t flags);

/* value_and_grad(loss_func) for every staged window (losses.py:108-205 + its reverse pass).
 *   theta  (n_windows, h, w, 2)     value (n_windows)     grad (n_windows, h, w, 2) or NULL (forward only)
 *   aux    (n_windows) or NULL                                                                         */
int eincm_loss_grad(eincm_ctx* ctx, const double* theta, int h, int w, const eincm_params* p,
                    double* value, double* grad, eincm_aux* aux);

/* Asynchronous form of eincm_loss_grad: _async copies theta, enqueues the whole evaluation on the context's stream and returns;
 * _wait synchronises that stream and hands over (value, grad, aux) exactly as eincm_loss_grad does.  One host thread can keep
 * several contexts in flight this way (one HIP stream each): the host's work for one context (e.g. a solver's line-search bookkeeping)
 * overlaps another context's kernels - the pipelined lockstep solver runs 8 pyramid solves in 0.18 s instead of 0.24 s on two contexts.
 * (Evaluation throughput alone does not gain since round 2: both event kernels are throughput-bound)  No other
 * call on the context is allowed between the two. */
int eincm_loss_grad_async(eincm_ctx* ctx, const double* theta, int h, int w, const eincm_params* p, int want_grad);
int eincm_loss_grad_wait(eincm_ctx* ctx, double* value, double* grad, eincm_aux* aux);

/* eincm_loss_grad for a SUBSET of the staged windows: active[b] != 0 selects window b (NULL = all).  The workgroups of the other
 * windows leave at once, so the call costs about what its active windows cost; their value comes back NaN and their gradient zero.
 * What a lockstep batch solver needs once some of its windows have converged (the reference has no such caller: it solves one window
 * at a time, src/eincm/solver.py:209-216).  Windows beyond the 64th are always evaluated. */
int eincm_loss_grad_masked(eincm_ctx* ctx, const double* theta, int h, int w, const eincm_params* p, const uint8_t* active,
                           double* value, double* grad, eincm_aux* aux);
/* ... and its asynchronous form (collected by eincm_loss_grad_wait): a lockstep solver that drives two contexts keeps the GPU busy with
 * one group of windows while the host advances the other group's line searches */
int eincm_loss_grad_masked_async(eincm_ctx* ctx, const double* theta, int h, int w, const eincm_params* p, const uint8_t* active, int want_grad);

/* handover_loss_func (losses.py:208-276) and d/d(alpha_handover) = <dL/dtheta_ho, prev - theta>:
 *   theta_ho = a*prev_theta + (1-a)*theta;  a (n_windows), value (n_windows), dvalue_da (n_windows) or NULL */
int eincm_handover_loss_grad(eincm_ctx* ctx, const double* alpha_handover, const double* prev_theta,
                             const double* theta, int h, int w, const eincm_params* p,
                             double* value, double* dvalue_da);

/* compute_loss_objectives (losses.py:49-105) on a full-resolution Theta (n_windows, H, W, 2): forward only. */
int eincm_objectives(eincm_ctx* ctx, const double* Theta, eincm_objectives_out* out /* n_windows */);

/* Device images of the last evaluation, copied to host (parity tests / plotting):
 *   iwes (n_windows, n_refs, H, W) float;  zero_iwe (n_windows, H, W) float;
 *   image_grad = dL/dIWE (n_windows, n_refs, H, W) float;  scaled_theta (n_windows, H, W, 2) double   */
int eincm_get_iwes(eincm_ctx* ctx, float* iwes);
int eincm_get_zero_iwe(eincm_ctx* ctx, float* zero_iwe);
int eincm_get_image_grad(eincm_ctx* ctx, float* image_grad);
int eincm_get_scaled_theta(eincm_ctx* ctx, double* scaled_theta);

/* Integer image of the rounded warped coordinates under the Theta of the last evaluation:
 * counts[b,r,ry,rx] = #{events of window b with round(warp(x,y,t; tau_r)) = (rx,ry)}, JAX wrap/drop index rule
 * (the centre tap of events_to_pdf_frame, src/utils/event_utils.py:32-33,59).  counts (n_windows, n_refs, H, W) uint32.
 * The IWE is a float image; this is its integer skeleton and must equal the reference's bit for bit.
 * Overwrites the dL/dIWE image of the last evaluation (eincm_get_image_grad must be called before it). */
int eincm_get_count_images(eincm_ctx* ctx, uint32_t* counts);

/* The warped coordinates themselves: warped_xs[r,i] = x_i - Theta[y_i,x_i,0]*(t_i - tau_r) (and ys with component 1) of ONE window's events,
 * in the order they were handed to eincm_set_windows, under the Theta of the last evaluation: the 'warped_xs' / 'warped_ys' entries of
 * compute_loss_objectives (src/eincm/losses.py:58,90-91; per_pix_warp, src/eincm/event_warpers.py:28-37), read only by plotters.
 * warped_xs, warped_ys: (n_refs, n_events[window]) float64 each.  Same operations as the evaluation's warp (product rounded first). */
int eincm_get_warped_events(eincm_ctx* ctx, int window, double* warped_xs, double* warped_ys);

/* compute_weights_for_multi_reference (losses.py:39-46) */
int eincm_multi_ref_weights(int n_refs, double* w);

/* weight matrix of jax.image.scale_and_translate along one axis (theta_utils.py:25-35): A (n_out, n_in) */
int eincm_resample_matrix(int n_in, int n_out, int method, double* A);

int eincm_get_timings(eincm_ctx* ctx, eincm_timings* t);
/* EINCM_CF_TIMING_DOMINANT contexts: which of the two event kernels carry timing events from the next evaluation on (both by
 * default).  A timed launch costs ~6 us of an evaluation; a throughput measurement times only the kernel it reports. */
int eincm_set_timed_kernels(eincm_ctx* ctx, int splat, int gather);
/* ... and on every `period`-th evaluation only (1 = every evaluation): a timed launch costs ~6 us of a 240 us step; eincm_get_timings_total
 * reports how many evaluations its sums cover.  The counter restarts with the call. */
int eincm_set_timing_period(eincm_ctx* ctx, int period);

/* Host-side wall time (microseconds, summed since the last reset) the calling thread spent in the phases of the evaluations of
 * this context, and their number: [EINCM_HP_BEGIN] argument checks, theta staging and the launches of the forward half,
 * [EINCM_HP_LAUNCH] the launches of the second half, [EINCM_HP_WAIT] waiting for the stream, [EINCM_HP_COLLECT] handing the
 * results over (incl. the host-side scalar assembly of 2-DoF evaluations).  A diagnostic: what an evaluation costs besides its kernels. */
#define EINCM_HP_BEGIN 0
#define EINCM_HP_LAUNCH 1
#define EINCM_HP_WAIT 2
#define EINCM_HP_COLLECT 3
#define EINCM_N_HOST_PHASES 4
int eincm_get_host_profile(eincm_ctx* ctx, double* us /* EINCM_N_HOST_PHASES */, int64_t* n_evals, int reset);
/* Diagnostic: how the staged batch is cut into work and how the last evaluation was launched (no counterpart in the reference; the
 * numbers behind DESIGN.md section 4.2 "where those rules hold").  Staging decides the segment lengths and whether the batch is in
 * the regime of the bank-aligned LDS pitch; every evaluation decides the LDS window capacities from max|theta| and the time span
 * that all but 3 % of the events' segments stay within.  Entries of the last evaluation are 0 before the first one. */
#define EINCM_LP_SEG_GATHER 0        /* events per segment: the theta-grid gather's list */
#define EINCM_LP_SEG_SPLAT 1         /* ... k_splat's list */
#define EINCM_LP_SEG_GATHER_2DOF 2   /* ... the 2-DoF gather's list */
#define EINCM_LP_SEG_SPLAT_SHORT 3   /* ... k_splat's short list for very large 2-DoF theta (0: none was built) */
#define EINCM_LP_PITCH_POLICY 4      /* 0: LDS windows at pitch = width; 1: k_splat's at the bank-aligned pitch where that costs no capacity class; 2: the 2-DoF gather's too */
#define EINCM_LP_SPAN_SPLAT 5        /* fraction of a window's duration the capacity of k_splat's windows is sized for */
#define EINCM_LP_SPAN_GATHER 6
#define EINCM_LP_SPAN_GATHER_2DOF 7
#define EINCM_LP_CAP_SPLAT 8         /* last evaluation: LDS window capacity in 32-bit words, k_splat */
#define EINCM_LP_CAP_GATHER 9        /* ... theta-grid gather */
#define EINCM_LP_CAP_GATHER_2DOF 10  /* ... 2-DoF gather */
#define EINCM_LP_PITCH_ALIGNED 11    /* last evaluation: bit 0 k_splat, bit 1 the 2-DoF gather took the aligned pitch */
#define EINCM_LP_SPLAT_SHORT 12      /* last evaluation: k_splat walked its short list */
#define EINCM_N_LAUNCH_POLICY 13
int eincm_get_launch_policy(eincm_ctx* ctx, double* out /* EINCM_N_LAUNCH_POLICY */);
/* sums of the per-evaluation timings since the last reset, and how many evaluations they cover (a bench reads them once after
 * its timed loop instead of calling eincm_get_timings inside it) */
int eincm_get_timings_total(eincm_ctx* ctx, eincm_timings* sum, int64_t* n_evals, int reset);

/* The evaluation with theta AND gradient resident in HBM, for a caller whose optimiser lives on the GPU: only the scalars cross PCIe
 * (a dense theta at 480x640 otherwise moves 2 x 4.9 MB per evaluation: ~380 of its ~590 us).  The reference's optimiser is on the host
 * (src/eincm/solver.py:165-173), so this has no counterpart there.
 *   theta_dev      (n_windows, h, w, 2) float64 in the memory of the context's device, complete when the call is made (the engine runs
 *                  on its own stream: synchronise the stream that produced theta first)
 *   theta_abs_max  an upper bound of |theta| (px per unit time) if the caller has one, < 0 otherwise: it only selects the capacity of
 *                  the LDS windows (any value is correct - results agree to a fixed-point quantum; a bound far too small or unknown costs speed)
 *   value          (n_windows) on the HOST; aux optional, on the host
 *   grad_dev       (n_windows, h, w, 2) float64 on the device, or NULL for a forward-only evaluation; complete on return
 * EINCM_ERR_NONFINITE reports a non-finite value or theta (the device gradient is not scanned). */
int eincm_loss_grad_device(eincm_ctx* ctx, const double* theta_dev, int h, int w, const eincm_params* p, double theta_abs_max,
                           double* value, double* grad_dev, eincm_aux* aux);

/* Event-sharded mode over a GPU collective (RCCL): keep the results of the finishing half in HBM so that the caller can all-reduce
 * the gradient there, instead of bouncing it through the host.  eincm_set_device_results(ctx, 1) once; then per evaluation
 *   eincm_forward_iwe -> [all-reduce the IWE accumulator] -> eincm_finish_launch (returns with the gradient complete in HBM)
 *   -> [all-reduce eincm_grad_device_ptr's n_doubles doubles] -> eincm_finish_collect (copies value / gradient / aux to the host).
 * 2-DoF evaluations run their scalar assembly on the GPU again in this mode (k_final), so that the gradient exists in HBM. */
int eincm_set_device_results(eincm_ctx* ctx, int on);
int eincm_finish_launch(eincm_ctx* ctx);
int eincm_grad_device_ptr(eincm_ctx* ctx, void** dptr, int64_t* n_doubles);
int eincm_finish_collect(eincm_ctx* ctx, double* value, double* grad, eincm_aux* aux);

/* Event-sharded evaluation: the events of the SAME windows are split over several contexts (one per GPU);
 * edges and edge_ts are replicated.  The IWE is additive over events (src/utils/event_utils.py:59 is a pure sum), so
 *   every shard:  eincm_forward_iwe(theta)            k_theta + k_splat on its own events; returns stream-synchronised
 *   caller:       all-reduce(sum) of the IWE stacks   (RCCL on eincm_iwe_device_ptr; (n_windows, n_refs, H, W) int64: the engine
 *                                                      accumulates pixel * 2^30 as 64-bit integers, so the sum over shards is exact
 *                                                      and independent of the reduction order)
 *   every shard:  eincm_finish_loss_grad              statistics ... gradient on the summed stack
 * gives the same value on every shard and gradients that SUM to the full gradient (pass EINCM_PF_NO_TV_GRAD on all but
 * one shard).  Staging: eincm_set_windows_ex(..., EINCM_SW_DEFER_CONSTANTS), all-reduce(max) of the event masks
 * (eincm_mask_device_ptr, uint8), eincm_forward_iwe(theta = NULL), all-reduce(sum) of the IWE stacks, eincm_finish_constants. */
int eincm_forward_iwe(eincm_ctx* ctx, const double* theta, int h, int w, const eincm_params* p, int want_grad);
int eincm_finish_loss_grad(eincm_ctx* ctx, double* value, double* grad, eincm_aux* aux);
int eincm_finish_constants(eincm_ctx* ctx);
int eincm_iwe_device_ptr(eincm_ctx* ctx, void** dptr, int64_t* n_words);   /* n_words 64-bit integers */
int eincm_mask_device_ptr(eincm_ctx* ctx, void** dptr, int64_t* n_bytes);

/* ---- the step that produces `edges`, and the tiled objectives ------------------------------------ */
#define EINCM_EDT_EXPONENTIAL 0   /* 1 - exp(-d / alpha)        img_utils.py:232, :382 */
#define EINCM_EDT_LINEAR 1        /* d                          img_utils.py:376       */
#define EINCM_EDT_LINEAR_BOUND 2  /* min(d, d_sat)              img_utils.py:378       */
#define EINCM_EDT_LOGARITHMIC 3   /* log(d + 1)                 img_utils.py:380       */

/* Inverse (exponential) distance transform of binary edge images: replaces eincm_inv_exp_dist_transform
 * (src/utils/img_utils.py:229-233, scipy.ndimage.distance_transform_edt) and RTEF_IEDT.compute_edge_iedt
 * (img_utils.py:236-410, Meijster transform) - both are  1 - minmax(f(d))  of the exact Euclidean distance d to the
 * nearest edge pixel.  edge_img (n, H, W) uint8, non-zero = edge, H x W = the context's sensor; out (n, H, W) double;
 * sqdist (n, H, W) int32 or NULL receives d^2 (integer work: equals the reference's bit for bit).
 * An image without any edge pixel has no distance transform: EINCM_ERR_ARG. */
int eincm_inv_dist_transform(eincm_ctx* ctx, const uint8_t* edge_img, int n, int formulation, double alpha, double d_sat,
                             double* out, int32_t* sqdist);

/* smoothen_edges (img_utils.py:210-220): cv.GaussianBlur of a float64 image with the kernel size derived from sigma
 * (round(8 sigma + 1) | 1 taps), separable, BORDER_REFLECT_101.  src, dst (n, H, W) double; may alias. */
int eincm_gaussian_blur(eincm_ctx* ctx, const double* src, int n, double sigma, double* dst);

/* extract_tiles (img_utils.py:105-120) + compute_adaptive_* (contrast_objectives.py:42-87, correlation_objectives.py:105-130)
 * and their pairwise siblings (correlation_objectives.py:28-102), on the images of the LAST evaluation, per (window, ref):
 * contrast-type objectives on the raw IWE (as losses.py:70 does), pair-type ones on (edges, min-max-normalised IWE)
 * (as losses.py:65 does).  Whole tiles only; the ragged remainder is ignored, as in the reference. */
typedef struct eincm_tiled_out {
    int32_t n_refs, n_tiles;
    double adaptive_mean_gradient_magnitude[EINCM_MAX_REFS];
    double adaptive_variance[EINCM_MAX_REFS];
    double adaptive_mean_squared_error[EINCM_MAX_REFS];
    double sum_squared_error[EINCM_MAX_REFS];
    double mean_hadamard_product[EINCM_MAX_REFS];
    double sum_hadamard_product[EINCM_MAX_REFS];
    double joint_contrast[EINCM_MAX_REFS];
} eincm_tiled_out;
int eincm_tiled_objectives(eincm_ctx* ctx, int tile_h, int tile_w, eincm_tiled_out* out /* n_windows */);

#ifdef __cplusplus
}
#endif
#endif /* EINCM_H */
