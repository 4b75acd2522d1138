/* eincm_ref.c — C / OpenMP port of the oracle's alpha/beta path (contrast = mean squared Scharr magnitude, correlation
 * = normalised-IWE MSE), float64 throughout.  TEST INFRASTRUCTURE ONLY (see oracle/eincm_oracle.py header): used as the
 * multi-core CPU baseline of bench.py (`cpu_baseline.kind = "port"`) and cross-checked against the numpy oracle in
 * tests/test_oracle_c_port.py.  PARITY UNPINNED like the rest of oracle/ (the reference cannot run here).
 *
 * Follows, like oracle/eincm_oracle.py:loss_and_grad with gamma = delta = 0:
 *   src/eincm/event_warpers.py:28-37   warp            src/utils/event_utils.py:31-61   3x3 Gaussian-pdf splat (JAX index rule)
 *   src/utils/img_utils.py:24-25       normalise       src/utils/img_utils.py:414-425   Scharr 'same' convolution
 *   src/eincm/losses.py:39-46,54-72,176-193            weights, objectives, final loss
 * and the hand-derived reverse pass of SURVEY Appendix A.2.  Theta is the FULL-RESOLUTION field (H,W,2); the caller
 * resamples theta and projects the gradient (oracle/eincm_oracle.py does that part in numpy).
 *
 * Build: make -C oracle   ->  oracle/libeincm_ref.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSN 2.220446049250313e-16
#define LOG_2PI 1.8378770664093453

static inline long wrap_drop(long p, long n) { if (p < 0) p += n; return (p >= 0 && p < n) ? p : -1; }

static inline long round_i(double v) {
    double r = nearbyint(v);                       /* FE_TONEAREST: half to even, like jnp.round */
    if (r > 2147483647.0) r = 2147483647.0;
    if (r < -2147483648.0) r = -2147483648.0;
    return (long)r;
}

/* frame += splat of events [lo,hi) */
static void splat_range(const double* wx, const double* wy, int64_t lo, int64_t hi, int H, int W, double* frame) {
    for (int64_t e = lo; e < hi; ++e) {
        const long rx = round_i(wx[e]), ry = round_i(wy[e]);
        for (int dx = -1; dx <= 1; ++dx) {
            const long cx = wrap_drop(rx + dx, W);
            if (cx < 0) continue;
            const double qx = (double)(rx + dx) - wx[e];
            for (int dy = -1; dy <= 1; ++dy) {
                const long cy = wrap_drop(ry + dy, H);
                if (cy < 0) continue;
                const double qy = (double)(ry + dy) - wy[e];
                frame[cy * W + cx] += exp(-0.5 * (qx * qx + qy * qy) - LOG_2PI);
            }
        }
    }
}

static void splat(const double* wx, const double* wy, int64_t N, int H, int W, double* frame, double* scratch, int nt) {
    const size_t HW = (size_t)H * W;
    memset(scratch, 0, sizeof(double) * HW * nt);
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t lo = N * t / nt, hi = N * (t + 1) / nt;
        splat_range(wx, wy, lo, hi, H, W, scratch + HW * t);
    }
#pragma omp parallel for num_threads(nt)
    for (size_t p = 0; p < HW; ++p) {
        double s = 0.0;
        for (int t = 0; t < nt; ++t) s += scratch[HW * t + p];
        frame[p] = s;
    }
}

static inline double at(const double* a, int H, int W, int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? a[(size_t)y * W + x] : 0.0; }

/* difference-first Scharr, zero padded (oracle scharr_grads) */
static void scharr(const double* a, int H, int W, double* gx, double* gy, int nt) {
#pragma omp parallel for num_threads(nt)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            gx[(size_t)y * W + x] = 3.0 * (at(a, H, W, y + 1, x + 1) - at(a, H, W, y + 1, x - 1)) + 10.0 * (at(a, H, W, y, x + 1) - at(a, H, W, y, x - 1))
                                  + 3.0 * (at(a, H, W, y - 1, x + 1) - at(a, H, W, y - 1, x - 1));
            gy[(size_t)y * W + x] = 3.0 * (at(a, H, W, y + 1, x + 1) - at(a, H, W, y - 1, x + 1)) + 10.0 * (at(a, H, W, y + 1, x) - at(a, H, W, y - 1, x))
                                  + 3.0 * (at(a, H, W, y + 1, x - 1) - at(a, H, W, y - 1, x - 1));
        }
}

static void minmax(const double* a, size_t n, double* mn, double* mx) {
    double lo = a[0], hi = a[0];
    for (size_t i = 1; i < n; ++i) { if (a[i] < lo) lo = a[i]; if (a[i] > hi) hi = a[i]; }
    *mn = lo; *mx = hi;
}

/* returns 0 on success.  g_Theta (H,W,2) may be NULL (forward only).  iwes_out (R,H,W) and G_out (R,H,W; dL/dIWE, only
 * filled when g_Theta is given) may be NULL: they expose the image of warped events and its cotangent so that the
 * full-size GPU parity tests can compare images, not only scalars. */
int eincm_ref_loss_grad_ex(int H, int W, int64_t N, int R, const int16_t* xs, const int16_t* ys, const double* ts,
                           const double* edges, const double* edge_ts, const double* Theta, double alpha, double beta,
                           double* value, double* g_Theta, double* iwes_out, double* G_out, int nthreads) {
    const size_t HW = (size_t)H * W;
    int nt = nthreads > 0 ? nthreads : 1;
#ifndef _OPENMP
    nt = 1;
#endif
    double* wx = malloc(sizeof(double) * (N > 0 ? N : 1));
    double* wy = malloc(sizeof(double) * (N > 0 ? N : 1));
    double* I0 = malloc(sizeof(double) * HW), *I = malloc(sizeof(double) * HW), *G = malloc(sizeof(double) * HW);
    double* gx = malloc(sizeof(double) * HW), *gy = malloc(sizeof(double) * HW), *n0 = malloc(sizeof(double) * HW);
    double* scratch = malloc(sizeof(double) * HW * nt * 2);
    double* w = malloc(sizeof(double) * R);
    if (!wx || !wy || !I0 || !I || !G || !gx || !gy || !n0 || !scratch || !w) return -1;

    double ws = 0.0;                                /* losses.py:39-46 */
    for (int r = 0; r < R; ++r) { const double x = (R > 1) ? -1.5 + 3.0 * r / (R - 1) : -1.5; w[r] = exp(-0.5 * x * x) / sqrt(2.0 * M_PI); ws += w[r]; }
    for (int r = 0; r < R; ++r) w[r] /= ws;

#pragma omp parallel for num_threads(nt)
    for (int64_t e = 0; e < N; ++e) { wx[e] = (double)xs[e]; wy[e] = (double)ys[e]; }
    splat(wx, wy, N, H, W, I0, scratch, nt);
    double m0, M0;
    minmax(I0, HW, &m0, &M0);
    const double D0 = M0 - m0 + EPSN;
    for (size_t p = 0; p < HW; ++p) n0[p] = (I0[p] - m0) / D0;
    scharr(I0, H, W, gx, gy, nt);
    double c0 = 0.0;
#pragma omp parallel for reduction(+:c0) num_threads(nt)
    for (size_t p = 0; p < HW; ++p) c0 += gx[p] * gx[p] + gy[p] * gy[p];
    c0 /= (double)HW;

    if (g_Theta) memset(g_Theta, 0, sizeof(double) * HW * 2);
    double sum_con = 0.0, sum_corr = 0.0;
    for (int r = 0; r < R; ++r) {
        const double tau = edge_ts[r];
        const double* E = edges + (size_t)r * HW;
#pragma omp parallel for num_threads(nt)
        for (int64_t e = 0; e < N; ++e) {
            const size_t o = ((size_t)ys[e] * W + xs[e]) * 2;
            const double dt = ts[e] - tau;
            wx[e] = (double)xs[e] - Theta[o] * dt * 1.0;
            wy[e] = (double)ys[e] - Theta[o + 1] * dt * 1.0;
        }
        splat(wx, wy, N, H, W, I, scratch, nt);
        if (iwes_out) memcpy(iwes_out + (size_t)r * HW, I, sizeof(double) * HW);
        double m, M;
        minmax(I, HW, &m, &M);
        const double D = M - m + EPSN;
        double mse = 0.0, mse0 = 0.0, cnt_m = 0.0, cnt_M = 0.0;
#pragma omp parallel for reduction(+:mse,mse0,cnt_m,cnt_M) num_threads(nt)
        for (size_t p = 0; p < HW; ++p) {
            const double n = (I[p] - m) / D;
            mse += (E[p] - n) * (E[p] - n);
            mse0 += (E[p] - n0[p]) * (E[p] - n0[p]);
            cnt_m += (I[p] == m); cnt_M += (I[p] == M);
        }
        mse /= (double)HW; mse0 /= (double)HW;
        scharr(I, H, W, gx, gy, nt);
        double c = 0.0;
#pragma omp parallel for reduction(+:c) num_threads(nt)
        for (size_t p = 0; p < HW; ++p) c += gx[p] * gx[p] + gy[p] * gy[p];
        c /= (double)HW;
        const double corr = -mse, zc = -mse0;
        sum_con += w[r] * c / (c0 + EPSN);
        sum_corr += w[r] * corr / (zc + EPSN);
        if (!g_Theta) continue;

        const double a_r = -alpha * w[r] / (R * (c0 + EPSN)), b_r = -beta * w[r] / (R * (zc + EPSN));
        double sGn_n1 = 0.0, sGn_n = 0.0;             /* sum Gn*(n-1), sum Gn*n */
#pragma omp parallel for reduction(+:sGn_n1,sGn_n) num_threads(nt)
        for (size_t p = 0; p < HW; ++p) {
            const double n = (I[p] - m) / D;
            const double Gn = b_r * (2.0 / (double)HW) * (E[p] - n);
            sGn_n1 += Gn * (n - 1.0); sGn_n += Gn * n;
        }
        const double dm = sGn_n1 / D, dM = -sGn_n / D;
#pragma omp parallel for num_threads(nt)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t p = (size_t)y * W + x;
                /* adjoint of the 'same' convolution = 'same' convolution with the flipped kernel = -conv for Scharr */
                const double ax = 3.0 * (at(gx, H, W, y + 1, x + 1) - at(gx, H, W, y + 1, x - 1)) + 10.0 * (at(gx, H, W, y, x + 1) - at(gx, H, W, y, x - 1))
                                + 3.0 * (at(gx, H, W, y - 1, x + 1) - at(gx, H, W, y - 1, x - 1));
                const double ay = 3.0 * (at(gy, H, W, y + 1, x + 1) - at(gy, H, W, y - 1, x + 1)) + 10.0 * (at(gy, H, W, y + 1, x) - at(gy, H, W, y - 1, x))
                                + 3.0 * (at(gy, H, W, y + 1, x - 1) - at(gy, H, W, y - 1, x - 1));
                const double n = (I[p] - m) / D;
                double g = a_r * (2.0 / (double)HW) * (-(ax + ay)) + b_r * (2.0 / (double)HW) * (E[p] - n) / D;
                if (I[p] == m) g += dm / cnt_m;
                if (I[p] == M) g += dM / cnt_M;
                G[p] = g;
            }
        if (G_out) memcpy(G_out + (size_t)r * HW, G, sizeof(double) * HW);
        /* gather + accumulate per source pixel: private accumulators per thread, then reduce */
        memset(scratch, 0, sizeof(double) * HW * 2 * nt);
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            double* acc = scratch + HW * 2 * t;
            const int64_t lo = N * t / nt, hi = N * (t + 1) / nt;
            for (int64_t e = lo; e < hi; ++e) {
                const long rx = round_i(wx[e]), ry = round_i(wy[e]);
                double gwx = 0.0, gwy = 0.0;
                for (int dx = -1; dx <= 1; ++dx) {
                    const long cx = wrap_drop(rx + dx, W);
                    if (cx < 0) continue;
                    const double qx = (double)(rx + dx) - wx[e];
                    for (int dy = -1; dy <= 1; ++dy) {
                        const long cy = wrap_drop(ry + dy, H);
                        if (cy < 0) continue;
                        const double qy = (double)(ry + dy) - wy[e];
                        const double k = exp(-0.5 * (qx * qx + qy * qy) - LOG_2PI);
                        const double g = G[cy * W + cx];
                        gwx += g * k * qx; gwy += g * k * qy;
                    }
                }
                const double dt = ts[e] - tau;
                const size_t o = ((size_t)ys[e] * W + xs[e]) * 2;
                acc[o] += -dt * gwx; acc[o + 1] += -dt * gwy;
            }
        }
#pragma omp parallel for num_threads(nt)
        for (size_t p = 0; p < HW * 2; ++p) {
            double s = 0.0;
            for (int t = 0; t < nt; ++t) s += scratch[HW * 2 * t + p];
            g_Theta[p] += s;
        }
    }
    *value = alpha * (-(sum_con / R)) + beta * (-(sum_corr / R));
    free(wx); free(wy); free(I0); free(I); free(G); free(gx); free(gy); free(n0); free(scratch); free(w);
    return 0;
}

int eincm_ref_loss_grad(int H, int W, int64_t N, int R, const int16_t* xs, const int16_t* ys, const double* ts,
                        const double* edges, const double* edge_ts, const double* Theta, double alpha, double beta,
                        double* value, double* g_Theta, int nthreads) {
    return eincm_ref_loss_grad_ex(H, W, N, R, xs, ys, ts, edges, edge_ts, Theta, alpha, beta, value, g_Theta, NULL, NULL, nthreads);
}

int eincm_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
