"""CPU oracle for the EINCM loss(theta, events, edges) -> (value, grad) path.  numpy, float64.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker / reported CPU baseline.  The product path (``edge-informed-contrast-maximization_amd``)
never imports this module and fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference (robotic-vision-lab/Edge-Informed-Contrast-Maximization) is pure
Python on JAX; jax/jaxlib/jaxopt are not installed here (``import jax`` -> ModuleNotFoundError) and
cannot be fetched, the reference ships no tests, golden vectors or fixtures, and pins no versions.
This file is therefore an op-for-op restatement of the reference *source text*, with the upstream
JAX semantics (scatter index normalisation, ``scale_and_translate``, min/max tie gradients) taken
from knowledge of JAX and listed below.  It is pinned only by (a) the source-derived known answers
in ``tests/test_oracle_known_answers.py`` (tap constants, multi-reference weights, zero-theta and
integer-shift identities), (b) an independent torch-float64 autograd restatement
(``oracle/eincm_torch.py``) and (c) central finite differences.

Reference lines followed (relative to /root/reference/src):
  eincm/losses.py:39-46      compute_weights_for_multi_reference
  eincm/losses.py:49-105     compute_loss_objectives
  eincm/losses.py:108-205    loss_func
  eincm/losses.py:269-274    handover_loss_func
  eincm/event_warpers.py:28-37   per_pix_warp
  utils/event_utils.py:31-61     events_to_pdf_frame
  utils/event_utils.py:64-76     make_event_mask
  utils/img_utils.py:24-25       normalize_to_unit_range
  utils/img_utils.py:414-425     sobel_scharr_optimized_image_grads
  utils/theta_utils.py:25-35     scale_theta_to_sensor_size
  utils/theta_utils.py:59-73     per_pix_theta_to_flow
  eincm/objectives/contrast_objectives.py:22-25,38   mean grad-mag^2, variance
  eincm/objectives/correlation_objectives.py:25-26   MSE
  eincm/objectives/event_collapse_objectives.py:10-20  iwe_divergence
  eincm/regularizers.py:16-38,43-58   TV, theta divergence
  eincm/contrast_metrics.py:17        FWL

JAX semantics mimicked (not executable here):
  S1  ``frame.at[rs, cs].add(v, mode='drop')``: negative indices are normalised first
      (i in [-n,-1] -> i+n), then out-of-range updates are dropped; duplicates accumulate.
  S3  ``jnp.round`` is round-half-to-even with zero derivative.
  S4  reverse-mode of ``min``/``max`` shares the cotangent equally among tied extremal elements.
  S5  ``abs'(0) = sign(0) = 0``.
  S6  ``jax.scipy.signal.convolve(a, k, 'same')`` is a true convolution (kernel flipped), zero padded.
  S7  ``jax.image.scale_and_translate(..., antialias=True)`` weight matrices (see resample_matrix).
  S8  ``multivariate_normal.pdf(q, 0, I2) = exp(-0.5*|q|^2 - log(2*pi))``.
  S9  ``jnp.var`` is the population variance.
"""
import math
import sys

import numpy as np

EPSN = sys.float_info.epsilon  # losses.py:24, img_utils.py:18, regularizers.py:11

SCHARR_GX = np.array([[3.0, 0.0, -3.0], [10.0, 0.0, -10.0], [3.0, 0.0, -3.0]])  # img_utils.py:417
SCHARR_GY = np.array([[3.0, 10.0, 3.0], [0.0, 0.0, 0.0], [-3.0, -10.0, -3.0]])  # img_utils.py:418
DIV_KERN = np.array([[1 / 12, 1 / 6, 1 / 12], [1 / 6, 0.0, 1 / 6], [1 / 12, 1 / 6, 1 / 12]])  # event_collapse_objectives.py:14

CONTRAST_GRAD_MAG = 0   # losses.py:70 (the live contrast objective)
CONTRAST_VARIANCE = 1   # contrast_objectives.py:38 (BASELINE config C1 "variance-only")


# --------------------------------------------------------------------------------------
# small dense ops
# --------------------------------------------------------------------------------------
def conv2_same(img, kern):
    """True 2-D convolution, zero padded, output the size of ``img`` (S6; img_utils.py:420-421)."""
    H, W = img.shape
    kh, kw = kern.shape
    ph, pw = kh // 2, kw // 2
    pad = np.zeros((H + 2 * ph, W + 2 * pw), dtype=np.float64)
    pad[ph:ph + H, pw:pw + W] = img
    out = np.zeros((H, W), dtype=np.float64)
    # out[y,x] = sum_{a,b} kern[a,b] * img[y-(a-ph), x-(b-pw)]
    for a in range(kh):
        for b in range(kw):
            if kern[a, b] == 0.0:
                continue
            ys = ph - (a - ph)
            xs = pw - (b - pw)
            out += kern[a, b] * pad[ys:ys + H, xs:xs + W]
    return out


def conv2_same_adjoint(cot, kern):
    """Adjoint of ``img -> conv2_same(img, kern)``: a 'same' convolution with the 180-degree flipped kernel."""
    return conv2_same(cot, kern[::-1, ::-1])


def scharr_grads(img):
    """img_utils.py:414-425 -> (I_x, I_y): true 'same' convolution with SCHARR_GX / SCHARR_GY, zero padded.

    Written difference-first,
        I_x[y,x] = 3(p[y+1,x+1]-p[y+1,x-1]) + 10(p[y,x+1]-p[y,x-1]) + 3(p[y-1,x+1]-p[y-1,x-1])
        I_y[y,x] = 3(p[y+1,x+1]-p[y-1,x+1]) + 10(p[y+1,x]-p[y-1,x]) + 3(p[y+1,x-1]-p[y-1,x-1])
    which equals conv2_same(img, SCHARR_G*) up to rounding.  The order matters in exactly one place:
    regularizers.py:26-29 counts pixels whose flow gradient is *non-zero* and :31-36 takes abs (sign in the
    gradient), so on locally constant flow (bilinear up-sampling clamps at the borders) a tap-by-tap sum can
    leave +-1e-16 residues that flip the count and the sign.  XLA's own summation order is unknowable here
    (parity unpinned); differences of equal values are exactly 0 under any FMA contraction, so this form gives
    the HIP kernels a well-defined target.
    """
    H, W = img.shape
    p = np.zeros((H + 2, W + 2), dtype=np.float64)
    p[1:H + 1, 1:W + 1] = img
    c = slice(1, W + 1); l = slice(0, W); r = slice(2, W + 2)
    m = slice(1, H + 1); u = slice(0, H); d = slice(2, H + 2)     # u = y-1, d = y+1
    gx = 3.0 * (p[d, r] - p[d, l]) + 10.0 * (p[m, r] - p[m, l]) + 3.0 * (p[u, r] - p[u, l])
    gy = 3.0 * (p[d, r] - p[u, r]) + 10.0 * (p[d, c] - p[u, c]) + 3.0 * (p[d, l] - p[u, l])
    return gx, gy


def normalize_to_unit_range(arr):
    """img_utils.py:24-25."""
    return (arr - arr.min()) / (arr.max() - arr.min() + EPSN)


def compute_weights_for_multi_reference(n_refs, n_sigma=1.5):
    """losses.py:39-46 (scipy.stats.norm.pdf restated: exp(-x^2/2)/sqrt(2*pi))."""
    x = np.linspace(-n_sigma, n_sigma, n_refs)
    w = np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
    return w / w.sum()


# --------------------------------------------------------------------------------------
# theta resampling (S7)
# --------------------------------------------------------------------------------------
def _kernel_triangle(x):
    return np.maximum(0.0, 1.0 - np.abs(x))


def _kernel_lanczos(radius):
    def k(x):
        y = radius * np.sin(np.pi * x) * np.sin(np.pi * x / radius)
        with np.errstate(divide='ignore', invalid='ignore'):
            out = np.where(x > 1e-3, y / (np.pi ** 2 * x ** 2), 1.0)
        return np.where(x > radius, 0.0, out)
    return k


def _kernel_cubic(x):
    # Keys cubic, A = -0.5 (jax.image 'cubic'/'bicubic')
    out = ((1.5 * x - 2.5) * x) * x + 1.0
    out = np.where(x >= 1.0, ((-0.5 * x + 2.5) * x - 4.0) * x + 2.0, out)
    return np.where(x >= 2.0, 0.0, out)


_KERNELS = {
    'linear': _kernel_triangle, 'bilinear': _kernel_triangle, 'triangle': _kernel_triangle,
    'trilinear': _kernel_triangle,
    'lanczos3': _kernel_lanczos(3.0), 'lanczos5': _kernel_lanczos(5.0),
    'cubic': _kernel_cubic, 'bicubic': _kernel_cubic, 'tricubic': _kernel_cubic,
}


def resample_matrix(n_in, n_out, scale, method='bilinear', translation=0.0, antialias=True):
    """Weight matrix ``A`` of shape (n_out, n_in) with ``out = A @ in`` along one axis (S7).

    Restates jax.image.scale_and_translate's per-axis weight computation as used at
    theta_utils.py:25-35 (scale = n_out/n_in, translation 0, antialias True).
    """
    kernel = _KERNELS[method]
    inv_scale = 1.0 / scale
    kernel_scale = max(inv_scale, 1.0) if antialias else 1.0
    sample_f = (np.arange(n_out, dtype=np.float64) + 0.5) * inv_scale - translation * inv_scale - 0.5
    x = np.abs(sample_f[None, :] - np.arange(n_in, dtype=np.float64)[:, None]) / kernel_scale
    weights = kernel(x)                                    # (n_in, n_out)
    total = weights.sum(axis=0, keepdims=True)
    weights = np.where(np.abs(total) > 1000.0 * float(np.finfo(np.float32).eps),
                       weights / np.where(total != 0, total, 1.0), 0.0)
    inside = np.logical_and(sample_f >= -0.5, sample_f <= n_in - 0.5)[None, :]
    weights = np.where(inside, weights, 0.0)
    return np.ascontiguousarray(weights.T)                 # (n_out, n_in)


def scale_theta_to_sensor_size(theta, sensor_size, method='bilinear'):
    """theta_utils.py:10-37: (h,w,2) -> (H,W,2).  The channel axis has scale 1 -> identity for the triangle kernel."""
    theta = np.asarray(theta, dtype=np.float64)
    h, w, _ = theta.shape
    H, W = sensor_size
    A_H = resample_matrix(h, H, H / h, method)
    A_W = resample_matrix(w, W, W / w, method)
    A_C = resample_matrix(2, 2, 1.0, method)
    return np.einsum('yi,xj,dc,ijc->yxd', A_H, A_W, A_C, theta, optimize=True)


def scale_theta_adjoint(g_Theta, theta_shape, method='bilinear'):
    """Adjoint of scale_theta_to_sensor_size: (H,W,2) cotangent -> (h,w,2)."""
    h, w, _ = theta_shape
    H, W, _ = g_Theta.shape
    A_H = resample_matrix(h, H, H / h, method)
    A_W = resample_matrix(w, W, W / w, method)
    A_C = resample_matrix(2, 2, 1.0, method)
    return np.einsum('yi,xj,dc,yxd->ijc', A_H, A_W, A_C, g_Theta, optimize=True)


# --------------------------------------------------------------------------------------
# events
# --------------------------------------------------------------------------------------
def per_pix_warp(Theta, xs, ys, ts, t_ref, delta_time=1.0):
    """event_warpers.py:28-37."""
    xi = np.asarray(xs).astype(np.int64)
    yi = np.asarray(ys).astype(np.int64)
    dts = np.asarray(ts, dtype=np.float64) - t_ref
    wx = xi - Theta[yi, xi, 0] * dts * delta_time
    wy = yi - Theta[yi, xi, 1] * dts * delta_time
    return wx, wy


def _tap_index(r, d, n):
    """S1: index normalisation + drop.  Returns (wrapped index, valid mask)."""
    p = r + d
    p = np.where(p < 0, p + n, p)
    valid = (p >= 0) & (p < n)
    return p, valid


def _round_i(v):
    # jnp.round(...).astype(int32); values far outside int32 are clipped (undefined upstream)
    return np.clip(np.rint(v), -2 ** 31, 2 ** 31 - 1).astype(np.int64)


def events_to_pdf_frame(wx, wy, sensor_size):
    """event_utils.py:13-61: 3x3 un-normalised Gaussian-pdf splat centred on the rounded coordinate."""
    H, W = sensor_size
    wx = np.asarray(wx, dtype=np.float64)
    wy = np.asarray(wy, dtype=np.float64)
    rx = _round_i(wx)
    ry = _round_i(wy)
    frame = np.zeros(H * W, dtype=np.float64)
    for dx in (-1, 0, 1):          # event_utils.py:42 (outer dx)
        for dy in (-1, 0, 1):      # event_utils.py:43 (inner dy)
            qx = (rx + dx) - wx
            qy = (ry + dy) - wy
            k = np.exp(-0.5 * (qx * qx + qy * qy) - math.log(2.0 * math.pi))   # S8
            cs, vx = _tap_index(rx, dx, W)
            rs, vy = _tap_index(ry, dy, H)
            v = vx & vy
            frame += np.bincount(rs[v] * W + cs[v], weights=k[v], minlength=H * W)
    return frame.reshape(H, W)


def rounded_count_image(wx, wy, sensor_size):
    """Integer skeleton of events_to_pdf_frame: histogram of the rounded coordinates (event_utils.py:32-33) with the
    scatter index rule of :59 (S1).  int64 (H, W)."""
    H, W = sensor_size
    cs, vx = _tap_index(_round_i(np.asarray(wx, dtype=np.float64)), 0, W)
    rs, vy = _tap_index(_round_i(np.asarray(wy, dtype=np.float64)), 0, H)
    v = vx & vy
    return np.bincount(rs[v] * W + cs[v], minlength=H * W).reshape(H, W)


def events_to_pdf_frame_adjoint(G, wx, wy):
    """Reverse-mode of events_to_pdf_frame w.r.t. (wx, wy) for an image cotangent G (round has zero derivative)."""
    H, W = G.shape
    rx = _round_i(wx)
    ry = _round_i(wy)
    gwx = np.zeros_like(wx)
    gwy = np.zeros_like(wy)
    Gf = G.reshape(-1)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            qx = (rx + dx) - wx
            qy = (ry + dy) - wy
            k = np.exp(-0.5 * (qx * qx + qy * qy) - math.log(2.0 * math.pi))
            cs, vx = _tap_index(rx, dx, W)
            rs, vy = _tap_index(ry, dy, H)
            v = vx & vy
            g = np.where(v, Gf[np.where(v, rs * W + cs, 0)], 0.0)
            # dk/dwx = k * qx  (q = p - w, dq/dw = -1, dk/dq = -k q)
            gwx += g * k * qx
            gwy += g * k * qy
    return gwx, gwy


def make_event_mask(xs, ys, sensor_size):
    """event_utils.py:64-76."""
    H, W = sensor_size
    m = np.zeros((H, W), dtype=bool)
    xi = np.asarray(xs).astype(np.int64)
    yi = np.asarray(ys).astype(np.int64)
    xi = np.where(xi < 0, xi + W, xi)
    yi = np.where(yi < 0, yi + H, yi)
    v = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
    m[yi[v], xi[v]] = True
    return m


# --------------------------------------------------------------------------------------
# objectives
# --------------------------------------------------------------------------------------
def compute_mean_gradient_magnitude(arr):
    """contrast_objectives.py:13-26 (no sqrt: mean of squared magnitude)."""
    gx, gy = scharr_grads(arr)
    return (gx * gx + gy * gy).mean()


def compute_variance(arr):
    """contrast_objectives.py:29-39."""
    return np.var(arr)


def compute_mean_squared_error(a, b):
    """correlation_objectives.py:12-27."""
    return ((a - b) ** 2).mean()


def iwe_divergence(iwe):
    """event_collapse_objectives.py:8-20."""
    gx, gy = scharr_grads(iwe)
    div = np.abs(conv2_same(gx, DIV_KERN) + conv2_same(gy, DIV_KERN))
    return div.mean()


def iwe_divergence_adjoint(iwe):
    """d iwe_divergence / d iwe  (S5: sign(0) = 0)."""
    H, W = iwe.shape
    gx, gy = scharr_grads(iwe)
    s = np.sign(conv2_same(gx, DIV_KERN) + conv2_same(gy, DIV_KERN)) / (H * W)
    t = conv2_same_adjoint(s, DIV_KERN)
    return conv2_same_adjoint(t, SCHARR_GX) + conv2_same_adjoint(t, SCHARR_GY)


def compute_fwl(iwe, zero_iwe):
    """contrast_metrics.py:6-18."""
    return np.var(iwe) / np.var(zero_iwe)


def per_pix_theta_to_flow(Theta, xs, ys):
    """theta_utils.py:40-73 (dt = 1): Theta at pixels holding >= 1 event, else 0."""
    mask = make_event_mask(xs, ys, Theta.shape[:2])
    return Theta * mask[:, :, None], mask


def per_pix_total_variation(Theta, xs, ys, return_grad=False):
    """regularizers.py:14-38.  With return_grad also d TV / d Theta (count is non-differentiable)."""
    flow, mask = per_pix_theta_to_flow(Theta, xs, ys)
    terms = []
    for c in (0, 1):
        gx, gy = scharr_grads(flow[:, :, c])
        terms.append((gx, gy))
    nz = np.zeros(mask.shape, dtype=bool)
    tot = 0.0
    for gx, gy in terms:
        nz |= (np.abs(gx) > 0) | (np.abs(gy) > 0)
    tot = np.sum((np.abs(terms[0][0]) * 0.25 + np.abs(terms[0][1]) * 0.25)
                 + (np.abs(terms[1][0]) * 0.25 + np.abs(terms[1][1]) * 0.25))
    denom = nz.sum() + EPSN
    tv = tot / denom
    if not return_grad:
        return tv
    g = np.zeros_like(Theta)
    for c, (gx, gy) in enumerate(terms):
        gf = (conv2_same_adjoint(np.sign(gx), SCHARR_GX) + conv2_same_adjoint(np.sign(gy), SCHARR_GY)) * (0.25 / denom)
        g[:, :, c] = gf * mask
    return tv, g


def per_pix_theta_divergence(Theta):
    """regularizers.py:41-58 (report only)."""
    acc = np.zeros(Theta.shape[:2])
    for c in (0, 1):
        gx, gy = scharr_grads(Theta[:, :, c])
        acc += conv2_same(gx, DIV_KERN) + conv2_same(gy, DIV_KERN)
    return np.abs(acc).mean()


def compute_loss_objectives(Theta, xs, ys, ts, edges, edge_ts, sensor_size):
    """losses.py:49-105.  ``Theta`` is the full-resolution (H, W, 2) field."""
    edges = np.asarray(edges, dtype=np.float64)
    edge_ts = np.asarray(edge_ts, dtype=np.float64)
    R = len(edge_ts)
    zero_iwe = events_to_pdf_frame(np.asarray(xs, dtype=np.float64), np.asarray(ys, dtype=np.float64), sensor_size)
    n0 = normalize_to_unit_range(zero_iwe)
    warped = [per_pix_warp(Theta, xs, ys, ts, edge_ts[r], 1.0) for r in range(R)]
    warped_xs = np.stack([w[0] for w in warped])
    warped_ys = np.stack([w[1] for w in warped])
    iwes = np.stack([events_to_pdf_frame(warped_xs[r], warped_ys[r], sensor_size) for r in range(R)])
    niwes = np.stack([normalize_to_unit_range(iwes[r]) for r in range(R)])
    corrs = np.array([compute_mean_squared_error(edges[r], niwes[r]) for r in range(R)]) * (-1)
    zero_corrs = np.array([compute_mean_squared_error(edges[r], n0) for r in range(R)]) * (-1)
    rel_corrs = corrs / (zero_corrs + EPSN)
    contrasts = np.array([compute_mean_gradient_magnitude(iwes[r]) for r in range(R)])
    zero_contrast = compute_mean_gradient_magnitude(zero_iwe)
    rel_contrasts = contrasts / (zero_contrast + EPSN)
    tv = per_pix_total_variation(Theta, xs, ys)
    theta_div = per_pix_theta_divergence(Theta)
    divs = np.array([iwe_divergence(niwes[r]) for r in range(R)])
    zero_div = iwe_divergence(n0)
    rel_divs = divs / (zero_div + EPSN)
    fwls = np.array([compute_fwl(iwes[r], zero_iwe) for r in range(R)])
    return {
        'warped_xs': warped_xs, 'warped_ys': warped_ys,
        'correlations': corrs, 'zero_correlations': zero_corrs, 'rel_correlations': rel_corrs,
        'contrasts': contrasts, 'zero_contrast': zero_contrast, 'rel_contrasts': rel_contrasts,
        'theta_total_variation': tv, 'theta_divergence': theta_div,
        'iwe_divergences': divs, 'zero_iwe_divergence': zero_div, 'rel_iwe_divergences': rel_divs,
        'flow_warp_losses': fwls, 'multi_ref_weights': compute_weights_for_multi_reference(R),
        # extras for kernel parity tests (not in the reference dict)
        '_iwes': iwes, '_zero_iwe': zero_iwe,
        '_variances': np.array([np.var(iwes[r]) for r in range(R)]), '_zero_variance': np.var(zero_iwe),
    }


def loss_func(theta, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta, cur_pyr_lvl, n_pyr_lvls,
              sensor_size, scale_to_sensor_size_method='bilinear', contrast_kind=CONTRAST_GRAD_MAG):
    """losses.py:108-205 -> (final_loss, aux).  contrast_kind=variance swaps losses.py:70-72 for compute_variance."""
    Theta = scale_theta_to_sensor_size(theta, sensor_size, scale_to_sensor_size_method)
    lo = compute_loss_objectives(Theta, xs, ys, ts, edges, edge_ts, sensor_size)
    w = lo['multi_ref_weights']
    if contrast_kind == CONTRAST_VARIANCE:
        contrasts, zero_contrast = lo['_variances'], lo['_zero_variance']
    else:
        contrasts, zero_contrast = lo['contrasts'], lo['zero_contrast']
    tv = lo['theta_total_variation'] if cur_pyr_lvl <= 0 else 0.0
    rel_corrs = (w * lo['correlations']) / (lo['zero_correlations'] + EPSN)
    rel_contrasts = (w * contrasts) / (zero_contrast + EPSN)
    rel_divs = (w * lo['iwe_divergences']) / (lo['zero_iwe_divergence'] + EPSN)
    mean_rel_corr = rel_corrs.mean()
    mean_rel_contrast = rel_contrasts.mean()
    mean_rel_div = rel_divs.mean()
    contrast_loss = mean_rel_contrast * (-1)
    correlation_loss = mean_rel_corr * (-1)
    final = (alpha * contrast_loss + beta * correlation_loss) + (gamma * tv + delta * mean_rel_div)
    aux = {'final_loss': final, 'scaled_theta': Theta, 'mean_rel_corr': mean_rel_corr,
           'mean_rel_contrast': mean_rel_contrast, 'mean_rel_iwe_divergence': mean_rel_div,
           'theta_total_variation': tv, 'multi_ref_weights': w}
    return final, aux


def loss_and_grad(theta, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta, cur_pyr_lvl, n_pyr_lvls,
                  sensor_size, scale_to_sensor_size_method='bilinear', contrast_kind=CONTRAST_GRAD_MAG,
                  return_intermediates=False):
    """value_and_grad(loss_func) by the hand-derived reverse pass.

    What jaxopt.ScipyMinimize(jit=True) computes through jax.value_and_grad at solver.py:165-173.
    Returns (value, grad (h,w,2), aux).
    """
    theta = np.asarray(theta, dtype=np.float64)
    edges = np.asarray(edges, dtype=np.float64)
    edge_ts = np.asarray(edge_ts, dtype=np.float64)
    ts = np.asarray(ts, dtype=np.float64)
    H, W = sensor_size
    HW = float(H * W)
    R = len(edge_ts)
    xi = np.asarray(xs).astype(np.int64)
    yi = np.asarray(ys).astype(np.int64)

    Theta = scale_theta_to_sensor_size(theta, sensor_size, scale_to_sensor_size_method)
    w = compute_weights_for_multi_reference(R)

    zero_iwe = events_to_pdf_frame(xi.astype(np.float64), yi.astype(np.float64), sensor_size)
    n0 = normalize_to_unit_range(zero_iwe)
    if contrast_kind == CONTRAST_VARIANCE:
        c0 = np.var(zero_iwe)
    else:
        c0 = compute_mean_gradient_magnitude(zero_iwe)
    d0 = iwe_divergence(n0)

    g_Theta = np.zeros_like(Theta)
    sum_rel_con = 0.0
    sum_rel_corr = 0.0
    sum_rel_div = 0.0
    inter = {'iwes': [], 'G': []}
    for r in range(R):
        wx, wy = per_pix_warp(Theta, xi, yi, ts, edge_ts[r], 1.0)
        I = events_to_pdf_frame(wx, wy, sensor_size)
        m, M = I.min(), I.max()
        D = M - m + EPSN
        n = (I - m) / D
        E = edges[r]
        mse = ((E - n) ** 2).mean()
        mse0 = ((E - n0) ** 2).mean()
        corr, zc = -mse, -mse0
        if contrast_kind == CONTRAST_VARIANCE:
            c = np.var(I)
            dc_dI = (2.0 / HW) * (I - I.mean())
        else:
            gx, gy = scharr_grads(I)
            c = (gx * gx + gy * gy).mean()
            dc_dI = (2.0 / HW) * (conv2_same_adjoint(gx, SCHARR_GX) + conv2_same_adjoint(gy, SCHARR_GY))
        d = iwe_divergence(n)
        sum_rel_con += w[r] * c / (c0 + EPSN)
        sum_rel_corr += w[r] * corr / (zc + EPSN)
        sum_rel_div += w[r] * d / (d0 + EPSN)

        a_r = -alpha * w[r] / (R * (c0 + EPSN))        # dL/dc_r
        b_r = -beta * w[r] / (R * (zc + EPSN))         # dL/dcorr_r
        e_r = delta * w[r] / (R * (d0 + EPSN))         # dL/dd_r
        Gn = b_r * (2.0 / HW) * (E - n)
        if delta != 0.0:
            Gn = Gn + e_r * iwe_divergence_adjoint(n)
        # through n = (I - m) / (M - m + eps), S4 tie sharing
        dm = np.sum(Gn * (n - 1.0)) / D
        dM = -np.sum(Gn * n) / D
        is_min = (I == m)
        is_max = (I == M)
        G = a_r * dc_dI + Gn / D + dm * is_min / is_min.sum() + dM * is_max / is_max.sum()
        gwx, gwy = events_to_pdf_frame_adjoint(G, wx, wy)
        dts = ts - edge_ts[r]
        # wx = x - Theta[y,x,0]*dt  ->  dL/dTheta[y,x,0] += -dt * dL/dwx
        g_Theta[:, :, 0] += np.bincount(yi * W + xi, weights=-dts * gwx, minlength=H * W).reshape(H, W)
        g_Theta[:, :, 1] += np.bincount(yi * W + xi, weights=-dts * gwy, minlength=H * W).reshape(H, W)
        if return_intermediates:
            inter['iwes'].append(I)
            inter['G'].append(G)

    mean_rel_con = sum_rel_con / R
    mean_rel_corr = sum_rel_corr / R
    mean_rel_div = sum_rel_div / R
    tv = 0.0
    if cur_pyr_lvl <= 0:
        tv, g_tv = per_pix_total_variation(Theta, xi, yi, return_grad=True)
        if gamma != 0.0:
            g_Theta += gamma * g_tv
    final = (alpha * (-mean_rel_con) + beta * (-mean_rel_corr)) + (gamma * tv + delta * mean_rel_div)
    grad = scale_theta_adjoint(g_Theta, theta.shape, scale_to_sensor_size_method)
    aux = {'final_loss': final, 'scaled_theta': Theta, 'mean_rel_corr': mean_rel_corr,
           'mean_rel_contrast': mean_rel_con, 'mean_rel_iwe_divergence': mean_rel_div,
           'theta_total_variation': tv, 'multi_ref_weights': w, 'g_Theta': g_Theta}
    if return_intermediates:
        aux['_iwes'] = np.stack(inter['iwes'])
        aux['_G'] = np.stack(inter['G'])
        aux['_zero_iwe'] = zero_iwe
    return final, grad, aux


def handover_loss_func(alpha_handover, prev_theta, theta, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta,
                       cur_pyr_lvl, n_pyr_lvls, sensor_size, scale_to_sensor_size_method='bilinear'):
    """losses.py:208-276 -> loss only."""
    theta_ho = alpha_handover * np.asarray(prev_theta) + (1 - alpha_handover) * np.asarray(theta)
    return loss_func(theta_ho, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta, cur_pyr_lvl, n_pyr_lvls,
                     sensor_size, scale_to_sensor_size_method)[0]


def handover_loss_and_grad(alpha_handover, prev_theta, theta, *args, **kw):
    """value and d/d(alpha_handover) of handover_loss_func: <dL/dtheta_ho, prev - theta>."""
    prev_theta = np.asarray(prev_theta, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64)
    theta_ho = alpha_handover * prev_theta + (1 - alpha_handover) * theta
    val, grad, _ = loss_and_grad(theta_ho, *args, **kw)
    return val, float(np.sum(grad * (prev_theta - theta)))
