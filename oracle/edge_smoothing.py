"""CPU oracle for SURVEY row f-4: edge smoothing and the tiled ("adaptive") objectives.  numpy / scipy, float64.

TEST INFRASTRUCTURE ONLY (same rule as oracle/eincm_oracle.py): imported by ``tests/`` only, never by the product.

Pinning, function by function:
  * ``eincm_inv_exp_dist_transform`` — PINNED to the third-party routine the reference itself calls:
    ``scipy.ndimage.distance_transform_edt`` (src/utils/img_utils.py:229-233) is importable here and is called the same
    way; the rest of the function is three numpy expressions.  (scipy's version is not pinned by the reference; the
    exact Euclidean distance transform is unique, so any correct version gives the same integers under the root.)
  * ``rtef_edt_squared`` / ``rtef_inv_exp_dist_transform`` — restatement of the pure-numpy class ``RTEF_IEDT``
    (img_utils.py:236-410, a two-phase Meijster transform on int32).  Pinned by known answers: its squared distances
    must equal the brute-force definition and scipy's transform squared (tests/test_oracle_edges.py).
  * ``smoothen_edges`` — PARITY UNPINNED.  The reference calls ``cv.GaussianBlur(edge_img, None, k_size, sigma, 0)``
    (img_utils.py:210-220); OpenCV is not installed here.  Restated from knowledge of OpenCV: positional binding makes
    that call ``ksize=None`` (-> derived from sigma), ``sigmaX=k_size``, ``dst=sigma`` (ignored), ``sigmaY=0``
    (-> = sigmaX); for CV_64F the kernel size is ``round(8*sigma + 1) | 1``, the kernel ``exp(-x^2 / (2 sigma^2))``
    normalised to sum 1, separable, border BORDER_REFLECT_101.  Only checked against analytic properties.
  * tiled objectives — restatement of ``extract_tiles`` (img_utils.py:105-120) and the ``compute_adaptive_*`` family
    (contrast_objectives.py:42-87, correlation_objectives.py:28-130); jax absent -> unpinned like the main oracle, but
    each is a composition of functions the main oracle already restates.
"""
import math
import sys

import numpy as np
from scipy import ndimage

from . import eincm_oracle as O

EPSN = sys.float_info.epsilon
BIG_INT = np.iinfo(np.int32).max          # img_utils.py:260


def normalize_to_unit_range(arr):
    """img_utils.py:24-25."""
    return (arr - arr.min()) / (arr.max() - arr.min() + EPSN)


# ---------------------------------------------------------------------------------------------- IEDT (scipy flavour)
def eincm_inv_exp_dist_transform(edge_img, alpha=6):
    """img_utils.py:229-233: distance to the nearest edge pixel -> 1 - exp(-d/alpha) -> 1 - minmax."""
    d = ndimage.distance_transform_edt(~(np.asarray(edge_img).astype('bool')))
    e = 1 - np.exp(-d / alpha)
    return 1 - normalize_to_unit_range(e)


def edt_squared_bruteforce(edge_img):
    """Definition of the squared Euclidean distance transform, O((HW)^2); tiny images only."""
    e = np.asarray(edge_img).astype(bool)
    ys, xs = np.nonzero(e)
    H, W = e.shape
    yy, xx = np.mgrid[0:H, 0:W]
    d2 = (yy[..., None] - ys) ** 2 + (xx[..., None] - xs) ** 2
    return d2.min(axis=-1).astype(np.int64)


# ---------------------------------------------------------------------------------------------- IEDT (RTEF flavour)
def _rtef_map_x(edge):
    """Phase 1 (img_utils.py:314-332): per row, distance to the nearest edge pixel of that row; BIG_INT if the row has none."""
    H, W = edge.shape
    g = np.full((H, W), BIG_INT, dtype=np.int64)
    for y in range(H):
        last = None
        for x in range(W):                              # left-to-right
            if edge[y, x]:
                last = x
            if last is not None:
                g[y, x] = x - last
        for x in range(W - 2, -1, -1):                  # right-to-left
            if g[y, x] > g[y, x + 1]:
                g[y, x] = g[y, x + 1] + 1
    return g


def rtef_edt_squared(edge_img):
    """Phase 2 (img_utils.py:335-370): per column, lower envelope of the parabolas f_i(j) = g[i]^2 + (j - i)^2 kept on a
    stack (s = apex rows, t = first row where the apex takes over), with the reference's integer conventions: floor
    division for the intersection abscissa, rows whose g is BIG_INT never enter the stack, start with s[0] = t[0] = 0."""
    edge = np.asarray(edge_img).astype(bool)
    H, W = edge.shape
    g = _rtef_map_x(edge)
    out = np.zeros((H, W), dtype=np.int64)

    for x in range(W):
        col = g[:, x]

        def f(i, j):                                    # parabola_ordinate, img_utils.py:271-288
            return BIG_INT if col[i] == BIG_INT else int(col[i]) ** 2 + (j - i) ** 2

        def sep(i, u):                                  # parabolas_intersection_abscissa, img_utils.py:291-311
            if col[i] == BIG_INT or col[u] == BIG_INT:
                return BIG_INT
            return (u * u - i * i + int(col[u]) ** 2 - int(col[i]) ** 2) // (2 * (u - i))

        q, s, t = 0, [0] * H, [0] * H
        for u in range(1, H):
            while q >= 0 and f(s[q], t[q]) > f(u, t[q]):
                q -= 1
            if q < 0:
                q, s[0] = 0, u
            else:
                w = sep(s[q], u)
                if w != BIG_INT:
                    w += 1
                    if 0 <= w < H:
                        q += 1
                        s[q], t[q] = u, w
        for j in range(H - 1, -1, -1):
            out[j, x] = f(s[q], j)
            if j == t[q]:
                q -= 1
    return out


def rtef_inv_exp_dist_transform(edge_img, dist_surf_saturation_distance=None, alpha_iedt=None, formulation='exponential'):
    """img_utils.py:223-226 -> RTEF_IEDT.compute_edge_iedt (:396-410): sqrt, formulation, min-max normalise, 1 - x."""
    e = np.asarray(edge_img)
    vals = set(e.flatten().tolist())
    assert e.ndim == 2 and len(vals) == 2 and 0 in {int(v) for v in vals}, 'Need 2D binary edge image'   # :397-399
    d_sat = dist_surf_saturation_distance if dist_surf_saturation_distance is not None else 6.0        # :256
    alpha = alpha_iedt if alpha_iedt is not None else d_sat / 5.541                                      # :257
    d = np.sqrt(np.abs(rtef_edt_squared(e).astype(np.float64)))                                          # :373
    if formulation == 'linear':
        pass
    elif formulation == 'linear-bound':
        d = np.minimum(d, d_sat)
    elif formulation == 'logarithmic':
        d = np.log(d + 1.0)
    elif formulation == 'exponential':
        d = 1 - np.exp(-d / alpha)
    else:
        raise NotImplementedError(formulation)
    return 1 - normalize_to_unit_range(d)


# ---------------------------------------------------------------------------------------------- Gaussian (OpenCV flavour)
def gaussian_kernel_cv(sigma):
    """cv::getGaussianKernel for the automatic size of a CV_64F image: n = round(8 sigma + 1) | 1."""
    n = int(round(sigma * 4 * 2 + 1)) | 1
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    k = np.exp(-0.5 / (sigma * sigma) * x * x)
    return k / k.sum()


def smoothen_edges(edge_img, k_size=1, sigma=1):
    """img_utils.py:210-220.  ``k_size`` lands in OpenCV's sigmaX slot, ``sigma`` in its ``dst`` slot (see module doc)."""
    del sigma
    img = np.asarray(edge_img).astype(np.float64)
    k = gaussian_kernel_cv(float(k_size))
    r = len(k) // 2
    H, W = img.shape
    if r >= W or r >= H:
        raise ValueError('image smaller than the kernel radius: BORDER_REFLECT_101 undefined')
    p = np.pad(img, ((0, 0), (r, r)), mode='reflect')
    rows = sum(k[i] * p[:, i:i + W] for i in range(len(k)))
    p = np.pad(rows, ((r, r), (0, 0)), mode='reflect')
    return sum(k[i] * p[i:i + H, :] for i in range(len(k)))


# ---------------------------------------------------------------------------------------------- tiled objectives
def extract_tiles(arr, tile_h, tile_w):
    """img_utils.py:105-120: whole tiles only, row-major; the ragged right/bottom remainder is ignored."""
    H, W = arr.shape
    return np.stack([arr[i * tile_h:(i + 1) * tile_h, j * tile_w:(j + 1) * tile_w]
                     for i in range(H // tile_h) for j in range(W // tile_w)])


def _tile_size(tile_size):
    return (32, 42) if tile_size is None else tuple(tile_size)       # contrast_objectives.py:56-59


def compute_adaptive_mean_gradient_magnitude(arr, tile_size=None):
    """contrast_objectives.py:42-64: each tile convolved on its own ('same', zero padded at the TILE border)."""
    th, tw = _tile_size(tile_size)
    return float(sum(O.compute_mean_gradient_magnitude(t) for t in extract_tiles(arr, th, tw)))


def compute_adaptive_variance(arr, tile_size=None):
    """contrast_objectives.py:67-87."""
    th, tw = _tile_size(tile_size)
    return float(sum(np.var(t.astype(np.float64)) for t in extract_tiles(arr, th, tw)))


def compute_adaptive_mean_squared_error(a, b, tile_size=None):
    """correlation_objectives.py:105-130."""
    th, tw = _tile_size(tile_size)
    return float(sum(O.compute_mean_squared_error(x, y) for x, y in zip(extract_tiles(a, th, tw), extract_tiles(b, th, tw))))


def compute_sum_squared_error(a, b):
    """correlation_objectives.py:28-43."""
    return float(((a - b) ** 2).sum())


def compute_mean_hadamard_product(a, b):
    """correlation_objectives.py:46-62."""
    return float((a * b).mean())


def compute_sum_hadamard_product(a, b):
    """correlation_objectives.py:65-81."""
    return float((a * b).sum())


def compute_joint_contrast(a, b):
    """correlation_objectives.py:84-102."""
    return float(O.compute_mean_gradient_magnitude(a + b))
