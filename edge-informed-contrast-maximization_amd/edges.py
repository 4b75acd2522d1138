"""Edge smoothing on the GPU: the last step that produces the ``edges`` stack the loss consumes (SURVEY f-4).

Mirrors the smoothing callables the reference's hydra group ``edge_extraction/smoothen`` binds
(src/experiments/e00/configs/edge_extraction/smoothen/{gaussian,iedt}.yaml), same names and arguments:

  smoothen_edges(edge_img, k_size=1, sigma=1)                         src/utils/img_utils.py:210-220
  eincm_inv_exp_dist_transform(edge_img, alpha=6)                      src/utils/img_utils.py:229-233
  rtef_inv_exp_dist_transform(edge_img, d_sat, alpha_iedt, formulation)   src/utils/img_utils.py:223-226, :236-410

The Canny detector and the photometric clean-up in front of it (OpenCV, img_utils.py:131-207) are not part of this
package: ``edge_img`` is the binary image they produce.  Everything here runs in libeincm_hip.so; no CPU fallback.
"""
import numpy as np

from .engine import Engine

_engines = {}


def _engine(shape, device=0):
    """A small context per sensor size: these entry points only need its stream and scratch buffers."""
    key = (int(shape[0]), int(shape[1]), int(device))
    if key not in _engines:
        _engines[key] = Engine(key[:2], max_events_total=1, max_refs=1, max_windows=1, device=device)
    return _engines[key]


def clear_engines():
    for e in _engines.values():
        e.close()
    _engines.clear()


def smoothen_edges(edge_img, k_size=1, sigma=1, engine=None):
    """Gaussian smoothing as the reference's call performs it.  The reference passes its arguments to OpenCV positionally
    (``cv.GaussianBlur(edge_img, None, k_size, sigma, 0)``, img_utils.py:218), which binds ``k_size`` to sigmaX and
    ``sigma`` to the (ignored) output array; the kernel size is then derived from sigmaX.  ``sigma`` is therefore accepted
    and unused here too."""
    del sigma
    img = np.asarray(edge_img).astype(np.float64)
    eng = engine or _engine(img.shape[-2:])
    return eng.gaussian_blur(img, float(k_size))


def eincm_inv_exp_dist_transform(edge_img, alpha=6, engine=None):
    e = np.asarray(edge_img)
    eng = engine or _engine(e.shape[-2:])
    return eng.inv_dist_transform(e.astype('bool'), 'exponential', alpha=float(alpha))


def rtef_inv_exp_dist_transform(edge_img, dist_surf_saturation_distance=None, alpha_iedt=None, formulation='exponential',
                                engine=None):
    e = np.asarray(edge_img)
    vals = np.unique(e)
    assert e.ndim == 2 and len(vals) == 2 and 0 in vals.astype('int'), 'Need 2D binary edge image'    # img_utils.py:397-399
    d_sat = 6.0 if dist_surf_saturation_distance is None else float(dist_surf_saturation_distance)   # img_utils.py:256
    alpha = d_sat / 5.541 if alpha_iedt is None else float(alpha_iedt)                               # img_utils.py:257
    eng = engine or _engine(e.shape)
    return eng.inv_dist_transform(e.astype('bool'), formulation, alpha=alpha, d_sat=d_sat)


def smooth_edge_stack(edge_imgs, smoothen_edges_func=smoothen_edges, **kw):
    """exp_mgr.py:343-350 for a stack of binary edge images: smooth each, then min-max normalise each to [0, 1]."""
    from .staging import normalize_edges
    return normalize_edges([smoothen_edges_func(e, **kw) for e in edge_imgs])
