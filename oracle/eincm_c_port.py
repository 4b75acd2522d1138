"""ctypes wrapper of oracle/libeincm_ref.so (C / OpenMP port of the alpha/beta path).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import eincm_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load(build=True):
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libeincm_ref.so')
        if build and (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, 'eincm_ref.c'))):
            subprocess.run(['make', '-s', '-C', _HERE], check=True)
        lib = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        lib.eincm_ref_loss_grad.restype = C.c_int
        lib.eincm_ref_loss_grad.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int16), dp, dp, dp,
                                            dp, C.c_double, C.c_double, dp, dp, C.c_int]
        lib.eincm_ref_loss_grad_ex.restype = C.c_int
        lib.eincm_ref_loss_grad_ex.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int16), dp, dp, dp,
                                               dp, C.c_double, C.c_double, dp, dp, dp, dp, C.c_int]
        lib.eincm_ref_max_threads.restype = C.c_int
        _LIB = lib
    return _LIB


def max_threads():
    return int(load().eincm_ref_max_threads())


def loss_and_grad(theta, xs, ys, ts, edges, edge_ts, alpha, beta, sensor_size, method='bilinear', nthreads=None, want_grad=True,
                  return_images=False):
    """(value, grad (h,w,2)) of loss_func with gamma = delta = 0 (any cur_pyr_lvl), float64, `nthreads` OpenMP threads.
    return_images: also return {'iwes': (R,H,W), 'G': (R,H,W) dL/dIWE (None without a gradient)}."""
    lib = load()
    H, W = sensor_size
    theta = np.asarray(theta, dtype=np.float64)
    Theta = np.ascontiguousarray(O.scale_theta_to_sensor_size(theta, (H, W), method))
    xs = np.ascontiguousarray(xs, dtype=np.int16); ys = np.ascontiguousarray(ys, dtype=np.int16)
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    edges = np.ascontiguousarray(edges, dtype=np.float64); edge_ts = np.ascontiguousarray(edge_ts, dtype=np.float64)
    val = C.c_double(0.0)
    g = np.zeros((H, W, 2)) if want_grad else None
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    R = len(edge_ts)
    iwes = np.zeros((R, H, W)) if return_images else None
    G = np.zeros((R, H, W)) if (return_images and want_grad) else None
    rc = lib.eincm_ref_loss_grad_ex(H, W, len(xs), R, xs.ctypes.data_as(C.POINTER(C.c_int16)),
                                    ys.ctypes.data_as(C.POINTER(C.c_int16)), dp(ts), dp(edges), dp(edge_ts), dp(Theta), float(alpha),
                                    float(beta), C.byref(val), dp(g) if want_grad else None,
                                    dp(iwes) if iwes is not None else None, dp(G) if G is not None else None,
                                    int(nthreads or max_threads()))
    if rc:
        raise MemoryError('eincm_ref_loss_grad failed')
    grad = O.scale_theta_adjoint(g, theta.shape, method) if want_grad else None
    if return_images:
        return val.value, grad, {'iwes': iwes, 'G': G}
    return val.value, grad
