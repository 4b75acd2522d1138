"""Wall time of the f-4 entry points on the GPU box beside the CPU routines the reference calls (scipy EDT) or the
oracle's numpy restatement (Gaussian).  Host buffers in, host buffers out (that is the reference's function shape)."""
import importlib
import os
import sys
import time

import numpy as np
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
E = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
from oracle import edge_smoothing as ES           # noqa: E402  (dev tool: oracle as the timed CPU side)
from tests.test_gpu_edges import _canny_like       # noqa: E402


def med(f, n=20):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for shape, R in [((260, 346), 5), ((480, 640), 3)]:
    imgs = np.stack([_canny_like(shape, s) for s in range(R)])
    f64 = imgs.astype(np.float64)
    with E.Engine(shape, 1, max_refs=1) as e:
        t_gpu = med(lambda: e.inv_dist_transform(imgs, alpha=6.0))
        t_cpu = med(lambda: [ES.eincm_inv_exp_dist_transform(i, 6.0) for i in imgs], 5)
        g_gpu = med(lambda: e.gaussian_blur(f64, 1.0))
        g_cpu = med(lambda: [ES.smoothen_edges(i, 1) for i in f64], 5)
        print(f'{shape} x{R}: IEDT gpu {t_gpu:.3f} ms, scipy+numpy {t_cpu:.2f} ms | gaussian gpu {g_gpu:.3f} ms, numpy {g_cpu:.2f} ms')
