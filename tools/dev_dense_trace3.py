import sys, os, time, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R = 480, 640, 1_000_000, 3
win = synth.make_window(7, (H, W), N, R, flow='smooth', flow_mag=20.0)
th0 = win['flow_gt'] * 0.9
ths = [th0 * (1 + .01 * k) for k in range(5)]
p = engine.make_params(20., 35., 2.5e-4, 0., 0)
T = time.perf_counter
def lg(self, theta, params):
    t = [T()]
    th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64)); t.append(T())
    th = th[None]; t.append(T())
    _, h, w, _ = th.shape
    value = np.empty(self.B, dtype=np.float64); t.append(T())
    grad = np.empty_like(th); t.append(T())
    a1 = th.ctypes.data; a2 = value.ctypes.data; a3 = grad.ctypes.data; pp = C.byref(params); t.append(T())
    rc = self._lib.eincm_loss_grad(self._ctx, a1, h, w, pp, a2, a3, None); t.append(T())
    self._check(rc, True); t.append(T())
    return value, grad, None, [round((b - a) * 1e6) for a, b in zip(t, t[1:])]
with engine.Engine((H, W), N, max_refs=R) as e:
    e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
    for k in range(5): e.loss_grad(ths[k], p)
    for k in range(6):
        t0 = T(); v, g, _, tt = lg(e, ths[k % 5], p); t1 = T()
        print('total %.0f us; asarray, [None], value, grad, ctypes-args, C call, check:' % ((t1 - t0) * 1e6), tt, flush=True)
    for k in range(4):
        t0 = T(); v, g, _ = e.loss_grad(ths[k % 5], p); print('engine.loss_grad %.0f us' % ((T() - t0) * 1e6), flush=True)
