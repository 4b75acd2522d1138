"""dev: set_window time at 1e7 events (480x640, R = 3), with and without pinning the caller's arrays in place."""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R = 480, 640, 10_000_000, 3
win = synth.make_window(7, (H, W), N, R, flow='smooth', flow_mag=20.0)
a = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
with engine.Engine((H, W), N, max_refs=R) as e:
    import ctypes as C
    n = np.array([N], dtype=np.int64); edges = np.ascontiguousarray(win['edges'][None]); ets = np.ascontiguousarray(win['edge_ts'][None])
    for k in range(5):
        t0 = time.perf_counter(); e.set_window(*a); t1 = time.perf_counter()
        rc = e._lib.eincm_set_windows_ex(e._ctx, 1, R, n.ctypes.data_as(C.POINTER(C.c_int64)), win['xs'].ctypes.data_as(C.POINTER(C.c_int16)),
                                         win['ys'].ctypes.data_as(C.POINTER(C.c_int16)), win['ts'].ctypes.data_as(C.POINTER(C.c_double)),
                                         edges.ctypes.data_as(C.POINTER(C.c_double)), ets.ctypes.data_as(C.POINTER(C.c_double)), 0)
        t2 = time.perf_counter()
        print('set_window 1e7: python+C %.2f ms, C call alone %.2f ms (rc %d)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, rc), flush=True)
    th = synth.theta_near_truth(7, win, (4, 4))
    v, g, _ = e.loss_grad(th, engine.make_params(20., 35., 0., 0., 2))
    print('loss', v[0])
