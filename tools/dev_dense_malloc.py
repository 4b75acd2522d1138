"""dev: dense-theta loss_grad wall time with the default glibc malloc policy and with large blocks kept on the heap."""
import sys, os, time, importlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R = 480, 640, 1_000_000, 3
win = synth.make_window(7, (H, W), N, R, flow='smooth', flow_mag=20.0)
th0 = win['flow_gt'] * 0.9
ths = [th0 * (1 + .01 * k) for k in range(5)]
p = engine.make_params(20., 35., 2.5e-4, 0., 0)
def run(tag):
    with engine.Engine((H, W), N, max_refs=R) as e:
        e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
        for k in range(5): e.loss_grad(ths[k], p)
        ts = []
        for k in range(30):
            t0 = time.perf_counter(); v, g, _ = e.loss_grad(ths[k % 5], p); ts.append(time.perf_counter() - t0)
        print(tag, 'loss_grad dense: median %.3f ms min %.3f ms' % (np.median(ts) * 1e3, min(ts) * 1e3), flush=True)
run('default malloc')
libc = ctypes.CDLL(None)
M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
print('mallopt', libc.mallopt(M_MMAP_THRESHOLD, 1 << 30), libc.mallopt(M_TRIM_THRESHOLD, 1 << 30))
run('heap-resident large blocks')
