import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'libeincm_hosttrace.so')
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R = 480, 640, 1_000_000, 3
win = synth.make_window(7, (H, W), N, R, flow='smooth', flow_mag=20.0)
th = win['flow_gt'] * 0.9
p = engine.make_params(20., 35., 2.5e-4, 0., 0)
with engine.Engine((H, W), N, max_refs=R) as e:
    e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
    for k in range(5): e.loss_grad(th * (1 + .01 * k), p)
    os.environ['EINCM_TRACE_HOST'] = '1'
    for k in range(4):
        x = th * (1 + .01 * (k % 5))
        t0 = time.perf_counter(); e.loss_grad(x, p); print('wall us', (time.perf_counter() - t0) * 1e6, flush=True)
