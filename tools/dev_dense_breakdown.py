"""C3 (dense theta, 480x640, 1e6 events, R=3): wall time of one loss+grad vs the GPU stage times, to see what the float64
host boundary costs.  Run on the GPU box."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, N, R = 480, 640, 1_000_000, 3
win = synth.make_window(7, (H, W), N, R, flow='smooth', flow_mag=20.0)
th = win['flow_gt'] * 0.9
p = engine.make_params(20., 35., 2.5e-4, 0., 0)
for timing in (False, True):
    with engine.Engine((H, W), N, max_refs=R, timing=timing) as e:
        e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
        for k in range(5): e.loss_grad(th * (1 + .01 * k), p)
        ts = []
        for k in range(30):
            x = th * (1 + .01 * (k % 5))
            t0 = time.perf_counter(); e.loss_grad(x, p); ts.append(time.perf_counter() - t0)
        msg = f'dense C3 timing={timing}: wall median {np.median(ts)*1e3:.3f} ms min {min(ts)*1e3:.3f} ms'
        if timing:
            msg += ' | ' + ' '.join(f'{k}={v*1e3:.0f}us' for k, v in e.timings().items() if v > 0)
        print(msg)
a = np.random.rand(H, W, 2); b = np.empty_like(a)
t0 = time.perf_counter()
for _ in range(20): np.copyto(b, a)
print(f'host memcpy of one dense theta ({a.nbytes/1e6:.1f} MB): {(time.perf_counter()-t0)/20*1e3:.3f} ms; isfinite scan: ', end='')
t0 = time.perf_counter()
for _ in range(20): np.isfinite(a).all()
print(f'{(time.perf_counter()-t0)/20*1e3:.3f} ms')
