import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, R = 260, 346, 5
for N, mag in ((100_000, 60.0), (1_000_000, 60.0), (1_000_000, 150.0)):
    win = synth.make_window(5, (H, W), N, R, flow='smooth', flow_mag=mag)
    th = synth.theta_near_truth(5, win, (8, 8))
    p = engine.make_params(20., 35., 0., 0., 1)
    with engine.Engine((H, W), N, max_refs=R) as e:
        e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
        for k in range(4): e.loss_grad(th, p)
        ts = []
        for k in range(20):
            t0 = time.perf_counter(); e.loss_grad(th * (1 + 0.01 * (k % 3)), p); ts.append(time.perf_counter() - t0)
    print(f'N={N} flow {mag}: {np.median(ts)*1e6:.0f} us', 'WINCAP=' + os.environ.get('EINCM_WINCAP', 'adaptive'))
