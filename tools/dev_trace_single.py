"""Single-window evaluation loop for a kernel trace: python3 tools/dev_trace_single.py [N] [h]  (run under rocprofv3 --kernel-trace)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W, R = 260, 346, 5
win = synth.make_window(0, (H, W), N, R, flow='constant', flow_mag=20.0)
th = synth.theta_near_truth(0, win, (h, h))
p = engine.make_params(20., 35., 0., 0., 4 if h == 1 else 1)
with engine.Engine((H, W), N, max_refs=R) as e:
    e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
    for k in range(10): e.loss_grad(th * (1 + .01 * k), p)
    ts = []
    for k in range(100):
        t0 = time.perf_counter(); e.loss_grad(th * (1 + .01 * (k % 5)), p); ts.append(time.perf_counter() - t0)
    print(f'N={N} theta=({h},{h}) wall median {np.median(ts)*1e6:.1f} us min {min(ts)*1e6:.1f} us')
    e.host_profile(reset=True)
    for k in range(200): e.loss_grad(th * (1 + .01 * (k % 5)), p)
    hp, n = e.host_profile()
    print('host phases, us per evaluation: ' + ', '.join(f'{k} {v / n:.1f}' for k, v in hp.items()))
