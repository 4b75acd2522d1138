"""k_splat / k_gather / step on the bench batch against the magnitude of a 2-DoF theta (px per window), for the library / environment in
effect: where do long splat segments stop paying?   python3 tools/dev_theta_mag.py"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, R, B, N = 260, 346, 5, 8, 1_000_000
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
base = np.stack([synth.theta_near_truth(b, w, (1, 1)) for b, w in enumerate(wins)])
mag = np.linalg.norm(base.reshape(B, 2), axis=1).reshape(B, 1, 1, 1) + 1e-9
p = engine.make_params(20., 35., 0., 0., 4)
with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing='dominant') as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    for px in (0.0, 20.0, 40.0, 60.0, 80.0, 117.0):
        th = base / mag * px
        for k in range(30): e.loss_grad(th * (1 + 0.001 * k), p)
        e.timings_total(reset=True)
        t0 = time.perf_counter()
        for k in range(20): e.loss_grad(th * (1 + 0.001 * k), p)
        dt = (time.perf_counter() - t0) / 20
        acc, cnt = e.timings_total()
        print(f'|theta| {px:6.1f} px: splat {acc["splat"] / cnt * 1e3:7.1f} us  gather {acc["gather"] / cnt * 1e3:7.1f} us  step {dt * 1e6:7.1f} us')
