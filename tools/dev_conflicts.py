import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, N, R, B = 260, 346, 1_000_000, 5, 8
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
truth = np.stack([w['flow_gt'][0, 0].reshape(1, 1, 2) for w in wins])
p = engine.make_params(20., 35., 0., 0., 4)
with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    for s in (0.0, 0.5, 0.9, 1.0, 1.1, 2.0, -3.0):
        acc = {}
        for k in range(6):
            e.loss_grad(truth * s, p)
            if k >= 1:
                for kk, vv in e.timings().items(): acc[kk] = acc.get(kk, 0) + vv / 5
        print(f'theta = {s:+.1f} x truth: splat {acc["splat"]*1e3:.0f} us  gather {acc["gather"]*1e3:.0f} us  total {acc["total"]*1e3:.0f} us')
