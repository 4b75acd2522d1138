import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, N, R, B = 260, 346, 1_000_000, 5, 8
p = engine.make_params(20., 35., 0., 0., 4)
for noise in (0.1, 0.5, 1.0):
    wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0, noise_frac=noise) for b in range(B)]
    th = np.stack([w['flow_gt'][0, 0].reshape(1, 1, 2) for w in wins])
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
        acc = {}
        for k in range(6):
            e.loss_grad(th * (1 + 0.01 * k), p)
            if k >= 1:
                for kk, vv in e.timings().items(): acc[kk] = acc.get(kk, 0) + vv / 5
        print(f'noise fraction {noise}: splat {acc["splat"]*1e3:.0f} us  gather {acc["gather"]*1e3:.0f} us')
