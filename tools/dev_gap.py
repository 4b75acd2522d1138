#!/usr/bin/env python3
"""Host turn-around of the bench step (dev tool): wall time per 8-window evaluation with the HIP-event bracketing off / 'dominant' / full."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R, B = 260, 346, 1000000, 5, 8
raw = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
th = np.stack([synth.theta_near_truth(b, w, (1, 1)) for b, w in enumerate(raw)])
ths = [np.ascontiguousarray(th * (1.0 + 0.01 * ((k % 7) - 3))) for k in range(7)]
p = engine.make_params(20., 35., 0., 0., 4)
for timing in (None, 'dominant', True, None, 'dominant'):
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=timing) as e:
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in raw])
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        n = 200
        t0 = time.perf_counter()
        for k in range(n):
            e.loss_grad(ths[k % 7], p)
        dt = (time.perf_counter() - t0) / n
        print('timing=%-9s %.1f us per step' % (timing, dt * 1e6), flush=True)
