#!/usr/bin/env python3
"""Per-stage HIP-event times of one loss+grad for a few workload shapes (dev tool): where a pyramid-level evaluation spends its time."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
CFG = [(260, 346, 1000000, 5, (1, 1), 1, 0.0), (260, 346, 1000000, 5, (16, 16), 1, 0.0), (260, 346, 1000000, 5, (16, 16), 1, 2.5e-4),
       (260, 346, 1000000, 5, (16, 16), 8, 0.0), (480, 640, 10000000, 3, (16, 16), 1, 0.0), (480, 640, 1000000, 3, 'dense', 1, 0.0)]
if os.environ.get('DEV_CFG'):                  # e.g. DEV_CFG=1,3 picks rows
    CFG = [CFG[int(i)] for i in os.environ['DEV_CFG'].split(',')]
if os.environ.get('DEV_LIB'):                  # an alternative build of the library
    importlib.import_module('edge-informed-contrast-maximization_amd._lib').LIB_PATH = os.path.abspath(os.environ['DEV_LIB'])
for (H, W, N, R, hw, B, gamma) in CFG:
    dense = hw == 'dense'
    wins = [synth.make_window(b, (H, W), N, R, flow='smooth' if dense else 'constant', flow_mag=20.0) for b in range(B)]
    th = np.stack([w['flow_gt'] if dense else synth.theta_near_truth(b, w, hw) for b, w in enumerate(wins)])
    p = engine.make_params(20., 35., gamma, 0., 0 if (dense or gamma) else (4 if hw == (1, 1) else 1))
    args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
        e.set_windows(args)
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        acc, n = {}, 10
        for k in range(n):
            e.loss_grad(th * (1 + 0.01 * (k % 5)), p)
            for kk, vv in e.timings().items():
                acc[kk] = acc.get(kk, 0.0) + vv * 1e3 / n
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B) as e:
        e.set_windows(args)
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        ts = []
        ths = [th * (1 + 0.01 * k) for k in range(5)]
        for k in range(30):
            t0 = time.perf_counter(); e.loss_grad(ths[k % 5], p); ts.append(time.perf_counter() - t0)
    print('%dx%d N=%g R=%d th=%s B=%d gamma=%g: wall %.0f us | ' % (H, W, N, R, hw, B, gamma, np.median(ts) * 1e6) +
          ' '.join('%s %.1f' % (k, v) for k, v in acc.items() if v > 0.05), flush=True)
