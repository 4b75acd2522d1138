#!/usr/bin/env python3
"""Where the lockstep batch solve spends its wall time: cProfile of BatchedMultipleLevelEINCMSolver.solve() on the bench's C4 leg
(8 x [260x346, 1e6 events, R = 5], pyramid 1..16) plus the engine's own host-phase profile.  python tools/dev_c4_profile.py [B]"""
import cProfile
import importlib
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402

ge.build()
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
sol = importlib.import_module('edge-informed-contrast-maximization_amd.solver')
bsol = importlib.import_module('edge-informed-contrast-maximization_amd.batch_solver')

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W, N, R, n_lvls, maxiter = 260, 346, 1_000_000, 5, 5, 40
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
loss = dict(alpha=20.0, beta=35.0, gamma=0.0, delta=0.0, scale_to_sensor_size_method='bilinear')
maxit = sol.growing_maxiters(n_lvls, maxiter / 5, maxiter)
sp = {'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {'pyr_lvl_0': 1, 'pyr_lvl_1': 1}}
for rep in range(2):
    bs = bsol.BatchedMultipleLevelEINCMSolver(B, (H, W), n_lvls, maxit, loss, sp, pyramid_bases=[2] * (n_lvls - 1))
    bs.set_datasamples(args)
    bs.engine.host_profile(reset=True)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep == 1:
        pr.enable()
    bs.solve()
    if rep == 1:
        pr.disable()
    t = time.perf_counter() - t0
    us, n = bs.engine.host_profile()
    print(f'rep {rep}: solve {t:.3f} s, {bs.n_batch_evals} engine calls, {bs.n_window_evals} window evaluations; '
          f'engine host phases (us, summed): {us}, calls {n}; in the engine: {sum(us.values()) / 1e6:.3f} s')
    bs.close()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
print(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
print(s.getvalue())
