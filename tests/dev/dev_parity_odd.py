#!/usr/bin/env python3
"""dev: parity against the C/OpenMP port on shapes outside the benchmark set (large sensor, 8 reference times, all events in one tile,
theta grids finer than the LDS staging of the resampling weights).  Run on the GPU box: python tests/dev/dev_parity_odd.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
from oracle import eincm_c_port as CP
rel = lambda x, y: np.abs(np.asarray(x, float) - y).max() / max(np.abs(y).max(), 1e-300)
CASES = [('720x1280 dense', (720, 1280), 400_000, 3, 'dense', 'smooth', None),
         ('720x1280 2-DoF R=8', (720, 1280), 400_000, 8, (1, 1), 'constant', None),
         ('720x1280 theta 40x70', (720, 1280), 400_000, 2, (40, 70), 'smooth', None),
         ('260x346 one hot tile', (260, 346), 600_000, 5, (4, 4), 'constant', 'hot'),
         ('260x346 theta 200x300 (direct resampling path)', (260, 346), 300_000, 2, (200, 300), 'smooth', None),
         ('100x3000 strip', (100, 3000), 300_000, 3, (3, 50), 'smooth', None)]
for name, (H, W), N, R, hw, flow, special in CASES:
    win = synth.make_window(77, (H, W), N, R, flow=flow, flow_mag=12.0)
    if special == 'hot':                                   # every event inside one 32x32 tile
        win['xs'] = (64 + win['xs'] % 32).astype(np.int16); win['ys'] = (96 + win['ys'] % 32).astype(np.int16)
    rng = np.random.default_rng(5)
    if hw == 'dense':
        th = win['flow_gt'] * rng.uniform(0.5, 1.5, (H, W, 2))
    elif hw[0] > 32:
        th = rng.normal(0.0, 4.0, hw + (2,))
    else:
        th = synth.theta_near_truth(77, win, hw)
    a = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
    v_ref, g_ref, im = CP.loss_and_grad(th, *a, 20.0, 35.0, (H, W), nthreads=16, return_images=True)
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*a)
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4 if hw == (1, 1) else 1))
        v2, g2, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4 if hw == (1, 1) else 1))
        print('%-48s loss %.1e grad %.1e iwe %.1e repeat identical %s' % (name, abs(v[0] - v_ref) / abs(v_ref), rel(g[0], g_ref),
              rel(eng.iwes()[0], im['iwes']), bool(np.array_equal(v, v2) and np.array_equal(g, g2))), flush=True)
