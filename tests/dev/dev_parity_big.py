#!/usr/bin/env python3
"""dev: parity numbers of one big case against the C port, for a given library build.  usage: dev_parity_big.py lib.so [case]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
L.LIB_PATH = os.path.abspath(sys.argv[1])
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
from oracle import eincm_c_port as CP
case = sys.argv[2] if len(sys.argv) > 2 else 'dsec_1e7_pyr16'
CASES = {'dsec_1e7_pyr16': ((480, 640), 10_000_000, 3, (16, 16), 'smooth', 2000.0, 4000.0, 0),
         'dsec_1e6_dense': ((480, 640), 1_000_000, 3, 'dense', 'smooth', 2000.0, 4000.0, 0),
         'mvsec_1e6_pyr16': ((260, 346), 1_000_000, 5, (16, 16), 'smooth', 20.0, 35.0, 0),
         'mvsec_1e6_2dof': ((260, 346), 1_000_000, 5, (1, 1), 'constant', 20.0, 35.0, 4)}
(H, W), N, R, hw, flow, al, be, lvl = CASES[case]
win = synth.make_window(21, (H, W), N, R, flow=flow, flow_mag=20.0)
th = win['flow_gt'] * np.random.default_rng(3).uniform(0.5, 1.5, (H, W, 2)) if hw == 'dense' else synth.theta_near_truth(21, win, hw)
a = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
v_ref, g_ref, im = CP.loss_and_grad(th, *a, al, be, (H, W), nthreads=16, return_images=True)
rel = lambda x, y: np.abs(np.asarray(x, float) - y).max() / np.abs(y).max()
with engine.Engine((H, W), N, max_refs=R) as eng:
    eng.set_window(*a)
    v, g, _ = eng.loss_grad(th, engine.make_params(al, be, 0.0, 0.0, lvl))
    print(os.path.basename(sys.argv[1]), case, 'loss', abs(v[0] - v_ref) / abs(v_ref), 'grad', rel(g[0], g_ref), 'iwe', rel(eng.iwes()[0], im['iwes']),
          'G', rel(eng.image_grad()[0], im['G']), 'max|G|', np.abs(im['G']).max(), 'median|G|', np.median(np.abs(im['G'])))
