"""Helper of tests/test_gpu_switches.py: evaluate a fixed set of cases in THIS process (whose environment carries the switch under test;
some switches are read once per process) and write the results to an .npz.  Not a test module."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

H, W, R, B = 120, 160, 3, 2
N = (40000, 26000)
CASES = [((1, 1), 0.0, 4), ((4, 4), 2.5e-4, 0), ((16, 16), 0.0, 1), ((8, 8), 2.5e-4, 0)]


def inputs():
    synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
    wins = [synth.make_window(120 + b, (H, W), N[b], R, flow='smooth', flow_mag=9.0) for b in range(B)]
    thetas = [np.stack([synth.theta_near_truth(120 + b, w, hw) for b, w in enumerate(wins)]) for hw, _, _ in CASES]
    return wins, thetas


def main(out_path):
    import __graft_entry__ as ge
    ge.build()
    engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
    wins, thetas = inputs()
    res = {}
    with engine.Engine((H, W), sum(N), max_refs=R, max_windows=B) as eng:
        eng.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
        for i, (th, (hw, gamma, lvl)) in enumerate(zip(thetas, CASES)):
            v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, gamma, 0.0, lvl))
            res[f'v{i}'], res[f'g{i}'] = v, g
            res[f'iwe{i}'] = eng.iwes()
            res[f'G{i}'] = eng.image_grad()
    np.savez(out_path, **res)


if __name__ == '__main__':
    main(sys.argv[1])
