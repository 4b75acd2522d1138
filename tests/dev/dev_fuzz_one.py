#!/usr/bin/env python3
"""dev: run one fuzz case (index in the test's sequence) with a given library and print where the gradient differs."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'dev'))
import numpy as np
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
L.LIB_PATH = os.path.abspath(sys.argv[1])
import fuzz_gpu
from oracle import eincm_oracle as O
idx = int(sys.argv[2])
rng = np.random.default_rng(2)
for i in range(idx + 1):
    c = fuzz_gpu.draw_case(rng)
print(c)
seed = 2000 + idx
H, W, R, B = c['H'], c['W'], c['R'], c['B']
rng = np.random.default_rng(seed)
synth, engine = fuzz_gpu.synth, fuzz_gpu.engine
wins, thetas = [], []
for b in range(B):
    n = c['N'][b]
    win = synth.make_window(seed + b, (H, W), max(n, 1), R, flow=c['flow'], flow_mag=max(c['mag'], 1e-3))
    for k in ('xs', 'ys', 'ts'):
        win[k] = win[k][:n]
    wins.append(win)
    h, w = c['hw']
    base = win['flow_gt'] if (h, w) == (H, W) else np.broadcast_to(win['flow_gt'].mean(axis=(0, 1)), (h, w, 2))
    thetas.append(base * rng.uniform(0.5, 1.5, (h, w, 2)) + rng.normal(0, 0.5, (h, w, 2)))
thetas = np.stack(thetas)
args = lambda w: (w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts'])
with engine.Engine((H, W), max(sum(c['N']), 1), max_refs=R, max_windows=B) as eng:
    eng.set_windows([args(w) for w in wins])
    v, g, _ = eng.loss_grad(thetas, engine.make_params(c['alpha'], c['beta'], c['gamma'], c['delta'], c['lvl'], c['method'], c['ck']))
    G = eng.image_grad()
for b in range(B):
    v_ref, g_ref, aux = O.loss_and_grad(thetas[b], *args(wins[b]), c['alpha'], c['beta'], c['gamma'], c['delta'], c['lvl'], 5,
                                        (H, W), c['method'], contrast_kind=c['ck'], return_intermediates=True)
    d = np.abs(g[b] - g_ref)
    print('window', b, 'events', wins[b]['xs'], wins[b]['ys'], wins[b]['ts'], 'v', v[b], v_ref, 'max|g_ref|', np.abs(g_ref).max(), 'max diff', d.max(), 'at', np.unravel_index(d.argmax(), d.shape))
    print('G err', np.abs(G[b] - aux['_G']).max() / np.abs(aux['_G']).max(), 'max|G|', np.abs(aux['_G']).max())
    nz = np.argwhere(np.abs(g_ref) > 1e-3 * np.abs(g_ref).max())
    for (i, j, k) in nz[:12]:
        print((i, j, k), g[b][i, j, k], g_ref[i, j, k])
