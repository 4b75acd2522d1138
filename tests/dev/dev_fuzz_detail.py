"""One case of fuzz_gpu.py, window by window, with the terms of the loss next to the oracle's (dev script):
python tests/dev/dev_fuzz_detail.py <seed> <index>   (EINCM_FUZZ_TINY as in fuzz_gpu.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_gpu as F
from oracle import eincm_oracle as O
seed0, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for i in range(idx + 1):
    c = F.draw_case(rng)
print(c)
H, W, R, B = c['H'], c['W'], c['R'], c['B']
seed = 1000 * seed0 + idx
rng = np.random.default_rng(seed)
wins, thetas = [], []
for b in range(B):
    n = c['N'][b]
    win = F.synth.make_window(seed + b, (H, W), max(n, 1), R, flow=c['flow'], flow_mag=max(c['mag'], 1e-3))
    for k in ('xs', 'ys', 'ts'):
        win[k] = win[k][:n]
    wins.append(win)
    h, w = c['hw']
    base = win['flow_gt'] if (h, w) == (H, W) else np.broadcast_to(win['flow_gt'].mean(axis=(0, 1)), (h, w, 2))
    thetas.append(base * rng.uniform(0.5, 1.5, (h, w, 2)) + rng.normal(0, 0.5, (h, w, 2)))
thetas = np.stack(thetas)
if c['mag'] == 0.0 and c['flow'] == 'zero':
    thetas[:] = 0.0
args = lambda w: (w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts'])   # noqa: E731
p = F.engine.make_params(c['alpha'], c['beta'], c['gamma'], c['delta'], c['lvl'], c['method'], c['ck'])
with F.engine.Engine((H, W), max(sum(c['N']), 1), max_refs=R, max_windows=B) as eng:
    eng.set_windows([args(w) for w in wins])
    v, g, aux = eng.loss_grad(thetas, p, want_aux=True)
    iwe = eng.iwes() if hasattr(eng, 'iwes') else None
for b in range(B):
    v_ref, g_ref, a = O.loss_and_grad(thetas[b], *args(wins[b]), c['alpha'], c['beta'], c['gamma'], c['delta'], c['lvl'], 5, (H, W),
                                      c['method'], contrast_kind=c['ck'], return_intermediates=True)
    print(f'window {b} N {c["N"][b]}: value {v[b]!r} oracle {v_ref!r}')
    for k in ('mean_rel_corr', 'mean_rel_contrast', 'theta_total_variation', 'mean_rel_iwe_divergence'):
        print(f'    {k:28s} hip {aux[b][k]!r}  oracle {a[k]!r}')
