"""Dev helper (GPU box): HIP engine vs oracle on a few shapes; prints errors and per-stage timings."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eincm_amd
from eincm_amd import engine, synth
from oracle import eincm_oracle as O

def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)

cases = [
    # H, W, N, R, theta(h,w), flow, mag, gamma, lvl, ck
    (60, 80, 20000, 3, (1, 1), 'constant', 8.0, 0.0, 4, 0),
    (60, 80, 20000, 3, (4, 4), 'smooth', 8.0, 2.5e-4, 0, 0),
    (60, 80, 20000, 3, (60, 80), 'smooth', 8.0, 2.5e-4, 0, 0),
    (180, 240, 10000, 1, (1, 1), 'constant', 20.0, 0.0, 4, 1),
    (260, 346, 100000, 5, (1, 1), 'constant', 20.0, 0.0, 4, 0),
    (260, 346, 100000, 5, (16, 16), 'smooth', 20.0, 2.5e-4, 0, 0),
    (260, 346, 100000, 5, (2, 2), 'smooth', 60.0, 0.0, 3, 0),   # big flow: wrap / drop
]
for (H, W, N, R, (h, w), flow, mag, gamma, lvl, ck) in cases:
    win = synth.make_window(1, (H, W), N, R, flow=flow, flow_mag=mag)
    if (h, w) == (H, W):
        th = win['flow_gt'] * np.random.default_rng(3).uniform(0.5, 1.5, (H, W, 2))
    else:
        th = synth.theta_near_truth(1, win, (h, w))
    args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
    t0 = time.time()
    v_ref, g_ref, aux = O.loss_and_grad(th, *args, 20., 35., gamma, 0.0, lvl, 5, (H, W), 'bilinear', contrast_kind=ck, return_intermediates=True)
    t_or = time.time() - t0
    with engine.Engine((H, W), N, max_refs=R, timing=True) as eng:
        t0 = time.time(); eng.set_window(*args); t_set = time.time() - t0
        p = engine.make_params(20., 35., gamma, 0.0, lvl, contrast_kind=ck)
        v, g, a = eng.loss_grad(th, p, want_aux=True)
        iw = eng.iwes()[0]; G = eng.image_grad()[0]; z = eng.zero_iwe()[0]
        ts = []
        for _ in range(5):
            t0 = time.time(); eng.loss_grad(th, p); ts.append(time.time() - t0)
        tm = eng.timings()
    print(f'H{H} W{W} N{N} R{R} th{(h,w)} {flow}{mag} g{gamma} lvl{lvl} ck{ck}: '
          f'val {v[0]:.8f} ref {v_ref:.8f} rel {abs(v[0]-v_ref)/abs(v_ref):.2e} | grad rel {rel(g[0], g_ref):.2e} '
          f'| iwe rel {rel(iw, aux["_iwes"]):.2e} zero {rel(z, aux["_zero_iwe"]):.2e} G rel {rel(G, aux["_G"]):.2e} '
          f'| oracle {t_or*1e3:.0f} ms set {t_set*1e3:.1f} ms eval wall {min(ts)*1e6:.0f} us')
    print('   stages(us):', {k: round(v_ * 1e3, 1) for k, v_ in tm.items() if v_ > 0})
