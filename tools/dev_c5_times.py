"""Event-kernel times and wall of one evaluation on the C5 shape (480x640, 1e7 events, R = 3, 16x16 theta) for the environment in effect
(EINCM_GATHER_ALL_R, EINCM_SEG_SPLAT, ...).  python3 tools/dev_c5_times.py [h] [gamma] [H W N [B]]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
h = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gamma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
H, W, N, R = 480, 640, 10_000_000, 3
if len(sys.argv) > 5: H, W, N = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ws = [synth.make_window(7 + b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
base = np.stack([synth.theta_near_truth(7 + b, w, (h, h)) for b, w in enumerate(ws)])
ths = [np.ascontiguousarray(base * (1.0 + 0.01 * ((k % 7) - 3))) for k in range(7)]
p = engine.make_params(20., 35., gamma, 0., 4 if h == 1 else (0 if gamma else 1))
res = {}
for mode in (False, 'dominant'):
    with engine.Engine((H, W), N * B, max_refs=R, max_windows=B, timing=mode) as e:
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in ws])
        for k in range(30): e.loss_grad(ths[k % 7], p)
        if mode: e.timings_total(reset=True)
        t0 = time.perf_counter()
        for k in range(40): e.loss_grad(ths[k % 7], p)
        wall = (time.perf_counter() - t0) / 40
        if mode:
            acc, cnt = e.timings_total(); res['k'] = {k: round(v / cnt * 1e3, 1) for k, v in acc.items() if v > 0}
        else:
            res['wall'] = round(wall * 1e6, 1)
print({k: os.environ[k] for k in os.environ if k.startswith('EINCM_')}, f'h={h} {H}x{W} N={N} B={B}: wall {res["wall"]} us, kernels {res["k"]}')
