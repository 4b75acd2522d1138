"""Host phases of the bench batch (8 x [260x346, 1e6 events, R = 5], 2-DoF theta unless argv[1] = h): wall per step against the time the
calling thread spends in each phase of eincm_loss_grad (eincm_get_host_profile).  python3 tools/dev_batch_profile.py [h] [B] [N]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
h = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
H, W, R = 260, 346, 5
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
base = np.stack([synth.theta_near_truth(b, w, (h, h)) for b, w in enumerate(wins)])
ths = [np.ascontiguousarray(base * (1.0 + 0.01 * ((k % 7) - 3))) for k in range(7)]
p = engine.make_params(20., 35., 0., 0., 4 if h == 1 else 1)
with engine.Engine((H, W), B * N, max_refs=R, max_windows=B) as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    t_end = time.perf_counter() + 0.3
    k = 0
    while time.perf_counter() < t_end:
        e.loss_grad(ths[k % 7], p); k += 1
    e.host_profile(reset=True)
    n = 200
    t0 = time.perf_counter()
    for k in range(n):
        e.loss_grad(ths[k % 7], p)
    wall = (time.perf_counter() - t0) / n
    hp, cnt = e.host_profile()
    print(f'B={B} N={N} theta=({h},{h}): wall {wall * 1e6:.1f} us per step; host phases us: ' + ', '.join(f'{k} {v / cnt:.1f}' for k, v in hp.items())
          + f'; outside the C call {wall * 1e6 - sum(hp.values()) / cnt:.1f}')
