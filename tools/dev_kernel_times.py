"""Per-kernel HIP-event times of the bench batch (timing='dominant': events attached to the two event kernels) and the wall per step
without any timing, for the library EINCM_LIB points at.  python3 tools/dev_kernel_times.py [h] [B] [N]"""
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
res = {}
for mode in (False, 'dominant', True):
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=mode) as e:
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
        t_end = time.perf_counter() + 0.3
        k = 0
        while time.perf_counter() < t_end:
            e.loss_grad(ths[k % 7], p); k += 1
        if mode: e.timings_total(reset=True)
        n = 100
        t0 = time.perf_counter()
        for k in range(n):
            e.loss_grad(ths[k % 7], p)
        wall = (time.perf_counter() - t0) / n
        if mode:
            acc, cnt = e.timings_total()
            res[mode] = {k: round(v / cnt * 1e3, 1) for k, v in acc.items() if v > 0}
        else:
            res['wall_us'] = round(wall * 1e6, 1)
print(f'{os.environ.get("EINCM_LIB", "product").split("/")[-1]} B={B} N={N} h={h}: wall {res["wall_us"]} us | attached events: {res["dominant"]} | every stage: {res[True]}')
