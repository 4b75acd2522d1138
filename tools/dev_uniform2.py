import importlib, os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R, B = 260, 346, 1000000, 5, 8
raw = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
th = np.stack([synth.theta_near_truth(b, w, (1, 1)) for b, w in enumerate(raw)])
p = engine.make_params(20., 35., 0., 0., 4)
def run(wins, th, label):
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
        e.set_windows(wins)
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        acc = {}; n = 20
        for k in range(n):
            e.loss_grad(th * (1.0 + 0.01 * ((k % 7) - 3)), p)
            for kk, vv in e.timings().items():
                acc[kk] = acc.get(kk, 0.0) + vv / n
    print('%-28s splat %.1f us  gather %.1f us' % (label, acc['splat'] * 1e3, acc['gather'] * 1e3), flush=True)
run([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in raw], th, 'edge events, theta ~ truth')
run([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in raw], th * 3.0, 'edge events, theta x 3')
rng = np.random.default_rng(0)
uni = [(rng.integers(0, W, N).astype(np.int16), rng.integers(0, H, N).astype(np.int16), w['ts'], w['edges'], w['edge_ts']) for w in raw]
run(uni, th, 'uniform random events')
