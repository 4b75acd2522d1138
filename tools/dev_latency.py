import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, R = 260, 346, 5
for N in (100_000, 1_000_000):
    win = synth.make_window(0, (H, W), N, R, flow='constant', flow_mag=20.0)
    for hw in ((1, 1), (16, 16)):
        th = synth.theta_near_truth(0, win, hw)
        p = engine.make_params(20., 35., 0., 0., 4 if hw == (1, 1) else 1)
        for timing in (False, True):
            with engine.Engine((H, W), N, max_refs=R, timing=timing) as e:
                e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
                for k in range(5): e.loss_grad(th * (1 + .01 * k), p)
                ts = []
                for k in range(40):
                    t0 = time.perf_counter(); e.loss_grad(th * (1 + .01 * (k % 5)), p); ts.append(time.perf_counter() - t0)
                msg = f'N={N} theta={hw} timing={timing}: wall median {np.median(ts)*1e6:.0f} us min {min(ts)*1e6:.0f} us'
                if timing:
                    msg += ' | ' + ' '.join(f'{k}={v*1e3:.0f}' for k, v in e.timings().items() if v > 0)
                print(msg)
