import sys, os, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, time; sys.path.insert(0, %r)
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, R = 260, 346, 5
for N in (100000, 1000000):
    win = synth.make_window(0, (H, W), N, R, flow="constant", flow_mag=20.0)
    for hw in ((1, 1), (16, 16)):
        th = synth.theta_near_truth(0, win, hw)
        p = engine.make_params(20., 35., 0., 0., 4 if hw == (1, 1) else 1)
        with engine.Engine((H, W), N, max_refs=R) as e:
            e.set_window(win["xs"], win["ys"], win["ts"], win["edges"], win["edge_ts"])
            for k in range(5): e.loss_grad(th * (1 + .01 * k), p)
            ts = []
            for k in range(40):
                t0 = time.perf_counter(); e.loss_grad(th * (1 + .01 * (k %% 5)), p); ts.append(time.perf_counter() - t0)
            print("N=%%d theta=%%s: %%.0f us" %% (N, hw, np.median(ts) * 1e6), end=" | ")
print()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for segs, seg in ((0, 0), (512, 2048), (1024, 2048), (1024, 4096), (2048, 4096), (2048, 8192), (4096, 8192)):
    env = dict(os.environ)
    if seg: env['EINCM_SEG'] = str(seg); env['EINCM_SEG_SPLAT'] = str(segs)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print(f'seg_s {segs} seg_g {seg}:', r.stdout.strip().split('\n')[-1] if r.stdout else r.stderr[-300:])
