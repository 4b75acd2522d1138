import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from concurrent.futures import ThreadPoolExecutor
from eincm_amd import engine, synth
H, W, N, R, B = 260, 346, 1_000_000, 5, 8
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
ths = np.stack([synth.theta_near_truth(b, w, (1, 1)) for b, w in enumerate(wins)])
p = engine.make_params(20., 35., 0., 0., 4)
def mk(ws):
    e = engine.Engine((H, W), len(ws) * N, max_refs=R, max_windows=len(ws))
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in ws]); return e
for nsplit in (1, 2, 4):
    per = B // nsplit
    engs = [mk(wins[i*per:(i+1)*per]) for i in range(nsplit)]
    pool = ThreadPoolExecutor(nsplit)
    def step(k):
        futs = [pool.submit(engs[i].loss_grad, ths[i*per:(i+1)*per] * (1 + 0.01 * (k % 5)), p) for i in range(nsplit)]
        return [f.result() for f in futs]
    for k in range(3): step(k)
    t0 = time.perf_counter()
    for k in range(20): step(k)
    dt = (time.perf_counter() - t0) / 20
    print(f'{nsplit} contexts x {per} windows: {dt*1e3:.3f} ms/step  {B*N*R/dt:.3e} warped-ev/s')
    for e in engs: e.close()
