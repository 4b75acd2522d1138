"""One process, the bench's 8 windows split over 1/2/4 engine contexts, each driven by its own free-running host thread for K
steps (no join between steps) - the way independent per-window solvers would drive the engine.  Run on the GPU box."""
import sys, os, time, threading; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, N, R, B, K = 260, 346, 1_000_000, 5, 8, 200
wins = [synth.make_window(1000 + b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
ths = np.stack([synth.theta_near_truth(1000 + b, w, (1, 1)) for b, w in enumerate(wins)])
p = engine.make_params(20., 35., 0., 0., 4)
def mk(ws):
    e = engine.Engine((H, W), len(ws) * N, max_refs=R, max_windows=len(ws))
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in ws]); return e
for nsplit in (1, 2, 4, 8):
    per = B // nsplit
    engs = [mk(wins[i*per:(i+1)*per]) for i in range(nsplit)]
    bar = threading.Barrier(nsplit + 1)
    def work(i):
        th = ths[i*per:(i+1)*per]
        for k in range(10): engs[i].loss_grad(th * (1 + 0.01 * (k % 5)), p)
        bar.wait()
        for k in range(K): engs[i].loss_grad(th * (1 + 0.01 * (k % 7 - 3)), p)
        bar.wait()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(nsplit)]
    for t in ts: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = (time.perf_counter() - t0) / K
    for t in ts: t.join()
    print(f'{nsplit} free-running contexts x {per} windows: {dt*1e3:.4f} ms per 8-window step  {B*N*R/dt:.4e} warped-ev/s', flush=True)
    for e in engs: e.close()
