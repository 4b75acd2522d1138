#!/usr/bin/env python3
"""dev: the 8-window bench batch as n contexts x (8/n) windows, each context driven by its own free-running host thread.
usage: dev_ctx.py lib.so"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
L.LIB_PATH = os.path.abspath(sys.argv[1])
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
z = np.load('/tmp/dev_ab_wins.npz'); B = 8
wins = [{k: z[f'{k}{b}'] for k in ('xs', 'ys', 'ts', 'edges', 'edge_ts', 'th')} for b in range(B)]
th = np.stack([w['th'] for w in wins])
p = engine.make_params(20., 35., 0., 0., 4)
H, W, N, R = 260, 346, 1000000, 5
for n_ctx in (1, 2, 4, 8):
    per = B // n_ctx
    engs = []
    for i in range(n_ctx):
        e = engine.Engine((H, W), per * N, max_refs=R, max_windows=per)
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins[i * per:(i + 1) * per]])
        engs.append(e)
    bar = threading.Barrier(n_ctx + 1)
    n_free = 60
    def drive(i):
        for k in range(10):
            engs[i].loss_grad(th[i * per:(i + 1) * per] * (1 + 0.01 * k), p)
        bar.wait()
        for k in range(n_free):
            engs[i].loss_grad(th[i * per:(i + 1) * per] * (1 + 0.01 * (k % 7)), p)
        bar.wait()
    ts = [threading.Thread(target=drive, args=(i,)) for i in range(n_ctx)]
    for t in ts: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = (time.perf_counter() - t0) / n_free
    for t in ts: t.join()
    for e in engs: e.close()
    print(os.path.basename(sys.argv[1]), 'contexts', n_ctx, 'windows each', per, 'ms per 8-window step %.4f' % (dt * 1e3), flush=True)
