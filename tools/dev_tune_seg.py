#!/usr/bin/env python3
"""Segment-length tuning over the workload shapes (dev tool): per (EINCM_SEG_SPLAT, EINCM_SEG) the HIP-event times of
k_splat / k_gather and the median wall latency of one loss+grad.  One child process per setting (the knobs are read at create)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
CFG = %(cfg)r
out = []
for (H, W, N, R, hw, B) in CFG:
    dense = hw == 'dense'
    wins = [synth.make_window(b, (H, W), N, R, flow='smooth' if dense else 'constant', flow_mag=20.0) for b in range(B)]
    th = np.stack([w['flow_gt'] if dense else synth.theta_near_truth(b, w, hw) for b, w in enumerate(wins)])
    p = engine.make_params(20., 35., 0., 0., 0 if dense else (4 if hw == (1, 1) else 1))
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        sp = ga = 0.0
        n = 10
        for k in range(n):
            e.loss_grad(th * (1 + 0.01 * (k %% 5)), p)
            t = e.timings(); sp += t['splat'] * 1e3 / n; ga += t['gather'] * 1e3 / n
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B) as e:
        e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        ts = []
        for k in range(30):
            thk = th * (1 + 0.01 * (k %% 5))
            t0 = time.perf_counter(); e.loss_grad(thk, p); ts.append(time.perf_counter() - t0)
    out.append('%%dx%%d N=%%g R=%%d th=%%s B=%%d: splat %%.1f gather %%.1f wall %%.1f us' %% (H, W, N, R, hw, B, sp, ga, np.median(ts) * 1e6))
print(' | '.join(out))
'''
CFG = [(260, 346, 100000, 5, (1, 1), 1), (260, 346, 1000000, 5, (1, 1), 1), (260, 346, 1000000, 5, (16, 16), 1),
       (260, 346, 1000000, 5, (1, 1), 2), (480, 640, 1000000, 3, 'dense', 1), (480, 640, 10000000, 3, (16, 16), 1)]


def main():
    settings = [(0, 0)] + [(s, g) for s in (4096, 8192) for g in (4096, 8192, 16384)]
    for segs, seg in settings:
        env = dict(os.environ)
        if seg:
            env['EINCM_SEG'] = str(seg); env['EINCM_SEG_SPLAT'] = str(segs)
        r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'cfg': CFG}], env=env, capture_output=True, text=True)
        print(f'seg_s {segs} seg {seg}: ' + (r.stdout.strip().split('\n')[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)


if __name__ == '__main__':
    main()
