#!/usr/bin/env python3
"""A/B timing of library variants on the bench workload (dev tool): per-stage HIP-event times of k_splat / k_gather.
usage: python tools/dev_ab.py tools/variants/libeincm_base.so tools/variants/libeincm_x.so ...   (one process per variant,
interleaved over --rounds rounds so that clock drift hits every variant alike)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, sys, json, os
sys.path.insert(0, %(root)r)
import numpy as np
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
L.LIB_PATH = %(lib)r
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R, B = 260, 346, 1000000, 5, 8
cache = '/tmp/dev_ab_wins.npz'
if os.path.exists(cache):
    z = np.load(cache); wins = [{k: z[f'{k}{b}'] for k in ('xs', 'ys', 'ts', 'edges', 'edge_ts', 'th')} for b in range(B)]
else:
    wins = []
    for b in range(B):
        w = synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0)
        w['th'] = synth.theta_near_truth(b, w, %(theta)r)
        wins.append(w)
    np.savez(cache, **{f'{k}{b}': w[k] for b, w in enumerate(wins) for k in ('xs', 'ys', 'ts', 'edges', 'edge_ts', 'th')})
th = np.stack([w['th'] for w in wins])
p = engine.make_params(20., 35., 0., 0., 4)
with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    import time
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        e.loss_grad(th, p)
    acc = {}
    n = 20
    for k in range(n):
        v, g, _ = e.loss_grad(th * (1.0 + 0.01 * ((k %% 7) - 3)), p)
        for kk, vv in e.timings().items():
            acc[kk] = acc.get(kk, 0.0) + vv / n
    print(json.dumps({'lib': os.path.basename(%(lib)r), 'splat_us': acc['splat'] * 1e3, 'gather_us': acc['gather'] * 1e3,
                      'theta_us': acc['theta'] * 1e3, 'stats_us': acc['stats'] * 1e3, 'imgrad_us': acc['imgrad'] * 1e3,
                      'final_us': acc['final'] * 1e3, 'total_us': acc['total'] * 1e3, 'v0': float(v[0]), 'g0': [float(x) for x in g[0].ravel()[:2]]}))
'''


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    rounds = 2
    theta = (1, 1)
    for a in sys.argv[1:]:
        if a.startswith('--rounds='):
            rounds = int(a.split('=')[1])
        if a.startswith('--theta='):
            theta = tuple(int(v) for v in a.split('=')[1].split('x'))
    for r in range(rounds):
        for lib in args:
            code = CHILD % {'root': ROOT, 'lib': os.path.abspath(lib), 'theta': theta}
            out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
            print(out.stdout.strip() or out.stderr[-400:], flush=True)


if __name__ == '__main__':
    main()
