#!/usr/bin/env python3
"""Experiment (dev tool): does k_splat speed up when the events a wavefront handles together come from distinct source pixels?
The bench batch is staged twice: as delivered (time order inside each tile), and with every block of BLK consecutive events of a tile
re-dealt so that events of one source pixel land in different 64-event groups.  Prints the HIP-event times of the event kernels."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
H, W, N, R, B = 260, 346, 1000000, 5, int(os.environ.get('B', '8'))
HW = tuple(int(v) for v in os.environ.get('THETA', '1x1').split('x'))
TS = 32


def spread(xs, ys, ts, blk):
    tx = (W + TS - 1) // TS
    tile = (ys.astype(np.int64) // TS) * tx + (xs.astype(np.int64) // TS)
    order = np.argsort(tile, kind='stable')                 # what the engine's binning does
    tile_s = tile[order]
    starts = np.flatnonzero(np.r_[True, tile_s[1:] != tile_s[:-1]])
    ends = np.r_[starts[1:], len(order)]
    out = order.copy()
    g = blk // 64
    for s, e in zip(starts, ends):
        for b0 in range(s, e, blk):
            b1 = min(b0 + blk, e)
            idx = order[b0:b1]
            pix = ys[idx].astype(np.int64) * W + xs[idx]
            srt = idx[np.argsort(pix, kind='stable')]
            n = len(srt)
            # deal sorted position k to group k % g, slot k // g
            k = np.arange(n)
            dest = (k % g) * 64 + (k // g)
            if n == blk:
                o = np.empty(n, dtype=idx.dtype); o[dest] = srt
            else:
                o = srt[np.argsort(dest, kind='stable')]
            out[b0:b1] = o
    return xs[out], ys[out], ts[out]


def run(wins, th, label):
    p = engine.make_params(20., 35., 0., 0., 4 if HW == (1, 1) else 1)
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
        e.set_windows(wins)
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            e.loss_grad(th, p)
        acc = {}
        n = 20
        for k in range(n):
            v, g, _ = e.loss_grad(th * (1.0 + 0.01 * ((k % 7) - 3)), p)
            for kk, vv in e.timings().items():
                acc[kk] = acc.get(kk, 0.0) + vv / n
    print('%-14s splat %.1f us  gather %.1f us  total %.1f us  v0 %.9f' % (label, acc['splat'] * 1e3, acc['gather'] * 1e3, acc['total'] * 1e3, v[0]), flush=True)


def main():
    raw = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
    th = np.stack([synth.theta_near_truth(b, w, HW) for b, w in enumerate(raw)])
    run([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in raw], th, 'time order')
    for blk in (256, 1024):
        wins = []
        for w in raw:
            xs, ys, ts = spread(w['xs'], w['ys'], w['ts'], blk)
            wins.append((xs, ys, ts, w['edges'], w['edge_ts']))
        run(wins, th, 'spread %d' % blk)
    run([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in raw], th * 0.0, 'time, theta=0')


if __name__ == '__main__':
    main()
