#!/bin/bash
cd $GRAFT_REPO_ROOT
cp edge-informed-contrast-maximization_amd/libeincm_hip.so /tmp/base.so
for v in base vF; do
  if [ $v != base ]; then cp tools/variants/libeincm_$v.so edge-informed-contrast-maximization_amd/libeincm_hip.so; touch edge-informed-contrast-maximization_amd/libeincm_hip.so; fi
  python - <<PY
import sys, os; sys.path.insert(0, '.')
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, N, R, B = 260, 346, 1000000, 5, 8
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
th = np.stack([w['flow_gt'][0, 0].reshape(1, 1, 2) for w in wins])
p = engine.make_params(20., 35., 0., 0., 4)
with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing=True) as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    acc = 0
    for k in range(6):
        e.loss_grad(th, p, want_grad=False)
        if k >= 1: acc += e.timings()['splat'] / 5
    print('$v: splat %.0f us' % (acc * 1e3))
PY
done
cp /tmp/base.so edge-informed-contrast-maximization_amd/libeincm_hip.so
