"""set_window in a loop at a small real-data size (256x336, 3*10^4 events, R = 5), for a kernel + memory-copy trace and a wall time.
python3 tools/dev_staging_trace.py [N] [reps]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H, W, R = 256, 336, 5
wins = [synth.make_window(k, (H, W), N, R, flow='smooth', flow_mag=8.0) for k in range(4)]
p = engine.make_params(20., 35., 0., 0., 4)
with engine.Engine((H, W), N, max_refs=R) as e:
    for k in range(5):
        w = wins[k % 4]; e.set_window(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts'])
    ts = []
    for k in range(reps):
        w = wins[k % 4]
        t0 = time.perf_counter(); e.set_window(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']); ts.append(time.perf_counter() - t0)
    print(f'N={N}: set_window median {np.median(ts)*1e6:.0f} us, min {min(ts)*1e6:.0f} us')
