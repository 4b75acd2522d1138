"""Round-3 experiment: k_splat as shipped against its run-merged form (EINCM_SPLAT_MERGE=1:
the gather's pixel-sorted copy of the events, same-destination taps of a thread summed in registers before the LDS atomics).
Prints the splat's HIP-event time and a checksum of the IWE stack (integer accumulation: the two forms must agree bit for bit).
python3 tools/dev_splat_merge.py H W N R h B"""
import sys, os, time, zlib; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W, N, R, h, B = (int(v) for v in sys.argv[1:7])
wins = [synth.make_window(b, (H, W), N, R, flow='constant' if h == 1 else 'smooth', flow_mag=20.0) for b in range(B)]
base = np.stack([synth.theta_near_truth(b, w, (h, h)) for b, w in enumerate(wins)])
p = engine.make_params(20., 35., 0., 0., 4 if h == 1 else 1)
with engine.Engine((H, W), B * N, max_refs=R, max_windows=B, timing='dominant') as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        e.loss_grad(base, p)
    e.timings_total(reset=True)
    for k in range(50):
        e.loss_grad(base * (1.0 + 0.01 * ((k % 7) - 3)), p)
    acc, cnt = e.timings_total()
    v, g, _ = e.loss_grad(base, p)
    iw = e.iwes()
print(f'{"merge" if os.environ.get("EINCM_SPLAT_MERGE") else "plain"} {H}x{W} N={N} R={R} theta={h}x{h} B={B}: k_splat {acc["splat"] / cnt * 1e3:.1f} us, k_gather {acc["gather"] / cnt * 1e3:.1f} us'
      f' | IWE crc {zlib.crc32(iw.tobytes()):08x} loss[0] {v[0]:.12g}')
