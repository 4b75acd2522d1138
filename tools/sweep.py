"""Measurement sweep of SURVEY 8(d): single-window loss+grad latency and throughput over N, R, theta shape, plus the C3
(dense theta) and C5 (10^7 events, pyramid + BFGS end to end) configurations.  Prints a markdown table.
Run on the GPU box:  python tools/sweep.py > gpurun_out/sweep.md"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import numpy as np
import eincm_amd
from eincm_amd import engine, synth, losses, solver as sol

PEAK = 8000.0


def b_alg(N, R, H, W, dense):
    return 2 * 8 * N + R * H * W * 4 * 5 + (H * W * 2 * 4 * 2 if dense else 0)


def measure(H, W, N, R, hw, gamma=0.0, lvl=4, alpha=20.0, beta=35.0, n_rep=30, flow='constant'):
    win = synth.make_window(7, (H, W), N, R, flow=flow, flow_mag=20.0)
    dense = hw == 'dense'
    th = win['flow_gt'] * 0.9 if dense else synth.theta_near_truth(7, win, hw)
    p = engine.make_params(alpha, beta, gamma, 0.0, lvl)
    with engine.Engine((H, W), N, max_refs=R) as e:
        t0 = time.perf_counter()
        e.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
        t_set = time.perf_counter() - t0
        ths = [th * (1 + 0.01 * k) for k in range(5)]          # theta changes every call; building it is the caller's business, not
        t_spin = time.perf_counter() + 0.2                     # the engine's (a fresh 4.9 MB numpy temporary costs 0.65 ms in page faults);
        k = 0                                                  # 0.2 s of evaluations first: a cold GPU runs its first ~100 ms some 8 % slower
        while time.perf_counter() < t_spin or k < 4:
            e.loss_grad(ths[k % 5], p); k += 1
        ts = []
        for k in range(n_rep):
            t0 = time.perf_counter(); e.loss_grad(ths[k % 5], p); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    gbps = b_alg(N, R, H, W, dense) / t / 1e9
    return t, t_set, N * R / t, gbps


print('| config | H×W | N | R | theta | eval ms (median) | warped-ev/s | B_alg GB/s | % of 8 TB/s | set_window ms |')
print('|---|---|---|---|---|---|---|---|---|---|')
rows = [('C1 (variance is a flag; grad-mag timed)', 180, 240, 10_000, 1, (1, 1)),
        ('C2', 260, 346, 100_000, 5, (1, 1)), ('C2 R=1', 260, 346, 100_000, 1, (1, 1)), ('C2 pyr16', 260, 346, 100_000, 5, (16, 16)),
        ('headline', 260, 346, 1_000_000, 5, (1, 1)), ('headline R=1', 260, 346, 1_000_000, 1, (1, 1)),
        ('headline pyr16', 260, 346, 1_000_000, 5, (16, 16)),
        ('sweep 1e7', 260, 346, 10_000_000, 5, (1, 1)), ('sweep 1e7 R=1', 260, 346, 10_000_000, 1, (1, 1)),
        ('MVSEC real size', 256, 336, 30_000, 5, (16, 16)), ('DSEC real size', 480, 640, 1_500_000, 3, (16, 16)),
        ('C3 dense', 480, 640, 1_000_000, 3, 'dense'), ('C5 shape', 480, 640, 10_000_000, 3, (16, 16))]
for name, H, W, N, R, hw in rows:
    dense = hw == 'dense'
    t, t_set, evs, gbps = measure(H, W, N, R, hw, gamma=2.5e-4 if (dense or hw == (16, 16)) else 0.0, lvl=0 if (dense or hw == (16, 16)) else 4,
                                  flow='smooth' if dense else 'constant', n_rep=30 if N <= 1_000_000 else 12)
    print(f'| {name} | {H}×{W} | {N:.0e} | {R} | {hw} | {t*1e3:.3f} | {evs:.3e} | {gbps:.0f} | {100*gbps/PEAK:.2f} | {t_set*1e3:.1f} |', flush=True)

# ---- C5: 480x640, 1e7 events, theta pyramid 1..16, ~50 BFGS iterations end to end on ONE GPU ----
H, W, N, R = 480, 640, 10_000_000, 3
win = synth.make_window(11, (H, W), N, R, flow='constant', flow_mag=4.0)
args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
kw = dict(alpha=2000.0, beta=4000.0, gamma=0.0, delta=0.0, n_pyr_lvls=5, sensor_size=(H, W), scale_to_sensor_size_method='bilinear')
maxit = {'pyr_lvl_4': 4, 'pyr_lvl_3': 6, 'pyr_lvl_2': 10, 'pyr_lvl_1': 13, 'pyr_lvl_0': 17}      # 50 iterations in total
n_eval = [0]
def counted(theta, *a, **k):
    n_eval[0] += 1
    return losses.value_and_grad_loss_func(theta, *a, **k)
s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=5, theta_opt_maxiters=maxit, theta_loss_pfunc=partial(counted, **kw),
                                 theta_opt_solver_params={'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {}},
                                 handover_settings={'use_handover': False, 'solve_handover_for_levels': [], 'use_downscaled_finest_priors': False,
                                                    'clip_solved_handover': False, 'alpha_handover': 0.0})
s.set_datasample(*args)
t0 = time.perf_counter()
losses.engine_for(*args, (H, W))              # staging (host binning + upload + theta = 0 constants)
t_stage = time.perf_counter() - t0
t0 = time.perf_counter()
out = s.solve()
t_solve = time.perf_counter() - t0
its = sum(st.iter_num for st in out['theta_opt_state_pyr'].values())
print()
print(f'C5 end to end on 1 GPU: 480×640, N=1e7, R=3, pyramid 1→16, {its} BFGS iterations, {n_eval[0]} loss+grad evaluations: '
      f'solve {t_solve:.3f} s ({t_solve/max(n_eval[0],1)*1e3:.2f} ms per evaluation incl. SciPy), staging {t_stage:.2f} s; '
      f'final loss {out["theta_opt_state_pyr"]["pyr_lvl_0"].fun_val:.4f}, coarse theta {out["final_theta_pyr"]["pyr_lvl_4"][0,0].round(3)} '
      f'(true flow {win["flow_gt"][0,0].round(3)})')
losses.clear_engine_cache()
