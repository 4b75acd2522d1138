"""How far apart do the lockstep batch solver and sequential SciPy solves end, level by level?  (8 x [260x346, N events], pyramid 1..16.)
python3 tools/dev_c4_agree.py [N] [exact]    exact = 1: SciPy's own n^3 expression of the BFGS update at every size"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import numpy as np, eincm_amd
from eincm_amd import engine, synth, solver as sol, batch_solver as bsol, losses
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
if len(sys.argv) > 2 and sys.argv[2] == '1':
    bsol._EXACT_UPDATE_MAX_N = 10 ** 9
H, W, R, B, n_lvls = 260, 346, 5, 4, 5
wins = [synth.make_window(1000 + b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
loss = dict(alpha=20.0, beta=35.0, gamma=0.0, delta=0.0, scale_to_sensor_size_method='bilinear')
maxit = sol.growing_maxiters(n_lvls, 8, 40)
sp = {'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {'pyr_lvl_0': 1, 'pyr_lvl_1': 1}}
bs = bsol.BatchedMultipleLevelEINCMSolver(B, (H, W), n_lvls, maxit, loss, sp, pyramid_bases=[2] * (n_lvls - 1))
bs.set_datasamples(args)
t0 = time.perf_counter(); ob = bs.solve(); tb = time.perf_counter() - t0
bs.close()
for b in range(B):
    s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=n_lvls, theta_opt_maxiters=maxit, theta_loss_pfunc=partial(losses.value_and_grad_loss_func, n_pyr_lvls=n_lvls, sensor_size=(H, W), **loss),
                                     theta_opt_solver_params=sp, pyramid_bases=[2] * (n_lvls - 1))
    s.set_datasample(*args[b])
    t0 = time.perf_counter(); o = s.solve(); ts = time.perf_counter() - t0
    line = []
    for k in (4, 3, 2, 1, 0):
        key = f'pyr_lvl_{k}'
        a, c = o['theta_opt_state_pyr'][key], ob[b]['theta_opt_state_pyr'][key]
        line.append(f'L{k}: seq {a.fun_val:.4f} (it {a.iter_num}, st {a.status}) bat {c.fun_val:.4f} (it {c.iter_num}, st {c.status})')
    print(f'window {b} [seq {ts:.2f} s]: ' + ' | '.join(line))
print(f'batched {tb:.2f} s for {B} windows')
