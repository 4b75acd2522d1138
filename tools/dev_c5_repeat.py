"""Run-to-run repeatability of the capped multi-level solve on the C5 shape (float atomics make evaluations differ in the last
bits; the objective is piecewise smooth, so capped BFGS trajectories can diverge).  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import numpy as np
import eincm_amd
from eincm_amd import engine, synth, losses, solver as sol
H, W, N, R = 480, 640, 10_000_000, 3
win = synth.make_window(11, (H, W), N, R, flow='constant', flow_mag=4.0)
args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
kw = dict(alpha=2000.0, beta=4000.0, gamma=0.0, delta=0.0, n_pyr_lvls=5, sensor_size=(H, W), scale_to_sensor_size_method='bilinear')
maxit = {'pyr_lvl_4': 4, 'pyr_lvl_3': 6, 'pyr_lvl_2': 10, 'pyr_lvl_1': 13, 'pyr_lvl_0': 17}
for rep in range(4):
    n_eval = [0]
    def counted(theta, *a, **k):
        n_eval[0] += 1
        return losses.value_and_grad_loss_func(theta, *a, **k)
    s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=5, theta_opt_maxiters=maxit, theta_loss_pfunc=partial(counted, **kw),
                                     theta_opt_solver_params={'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {}},
                                     handover_settings={'use_handover': False, 'solve_handover_for_levels': [], 'use_downscaled_finest_priors': False,
                                                        'clip_solved_handover': False, 'alpha_handover': 0.0})
    s.set_datasample(*args)
    out = s.solve()
    st = out['theta_opt_state_pyr']
    print(f'rep {rep}: evals {n_eval[0]}  ' + '  '.join(f'{k[-1]}: it {v.iter_num} st {v.status} f {v.fun_val:.4f}' for k, v in st.items()), flush=True)
