"""Does SciPy BFGS stop with status 2 ("precision loss") on the fine pyramid levels because of the objective (piecewise smooth:
`rint` in the warp, zero derivative) or because of the fp32-image evaluation noise of the HIP engine?  The same multi-level
solve is run on the fp64 C/OpenMP oracle (deterministic) and, several times, on the HIP engine.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from functools import partial
import numpy as np
import eincm_amd
from eincm_amd import synth, losses, solver as sol
from oracle import eincm_c_port as CP

H, W, R = 260, 346, 3
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
mag = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
win = synth.make_window(11, (H, W), N, R, flow='constant', flow_mag=mag)
args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
AL, BE = 2000.0, 4000.0
kw = dict(alpha=AL, beta=BE, gamma=0.0, delta=0.0, n_pyr_lvls=5, sensor_size=(H, W), scale_to_sensor_size_method='bilinear')
maxit = {'pyr_lvl_4': 4, 'pyr_lvl_3': 6, 'pyr_lvl_2': 10, 'pyr_lvl_1': 13, 'pyr_lvl_0': 17}


def oracle_vg(theta, xs, ys, ts, edges, edge_ts, cur_pyr_lvl=0, **k):
    v, g = CP.loss_and_grad(theta, xs, ys, ts, edges, edge_ts, AL, BE, (H, W), nthreads=16)
    return (v, {}), g


def run(name, pfunc):
    n_eval = [0]
    def counted(theta, *a, **k):
        n_eval[0] += 1
        return pfunc(theta, *a, **k)
    s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=5, theta_opt_maxiters=maxit, theta_loss_pfunc=counted,
                                     theta_opt_solver_params={'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {}},
                                     handover_settings={'use_handover': False, 'solve_handover_for_levels': [], 'use_downscaled_finest_priors': False,
                                                        'clip_solved_handover': False, 'alpha_handover': 0.0})
    s.set_datasample(*args)
    t0 = time.perf_counter()
    out = s.solve()
    st = out['theta_opt_state_pyr']
    print(f'{name:8s} evals {n_eval[0]:4d} {time.perf_counter()-t0:6.2f}s  ' + '  '.join(f'L{k[-1]}: it {v.iter_num:2d} st {v.status} f {v.fun_val:.5f}' for k, v in st.items()), flush=True)
    return out


print(f'N={N} flow {mag} px, {H}x{W}, R={R}')
o = run('oracle', oracle_vg)
for rep in range(3):
    h = run(f'hip #{rep}', partial(losses.value_and_grad_loss_func, **kw))
th_o, th_h = o['final_theta_pyr']['pyr_lvl_0'], h['final_theta_pyr']['pyr_lvl_0']
vo = CP.loss_and_grad(th_o, *args, AL, BE, (H, W), nthreads=16, want_grad=False)[0]
vh = CP.loss_and_grad(th_h, *args, AL, BE, (H, W), nthreads=16, want_grad=False)[0]
print(f'fp64 loss at the oracle solution {vo:.5f}, at the last hip solution {vh:.5f}; max |theta diff| {np.abs(th_o - th_h).max():.3f} px')
