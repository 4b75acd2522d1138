"""Level by level: the lockstep batch solve against the sequential driver (SciPy's BFGS update, and the rank-two form) on the bench's
C4 windows (8 x [260x346, 1e6 events, R = 5], pyramid 1..16).  python3 tools/dev_c4_levels.py [n_seq]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import numpy as np, eincm_amd
from eincm_amd import engine, synth, solver as sol, batch_solver as bsol, losses
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fallback = not (len(sys.argv) > 2 and sys.argv[2] == 'nofallback')
H, W, R, B, n_lvls, N = 260, 346, 5, 8, 5, 1_000_000
wins = [synth.make_window(b, (H, W), N, R, flow='constant', flow_mag=20.0) for b in range(B)]
args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
loss = dict(alpha=20.0, beta=35.0, gamma=0.0, delta=0.0, scale_to_sensor_size_method='bilinear')
maxit = sol.growing_maxiters(n_lvls, 8, 40)
sp = {'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {'pyr_lvl_0': 1, 'pyr_lvl_1': 1}, 'wolfe2_fallback': fallback}
bs = bsol.BatchedMultipleLevelEINCMSolver(B, (H, W), n_lvls, maxit, loss, sp, pyramid_bases=[2] * (n_lvls - 1))
bs.set_datasamples(args)
t0 = time.perf_counter(); ob = bs.solve(); tb = time.perf_counter() - t0
bs.close()
def fmt(s):
    return f'{s.fun_val:9.4f} it {s.iter_num:2d} st {s.status} nf {s.num_fun_eval:3d}'
for b in range(n_seq):
    outs = {}
    for upd in ('scipy', 'rank2'):
        s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=n_lvls, theta_opt_maxiters=maxit, theta_loss_pfunc=partial(losses.value_and_grad_loss_func, n_pyr_lvls=n_lvls, sensor_size=(H, W), **loss),
                                         theta_opt_solver_params={**sp, 'bfgs_update': upd}, pyramid_bases=[2] * (n_lvls - 1))
        s.set_datasample(*args[b])
        t0 = time.perf_counter(); outs[upd] = s.solve(); outs[upd + '_t'] = time.perf_counter() - t0
    print(f'window {b}: true flow {wins[b]["flow_gt"][0, 0]}  seq scipy {outs["scipy_t"]:.2f} s, seq rank2 {outs["rank2_t"]:.2f} s')
    for k in (4, 3, 2, 1, 0):
        key = f'pyr_lvl_{k}'
        print(f'   L{k}: scipy {fmt(outs["scipy"]["theta_opt_state_pyr"][key])} | rank2 {fmt(outs["rank2"]["theta_opt_state_pyr"][key])} | batched {fmt(ob[b]["theta_opt_state_pyr"][key])}'
              f' | mean theta scipy {outs["scipy"]["final_theta_pyr"][key].reshape(-1, 2).mean(0).round(3)} batched {ob[b]["final_theta_pyr"][key].reshape(-1, 2).mean(0).round(3)}')
print(f'batched {tb:.3f} s for {B} windows, {bs.n_batch_evals} engine calls; level-0 losses {[round(float(o["theta_opt_state_pyr"]["pyr_lvl_0"].fun_val), 3) for o in ob]}')
