#!/usr/bin/env python3
"""cProfile of ONE window's pyramid solve by the sequential driver (solver.MultipleLevelEINCMSolver, bfgs_update='rank2') on a bench
window: where the host time of the reference's own call pattern goes.  python tools/dev_seq_profile.py [scipy|rank2]"""
import cProfile, importlib, io, os, pstats, sys, time
from functools import partial
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import eincm_amd
from eincm_amd import synth, solver as sol, losses
upd = sys.argv[1] if len(sys.argv) > 1 else 'rank2'
H, W, N, R, n_lvls = 260, 346, 1_000_000, 5, 5
w = synth.make_window(0, (H, W), N, R, flow='constant', flow_mag=20.0)
args = (w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts'])
loss = dict(alpha=20.0, beta=35.0, gamma=0.0, delta=0.0, scale_to_sensor_size_method='bilinear')
maxit = sol.growing_maxiters(n_lvls, 8, 40)
sp = {'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {'pyr_lvl_0': 1, 'pyr_lvl_1': 1}, 'bfgs_update': upd}
losses.engine_for(*args, (H, W))
for rep in range(2):
    s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=n_lvls, theta_opt_maxiters=maxit, theta_loss_pfunc=partial(losses.value_and_grad_loss_func, n_pyr_lvls=n_lvls, sensor_size=(H, W), **loss),
                                     theta_opt_solver_params=sp, pyramid_bases=[2] * (n_lvls - 1))
    s.set_datasample(*args)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep: pr.enable()
    s.solve()
    if rep: pr.disable()
    print(f'rep {rep}: {time.perf_counter() - t0:.3f} s')
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(30); print(st.getvalue())
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(18); print(st.getvalue())
