import sys, os, time, cProfile, pstats; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import numpy as np, eincm_amd
from eincm_amd import engine, synth, losses, solver as sol
H, W = 256, 336
win = synth.make_window(900, (H, W), 30000, 5, flow='constant', flow_mag=4.0)
args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
kw = dict(alpha=20., beta=35., gamma=2.5e-4, delta=0., n_pyr_lvls=5, sensor_size=(H, W), scale_to_sensor_size_method='bilinear')
th = np.zeros((16, 16, 2)) + 0.5
eng = losses.engine_for(*args, (H, W))
p = engine.make_params(20., 35., 2.5e-4, 0., 0)
def timeit(f, n=300):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print('raw eng.loss_grad            : %.0f us' % timeit(lambda: eng.loss_grad(th, p)))
print('eng.loss_grad want_aux       : %.0f us' % timeit(lambda: eng.loss_grad(th, p, want_aux=True)))
print('losses.value_and_grad_loss_func: %.0f us' % timeit(lambda: losses.value_and_grad_loss_func(th, *args, cur_pyr_lvl=0, **kw)))
s = sol.ScipyMinimize(fun=partial(losses.value_and_grad_loss_func, cur_pyr_lvl=0, **kw), method='BFGS', maxiter=40, has_aux=True, options={'gtol': 1e-7})
t0 = time.perf_counter(); th_opt, st = s.run(th, *args); dt = time.perf_counter() - t0
print('BFGS 16x16: %d iters, %d evals, %.1f ms -> %.0f us per evaluation' % (st.iter_num, st.num_fun_eval, dt * 1e3, dt / st.num_fun_eval * 1e6))
pr = cProfile.Profile(); pr.enable(); s.run(th, *args); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
