import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from functools import partial
import numpy as np, eincm_amd
from eincm_amd import losses, solver as sol, synth, engine
from oracle import eincm_oracle as O
H, W = 256, 336
win = synth.make_window(900, (H, W), 30000, 5, flow='constant', flow_mag=4.0)
args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
kw = dict(alpha=20., beta=35., gamma=2.5e-4, delta=0., n_pyr_lvls=5, sensor_size=(H, W), scale_to_sensor_size_method='bilinear')
def oracle_vg(theta, xs, ys, ts, edges, edge_ts, cur_pyr_lvl):
    v, g, aux = O.loss_and_grad(theta, xs, ys, ts, edges, edge_ts, 20., 35., 2.5e-4, 0., cur_pyr_lvl, 5, (H, W))
    return (v, aux), g
hip_vg = partial(losses.value_and_grad_loss_func, **kw)
# repeatability of the HIP objective at a fixed theta
eng = losses.engine_for(*args, (H, W))
th = np.array([[[-2.0, -1.0]]])
p = engine.make_params(20., 35., 2.5e-4, 0., 4)
vals = [eng.loss_grad(th, p)[0][0] for _ in range(8)]
print('repeatability: max-min =', max(vals) - min(vals), 'value', vals[0])
for lvl, hw, maxiter in ((4, (1, 1), 8), (3, (2, 2), 11), (2, (4, 4), 19)):
    for name, f in (('oracle', oracle_vg), ('hip', hip_vg)):
        n = [0]
        def cnt(theta, *a, _f=f, **k):
            n[0] += 1; return _f(theta, *a, **k)
        s = sol.ScipyMinimize(fun=partial(cnt, cur_pyr_lvl=lvl), method='BFGS', maxiter=maxiter, has_aux=True, options={'gtol': 1e-7})
        t0 = time.perf_counter()
        th_opt, st = s.run(np.zeros(hw + (2,)), *args)
        print(f'lvl {lvl} {name:6s}: status {st.status} success {st.success} iters {st.iter_num} evals {n[0]} loss {st.fun_val:.8f} '
              f'theta[0,0] {th_opt[0,0].round(4)} time {time.perf_counter()-t0:.2f}s')
