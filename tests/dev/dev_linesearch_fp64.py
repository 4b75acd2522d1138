"""Do BFGS line searches fail on the HIP engine because its images are fp32, or because of the objective itself?
One of the bench's C4 windows (260x346, 1e6 events, R = 5; seed as bench.py), pyramid levels 4 (1x1) and 3 (2x2) from theta = 0:
SciPy BFGS on (a) the HIP engine, (b) the float64 C/OpenMP port of the same objective (oracle/eincm_ref.c).  Then the objective's own
roughness: f(theta + eps d) - f(theta) against the linear prediction eps g.d, in float64, for eps from 1e-8 to 1e-2 - events_to_pdf_frame
truncates the Gaussian to the 3x3 taps around the ROUNDED coordinate (event_utils.py:32-61), so f jumps whenever an event crosses a
half-pixel line; no float precision removes that.           python3 tests/dev/dev_linesearch_fp64.py [window seed]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from functools import partial
import numpy as np, eincm_amd
from eincm_amd import losses, solver as sol, synth
from oracle import eincm_c_port as CP
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W, N, R = 260, 346, 1_000_000, 5
win = synth.make_window(seed, (H, W), N, R, flow='constant', flow_mag=20.0)
args = (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
kw = dict(alpha=20., beta=35., gamma=0., delta=0., n_pyr_lvls=5, sensor_size=(H, W), scale_to_sensor_size_method='bilinear')
def port_vg(theta, xs, ys, ts, edges, edge_ts, cur_pyr_lvl):
    v, g = CP.loss_and_grad(theta, xs, ys, ts, edges, edge_ts, 20., 35., (H, W))
    return (v, {}), g
hip_vg = partial(losses.value_and_grad_loss_func, **kw)
print(f'window seed {seed}: true flow {win["flow_gt"][0, 0].round(3)}')
start = np.zeros((1, 1, 2))
for lvl, hw, maxiter in ((4, (1, 1), 8), (3, (2, 2), 11)):
    nxt = {}
    for name, f in (('hip fp32 images', hip_vg), ('fp64 C port', port_vg)):
        n = [0]
        def cnt(theta, *a, _f=f, **k):
            n[0] += 1; return _f(theta, *a, **k)
        s = sol.ScipyMinimize(fun=partial(cnt, cur_pyr_lvl=lvl), method='BFGS', maxiter=maxiter, has_aux=True, options={'gtol': 1e-7})
        t0 = time.perf_counter()
        x0 = np.repeat(np.repeat(start, hw[0] // start.shape[0], 0), hw[1] // start.shape[1], 1)
        th_opt, st = s.run(x0, *args)
        nxt[name] = th_opt
        print(f'lvl {lvl} {name:16s}: status {st.status} iters {st.iter_num} evals {n[0]} loss {st.fun_val:.8f} theta[0,0] {th_opt[0, 0].round(4)} '
              f'({time.perf_counter() - t0:.1f} s)', flush=True)
    start = nxt['fp64 C port']
# roughness of the float64 objective along the gradient direction at the level-4 end point
th = np.asarray(start[:1, :1], dtype=np.float64)
f0, g0 = CP.loss_and_grad(th, *args, 20., 35., (H, W))
d = -g0 / np.linalg.norm(g0)
slope = float((g0 * d).sum())
print(f'\nfloat64 objective at theta {th[0, 0].round(4)}: f = {f0:.10f}, |g| = {np.linalg.norm(g0):.4e}; along d = -g/|g|:')
print('      eps      f(theta+eps d)-f       eps*g.d        ratio     (HIP engine: same difference)')
eng = losses.engine_for(*args, (H, W))
from eincm_amd import engine as E
p = E.make_params(20., 35., 0., 0., 4)
h0 = eng.loss_grad(th[None], p)[0][0]
for eps in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
    f1, _ = CP.loss_and_grad(th + eps * d, *args, 20., 35., (H, W), want_grad=False)
    h1 = eng.loss_grad((th + eps * d)[None], p)[0][0]
    print(f'  {eps:8.0e}   {f1 - f0:+.6e}   {eps * slope:+.6e}   {(f1 - f0) / (eps * slope):+9.3f}     {h1 - h0:+.6e}', flush=True)
