import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, eincm_amd
from eincm_amd import engine, synth
H, W, R, B = 96, 128, 3, 2
wins = [synth.make_window(90 + b, (H, W), 20000 + 3000 * b, R, flow='smooth', flow_mag=8.0) for b in range(B)]
args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
with engine.Engine((H, W), 50000, max_refs=R, max_windows=B) as eng:
    eng.set_windows(args)
    th = np.stack([synth.theta_near_truth(90 + b, w, (1, 1)) for b, w in enumerate(wins)])
    p = engine.make_params(20.0, 35.0, 0.0, 0.0, 4)
    t = torch.from_numpy(th).cuda()
    out = {}
    for name, bound in (('auto', None), ('exact', float(np.abs(th).max())), ('zero', 0.0), ('huge', 1e4)):
        v, g, _ = eng.loss_grad_device(t, p, theta_abs_max=bound)
        out[name] = (v.copy(), g.cpu().numpy().copy(), eng.iwes().copy(), eng.image_grad().copy())
    a = out['auto']
    for name in ('exact', 'zero', 'huge'):
        b = out[name]
        print(name, 'v', np.abs(b[0] - a[0]).max(), 'g', np.abs(b[1] - a[1]).max(), 'iwe mismatches', int((b[2] != a[2]).sum()), 'max', np.abs(b[2] - a[2]).max(),
              'G mismatches', int((b[3] != a[3]).sum()))
    # host boundary for reference
    v, g, _ = eng.loss_grad(th, p)
    print('host', np.abs(v - a[0]).max(), np.abs(g - a[1]).max(), int((eng.iwes() != a[2]).sum()))
