import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, eincm_amd
from eincm_amd import engine, synth
H, W = 256, 336
for mag in (4.0, 12.0):
    win = synth.make_window(3, (H, W), 30000, 5, flow='constant', flow_mag=mag)
    v = win['flow_gt'][0, 0]
    with engine.Engine((H, W), 30000, max_refs=5) as eng:
        eng.set_window(win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])
        for (al, be) in ((20., 0.), (0., 35.), (20., 35.)):
            p = engine.make_params(al, be, 0., 0., 4)
            row = []
            for s in np.linspace(-0.5, 1.5, 9):
                val, g, aux = eng.loss_grad((s * v).reshape(1, 1, 2), p, want_aux=True)
                row.append(f'{s:+.2f}:{val[0]:8.3f}')
            print(f'mag {mag} v {v.round(2)} alpha {al} beta {be} |', ' '.join(row))
