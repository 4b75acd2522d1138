import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False))
    d['edges'] = d['edges'].astype(np.float64)
    al, be, ga, de, lvl, ck = d['params']
    d['kw'] = dict(alpha=float(al), beta=float(be), gamma=float(ga), delta=float(de), cur_pyr_lvl=int(lvl),
                   contrast_kind=int(ck), method=str(d['method']))
    d['sensor_size'] = tuple(d['edges'].shape[1:])
    return d
