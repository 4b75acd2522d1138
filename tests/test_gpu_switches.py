"""The environment switches DESIGN.md documents (experiments kept as a record, fallbacks, tuning knobs) still produce the objective:
every one of them is run in its own process (three are read once per process) on a two-window batch over four theta shapes and
compared with the default configuration and with the oracle.  Integer accumulation makes the IWE stack independent of layouts and
window capacities; segment lengths change the fixed-point scale of a tap, hence 'equal to ~1e-7' rather than bit for bit there."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('_switch_probe', os.path.join(HERE, '_switch_probe.py'))
probe = importlib.util.module_from_spec(spec)
spec.loader.exec_module(probe)

SWITCHES = [
    ({'EINCM_COMPOSE': '1'}, 'fused statistics + composing gather (DESIGN 4.3)'),
    ({'EINCM_SPLAT_MERGE': '1'}, 'run-merged forward accumulation (DESIGN 4.4)'),
    ({'EINCM_NO_HOST_ASM': '1'}, 'scalar assembly on the device (k_final)'),
    ({'EINCM_NO_BIG_THETA_ARG': '1'}, 'k_theta reads theta from the pinned staging buffer instead of its kernel arguments'),
    ({'EINCM_NO_PROJ_IN_GATHER': '1'}, 'dL/dTheta image + k_project instead of the in-gather projection'),
    ({'EINCM_NO_SEGSORT': '1', 'EINCM_NO_SPREAD': '1'}, 'both event copies in plain time order'),
    ({'EINCM_SEG': '4096', 'EINCM_SEG_SPLAT': '2048', 'EINCM_SEG_2DOF': '8192'}, 'other segment lengths'),
    ({'EINCM_GATHER_PARTS': '2'}, '2-DoF gather segments shared by two workgroups'),
    ({'EINCM_HOST_BINNING': '1'}, 'host-side counting sort'),
    ({'EINCM_GATHER_ALL_R': '1'}, 'theta-grid gather: one workgroup per segment walks all reference times'),
    ({'EINCM_PITCH_ALIGNED': '2'}, 'LDS windows of both 2-DoF event kernels at the bank-aligned row pitch (k_splat: the policy of a large batch, forced on a small one)'),
]


def run_probe(tmp_path, tag, env_extra):
    out = os.path.join(tmp_path, f'{tag}.npz')
    env = {k: v for k, v in os.environ.items() if not k.startswith('EINCM_') or k == 'EINCM_LIB'}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, '_switch_probe.py'), out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (tag, r.stdout[-2000:], r.stderr[-2000:])
    return dict(np.load(out))


def rel(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.mark.timeout(900)
def test_switches_reproduce_the_default_and_the_oracle(built_lib, tmp_path):
    from oracle import eincm_oracle as O
    base = run_probe(str(tmp_path), 'default', {})
    wins, thetas = probe.inputs()
    for i, (th, (hw, gamma, lvl)) in enumerate(zip(thetas, probe.CASES)):          # the default against the oracle
        for b, w in enumerate(wins):
            a = (w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts'])
            v_o, g_o, _ = O.loss_and_grad(th[b], *a, 20.0, 35.0, gamma, 0.0, lvl, 5, (probe.H, probe.W))
            assert abs(base[f'v{i}'][b] - v_o) <= 1e-5 * abs(v_o), (hw, b)
            assert rel(base[f'g{i}'][b], g_o) <= 1e-5, (hw, b)
    for k, (env, what) in enumerate(SWITCHES):
        got = run_probe(str(tmp_path), f'switch{k}', env)
        for i, (hw, _, _) in enumerate(probe.CASES):
            assert rel(got[f'v{i}'], base[f'v{i}']) <= 2e-6, (what, hw, rel(got[f'v{i}'], base[f'v{i}']))
            assert rel(got[f'g{i}'], base[f'g{i}']) <= 2e-5, (what, hw, rel(got[f'g{i}'], base[f'g{i}']))
            assert rel(got[f'iwe{i}'], base[f'iwe{i}']) <= 1e-6, (what, hw)
            assert rel(got[f'G{i}'], base[f'G{i}']) <= 2e-5, (what, hw)
