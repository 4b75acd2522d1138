"""Bit-reproducibility of the HIP path.  Every accumulation that crosses workgroups is integer (u64 IWE accumulator, i64 gradient
accumulators) or an ordered fp64 sum, and staging is a stable sort, so repeating an evaluation - or staging the same window again,
or evaluating it in a fresh context - gives the same bits.  The reference asks for float64 because "BFGS converges correctly only with
float64" (/root/reference/src/experiments/e00/configs/main.yaml:34); what its optimiser needs from the objective is that the same theta
gives the same value and gradient, which XLA's deterministic CPU scatter provides and float atomics on a GPU do not."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')


def win_args(win):
    return (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])


CASES = [
    # id, (H,W), N, R, theta, gamma, lvl
    ('headline_2dof', (260, 346), 1_000_000, 5, (1, 1), 0.0, 4),
    ('pyr16_tv', (260, 346), 1_000_000, 5, (16, 16), 2.5e-4, 0),
    ('dense_tv', (240, 320), 400_000, 3, 'dense', 2.5e-4, 0),
    ('pyr4_delta', (120, 160), 60_000, 3, (4, 4), 0.0, 2),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_repeated_evaluations_are_bit_identical(built_lib, case):
    _, (H, W), N, R, hw, gamma, lvl = case
    win = synth.make_window(31, (H, W), N, R, flow='smooth' if hw != (1, 1) else 'constant', flow_mag=20.0)
    th = win['flow_gt'] * 0.9 if hw == 'dense' else synth.theta_near_truth(31, win, hw)
    delta = 0.5 if case[0] == 'pyr4_delta' else 0.0
    p = engine.make_params(20.0, 35.0, gamma, delta, lvl)
    ref = None
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        for k in range(5):
            if k == 2:
                eng.loss_grad(th * 1.1, p)                       # something else in between
            if k == 3:
                eng.set_window(*win_args(win))                   # staged again: the stable sort gives the same binned arrays
            v, g, _ = eng.loss_grad(th, p)
            cur = (v.copy(), g.copy(), eng.iwes(), eng.image_grad())
            if ref is None:
                ref = cur
            else:
                for a, b in zip(ref, cur):
                    assert np.array_equal(a, b), k
    with engine.Engine((H, W), N, max_refs=R) as e2:             # a fresh context
        e2.set_window(*win_args(win))
        v, g, _ = e2.loss_grad(th, p)
        assert np.array_equal(v, ref[0]) and np.array_equal(g, ref[1]) and np.array_equal(e2.iwes(), ref[2])


def test_scipy_bfgs_solves_are_identical(built_lib):
    """Three solves of the same window from the same start end at the same theta, bit for bit (the round-1 engine ended at
    -2945.6 / -2674.4 / -2718.8 because of float atomics)."""
    solver = importlib.import_module('edge-informed-contrast-maximization_amd.solver')
    losses = importlib.import_module('edge-informed-contrast-maximization_amd.losses')
    from functools import partial
    H, W, N, R = 180, 240, 100_000, 3
    win = synth.make_window(5, (H, W), N, R, flow='smooth', flow_mag=12.0)
    outs = []
    for _ in range(3):
        losses.clear_engine_cache()
        fun = partial(losses.value_and_grad_loss_func, alpha=20.0, beta=35.0, gamma=0.0, delta=0.0, cur_pyr_lvl=2, n_pyr_lvls=5,
                      sensor_size=(H, W))
        s = solver.ScipyMinimize(fun=fun, method='BFGS', maxiter=12, has_aux=True, value_and_grad=True, options={'gtol': 1e-7})
        x, st = s.run(np.zeros((4, 4, 2)), *win_args(win))
        outs.append((np.asarray(x).copy(), float(st.fun_val), int(st.iter_num)))
    losses.clear_engine_cache()
    for x, f, it in outs[1:]:
        assert np.array_equal(x, outs[0][0]) and f == outs[0][1] and it == outs[0][2]


def test_event_order_inside_a_tile_only_moves_roundings(built_lib, monkeypatch):
    """Staging sorts the events of every segment by source pixel and deals the sorted sequence to the threads that walk it (k_segsort: runs of
    one pixel per thread, different pixels across a wavefront).  Integer accumulation makes the images independent of the order of the events:
    the IWE and the count images are bit-identical with and without the sort, the loss therefore too; only the gradient's per-thread sums
    (2-DoF) and per-run sums (theta grids) may round differently."""
    H, W, N, R = 260, 346, 300_000, 3
    win = synth.make_window(33, (H, W), N, R, flow='constant', flow_mag=15.0)
    out = {}
    for hw, lvl in (((1, 1), 4), ((8, 8), 1)):
        th = synth.theta_near_truth(33, win, hw)
        p = engine.make_params(20.0, 35.0, 0.0, 0.0, lvl)
        for mode in ('redeal', 'time_order'):
            if mode == 'time_order':
                monkeypatch.setenv('EINCM_NO_SEGSORT', '1')
            else:
                monkeypatch.delenv('EINCM_NO_SEGSORT', raising=False)
            with engine.Engine((H, W), N, max_refs=R) as eng:
                eng.set_window(*win_args(win))
                v, g, _ = eng.loss_grad(th, p)
                out[mode] = (v.copy(), g.copy(), eng.iwes().copy(), eng.count_images())
        a, b = out['redeal'], out['time_order']
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])
        assert np.array_equal(a[0], b[0]), (a[0], b[0])
        np.testing.assert_allclose(a[1], b[1], rtol=0, atol=2e-6 * np.abs(b[1]).max())
    monkeypatch.delenv('EINCM_NO_SEGSORT', raising=False)
