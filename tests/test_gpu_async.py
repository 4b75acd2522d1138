"""eincm_loss_grad_async / eincm_loss_grad_wait and EngineGroup: same numbers as the synchronous call, any grouping."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')


def _batch(B, H=96, W=128, N=20000, R=3):
    wins = [synth.make_window(50 + b, (H, W), N + 1000 * b, R, flow='smooth', flow_mag=6.0) for b in range(B)]
    args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
    return wins, args


@pytest.mark.parametrize('hw,gamma,lvl', [((1, 1), 0.0, 4), ((4, 4), 2.5e-4, 0)])
def test_group_equals_single_context(built_lib, hw, gamma, lvl):
    B, H, W = 5, 96, 128
    wins, args = _batch(B)
    th = np.stack([synth.theta_near_truth(50 + b, w, hw) for b, w in enumerate(wins)])
    p = engine.make_params(20.0, 35.0, gamma, 0.0, lvl)
    n_tot = sum(len(a[0]) for a in args)
    with engine.Engine((H, W), n_tot, max_refs=3, max_windows=B) as e:
        e.set_windows(args)
        v0, g0, a0 = e.loss_grad(th, p, want_aux=True)
    for n_groups in (1, 2, 3, 5):
        with engine.EngineGroup((H, W), n_tot, max_refs=3, max_windows=B, n_groups=n_groups) as grp:
            grp.set_windows(args)
            for _ in range(2):                                    # twice: the contexts are reusable after a wait
                v, g, a = grp.loss_grad(th, p, want_aux=True)
                np.testing.assert_allclose(v, v0, rtol=1e-6)
                np.testing.assert_allclose(g, g0, rtol=0, atol=1e-6 * np.abs(g0).max())
                assert [x['mean_rel_corr'] for x in a] == pytest.approx([x['mean_rel_corr'] for x in a0], rel=1e-6)
            v, g, _ = grp.loss_grad(th, p, want_grad=False)
            assert g is None
            np.testing.assert_allclose(v, v0, rtol=1e-6)


def test_async_state_errors(built_lib):
    wins, args = _batch(1)
    p = engine.make_params(20.0, 35.0, 0.0, 0.0, 4)
    th = synth.theta_near_truth(50, wins[0], (1, 1))
    with engine.Engine((96, 128), len(args[0][0]), max_refs=3) as e:
        e.set_windows(args)
        e._async = ((1, 1, 1, 2), True)
        with pytest.raises(engine.EincmError, match='without eincm_loss_grad_async'):
            e.loss_grad_wait()
        e.loss_grad_async(th, p)
        with pytest.raises(engine.EincmError, match='in flight'):        # nothing else may start on this context before the wait
            e.loss_grad(th, p)
        with pytest.raises(engine.EincmError, match='in flight'):
            e.set_windows(args)
        v1, g1, _ = e.loss_grad_wait()
        v2, g2, _ = e.loss_grad(th, p)                           # the synchronous call still works afterwards
        np.testing.assert_allclose(v1, v2, rtol=1e-6)
        np.testing.assert_allclose(g1, g2, rtol=0, atol=1e-6 * np.abs(g2).max())
        with pytest.raises(engine.EincmError, match='without eincm_loss_grad_async'):
            e.loss_grad_wait()
