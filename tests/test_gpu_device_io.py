"""eincm_loss_grad_device: theta and gradient resident in HBM (no reference counterpart: its optimiser lives on the host,
src/eincm/solver.py:165-173).  The evaluation is the one eincm_loss_grad runs - same kernels, device-side scalar assembly - so the
results must agree with the host boundary to rounding, and with the oracle to the north-star tolerance."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')

TOL = 1e-5      # north star: 1e-5 relative


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def test_device_resident_theta_and_gradient(built_lib):
    import torch
    from oracle import eincm_oracle as O
    H, W, R, B = 96, 128, 3, 2
    wins = [synth.make_window(90 + b, (H, W), 20000 + 3000 * b, R, flow='smooth', flow_mag=8.0) for b in range(B)]
    args = [(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins]
    cases = [((1, 1), 0.0, 4), ((4, 4), 2.5e-4, 0), ((16, 16), 0.0, 1), ('dense', 2.5e-4, 0)]
    with engine.Engine((H, W), 50000, max_refs=R, max_windows=B) as eng:
        eng.set_windows(args)
        for hw, gamma, lvl in cases:
            th = np.stack([np.ascontiguousarray(w['flow_gt'] * 0.9) if hw == 'dense' else synth.theta_near_truth(90 + b, w, hw)
                           for b, w in enumerate(wins)])
            p = engine.make_params(20.0, 35.0, gamma, 0.0, lvl)
            v_h, g_h, aux_h = eng.loss_grad(th, p, want_aux=True)
            t = torch.from_numpy(th).cuda()
            v_d, g_d, aux_d = eng.loss_grad_device(t, p, want_aux=True)
            assert isinstance(g_d, torch.Tensor) and g_d.is_cuda and g_d.shape == t.shape and g_d.dtype == torch.float64
            g_dn = g_d.cpu().numpy()
            assert rel(v_d, v_h) <= 1e-10 and rel(g_dn, g_h) <= 1e-9, (hw, rel(v_d, v_h), rel(g_dn, g_h))
            for k in ('mean_rel_corr', 'mean_rel_contrast', 'theta_total_variation'):
                assert aux_d[0][k] == pytest.approx(aux_h[0][k], rel=1e-10, abs=1e-300), (hw, k)
            # a bound on |theta| only selects LDS window capacities: as long as the windows hold every tap the results are bit for bit the same
            v_b, g_b, _ = eng.loss_grad_device(t, p, theta_abs_max=float(np.abs(th).max()))
            assert np.array_equal(v_b, v_d) and torch.equal(g_b, g_d), hw
            # a bound far too small: taps that miss the window go straight to HBM, rounded at the accumulator's scale (2^-30) instead of
            # the segment's (2^-21): the same image to a fixed-point quantum
            v_s, g_s, _ = eng.loss_grad_device(t, p, theta_abs_max=0.0)
            assert rel(v_s, v_d) <= 1e-7 and rel(g_s.cpu().numpy(), g_dn) <= 1e-6, hw
            for b in range(B):                                                   # and against the oracle
                v_o, g_o, _ = O.loss_and_grad(th[b], *args[b], 20.0, 35.0, gamma, 0.0, lvl, 5, (H, W))
                assert abs(v_d[b] - v_o) <= TOL * abs(v_o) and rel(g_dn[b], g_o) <= TOL, (hw, b)
            v_f, g_f, _ = eng.loss_grad_device(t, p, want_grad=False)            # forward only
            assert g_f is None and rel(v_f, v_d) <= 1e-7          # (forward-only evaluations sum the contrast in another kernel)
        # a later host-boundary call is unaffected, and the accessors see the device theta's image
        th = np.stack([synth.theta_near_truth(90 + b, w, (1, 1)) for b, w in enumerate(wins)])
        p = engine.make_params(20.0, 35.0, 0.0, 0.0, 4)
        v_d, g_d, _ = eng.loss_grad_device(torch.from_numpy(th).cuda(), p)
        T = eng.scaled_theta()
        assert np.array_equal(T[1, 5, 7], th[1, 0, 0])
        v_h, g_h, _ = eng.loss_grad(th, p)
        assert rel(v_d, v_h) <= 1e-10 and rel(g_d.cpu().numpy(), g_h) <= 1e-9


def test_device_entry_rejects_host_memory(built_lib):
    import ctypes as C
    H, W, R = 48, 64, 2
    w = synth.make_window(95, (H, W), 3000, R, flow='constant', flow_mag=3.0)
    with engine.Engine((H, W), 3000, max_refs=R) as eng:
        eng.set_window(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts'])
        th = np.zeros((1, 1, 1, 2))
        val = np.empty(1)
        p = engine.make_params(20.0, 35.0, 0.0, 0.0, 4)
        rc = eng._lib.eincm_loss_grad_device(eng._ctx, C.c_void_p(th.ctypes.data), 1, 1, C.byref(p), -1.0,
                                             val.ctypes.data_as(C.POINTER(C.c_double)), None, None)
        assert rc == engine.L.ERR_ARG
        with pytest.raises(TypeError):
            eng.loss_grad_device(th, p)
        v, _, _ = eng.loss_grad(th, p)                                     # the context is still usable
        assert np.isfinite(v[0])
