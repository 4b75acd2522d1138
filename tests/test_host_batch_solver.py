"""LockstepBFGS (batch_solver.py) against SciPy's BFGS: given the same (value, grad), every window takes SciPy's steps.
CPU only (analytic objectives); the GPU side is tests/test_gpu_batch_solver.py."""
import importlib

import numpy as np
import pytest
import scipy.optimize as spo

bs = importlib.import_module('edge-informed-contrast-maximization_amd.batch_solver')


def rosen_like(scale):
    def f(x):
        x = np.asarray(x, dtype=np.float64)
        v = scale * np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] += scale * (-400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1]))
        g[1:] += scale * 200.0 * (x[1:] - x[:-1] ** 2)
        return v, g
    return f


def noisy_quadratic(seed, n):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
    b = rng.standard_normal(n)

    def f(x):                                   # a smooth bowl plus a deterministic ripple at the 1e-9 level: line searches fail near the optimum
        x = np.asarray(x, dtype=np.float64)
        v = 0.5 * x @ A @ x - b @ x + 1e-9 * np.sum(np.sin(1e5 * x))
        g = A @ x - b + 1e-4 * np.cos(1e5 * x)
        return float(v), g
    return f


def run_pair(funs, x0, maxiter, gtol):
    B = len(funs)

    def fun_batch(X, mask):
        vg = [funs[b](X[b]) for b in range(B)]
        return np.array([v for v, _ in vg]), np.stack([g for _, g in vg])
    drv = bs.LockstepBFGS(fun_batch, x0, maxiter, gtol)
    res = drv.run()
    ref = [spo.minimize(funs[b], x0[b], jac=True, method='BFGS', options={'maxiter': maxiter, 'gtol': gtol}) for b in range(B)]
    return res, ref, drv


@pytest.mark.parametrize('n', [2, 10, 32])
def test_lockstep_equals_scipy_bit_for_bit(n):
    """n <= 64: SciPy's own update expression; iterates, values, counts and status codes are identical."""
    rng = np.random.default_rng(n)
    funs = [rosen_like(s) for s in (1.0, 0.3, 2.5, 1e-2)]
    x0 = rng.uniform(-1.5, 1.5, (4, n))
    res, ref, drv = run_pair(funs, x0, maxiter=60, gtol=1e-7)
    for a, b in zip(res, ref):
        assert np.array_equal(a.x, b.x) and a.fun == b.fun
        assert (a.nit, a.status, a.success) == (b.nit, b.status, b.success)
        assert a.nfev == b.nfev
    # windows finish at different ticks; one batched call serves every tick
    assert drv.n_batch_evals >= max(r.nfev for r in ref) and drv.n_batch_evals <= sum(r.nfev for r in ref)


def test_rank_two_update_beyond_64_dimensions():
    """n > 64 uses the O(n^2) rank-two form of the same update: same minimiser to rounding, far fewer flops."""
    n = 96

    def bowl(seed):
        rng = np.random.default_rng(seed)
        A = rng.standard_normal((n, n)); A = A @ A.T / n + np.eye(n)
        b = rng.standard_normal(n)
        return lambda x: (float(0.5 * x @ A @ x - b @ x + 0.25 * np.sum(x ** 4)), A @ x - b + x ** 3)
    funs = [bowl(1), bowl(2)]
    x0 = np.random.default_rng(5).uniform(-0.5, 0.5, (2, n))
    res, ref, _ = run_pair(funs, x0, maxiter=400, gtol=1e-7)
    for a, b in zip(res, ref):
        assert a.status == b.status == 0
        assert np.abs(a.x - b.x).max() < 1e-6 and a.fun == pytest.approx(b.fun, abs=1e-10)


def test_line_search_failure_takes_scipys_fallback_and_status():
    """A rippled bowl: DCSRCH gives up near the optimum, line_search_wolfe2 is tried, and the solve ends with SciPy's status
    (2 = precision loss) at SciPy's point."""
    funs = [noisy_quadratic(1, 6), noisy_quadratic(2, 6), rosen_like(1.0)]
    x0 = np.random.default_rng(9).uniform(-1, 1, (3, 6))
    res, ref, _ = run_pair(funs, x0, maxiter=200, gtol=1e-12)
    assert any(r.status == 2 for r in ref), 'the construction no longer provokes a line-search failure'
    for a, b in zip(res, ref):
        assert a.status == b.status and a.nit == b.nit
        assert np.array_equal(a.x, b.x) and a.fun == b.fun


def test_inactive_windows_ride_along():
    funs = [rosen_like(1.0)] * 3
    x0 = np.array([[0.5, 0.5], [-1.0, 1.0], [1.2, 1.2]])
    calls = []

    def fun_batch(X, mask):
        calls.append(X.copy())
        assert not mask[1]
        vg = [funs[b](X[b]) for b in range(3)]
        return np.array([v for v, _ in vg]), np.stack([g for _, g in vg])
    drv = bs.LockstepBFGS(fun_batch, x0, 50, 1e-8, active=[True, False, True])
    res = drv.run()
    assert res[1] is None and res[0].success and res[2].success
    assert all(np.array_equal(c[1], x0[1]) for c in calls)          # the rider's point never moves


def test_triangular_blas_update_equals_the_numpy_form(monkeypatch):
    """Above 64 unknowns the inverse Hessian is kept in one triangle and updated by dsyr2 / read by dsymv (with BLAS threading off);
    without threadpoolctl the same rank-two correction runs in plain numpy.  Same iterates to rounding, same symmetric hess_inv."""
    n = 96
    rng = np.random.default_rng(1)
    A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
    b = rng.normal(size=n)

    def fun_batch(X, mask):
        return np.array([0.5 * x @ A @ x - b @ x for x in X]), np.stack([A @ x - b for x in X])
    x0 = rng.normal(size=(2, n))
    r_blas = bs.LockstepBFGS(fun_batch, x0, 200, 1e-9).run()
    monkeypatch.setattr(bs, 'threadpool_limits', None)
    r_np = bs.LockstepBFGS(fun_batch, x0, 200, 1e-9).run()
    for a, c in zip(r_blas, r_np):
        assert a.success and c.success and a.nit == c.nit
        assert np.abs(a.x - c.x).max() <= 1e-12 and np.abs(a.x - np.linalg.solve(A, b)).max() <= 1e-9
        assert np.array_equal(a.hess_inv, a.hess_inv.T)
        assert np.abs(a.hess_inv - c.hess_inv).max() <= 1e-9 * np.abs(c.hess_inv).max()


def test_an_evaluation_error_leaves_no_helper_thread_behind():
    """The wolfe2 fallback runs SciPy's function in a helper thread; if the batched evaluation raises while such a thread waits for
    its answer, run() re-raises and the thread is woken and ended."""
    import threading
    funs = [noisy_quadratic(1, 6), noisy_quadratic(2, 6)]
    x0 = np.random.default_rng(9).uniform(-1, 1, (2, 6))
    drv_holder = {}

    def fun_batch(X, mask):
        drv = drv_holder['drv']
        if any(w is not None and getattr(w, 'phase', '') == 'ls2' for w in drv.windows):
            raise RuntimeError('engine failure')
        vg = [funs[b](X[b]) for b in range(2)]
        return np.array([v for v, _ in vg]), np.stack([g for _, g in vg])
    before = threading.active_count()
    drv_holder['drv'] = bs.LockstepBFGS(fun_batch, x0, 200, 1e-12)
    with pytest.raises(RuntimeError, match='engine failure'):
        drv_holder['drv'].run()
    assert threading.active_count() == before


@pytest.mark.parametrize('groups', [[[0, 1], [2, 3]], [[2], [0, 3], [1]]])
def test_pipelined_groups_take_the_same_steps(groups):
    """The pipelined form (several evaluation contexts, launch / collect) only changes WHEN a window is evaluated: every window's
    iterates, counts and status equal the plain lockstep's and SciPy's."""
    n = 10
    funs = [rosen_like(s) for s in (1.0, 0.3, 2.5, 1e-2)]
    x0 = np.random.default_rng(3).uniform(-1.5, 1.5, (4, n))
    res, ref, _ = run_pair(funs, x0, maxiter=60, gtol=1e-7)
    pending, log = {}, []

    def launch(gi, X, mask):
        assert gi not in pending                                   # one evaluation in flight per context
        ix = groups[gi]
        assert not mask[[b for b in range(4) if b not in ix]].any()
        pending[gi] = [(funs[b](X[b]) if mask[b] else (np.nan, np.zeros(n))) for b in ix]
        log.append(('launch', gi))

    def collect(gi):
        vg = pending.pop(gi)
        log.append(('collect', gi))
        return np.array([v for v, _ in vg]), np.stack([g for _, g in vg])
    drv = bs.LockstepBFGS(None, x0, 60, 1e-7, groups=groups, launch=launch, collect=collect)
    out = drv.run()
    assert not pending
    for a, b, c in zip(out, res, ref):
        assert np.array_equal(a.x, b.x) and np.array_equal(a.x, c.x) and (a.nit, a.nfev, a.status) == (c.nit, c.nfev, c.status)
    # every group is launched before the first one is collected: their evaluations overlap
    assert [e for e in log[:len(groups)]] == [('launch', gi) for gi in range(len(groups))]
