"""Lockstep solver for a BATCH of independent event windows: the caller of the engine's batch path (BASELINE config C4).

The reference solves one window at a time (/root/reference/src/eincm/solver.py:197-267, src/experiments/e00/exp_mgr.py:615-659):
every BFGS function evaluation is one loss+grad of one window.  The HIP engine evaluates B windows in ONE call at 2-3 times the
per-window rate of B single calls (bench.py: 8 x 10^6 events in 0.24 ms against 8 x 0.07 ms), but only a caller that has B thetas
ready at the same moment can use that.  This module is that caller:

* ``LockstepBFGS`` runs B independent BFGS minimisations - SciPy's own algorithm (scipy.optimize._optimize._minimize_bfgs: inverse
  Hessian update, initial step guess, strong-Wolfe line search by MINPACK's DCSRCH with the fallback to line_search_wolfe2, the same
  stopping rules and status codes), restated as a state machine that asks for ONE function evaluation at a time - and answers all
  the requests of a tick with one batched evaluation.  Windows that have converged ride along with their last theta.  Given the
  same (value, grad) a window takes exactly the steps SciPy's BFGS would take.
* ``BatchedMultipleLevelEINCMSolver`` drives the theta pyramid of B windows (or of the current windows of B independent sequences)
  level by level with it: same constructor keywords, state and per-window result dict as ``solver.MultipleLevelEINCMSolver``
  (reference solver.py:16-126, :254-267), retries included; a solved handover weight (L-BFGS-B on one scalar, :325-335) is found
  window by window with the other windows riding along.
"""
import numpy as np
import scipy.optimize as spo
from scipy.optimize._dcsrch import DCSRCH            # MINPACK-2 dcsrch as SciPy ships it: reverse communication, one step per call
from scipy.optimize._linesearch import line_search_wolfe2
from scipy.linalg.blas import dsymv, dsyr2

if not hasattr(DCSRCH, '_iterate'):                  # private SciPy API (1.12 ... 1.15 have it): fail at import, not mid-solve
    raise ImportError('batch_solver needs scipy.optimize._dcsrch.DCSRCH._iterate (SciPy >= 1.12); found SciPy ' + __import__('scipy').__version__)

try:                                                 # level-2 BLAS on a 512 x 512 matrix must not fan out over the host's cores
    import threadpoolctl                             # (OpenBLAS with 8+ threads: 15 ms per dsyr2 instead of 0.1 ms)
    _TPC = [None]

    def threadpool_limits(limits, user_api):
        """threadpoolctl's limiter on ONE controller per process: building a controller walks every loaded shared library (2 ms
        with torch in the process); limiting through an existing one takes 10 us."""
        if _TPC[0] is None:
            _TPC[0] = threadpoolctl.ThreadpoolController()
        return _TPC[0].limit(limits=limits, user_api=user_api)
except ImportError:                                  # without it the update stays in plain numpy
    threadpool_limits = None

from .engine import Engine, make_params
from .solver import ScipyMinimizeInfo, EmptyCallback, rescale_theta, _canon

_BFGS_C1, _BFGS_C2, _BFGS_XTOL, _BFGS_AMIN, _BFGS_AMAX, _LS_MAXITER = 1e-4, 0.9, 1e-14, 1e-100, 1e100, 100
_EXACT_UPDATE_MAX_N = 64          # up to this many unknowns the inverse-Hessian update is SciPy's own expression (two n x n products)


class _QuietLineSearch:
    """``warnings.catch_warnings`` is process-wide state: helper threads that enter and leave it independently restore each other's
    filters (the first one out switches the warning back on for the others).  This one counts: the first thread in installs the
    filter for scipy's LineSearchWarning, the last one out restores what was there."""

    def __init__(self):
        import threading
        self._lock, self._n, self._cw = threading.Lock(), 0, None

    def __enter__(self):
        import warnings
        from scipy.optimize._linesearch import LineSearchWarning
        with self._lock:
            if self._n == 0:
                self._cw = warnings.catch_warnings()
                self._cw.__enter__()
                warnings.simplefilter('ignore', LineSearchWarning)
            self._n += 1

    def __exit__(self, *exc):
        with self._lock:
            self._n -= 1
            if self._n == 0:
                self._cw.__exit__(None, None, None)
                self._cw = None
        return False


_quiet_line_search = _QuietLineSearch()


class _Abandoned(BaseException):
    """Raised inside a helper thread whose owner abandoned the minimisation."""


class _CoroutineCall:
    """Runs target(f, fprime) in a helper thread and turns its calls of f(x) / fprime(x) into requests the owner answers:
    next() -> ('request', x) or ('done', result); answer(value, grad) resumes the target.  One of the two threads runs at a time."""

    def __init__(self, target):
        import threading
        self._cv = threading.Condition()
        self._req = self._ans = self._res = None
        self._state = 'running'                  # running | waiting (a request is posted) | done
        self._cache = None

        def f(x):
            x = np.array(x, dtype=np.float64, copy=True)
            with self._cv:
                self._req, self._state = x, 'waiting'
                self._cv.notify_all()
                self._cv.wait_for(lambda: self._state in ('running', 'aborted'))
                if self._state == 'aborted':
                    raise _Abandoned()
                v, g = self._ans
            self._cache = (x, g)
            return v

        def fp(x):
            if self._cache is not None and np.array_equal(self._cache[0], x):
                return self._cache[1]
            f(x)
            return self._cache[1]

        def run():
            try:
                with _quiet_line_search:                        # (SciPy's BFGS silences the fallback's LineSearchWarning too)
                    res = target(f, fp)
            except BaseException as e:          # noqa: BLE001 - handed to the owner
                res = e
            with self._cv:
                self._res, self._state = res, 'done'
                self._cv.notify_all()
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def next(self):
        with self._cv:
            self._cv.wait_for(lambda: self._state in ('waiting', 'done'))
            if self._state == 'waiting':
                return 'request', self._req
        self._thread.join()
        if isinstance(self._res, BaseException):
            raise self._res
        return 'done', self._res

    def answer(self, value, grad):
        with self._cv:
            self._ans = (float(value), np.array(grad, dtype=np.float64).reshape(-1))
            self._state = 'running'
            self._cv.notify_all()

    def abandon(self):
        """The owner gives up (an evaluation raised): wake the helper with an exception so that its thread ends instead of waiting forever."""
        with self._cv:
            if self._state == 'done':
                return
            self._state = 'aborted'
            self._cv.notify_all()
        self._thread.join(timeout=5.0)


class _WindowBFGS:
    """One window's BFGS, one function evaluation at a time (scipy.optimize._optimize._minimize_bfgs with jac=True)."""

    def __init__(self, x0, maxiter, gtol, callback=None, wolfe2_fallback=True):
        self.wolfe2_fallback = bool(wolfe2_fallback)
        self.x0 = np.array(x0, dtype=np.float64).reshape(-1)
        self.n = self.x0.size
        self.maxiter = int(maxiter) if maxiter is not None else self.n * 200
        self.gtol = float(gtol)
        self.callback = callback
        self.phase = 'init'
        self.request = self.x0                  # the point whose (value, grad) this window wants next; None = finished
        self.k = 0
        self.nfev = 0
        self.warnflag = 0
        self.result = None

    # -- what the driver calls ----------------------------------------------------------------------------------
    def feed(self, f, g, single_eval):
        """(value, grad) at ``self.request``.  single_eval(x) -> (f, g) evaluates this window alone (the rare wolfe2 fallback)."""
        self.nfev += 1
        f = float(f)
        g = np.array(g, dtype=np.float64).reshape(-1)
        if self.phase == 'init':
            self.xk, self.old_fval, self.gfk = self.x0, f, g
            self.sym = threadpool_limits is not None and self.n > _EXACT_UPDATE_MAX_N
            # sym: the inverse Hessian lives in the UPPER triangle of a Fortran-ordered array (dsymv / dsyr2 touch half the matrix)
            self.Hk = np.asfortranarray(np.eye(self.n)) if self.sym else np.eye(self.n)
            self.old_old_fval = self.old_fval + np.linalg.norm(self.gfk) / 2          # initial step guess dx ~ 1
            self.gnorm = np.abs(self.gfk).max() if self.n else 0.0
            self._begin_iteration(single_eval)
        elif self.phase == 'ls':
            self.phi1, self.gval = f, g
            self.derphi1 = float(np.dot(g, self.pk))
            self._ls_step(single_eval)
        elif self.phase == 'ls2':
            self.ls2.answer(f, g)
            self._ls2_advance(single_eval)
        else:
            raise RuntimeError('feed() on a finished window')

    # -- BFGS iteration -------------------------------------------------------------------------------------------
    def _begin_iteration(self, single_eval):
        if not (self.gnorm > self.gtol and self.k < self.maxiter):
            return self._finish()
        self.pk = -dsymv(1.0, self.Hk, self.gfk, lower=0) if self.sym else -np.dot(self.Hk, self.gfk)
        derphi0 = float(np.dot(self.gfk, self.pk))
        # scalar_search_wolfe1: the first trial step
        if self.old_old_fval is not None and derphi0 != 0:
            alpha1 = min(1.0, 1.01 * 2 * (self.old_fval - self.old_old_fval) / derphi0)
            if alpha1 < 0:
                alpha1 = 1.0
        else:
            alpha1 = 1.0
        self.dcsrch = DCSRCH(None, None, _BFGS_C1, _BFGS_C2, _BFGS_XTOL, _BFGS_AMIN, _BFGS_AMAX)
        self.task, self.alpha1, self.phi1, self.derphi1, self.derphi0 = b'START', alpha1, self.old_fval, derphi0, derphi0
        self.gval = self.gfk
        self.ls_iter = 0
        self._ls_step(single_eval)

    def _ls_step(self, single_eval):
        """One pass of the loop of DCSRCH.__call__; leaves a request behind, or ends the line search."""
        if self.ls_iter >= _LS_MAXITER:
            return self._ls_done(None, single_eval)
        self.ls_iter += 1
        stp, self.phi1, self.derphi1, self.task = self.dcsrch._iterate(self.alpha1, self.phi1, self.derphi1, self.task)
        if not np.isfinite(stp):
            return self._ls_done(None, single_eval)
        if self.task[:2] == b'FG':
            self.alpha1 = stp
            self.phase, self.request = 'ls', self.xk + stp * self.pk
            return
        if self.task[:5] == b'ERROR' or self.task[:4] == b'WARN':
            stp = None
        self._ls_done(stp, single_eval)

    def _ls_done(self, stp, single_eval):
        if stp is not None:
            return self._step_taken(stp, self.phi1, self.gval, single_eval)
        if not self.wolfe2_fallback:                               # opt-out of SciPy's second line search: precision loss here and now
            self.warnflag = 2
            return self._finish()
        # _line_search_wolfe12: DCSRCH found no step, SciPy tries its other line search.  That one is not written for reverse
        # communication, so it runs in a helper thread whose f / fprime calls become this window's requests: the evaluations
        # stay in lockstep with the other windows (with the engine's fp32-level noise this fallback is the common end of a level)
        self.ls2 = _CoroutineCall(lambda fv, fg: line_search_wolfe2(fv, fg, self.xk, self.pk, self.gfk, self.old_fval, self.old_old_fval,
                                                                     c1=_BFGS_C1, c2=_BFGS_C2, amax=_BFGS_AMAX))
        self._ls2_advance(single_eval)

    def _ls2_advance(self, single_eval):
        kind, payload = self.ls2.next()
        if kind == 'request':
            self.phase, self.request = 'ls2', payload
            return
        ret = payload
        if ret[0] is None:
            self.warnflag = 2                                      # precision loss: no step satisfies the Wolfe conditions
            return self._finish()
        alpha_k, new_fval, gfkp1 = ret[0], ret[3], ret[5]
        if gfkp1 is None:                                          # (line_search_wolfe2 returns the gradient of its last evaluation)
            gfkp1 = np.asarray(single_eval(self.xk + alpha_k * self.pk)[1], dtype=np.float64).reshape(-1)
        self._step_taken(alpha_k, new_fval, gfkp1, single_eval)

    def _step_taken(self, alpha_k, new_fval, gfkp1, single_eval):
        self.old_fval, self.old_old_fval = new_fval, self.old_fval
        sk = alpha_k * self.pk
        self.xk = self.xk + sk
        yk = gfkp1 - self.gfk
        self.gfk = gfkp1
        self.k += 1
        if self.callback is not None:
            self.callback(spo.OptimizeResult(x=self.xk, fun=self.old_fval))
        self.gnorm = np.abs(self.gfk).max()
        if self.gnorm <= self.gtol:
            return self._finish()
        if alpha_k * np.linalg.norm(self.pk) <= 0.0:              # xrtol = 0
            return self._finish()
        if not np.isfinite(self.old_fval):
            self.warnflag = 2
            return self._finish()
        rhok_inv = float(np.dot(yk, sk))
        rhok = 1000.0 if rhok_inv == 0.0 else 1.0 / rhok_inv
        if self.n <= _EXACT_UPDATE_MAX_N:                          # SciPy's own expression (bit for bit the same inverse Hessian)
            I = np.eye(self.n, dtype=int)
            A1 = I - sk[:, np.newaxis] * yk[np.newaxis, :] * rhok
            A2 = I - yk[:, np.newaxis] * sk[np.newaxis, :] * rhok
            self.Hk = np.dot(A1, np.dot(self.Hk, A2)) + (rhok * sk[:, np.newaxis] * sk[np.newaxis, :])
        else:
            # the same update as ONE symmetric rank-two correction, O(n^2) instead of the two n x n products (n = 512 at a 16x16
            # theta: 10 ms per iteration in SciPy's form, the evaluation itself takes 0.1 ms):
            #   (I - r s y^T) H (I - r y s^T) + r s s^T = H - r (s (Hy)^T + (Hy) s^T) + r (1 + r y^T H y) s s^T     (H symmetric)
            #                                         = H + s w^T + w s^T,   w = (c / 2) s - r Hy,  c = r (1 + r y^T H y)
            Hy = dsymv(1.0, self.Hk, yk, lower=0) if self.sym else np.dot(self.Hk, yk)
            w = (0.5 * rhok * (1.0 + rhok * float(np.dot(yk, Hy)))) * sk - rhok * Hy
            if self.sym:
                self.Hk = dsyr2(1.0, sk, w, a=self.Hk, overwrite_a=1, lower=0)     # in place, upper triangle
            else:
                sw = np.outer(sk, w)
                self.Hk = self.Hk + sw + sw.T
        self._begin_iteration(single_eval)

    def _finish(self):
        fval = self.old_fval
        if self.warnflag == 2:
            pass
        elif self.k >= self.maxiter:
            self.warnflag = 1
        elif np.isnan(self.gnorm) or np.isnan(fval) or np.isnan(self.xk).any():
            self.warnflag = 3
        self.phase, self.request = 'done', None
        if getattr(self, 'sym', False):                             # hand out the full matrix
            d = self.Hk.diagonal().copy()                          # (the strictly lower triangle is still the identity's: zero)
            self.Hk = self.Hk + self.Hk.T
            self.Hk[np.diag_indices(self.n)] = d
            self.sym = False
        self.result = spo.OptimizeResult(fun=fval, jac=self.gfk, hess_inv=self.Hk, nfev=self.nfev, njev=self.nfev,
                                         status=self.warnflag, success=(self.warnflag == 0), x=self.xk, nit=self.k)


class LockstepBFGS:
    """B independent BFGS minimisations that evaluate in lockstep.

    fun_batch(X, mask) with X of shape (B, n) returns (values (B,), grads (B, n)); it is called once per tick with every window's
    current request (finished or inactive windows: their last point) and the mask of the windows that asked - the engine evaluates
    only those (Engine.loss_grad(active=...)), so a tick costs what its requesting windows cost.  ``active``: which windows are
    minimised at all.  ``wolfe2_fallback=False`` ends a minimisation with status 2 where DCSRCH finds no step, instead of trying SciPy's
    second line search first (line_search_wolfe2: some 40 evaluations that rarely find a step once the first search has failed on the
    objective's own roughness; profiles/r03/linesearch_fp64_vs_fp32.txt) - a deviation from SciPy, off by default.
    """

    def __init__(self, fun_batch, x0, maxiter, gtol, callbacks=None, active=None, wolfe2_fallback=True, groups=None, launch=None,
                 collect=None):
        """Pipelined form: ``groups`` (index arrays that partition the windows), ``launch(gi, X, mask)`` (enqueue the evaluation of group
        gi's masked windows at X[groups[gi]] and return at once) and ``collect(gi)`` -> (values, grads) of that group.  While the host
        advances the line searches of one group, the other groups' evaluations run on the GPU; ``fun_batch`` is not used then."""
        x0 = np.asarray(x0, dtype=np.float64)
        self.B, self.n = x0.shape
        self.fun_batch = fun_batch
        self.groups = [np.asarray(ix, dtype=int) for ix in groups] if groups is not None else None
        self.launch, self.collect = launch, collect
        if self.groups is not None:
            assert launch is not None and collect is not None
            assert sorted(int(b) for ix in self.groups for b in ix) == list(range(self.B)), 'groups must partition the windows'
            self._group_of = {int(b): gi for gi, ix in enumerate(self.groups) for b in ix}
            self._pos = {int(b): i for ix in self.groups for i, b in enumerate(ix)}
            self._inflight = [None] * len(self.groups)
        act = np.ones(self.B, bool) if active is None else np.asarray(active, bool)
        maxiters = np.broadcast_to(np.asarray(maxiter), (self.B,))
        cbs = callbacks if callbacks is not None else [None] * self.B
        self.windows = [(_WindowBFGS(x0[b], maxiters[b], gtol, cbs[b], wolfe2_fallback) if act[b] else None) for b in range(self.B)]
        self.last = x0.copy()
        self.n_batch_evals = 0                  # engine calls
        self.n_window_evals = 0                 # windows evaluated over all calls

    def _single(self, b):
        def ev(x):
            X = self.last.copy()
            X[b] = np.asarray(x, dtype=np.float64).reshape(-1)
            m = np.zeros(self.B, bool); m[b] = True
            self.n_batch_evals += 1; self.n_window_evals += 1
            if self.groups is not None:                 # (called while group gi is being fed: its context is idle)
                gi = self._group_of[b]
                self.launch(gi, X, m)
                v, g = self.collect(gi)
                return float(v[self._pos[b]]), np.array(g[self._pos[b]], dtype=np.float64)
            v, g = self.fun_batch(X, m)
            return float(v[b]), np.array(g[b], dtype=np.float64)
        return ev

    def run(self):
        """List of scipy OptimizeResult (None for inactive windows)."""
        run = self._run if self.groups is None else self._run_pipelined
        try:
            if threadpool_limits is not None:
                with threadpool_limits(limits=1, user_api='blas'):
                    return run()
            return run()
        finally:                                   # an evaluation raised mid-solve: no helper thread of a line-search fallback stays behind
            if self.groups is not None:            # ... and no evaluation in flight
                for gi, req in enumerate(self._inflight):
                    if req is not None:
                        self._inflight[gi] = None
                        try:
                            self.collect(gi)
                        except Exception:          # noqa: BLE001 - the first error is the one that propagates
                            pass
            for w in self.windows:
                ls2 = getattr(w, 'ls2', None) if w is not None else None
                if ls2 is not None:
                    ls2.abandon()

    def _run(self):
        while True:
            req = [(b, w) for b, w in enumerate(self.windows) if w is not None and w.request is not None]
            if not req:
                break
            m = np.zeros(self.B, bool)
            for b, w in req:
                self.last[b] = w.request
                m[b] = True
            self.n_batch_evals += 1; self.n_window_evals += len(req)
            v, g = self.fun_batch(self.last, m)
            for b, w in req:
                w.feed(v[b], g[b], self._single(b))
        return self._results()

    def _results(self):
        for b, w in enumerate(self.windows):          # riders keep their final point in `last`
            if w is not None:
                self.last[b] = w.result.x
        return [w.result if w is not None else None for w in self.windows]

    def _start_group(self, gi):
        req = [(int(b), self.windows[b]) for b in self.groups[gi] if self.windows[b] is not None and self.windows[b].request is not None]
        if not req:
            self._inflight[gi] = None
            return
        m = np.zeros(self.B, bool)
        for b, w in req:
            self.last[b] = w.request
            m[b] = True
        self.n_batch_evals += 1; self.n_window_evals += len(req)
        self.launch(gi, self.last, m)
        self._inflight[gi] = req

    def _run_pipelined(self):
        for gi in range(len(self.groups)):
            self._start_group(gi)
        while any(r is not None for r in self._inflight):
            for gi in range(len(self.groups)):
                req = self._inflight[gi]
                if req is None:
                    continue
                v, g = self.collect(gi)                # waits for group gi; the other groups' evaluations keep running
                self._inflight[gi] = None
                for b, w in req:
                    w.feed(v[self._pos[b]], g[self._pos[b]], self._single(b))
                self._start_group(gi)                  # back on the GPU before the next group is collected and fed
        return self._results()


def _info(res):
    return ScipyMinimizeInfo(fun_val=float(res.fun), success=bool(res.success), status=int(res.status), iter_num=int(res.nit),
                             hess_inv=getattr(res, 'hess_inv', None), num_fun_eval=int(res.nfev), num_jac_eval=int(res.njev),
                             num_hess_eval=0)


class BatchedMultipleLevelEINCMSolver:
    """The coarse-to-fine theta pyramid of B windows, solved level by level in lockstep on one engine context (or ``n_groups`` of them).

    Constructor keywords follow ``solver.MultipleLevelEINCMSolver`` (reference solver.py:16-126); instead of loss callables it
    takes the loss parameters (``loss_kwargs``: alpha, beta, gamma, delta, scale_to_sensor_size_method[, contrast_kind]), because
    the objective is the engine's batched loss+grad.  ``set_datasamples`` stages the B windows (one per sequence, or B independent
    windows); ``solve`` returns one result dict per window with the reference's keys (solver.py:254-267).  Calling set_datasamples /
    solve again continues every sequence with its own prior (handover), exactly as the single-window solver does window after window.
    """

    def __init__(self, n_windows, sensor_size, n_pyr_lvls, theta_opt_maxiters, loss_kwargs, theta_opt_solver_params,
                 handover_opt_maxiters=None, handover_opt_solver_params=None, handover_settings=None,
                 pyramid_downscale_method='bilinear', pyramid_upscale_method='repeat', pyramid_bases=None, device=0,
                 theta_solver_callbacks=None, n_groups=1):
        """n_groups > 1: the windows are split over that many engine contexts (HIP streams) and the lockstep is pipelined - while the
        host advances one group's line searches the other groups' evaluations run (LockstepBFGS, pipelined form)."""
        hs = handover_settings
        if hs is None:
            hs = {'use_handover': False, 'solve_handover_for_levels': [], 'use_downscaled_finest_priors': False,
                  'clip_solved_handover': False, 'alpha_handover': 0.0}
        assert all(k in hs for k in ('use_handover', 'solve_handover_for_levels', 'use_downscaled_finest_priors',
                                     'clip_solved_handover', 'alpha_handover'))
        assert len(theta_opt_maxiters) == n_pyr_lvls, 'theta_opt_maxiters should be provided for each pyramid level'
        assert theta_opt_solver_params['method'] == 'BFGS', 'the lockstep driver restates SciPy BFGS'
        self.B, self.sensor_size, self.n_pyr_lvls = int(n_windows), tuple(sensor_size), int(n_pyr_lvls)
        self.theta_opt_maxiters, self.theta_opt_solver_params = theta_opt_maxiters, theta_opt_solver_params
        self.handover_opt_maxiters = handover_opt_maxiters or {}
        self.handover_opt_solver_params = handover_opt_solver_params
        self.handover_settings = hs
        self.loss_kwargs = dict(loss_kwargs)
        self.pyramid_downscale_method, self.pyramid_upscale_method = pyramid_downscale_method, pyramid_upscale_method
        self.pyramid_bases = pyramid_bases if pyramid_bases is not None else [2] * (n_pyr_lvls - 1)
        self.callbacks = theta_solver_callbacks if theta_solver_callbacks is not None else [EmptyCallback() for _ in range(self.B)]
        self.device = device
        self.n_groups = max(1, min(int(n_groups), self.B))
        self.groups = [np.asarray(ix, dtype=int) for ix in np.array_split(np.arange(self.B), self.n_groups)]
        self.engines = []
        self.engine = None                      # the first group's context (the only one when n_groups == 1)
        self._first = True
        top = np.zeros((1, 1, 2))
        self.prior = [self._pyramid_from_top(top) for _ in range(self.B)]        # prior_theta_pyr per window
        self.n_batch_evals = 0
        self.n_window_evals = 0

    # -- pyramids (solver.py:132-151, :350-377) ---------------------------------------------------------------------
    def _upscale(self, theta, base):
        theta = np.asarray(theta, dtype=np.float64)
        if self.pyramid_upscale_method == 'repeat':
            return np.repeat(np.repeat(theta, base, axis=0), base, axis=1)
        return rescale_theta(theta, (int(theta.shape[0] * base), int(theta.shape[1] * base)), _canon(self.pyramid_upscale_method))

    def _downscale(self, theta, base):
        theta = np.asarray(theta, dtype=np.float64)
        return rescale_theta(theta, (int(theta.shape[0] / base), int(theta.shape[1] / base)), _canon(self.pyramid_downscale_method))

    def _pyramid_from_top(self, top):
        pyr = {f'pyr_lvl_{self.n_pyr_lvls - 1}': np.array(top, copy=True)}
        for k in reversed(range(self.n_pyr_lvls - 1)):
            pyr[f'pyr_lvl_{k}'] = self._upscale(pyr[f'pyr_lvl_{k + 1}'], self.pyramid_bases[-k - 1])
        return pyr

    # -- staging -------------------------------------------------------------------------------------------------------
    def set_datasamples(self, windows):
        """windows: B tuples (xs, ys, ts, edges, edge_ts)."""
        assert len(windows) == self.B
        n_tot = max(sum(len(windows[b][0]) for b in ix) for ix in self.groups)
        R = len(np.atleast_1d(windows[0][4]))
        if not self.engines or n_tot > self._cap or R > self._cap_r:
            self.close()
            self._cap, self._cap_r = max(n_tot, 1), R
            self.engines = [Engine(self.sensor_size, self._cap, max_refs=R, max_windows=len(ix), device=self.device) for ix in self.groups]
            self.engine = self.engines[0]
        for eng, ix in zip(self.engines, self.groups):
            eng.set_windows([windows[b] for b in ix])

    def close(self):
        for eng in self.engines:
            eng.close()
        self.engines, self.engine = [], None

    def _params(self, lvl):
        kw = self.loss_kwargs
        return make_params(kw['alpha'], kw['beta'], kw['gamma'], kw['delta'], lvl, kw.get('scale_to_sensor_size_method', 'bilinear'),
                           kw.get('contrast_kind', 0))

    # -- one level: B BFGS solves in lockstep, with the reference's retries (solver.py:209-239) ---------------------------
    def _solve_level(self, k, starts):
        key = f'pyr_lvl_{k}'
        shape = starts[0].shape
        p = self._params(k)

        def fun_batch(X, mask):
            v, g, _ = self.engine.loss_grad(X.reshape((self.B,) + shape), p, active=mask)
            return v, g.reshape(self.B, -1)

        def launch(gi, X, mask):
            ix = self.groups[gi]
            self.engines[gi].loss_grad_async(X[ix].reshape((len(ix),) + shape), p, active=mask[ix])

        def collect(gi):
            v, g, _ = self.engines[gi].loss_grad_wait()
            return v, g.reshape(len(self.groups[gi]), -1)
        pipe = dict(groups=self.groups, launch=launch, collect=collect) if self.n_groups > 1 else {}
        gtol = self.theta_opt_solver_params['options']['gtol']
        extra = (self.theta_opt_solver_params.get('n_extra_attempts', {}) or {}).get(key, 0)
        x = np.stack([np.asarray(s, dtype=np.float64).reshape(-1) for s in starts])
        active = np.ones(self.B, bool)
        states = [None] * self.B
        for attempt in range(1 + extra):
            for b in range(self.B):
                if active[b]:
                    self.callbacks[b].set_cur_pyr_lvl(k)
                    self.callbacks[b].reset_opt_iter()
            drv = LockstepBFGS(fun_batch, x, self.theta_opt_maxiters[key], gtol, callbacks=[
                (lambda r, cb=self.callbacks[b], sh=shape: cb(spo.OptimizeResult(x=np.asarray(r.x).reshape(sh), fun=r.fun)))
                for b in range(self.B)], active=active, wolfe2_fallback=self.theta_opt_solver_params.get('wolfe2_fallback', True), **pipe)
            res = drv.run()
            self.n_batch_evals += drv.n_batch_evals; self.n_window_evals += drv.n_window_evals
            for b in range(self.B):
                if active[b]:
                    x[b], states[b] = res[b].x, _info(res[b])
            # another attempt from the last iterate for the windows that stopped without converging (solver.py:218-239)
            active = np.array([active[b] and (not states[b].success) and states[b].iter_num > 0 for b in range(self.B)])
            if not active.any():
                break
        return [x[b].reshape(shape) for b in range(self.B)], states

    # -- handover (solver.py:302-347) ----------------------------------------------------------------------------------------
    def _handover(self, k, opt, ho_states, ho_weights):
        key, finer = f'pyr_lvl_{k}', f'pyr_lvl_{k - 1}'
        hs = self.handover_settings
        if self._first or not hs['use_handover']:
            return [opt[b] for b in range(self.B)]
        out = []
        solve = k in hs['solve_handover_for_levels']
        if solve:
            lvl = k - 1 if k > 0 else 0
            priors = [self.prior[b][finer if k > 0 else key] for b in range(self.B)]
            thetas = [self._upscale(opt[b], self.pyramid_bases[-k]) if k > 0 else opt[b] for b in range(self.B)]
            p = self._params(lvl)
            limits = tuple(hs.get('handover_limits', (0.0, 1.0)))
            a_cur = np.full(self.B, 0.5)
            hkey = f'pyr_lvl_{lvl}'
            for b in range(self.B):                 # a scalar L-BFGS-B per window; the other windows ride along at their weight
                def f(a, b=b):
                    aa = a_cur.copy(); aa[b] = float(np.asarray(a).reshape(-1)[0])
                    self.n_batch_evals += 1
                    gi = next(g for g, ix in enumerate(self.groups) if b in ix)      # the window's own context; its group rides along
                    ix = self.groups[gi]
                    v, dv = self.engines[gi].handover_loss_grad(aa[ix], np.stack([priors[i] for i in ix]),
                                                                np.stack([thetas[i] for i in ix]), p, want_grad=True)
                    k_loc = int(np.where(ix == b)[0][0])
                    return float(v[k_loc]), np.array([dv[k_loc]])
                r = spo.minimize(f, np.array([0.5]), jac=True, method=self.handover_opt_solver_params['method'],
                                 bounds=spo.Bounds([limits[0]], [limits[1]]),
                                 options={'gtol': self.handover_opt_solver_params['options']['gtol'],
                                          'maxiter': self.handover_opt_maxiters[hkey]})
                w = float(r.x[0])
                if hs['clip_solved_handover']:
                    w = float(np.clip(w, *hs['clip_solved_handover_limits']))
                a_cur[b] = w
                ho_states[b][key] = _info(r)
                ho_weights[b][key] = w
        for b in range(self.B):
            a = ho_weights[b][key] if solve else hs['alpha_handover']
            ho_weights[b][key] = a
            out.append(a * self.prior[b][key] + (1 - a) * opt[b])
        return out

    # -- solve (solver.py:197-267) -----------------------------------------------------------------------------------------------
    def solve(self):
        hs = self.handover_settings
        B, top = self.B, f'pyr_lvl_{self.n_pyr_lvls - 1}'
        if hs['use_downscaled_finest_priors']:
            for b in range(B):
                for k in range(1, self.n_pyr_lvls):
                    self.prior[b][f'pyr_lvl_{k}'] = self._downscale(self.prior[b][f'pyr_lvl_{k - 1}'], self.pyramid_bases[-(k - 1) - 1])
        for cb in self.callbacks:
            cb.reset()
        pre_opt = [self._pyramid_from_top(np.zeros((1, 1, 2))) for _ in range(B)]
        for b in range(B):
            pre_opt[b][top] = self.prior[b][top]
        opt = [dict() for _ in range(B)]
        ho_opt = [dict() for _ in range(B)]
        states = [dict() for _ in range(B)]
        ho_states = [dict() for _ in range(B)]
        ho_weights = [{f'pyr_lvl_{k}': 0.5 for k in range(self.n_pyr_lvls)} for _ in range(B)]
        for k in reversed(range(self.n_pyr_lvls)):
            key, nxt = f'pyr_lvl_{k}', f'pyr_lvl_{k - 1}'
            xs, sts = self._solve_level(k, [pre_opt[b][key] for b in range(B)])
            for b in range(B):
                opt[b][key], states[b][key] = xs[b], sts[b]
            hov = self._handover(k, [opt[b][key] for b in range(B)], ho_states, ho_weights)
            for b in range(B):
                ho_opt[b][key] = hov[b]
                if k != 0:
                    pre_opt[b][nxt] = self._upscale(hov[b], self.pyramid_bases[-k])
        results = []
        for b in range(B):
            results.append({'prior_theta_pyr': dict(self.prior[b]), 'pre_opt_theta_pyr': pre_opt[b], 'theta_opt_state_pyr': states[b],
                            'pre_handover_theta_pyr': opt[b], 'ho_opt_state_pyr': ho_states[b],
                            'final_handover_weight_pyr': ho_weights[b], 'final_theta_pyr': ho_opt[b]})
            self.prior[b] = dict(ho_opt[b])
        self._first = False
        return results
