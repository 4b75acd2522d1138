"""Host-side callers of the loss+grad path: the jaxopt-shaped SciPy wrappers and the multi-level solver.

Mirrors /root/reference/src/eincm/solver.py (MultipleLevelEINCMSolver :10-383) and the two jaxopt classes it uses
(jaxopt.ScipyMinimize / ScipyBoundedMinimize, call sites solver.py:165-183,209-216,325-335; wrapper shape shown in
README.md:108-126).  jaxopt differentiates ``fun`` with JAX; the HIP engine delivers the gradient itself, so the wrappers
here are used in jaxopt's own ``value_and_grad=True`` mode: ``fun(params, *args)`` returns ``(value, grad)`` — or
``((value, aux), grad)`` with ``has_aux=True`` — which is what ``losses.value_and_grad_loss_func`` /
``losses.value_and_grad_handover_loss_func`` return.  The BFGS / L-BFGS-B iterations themselves stay in SciPy on the
host in float64, exactly as in the reference.
"""
from collections import namedtuple
from functools import partial

import numpy as np
import scipy.optimize as spo

from .engine import resample_matrix

# fields the reference reads: fun_val, success, status, iter_num (solver.py:218-221,380-384)
ScipyMinimizeInfo = namedtuple('ScipyMinimizeInfo', 'fun_val success status iter_num hess_inv num_fun_eval num_jac_eval num_hess_eval')


class ScipyMinimize:
    """jaxopt.ScipyMinimize-shaped wrapper over scipy.optimize.minimize(jac=True).

    Same constructor keywords as the reference passes (solver.py:165-173): fun, method, maxiter, jit, has_aux, options,
    callback (+ tol, dtype).  ``jit`` is accepted and ignored (nothing to trace).  ``value_and_grad`` defaults to True:
    ``fun`` must return the gradient (see module docstring).
    """

    def __init__(self, fun, method=None, maxiter=500, jit=True, has_aux=False, options=None, callback=None, tol=None,
                 dtype=np.float64, value_and_grad=True, bfgs_update='scipy', wolfe2_fallback=True):
        """bfgs_update (method 'BFGS' only): 'scipy' = scipy.optimize.minimize itself; 'rank2' = the same algorithm restated in
        batch_solver.LockstepBFGS, whose inverse-Hessian update costs O(n^2) instead of SciPy's two n x n products - identical iterates
        up to 64 unknowns, equal to rounding beyond (at a 16x16 theta, n = 512, SciPy's update takes 10 ms per iteration, the
        evaluation 0.1 ms)."""
        if not value_and_grad:
            raise ValueError('this wrapper needs fun to return (value, grad): the HIP engine cannot be differentiated by tracing')
        if bfgs_update not in ('scipy', 'rank2'):
            raise ValueError(f'bfgs_update {bfgs_update!r}: scipy or rank2')
        self.bfgs_update = bfgs_update
        self.wolfe2_fallback = bool(wolfe2_fallback)          # 'rank2' only: see batch_solver.LockstepBFGS
        self.fun, self.method, self.maxiter, self.has_aux = fun, method, int(maxiter), bool(has_aux)
        self.options = dict(options or {})
        self.callback, self.tol, self.dtype = callback, tol, dtype
        self.num_fun_eval = 0

    def _scipy_fun(self, shape, args):
        def f(x):
            out = self.fun(x.reshape(shape), *args)
            self.num_fun_eval += 1
            val, grad = out
            if self.has_aux:
                val = val[0]
            return float(val), np.asarray(grad, dtype=np.float64).reshape(-1)
        return f

    def _scipy_callback(self, shape):
        if self.callback is None:
            return None

        def cb(intermediate_result):          # SciPy >= 1.11 passes an OptimizeResult to a callback with this parameter name
            intermediate_result.x = np.asarray(intermediate_result.x).reshape(shape)
            return self.callback(intermediate_result)
        return cb

    def _minimize(self, init_params, bounds, args):
        x0 = np.asarray(init_params, dtype=self.dtype)
        shape = x0.shape
        self.num_fun_eval = 0
        if self.bfgs_update == 'rank2' and self.method == 'BFGS' and bounds is None:
            from .batch_solver import LockstepBFGS          # (imports this module)
            f = self._scipy_fun(shape, args)

            def one(X, mask):
                v, g = f(X[0])
                return np.array([v]), g[None]
            gtol = self.options.get('gtol', self.tol if self.tol is not None else 1e-5)
            res = LockstepBFGS(one, x0.reshape(1, -1).astype(np.float64), self.maxiter, gtol,
                               callbacks=[self._scipy_callback(shape)], wolfe2_fallback=self.wolfe2_fallback).run()[0]
        else:
            res = spo.minimize(self._scipy_fun(shape, args), x0.reshape(-1).astype(np.float64), jac=True, tol=self.tol,
                               bounds=bounds, method=self.method, callback=self._scipy_callback(shape),
                               options={**self.options, 'maxiter': self.maxiter})
        params = np.asarray(res.x, dtype=np.float64).reshape(shape)
        info = ScipyMinimizeInfo(fun_val=float(res.fun), success=bool(res.success), status=int(res.status),
                                 iter_num=int(res.nit), hess_inv=getattr(res, 'hess_inv', None),
                                 num_fun_eval=int(getattr(res, 'nfev', self.num_fun_eval)),
                                 num_jac_eval=int(getattr(res, 'njev', 0)), num_hess_eval=int(getattr(res, 'nhev', 0)))
        return params, info

    def run(self, init_params, *args):
        """(params, state) like jaxopt: SciPy BFGS status 0 converged / 1 maxiter / 2 precision loss / 3 NaN."""
        return self._minimize(init_params, None, args)


class ScipyBoundedMinimize(ScipyMinimize):
    """jaxopt.ScipyBoundedMinimize-shaped: run(init_params, bounds=(lb, ub), *args) (solver.py:325-335)."""

    def run(self, init_params, bounds, *args):
        lb, ub = bounds
        x0 = np.asarray(init_params, dtype=np.float64)
        b = spo.Bounds(np.broadcast_to(np.asarray(lb, dtype=np.float64), x0.shape).reshape(-1),
                       np.broadcast_to(np.asarray(ub, dtype=np.float64), x0.shape).reshape(-1))
        return self._minimize(init_params, b, args)


def growing_maxiters(n_pyr_lvls, miniter, maxiter, order=1.413, use_growing=True):
    """Per-level iteration budget, keys 'pyr_lvl_k' (exp_mgr.py:169-187): ceil(miniter*p^ord + maxiter*(1-p)^ord)."""
    out = {}
    for k in range(n_pyr_lvls):
        p = k / (n_pyr_lvls - 1) if n_pyr_lvls > 1 else 0.0
        out[f'pyr_lvl_{k}'] = int(np.ceil(miniter * p ** order + maxiter * (1 - p) ** order)) if use_growing else int(maxiter)
    return out


def rescale_theta(theta, out_hw, method):
    """jax.image.scale_and_translate(theta, (h', w', 2), spatial_dims=(0,1,2), scale=(h'/h, w'/w, 1)) as used by
    _upscale_theta / _downscale_theta (solver.py:350-377); the channel axis goes through the same kernel at scale 1."""
    theta = np.asarray(theta, dtype=np.float64)
    h, w, _ = theta.shape
    A_H = resample_matrix(h, out_hw[0], method)
    A_W = resample_matrix(w, out_hw[1], method)
    A_C = resample_matrix(2, 2, method)
    return np.einsum('yi,xj,dc,ijc->yxd', A_H, A_W, A_C, theta)


class EmptyCallback:
    """Bookkeeping interface the solver drives (callbacks.py:8-96): no collection."""

    def __init__(self):
        self.iters = {}
        self.cur_key = None

    def reset(self):
        self.iters = {}

    def set_cur_pyr_lvl(self, pyr_lvl):
        self.cur_key = f'pyr_lvl_{pyr_lvl}'
        self.iters.setdefault(self.cur_key, 0)

    def reset_opt_iter(self):
        pass

    def get_iters(self):
        return self.iters

    def set_prior_and_current_thetas(self, prior_theta, theta):
        pass

    def __call__(self, intermediate_result):
        self.iters[self.cur_key] = self.iters.get(self.cur_key, 0) + 1


class CollectingCallback(EmptyCallback):
    """Collects theta_k / loss_k per level like EINCMThetaSolverCallback (callbacks.py:131-151)."""

    def __init__(self):
        super().__init__()
        self.thetas, self.losses = {}, {}

    def reset(self):
        super().reset()
        self.thetas, self.losses = {}, {}

    def __call__(self, intermediate_result):
        super().__call__(intermediate_result)
        self.thetas.setdefault(self.cur_key, []).append(np.array(intermediate_result.x, copy=True))
        self.losses.setdefault(self.cur_key, []).append(float(intermediate_result.fun))


class MultipleLevelEINCMSolver:
    """Coarse-to-fine theta pyramid solver; same constructor keywords, state and result dict as the reference
    (solver.py:16-126, solve :197-267, handover :302-347).  ``theta_loss_pfunc`` / ``handover_loss_pfunc`` are
    functools.partial objects over the value-and-grad callables of losses.py with everything bound except
    (theta, xs, ys, ts, edges, edge_ts, cur_pyr_lvl) / (alpha_handover, prev_theta, theta, ..., cur_pyr_lvl)."""

    _SCALE_METHODS = ['linear', 'bilinear', 'trilinear', 'cubic', 'bicubic', 'tricubic', 'lanczos3', 'lanczos5']

    def __init__(self, n_pyr_lvls, theta_opt_maxiters, theta_loss_pfunc, theta_opt_solver_params, handover_opt_maxiters=None,
                 handover_loss_pfunc=None, handover_opt_solver_params=None, handover_settings=None,
                 pyramid_downscale_method='bilinear', pyramid_upscale_method='repeat', pyramid_bases=None,
                 theta_solver_callback=None, handover_solver_callback=None):
        hs = handover_settings
        assert hs is None or all(k in hs for k in ('use_handover', 'solve_handover_for_levels', 'use_downscaled_finest_priors',
                                                   'clip_solved_handover', 'alpha_handover')), \
            'handover_settings should contain: use_handover, solve_handover_for_levels, use_downscaled_finest_priors, ' \
            'clip_solved_handover, alpha_handover'
        assert hs is None or (not hs['clip_solved_handover']) or len(hs.get('clip_solved_handover_limits', ())) == 2, \
            'valid clip_solved_handover_limits should be provided if clip_solved_handover is set to True'
        assert theta_opt_maxiters is not None and len(theta_opt_maxiters) == n_pyr_lvls, \
            'theta_opt_maxiters should be provided for each pyramid level'
        assert handover_opt_maxiters is None or len(handover_opt_maxiters) == n_pyr_lvls, \
            'handover_opt_maxiters should be provided for each pyramid level'
        assert pyramid_upscale_method in ['repeat'] + self._SCALE_METHODS, f'Invalid pyramid_upscale_method: "{pyramid_upscale_method}"'
        assert pyramid_downscale_method in self._SCALE_METHODS, f'Invalid pyramid_downscale_method: "{pyramid_downscale_method}"'
        if hs is None:
            hs = {'use_handover': False, 'solve_handover_for_levels': [], 'use_downscaled_finest_priors': False,
                  'clip_solved_handover': False, 'alpha_handover': 0.0}
        self.n_pyr_lvls = n_pyr_lvls
        self.theta_opt_maxiters = theta_opt_maxiters
        self.theta_opt_solver_params = theta_opt_solver_params
        self.theta_loss_pfunc = theta_loss_pfunc
        self.handover_opt_maxiters = handover_opt_maxiters if handover_opt_maxiters is not None else {}
        self.handover_opt_solver_params = handover_opt_solver_params
        self.handover_loss_pfunc = handover_loss_pfunc
        self.handover_settings = hs
        self.use_handover = hs['use_handover']
        self.solve_handover_switch_per_level = {f'pyr_lvl_{k}': (k in hs['solve_handover_for_levels']) for k in range(n_pyr_lvls)}
        self.use_downscaled_finest_priors = hs['use_downscaled_finest_priors']
        self.clip_solved_handover = hs['clip_solved_handover']
        self.clip_solved_handover_limits = hs['clip_solved_handover_limits'] if self.clip_solved_handover else None
        self.alpha_handover = hs['alpha_handover']
        self.pyramid_downscale_method = pyramid_downscale_method
        self.pyramid_upscale_method = pyramid_upscale_method
        self.pyramid_bases = pyramid_bases if pyramid_bases is not None else [2] * (n_pyr_lvls - 1)
        self.theta_solver_callback = theta_solver_callback if theta_solver_callback is not None else EmptyCallback()
        self.handover_solver_callback = handover_solver_callback if handover_solver_callback is not None else EmptyCallback()

        self.pre_opt_theta_pyr, self.opt_theta_pyr, self.handover_opt_theta_pyr, self.prior_theta_pyr = {}, {}, {}, {}
        self._initialize_theta_pyramids()
        self.init_handover_weight_pyr = {f'pyr_lvl_{k}': 0.5 for k in range(n_pyr_lvls)}
        self.final_handover_weight_pyr = {f'pyr_lvl_{k}': 0.5 for k in range(n_pyr_lvls)}
        self.single_lvl_theta_solvers, self.single_lvl_handover_solvers = {}, {}
        self._construct_solver_for_each_pyramid_level()
        self.datasample = {}
        self._IS_FIRST_SAMPLE = True
        self.theta_opt_state_pyr, self.ho_opt_state_pyr = {}, {}

    def not_first_sample(self):
        self._IS_FIRST_SAMPLE = False

    # ---- pyramids (solver.py:132-151) -------------------------------------------------------------------------
    def _initialize_theta_pyramids(self, theta_pyr_init=None):
        top = f'pyr_lvl_{self.n_pyr_lvls - 1}'
        for pyr in (self.pre_opt_theta_pyr, self.opt_theta_pyr, self.handover_opt_theta_pyr):
            pyr[top] = np.zeros((1, 1, 2))
        if theta_pyr_init is not None:
            self.prior_theta_pyr = theta_pyr_init
        else:
            self.prior_theta_pyr[top] = np.zeros((1, 1, 2))
        for k in reversed(range(self.n_pyr_lvls - 1)):
            key, coarser = f'pyr_lvl_{k}', f'pyr_lvl_{k + 1}'
            base = self.pyramid_bases[-k - 1]
            for pyr in (self.pre_opt_theta_pyr, self.opt_theta_pyr, self.handover_opt_theta_pyr):
                pyr[key] = self._upscale_theta(pyr[coarser], base=base)
            if theta_pyr_init is None:
                self.prior_theta_pyr[key] = self._upscale_theta(self.prior_theta_pyr[coarser], base=base)

    def _construct_solver_for_each_pyramid_level(self):
        for k in range(self.n_pyr_lvls):
            key = f'pyr_lvl_{k}'
            self.single_lvl_theta_solvers[key] = ScipyMinimize(
                fun=partial(self.theta_loss_pfunc, cur_pyr_lvl=k), method=self.theta_opt_solver_params['method'],
                maxiter=self.theta_opt_maxiters[key], jit=True, has_aux=True,
                options={'gtol': self.theta_opt_solver_params['options']['gtol'], 'return_all': True},
                callback=self.theta_solver_callback, bfgs_update=self.theta_opt_solver_params.get('bfgs_update', 'scipy'),
                wolfe2_fallback=self.theta_opt_solver_params.get('wolfe2_fallback', True))
            if self.handover_loss_pfunc is not None and self.handover_opt_solver_params is not None:
                self.single_lvl_handover_solvers[key] = ScipyBoundedMinimize(
                    fun=partial(self.handover_loss_pfunc, cur_pyr_lvl=k), method=self.handover_opt_solver_params['method'],
                    maxiter=self.handover_opt_maxiters[key], jit=True, has_aux=False,
                    options={'gtol': self.handover_opt_solver_params['options']['gtol']},
                    callback=self.handover_solver_callback)

    def set_datasample(self, xs, ys, ts, edges, edge_ts):
        self.datasample = {'events': {'x': xs, 'y': ys, 't': ts}, 'edges': edges, 'edge_ts': edge_ts}

    def _args(self):
        d = self.datasample
        return (d['events']['x'], d['events']['y'], d['events']['t'], d['edges'], d['edge_ts'])

    # ---- solve (solver.py:197-267) --------------------------------------------------------------------------------
    def solve(self, verbose=False):
        self._pre_solve()
        extra = self.theta_opt_solver_params.get('n_extra_attempts', {}) or {}
        for k in reversed(range(self.n_pyr_lvls)):
            key, next_key = f'pyr_lvl_{k}', f'pyr_lvl_{k - 1}'
            self._update_callback_pyr_lvl(k)
            n_extra = 0
            solver = self.single_lvl_theta_solvers[key]
            self.opt_theta_pyr[key], self.theta_opt_state_pyr[key] = solver.run(self.pre_opt_theta_pyr[key], *self._args())
            while ((not self.theta_opt_state_pyr[key].success) and self.theta_opt_state_pyr[key].iter_num > 0
                   and key in extra and n_extra < extra[key]):
                n_extra += 1
                self.opt_theta_pyr[key], self.theta_opt_state_pyr[key] = solver.run(self.opt_theta_pyr[key], *self._args())
            self.handover_opt_theta_pyr[key] = self._perform_handover_at_level(k)
            if k != 0:
                self.pre_opt_theta_pyr[next_key] = self._upscale_theta(self.handover_opt_theta_pyr[key], base=self.pyramid_bases[-k])
            if verbose:
                st = self.theta_opt_state_pyr[key]
                print(f'{key} done | loss={st.fun_val:8.4f} status={st.status} n_iters={st.iter_num}')
        old_prior = dict(self.prior_theta_pyr)
        self.prior_theta_pyr = dict(self.handover_opt_theta_pyr)
        self._IS_FIRST_SAMPLE = False
        return {
            'prior_theta_pyr': old_prior,
            'pre_opt_theta_pyr': dict(self.pre_opt_theta_pyr),
            'theta_opt_state_pyr': dict(self.theta_opt_state_pyr),
            'pre_handover_theta_pyr': dict(self.opt_theta_pyr),
            'ho_opt_state_pyr': dict(self.ho_opt_state_pyr),
            'final_handover_weight_pyr': dict(self.final_handover_weight_pyr),
            'final_theta_pyr': dict(self.handover_opt_theta_pyr),
        }

    def _pre_solve(self):
        self._stage_prior_theta_pyr()
        top = f'pyr_lvl_{self.n_pyr_lvls - 1}'
        self.pre_opt_theta_pyr[top] = self.prior_theta_pyr[top]
        self.theta_solver_callback.reset()
        self.handover_solver_callback.reset()
        self.theta_opt_state_pyr, self.ho_opt_state_pyr = {}, {}

    def _stage_prior_theta_pyr(self):
        if self.use_downscaled_finest_priors:
            for k in range(1, self.n_pyr_lvls):
                key, finer = f'pyr_lvl_{k}', f'pyr_lvl_{k - 1}'
                self.prior_theta_pyr[key] = self._downscale_theta(self.prior_theta_pyr[finer], base=self.pyramid_bases[-(k - 1) - 1])

    def _update_callback_pyr_lvl(self, pyr_lvl):
        for cb in (self.theta_solver_callback, self.handover_solver_callback):
            cb.set_cur_pyr_lvl(pyr_lvl)
            cb.reset_opt_iter()

    # ---- handover (solver.py:302-347) -------------------------------------------------------------------------------
    def _perform_handover_at_level(self, k):
        key, finer = f'pyr_lvl_{k}', f'pyr_lvl_{k - 1}'
        self.handover_solver_callback.set_prior_and_current_thetas(self.prior_theta_pyr[key], self.opt_theta_pyr[key])
        if self._IS_FIRST_SAMPLE or not self.use_handover:
            return self.opt_theta_pyr[key]
        if self.solve_handover_switch_per_level[key]:
            if k > 0:       # solved at the finer resolution, because the up-scaling follows the handover
                prior_theta = self.prior_theta_pyr[finer]
                theta = self._upscale_theta(self.opt_theta_pyr[key], self.pyramid_bases[-k])
                ho_solver = self.single_lvl_handover_solvers[finer]
            else:
                prior_theta, theta, ho_solver = self.prior_theta_pyr[key], self.opt_theta_pyr[key], self.single_lvl_handover_solvers[key]
            limits = tuple(self.handover_settings.get('handover_limits', (0.0, 1.0)))
            w, self.ho_opt_state_pyr[key] = ho_solver.run(self.init_handover_weight_pyr[key], limits, prior_theta, theta, *self._args())
            w = float(np.asarray(w).reshape(-1)[0])
            if self.clip_solved_handover:
                w = float(np.clip(w, *self.clip_solved_handover_limits))
            self.final_handover_weight_pyr[key] = w
        else:
            self.final_handover_weight_pyr[key] = self.alpha_handover
        a = self.final_handover_weight_pyr[key]
        return a * self.prior_theta_pyr[key] + (1 - a) * self.opt_theta_pyr[key]

    # ---- pyramid resampling (solver.py:350-377) ------------------------------------------------------------------
    def _upscale_theta(self, theta, base=2):
        theta = np.asarray(theta, dtype=np.float64)
        if self.pyramid_upscale_method == 'repeat':
            return np.repeat(np.repeat(theta, base, axis=0), base, axis=1)
        return rescale_theta(theta, (int(theta.shape[0] * base), int(theta.shape[1] * base)), _canon(self.pyramid_upscale_method))

    def _downscale_theta(self, theta, base=2):
        theta = np.asarray(theta, dtype=np.float64)
        return rescale_theta(theta, (int(theta.shape[0] / base), int(theta.shape[1] / base)), _canon(self.pyramid_downscale_method))


def _canon(method):
    return {'linear': 'bilinear', 'trilinear': 'bilinear', 'bicubic': 'cubic', 'tricubic': 'cubic'}.get(method, method)
