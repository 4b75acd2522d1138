"""Python handle on one eincm_ctx (include/eincm.h): a batch of event windows resident on one MI355X."""
import ctypes as C

import numpy as np

from . import _lib as L


class EincmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'eincm error {code}: {msg}')
        self.code = code


class NonFiniteLoss(EincmError):
    pass


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def as_int16_coords(a, what='coordinates'):
    """Event coordinates as int16, the reference's wire format.  Integer arrays are range-checked and cast; float arrays
    (e.g. rectified coordinates handed straight to loss_func) are rounded half-to-even first, like ``jnp.round(xs).astype(int16)``
    in per_pix_warp (/root/reference/src/eincm/event_warpers.py:29-30).  Values outside the int16 range raise instead of wrapping."""
    a = np.asarray(a)
    if a.dtype == np.int16:
        return a
    if a.dtype.kind not in 'iuf':
        raise TypeError(f'{what} must be integer or floating point, got {a.dtype}')
    if a.dtype.kind == 'f':
        if a.size and not np.all(np.isfinite(a)):
            raise ValueError(f'{what} contain non-finite values')
        a = np.rint(a)
    if a.size and (a.min() < -32768 or a.max() > 32767):
        raise ValueError(f'{what} outside the int16 range [-32768, 32767]')
    return a.astype(np.int16)


def make_params(alpha, beta, gamma, delta, cur_pyr_lvl, method='bilinear', contrast_kind=L.CONTRAST_GRAD_MAG,
                full_aux=False):
    if isinstance(method, str):
        if method not in L.METHODS:
            raise ValueError(f'scale_to_sensor_size_method {method!r} not supported; one of {sorted(L.METHODS)}')
        method = L.METHODS[method]
    return L.Params(float(alpha), float(beta), float(gamma), float(delta), int(cur_pyr_lvl), int(method),
                    int(contrast_kind), L.PF_FULL_AUX if full_aux else 0)


class Engine:
    """One GPU context.  ``set_windows`` stages a batch of B independent windows (B = 1 for the reference's
    single-window solver, solver.py:185-194); ``loss_grad`` evaluates value_and_grad(loss_func) for all of them."""

    def __init__(self, sensor_size, max_events_total, max_refs=8, max_windows=1, device=0, timing=False):
        self._lib = L.load()
        self.H, self.W = int(sensor_size[0]), int(sensor_size[1])
        self.max_windows = int(max_windows)
        self.max_refs = int(max_refs)
        # timing: False | True (every stage bracketed by marker events, ~20 % slower) | 'dominant' (the two event kernels
        # launched with their own start/stop events, read out on demand: see set_timed_kernels)
        flags = 0 if not timing else (L.CF_TIMING_DOMINANT if timing == 'dominant' else L.CF_TIMING)
        self._ctx = self._lib.eincm_create(int(device), self.H, self.W, int(max_refs), int(max_windows),
                                           int(max_events_total), flags)
        if not self._ctx:
            raise EincmError(L.ERR_HIP, self._lib.eincm_last_error(None).decode())
        self.B = 0
        self.R = 0
        self.timing = bool(timing)
        self._io = {}                      # (h, w) -> staging buffers of loss_grad with their addresses

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_ctx', None):
            self._lib.eincm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, allow_nonfinite=False):
        if rc == L.OK:
            return
        msg = self._lib.eincm_last_error(self._ctx).decode()
        if rc == L.ERR_NONFINITE:
            if allow_nonfinite:
                return
            raise NonFiniteLoss(rc, msg)
        raise EincmError(rc, msg)

    # -- staging ----------------------------------------------------------------------------------
    def set_windows(self, windows, defer_constants=False):
        """windows: list of (xs, ys, ts, edges, edge_ts) tuples (the reference's datasample tuple).
        defer_constants: event-sharded mode (see sharding.ShardedEngine): stage only, finish with finish_constants()."""
        B = len(windows)
        R = len(np.atleast_1d(windows[0][4]))
        n = np.array([len(w[0]) for w in windows], dtype=np.int64)
        # every window's arrays go over as they are (one pointer per window): no host-side concatenation of the batch
        xs = [np.ascontiguousarray(as_int16_coords(w[0], 'xs')) for w in windows]
        ys = [np.ascontiguousarray(as_int16_coords(w[1], 'ys')) for w in windows]
        ts = [np.ascontiguousarray(np.asarray(w[2], dtype=np.float64)) for w in windows]
        edges = [np.ascontiguousarray(np.asarray(w[3], dtype=np.float64)) for w in windows]
        edge_ts = np.ascontiguousarray(np.stack([np.atleast_1d(np.asarray(w[4], dtype=np.float64)) for w in windows]))
        for b, e in enumerate(edges):
            if e.shape != (R, self.H, self.W):
                raise ValueError(f'edges must be (R,{self.H},{self.W}) per window, got {e.shape} for window {b}')
        if edge_ts.shape != (B, R):
            raise ValueError('every window needs the same number of reference times')
        for b in range(B):
            if not (len(xs[b]) == len(ys[b]) == len(ts[b])):
                raise ValueError(f'window {b}: xs, ys, ts differ in length')
        one = np.zeros(1, np.int16), np.zeros(1, np.float64)           # a valid address for empty windows
        def ptrs(arrs, dummy):
            return (C.c_void_p * B)(*[(a if a.size else dummy).ctypes.data for a in arrs])
        rc = self._lib.eincm_set_windows_ptrs(self._ctx, B, R, n.ctypes.data_as(C.POINTER(C.c_int64)),
                                              ptrs(xs, one[0]), ptrs(ys, one[0]), ptrs(ts, one[1]), ptrs(edges, one[1]),
                                              _dp(edge_ts), L.SW_DEFER_CONSTANTS if defer_constants else 0)
        self._check(rc)
        if B != self.B:
            self._io = {}
        self.B, self.R = B, R
        self.n_events = n

    def set_window(self, xs, ys, ts, edges, edge_ts):
        self.set_windows([(xs, ys, ts, edges, edge_ts)])

    # -- evaluation -------------------------------------------------------------------------------
    def loss_grad(self, theta, params, want_grad=True, want_aux=False, allow_nonfinite=True, active=None):
        """theta: (B,h,w,2) or (h,w,2) when B == 1.  Returns (value (B,), grad (B,h,w,2) | None, aux list | None).
        active: optional (B,) mask - only those windows are evaluated (the others: value NaN, gradient 0), at about their share of the cost."""
        th = theta if isinstance(theta, np.ndarray) else np.asarray(theta, dtype=np.float64)
        if th.ndim == 3:
            th = th[None]
        if th.ndim != 4 or th.shape[0] != self.B or th.shape[3] != 2:
            raise ValueError(f'theta must be ({self.B},h,w,2), got {th.shape}')
        _, h, w, _ = th.shape
        # Staging buffers per theta shape with their addresses cached: `ndarray.ctypes.data` costs ~1 us per use, three of them per
        # call were 3 of the ~5 us this wrapper added to a 70 us evaluation.  The caller gets copies (a few hundred bytes at the
        # pyramid's sizes); a dense theta goes straight through.
        small = th.size <= 8192
        if small:
            bufs = self._io.get((h, w))
            if bufs is None:
                tb, vb, gb = np.empty((self.B, h, w, 2)), np.empty(self.B), np.empty((self.B, h, w, 2))
                bufs = self._io[(h, w)] = (tb, vb, gb, tb.ctypes.data, vb.ctypes.data, gb.ctypes.data)
            tb, vb, gb, p_th, p_v, p_g = bufs
            np.copyto(tb, th)
        else:
            th = np.ascontiguousarray(th, dtype=np.float64)
            vb, gb = np.empty(self.B, dtype=np.float64), (np.empty_like(th) if want_grad else None)
            p_th, p_v, p_g = th.ctypes.data, vb.ctypes.data, (gb.ctypes.data if want_grad else None)
        aux = (L.Aux * self.B)() if want_aux else None
        if active is not None:
            act = np.ascontiguousarray(np.asarray(active).astype(np.uint8))
            if act.shape != (self.B,):
                raise ValueError(f'active must be ({self.B},), got {act.shape}')
            rc = self._lib.eincm_loss_grad_masked(self._ctx, p_th, h, w, C.byref(params), act.ctypes.data, p_v, p_g if want_grad else None, aux)
        else:
            rc = self._lib.eincm_loss_grad(self._ctx, p_th, h, w, C.byref(params), p_v, p_g if want_grad else None, aux)
        self._check(rc, allow_nonfinite)
        value = vb.copy() if small else vb
        grad = (gb.copy() if small else gb) if want_grad else None
        auxl = None
        if want_aux:
            auxl = [{k: getattr(a, k) for k, _ in L.Aux._fields_} for a in aux]
        return value, grad, auxl

    # -- asynchronous evaluation: enqueue now, collect later (several contexts in flight, see EngineGroup) ----
    def loss_grad_async(self, theta, params, want_grad=True, active=None):
        """Enqueue an evaluation and return at once (theta is copied before the call returns); ``active`` as in loss_grad."""
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
        if th.ndim == 3:
            th = th[None]
        if th.ndim != 4 or th.shape[0] != self.B or th.shape[3] != 2:
            raise ValueError(f'theta must be ({self.B},h,w,2), got {th.shape}')
        act = None
        if active is not None:
            act = np.ascontiguousarray(np.asarray(active).astype(np.uint8))
            if act.shape != (self.B,):
                raise ValueError(f'active must be ({self.B},), got {act.shape}')
        self._async = None
        self._check(self._lib.eincm_loss_grad_masked_async(self._ctx, th.ctypes.data, th.shape[1], th.shape[2], C.byref(params),
                                                           act.ctypes.data if act is not None else None, 1 if want_grad else 0))
        self._async = (th.shape, bool(want_grad))

    def loss_grad_wait(self, want_aux=False, allow_nonfinite=True):
        if getattr(self, '_async', None) is None:
            raise EincmError(L.ERR_STATE, 'eincm_loss_grad_wait without eincm_loss_grad_async')
        shape, want_grad = self._async
        self._async = None
        value = np.empty(self.B, dtype=np.float64)
        grad = np.empty(shape, dtype=np.float64) if want_grad else None
        aux = (L.Aux * self.B)() if want_aux else None
        rc = self._lib.eincm_loss_grad_wait(self._ctx, value.ctypes.data, grad.ctypes.data if want_grad else None, aux)
        self._check(rc, allow_nonfinite)
        auxl = [{k: getattr(a, k) for k, _ in L.Aux._fields_} for a in aux] if want_aux else None
        return value, grad, auxl

    # -- the two halves of an evaluation (event-sharded mode) ------------------------------------------
    def forward_iwe(self, theta, params, want_grad=True):
        """k_theta + k_splat only; returns with the IWE stack complete in HBM.  theta=None: the theta = 0 constants pass."""
        if theta is None:
            rc = self._lib.eincm_forward_iwe(self._ctx, None, 1, 1, C.byref(make_params(1, 1, 0, 0, 1)), 0)
            self._check(rc)
            return None
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
        if th.ndim == 3:
            th = th[None]
        if th.ndim != 4 or th.shape[0] != self.B or th.shape[3] != 2:
            raise ValueError(f'theta must be ({self.B},h,w,2), got {th.shape}')
        self._check(self._lib.eincm_forward_iwe(self._ctx, _dp(th), th.shape[1], th.shape[2], C.byref(params), 1 if want_grad else 0))
        return th.shape

    def finish_loss_grad(self, theta_shape, want_grad=True, want_aux=False, allow_nonfinite=True):
        value = np.empty(self.B, dtype=np.float64)
        grad = np.empty(theta_shape, dtype=np.float64) if want_grad else None
        aux = (L.Aux * self.B)() if want_aux else None
        rc = self._lib.eincm_finish_loss_grad(self._ctx, _dp(value), _dp(grad) if want_grad else None, aux)
        self._check(rc, allow_nonfinite)
        auxl = [{k: getattr(a, k) for k, _ in L.Aux._fields_} for a in aux] if want_aux else None
        return value, grad, auxl

    # -- the finishing half with the results kept in HBM (event-sharded mode over a GPU collective) ------------------
    def set_device_results(self, on=True):
        self._check(self._lib.eincm_set_device_results(self._ctx, 1 if on else 0))

    def finish_launch(self):
        self._check(self._lib.eincm_finish_launch(self._ctx))

    def grad_tensor(self, theta_shape):
        """torch view of the gradient of the launched evaluation in HBM, shaped like theta: all-reduce it in place."""
        return self._device_view(self._lib.eincm_grad_device_ptr, '<f8', 8, tuple(theta_shape))

    def finish_collect(self, theta_shape, want_grad=True, want_aux=False, allow_nonfinite=True):
        value = np.empty(self.B, dtype=np.float64)
        grad = np.empty(theta_shape, dtype=np.float64) if want_grad else None
        aux = (L.Aux * self.B)() if want_aux else None
        rc = self._lib.eincm_finish_collect(self._ctx, _dp(value), _dp(grad) if want_grad else None, aux)
        self._check(rc, allow_nonfinite)
        auxl = [{k: getattr(a, k) for k, _ in L.Aux._fields_} for a in aux] if want_aux else None
        return value, grad, auxl

    # -- theta and gradient resident in HBM (an optimiser that lives on the GPU) -------------------------------------------
    def loss_grad_device(self, theta, params, theta_abs_max=None, want_grad=True, want_aux=False, allow_nonfinite=True):
        """theta: torch float64 CUDA tensor (B,h,w,2) (or (h,w,2) when B == 1) on the engine's device.  Returns (value ndarray (B,),
        grad torch tensor like theta | None, aux list | None); nothing but the scalars crosses PCIe.  theta_abs_max: an upper bound of
        |theta| if the caller has one on the host (it only selects LDS window capacities; None = unknown)."""
        import torch
        if not (isinstance(theta, torch.Tensor) and theta.is_cuda and theta.dtype == torch.float64):
            raise TypeError('theta must be a float64 CUDA tensor')
        th = theta.contiguous()
        if th.dim() == 3:
            th = th[None]
        if th.dim() != 4 or th.shape[0] != self.B or th.shape[3] != 2:
            raise ValueError(f'theta must be ({self.B},h,w,2), got {tuple(theta.shape)}')
        grad = torch.empty_like(th) if want_grad else None
        torch.cuda.current_stream(th.device).synchronize()              # the engine's kernels run on its own stream
        value = np.empty(self.B, dtype=np.float64)
        aux = (L.Aux * self.B)() if want_aux else None
        rc = self._lib.eincm_loss_grad_device(self._ctx, C.c_void_p(th.data_ptr()), int(th.shape[1]), int(th.shape[2]), C.byref(params),
                                              -1.0 if theta_abs_max is None else float(theta_abs_max), _dp(value),
                                              C.c_void_p(grad.data_ptr()) if want_grad else None, aux)
        self._check(rc, allow_nonfinite)
        if want_grad and theta.dim() == 3:
            grad = grad[0]
        auxl = [{k: getattr(a, k) for k, _ in L.Aux._fields_} for a in aux] if want_aux else None
        return value, grad, auxl

    def finish_constants(self):
        self._check(self._lib.eincm_finish_constants(self._ctx))

    def _device_view(self, getter, typestr, itemsize, shape):
        import torch
        ptr, n = C.c_void_p(), C.c_int64()
        self._check(getter(self._ctx, C.byref(ptr), C.byref(n)))

        class _View:            # the CUDA array interface: a zero-copy torch tensor over the engine's HBM buffer
            __cuda_array_interface__ = {'shape': shape, 'typestr': typestr, 'data': (int(ptr.value), False), 'version': 2,
                                        'strides': None}
        return torch.as_tensor(_View(), device=torch.device('cuda', torch.cuda.current_device()))

    def iwe_tensor(self):
        """torch view of the IWE accumulator (B,R,H,W) in HBM, for an RCCL all-reduce(sum) between the two halves.  The engine sums
        pixel * 2^30 as 64-bit integers (exact, order-independent); the view is int64 (values stay below 2^63)."""
        return self._device_view(self._lib.eincm_iwe_device_ptr, '<i8', 8, (self.B, self.R, self.H, self.W))

    def mask_tensor(self):
        """torch view of the event-presence mask (B,H,W) uint8."""
        return self._device_view(self._lib.eincm_mask_device_ptr, '|u1', 1, (self.B, self.H, self.W))

    def handover_loss_grad(self, alpha_handover, prev_theta, theta, params, want_grad=True, allow_nonfinite=True):
        pt = np.ascontiguousarray(np.asarray(prev_theta, dtype=np.float64))
        th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
        if th.ndim == 3:
            th, pt = th[None], pt[None]
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_handover, dtype=np.float64), (self.B,)))
        _, h, w, _ = th.shape
        value = np.empty(self.B)
        dv = np.empty(self.B) if want_grad else None
        rc = self._lib.eincm_handover_loss_grad(self._ctx, _dp(a), _dp(pt), _dp(th), h, w, C.byref(params), _dp(value),
                                                _dp(dv) if want_grad else None)
        self._check(rc, allow_nonfinite)
        return value, dv

    def objectives(self, Theta):
        """compute_loss_objectives (losses.py:49-105) on full-resolution Theta (B,H,W,2); list of dicts."""
        T = np.ascontiguousarray(np.asarray(Theta, dtype=np.float64))
        if T.ndim == 3:
            T = T[None]
        if T.shape != (self.B, self.H, self.W, 2):
            raise ValueError(f'Theta must be ({self.B},{self.H},{self.W},2), got {T.shape}')
        out = (L.ObjectivesOut * self.B)()
        rc = self._lib.eincm_objectives(self._ctx, _dp(T), out)
        self._check(rc, True)
        res = []
        for o in out:
            d = {}
            for k, t in L.ObjectivesOut._fields_:
                if k in ('n_refs', '_pad'):
                    continue
                v = getattr(o, k)
                d[k] = np.array(v[:o.n_refs]) if t is L._A else float(v)
            res.append(d)
        return res

    def tiled_objectives(self, tile_size=None):
        """extract_tiles + compute_adaptive_* and their pairwise siblings (contrast_objectives.py:42-87,
        correlation_objectives.py:28-130) on the images of the last evaluation; list of dicts of (R,) arrays."""
        th, tw = (32, 42) if tile_size is None else tile_size        # contrast_objectives.py:56-59
        out = (L.TiledOut * self.B)()
        self._check(self._lib.eincm_tiled_objectives(self._ctx, int(th), int(tw), out))
        res = []
        for o in out:
            d = {'n_tiles': int(o.n_tiles)}
            for k, t in L.TiledOut._fields_:
                if t is L._A:
                    d[k] = np.array(getattr(o, k)[:o.n_refs])
            res.append(d)
        return res

    # -- the step before the path: edge smoothing (SURVEY f-4) --------------------------------------
    def inv_dist_transform(self, edge_imgs, formulation='exponential', alpha=6.0, d_sat=6.0, return_sqdist=False):
        """1 - minmax(f(d)) of the exact Euclidean distance to the nearest edge pixel (img_utils.py:229-233, :236-410).
        edge_imgs: (n,H,W) or (H,W), non-zero = edge.  Returns float64 (and the int32 squared distances if asked)."""
        e = np.asarray(edge_imgs)
        single = e.ndim == 2
        e = np.ascontiguousarray((e[None] if single else e) != 0).astype(np.uint8)
        if e.shape[1:] != (self.H, self.W):
            raise ValueError(f'edge images must be ({self.H},{self.W}), got {e.shape[1:]}')
        if formulation not in L.EDT_FORMULATIONS:
            raise NotImplementedError(f'Invalid option: formulation={formulation!r}')      # img_utils.py:383-385
        out = np.empty(e.shape, dtype=np.float64)
        sq = np.empty(e.shape, dtype=np.int32) if return_sqdist else None
        self._check(self._lib.eincm_inv_dist_transform(
            self._ctx, e.ctypes.data_as(C.POINTER(C.c_uint8)), e.shape[0], L.EDT_FORMULATIONS[formulation], float(alpha),
            float(d_sat), _dp(out), sq.ctypes.data_as(C.POINTER(C.c_int32)) if return_sqdist else None))
        if single:
            out, sq = out[0], (sq[0] if return_sqdist else None)
        return (out, sq) if return_sqdist else out

    def gaussian_blur(self, imgs, sigma):
        """cv.GaussianBlur(float64 image, ksize from sigma, BORDER_REFLECT_101) (img_utils.py:210-220)."""
        a = np.asarray(imgs, dtype=np.float64)
        single = a.ndim == 2
        a = np.ascontiguousarray(a[None] if single else a)
        if a.shape[1:] != (self.H, self.W):
            raise ValueError(f'images must be ({self.H},{self.W}), got {a.shape[1:]}')
        out = np.empty_like(a)
        self._check(self._lib.eincm_gaussian_blur(self._ctx, _dp(a), a.shape[0], float(sigma), _dp(out)))
        return out[0] if single else out

    # -- device images ----------------------------------------------------------------------------
    def iwes(self):
        a = np.empty((self.B, self.R, self.H, self.W), dtype=np.float32)
        self._check(self._lib.eincm_get_iwes(self._ctx, a.ctypes.data_as(C.POINTER(C.c_float))))
        return a

    def zero_iwe(self):
        a = np.empty((self.B, self.H, self.W), dtype=np.float32)
        self._check(self._lib.eincm_get_zero_iwe(self._ctx, a.ctypes.data_as(C.POINTER(C.c_float))))
        return a

    def image_grad(self):
        a = np.empty((self.B, self.R, self.H, self.W), dtype=np.float32)
        self._check(self._lib.eincm_get_image_grad(self._ctx, a.ctypes.data_as(C.POINTER(C.c_float))))
        return a

    def count_images(self):
        """(B,R,H,W) uint32 histogram of the rounded warped coordinates under the last evaluation's Theta."""
        a = np.empty((self.B, self.R, self.H, self.W), dtype=np.uint32)
        self._check(self._lib.eincm_get_count_images(self._ctx, a.ctypes.data_as(C.POINTER(C.c_uint32))))
        return a

    def warped_events(self, window=0):
        """(warped_xs, warped_ys), (R, n_events) float64 each: per_pix_warp of one window's events (caller's order) at every reference
        time under the last evaluation's Theta - the two per-event entries of compute_loss_objectives (losses.py:58,90-91)."""
        n = int(self.n_events[window]) if 0 <= window < self.B else 0        # (the library reports a bad index)
        wx = np.empty((self.R, n), dtype=np.float64)
        wy = np.empty((self.R, n), dtype=np.float64)
        self._check(self._lib.eincm_get_warped_events(self._ctx, int(window), _dp(wx), _dp(wy)))
        return wx, wy

    def scaled_theta(self):
        a = np.empty((self.B, self.H, self.W, 2), dtype=np.float64)
        self._check(self._lib.eincm_get_scaled_theta(self._ctx, _dp(a)))
        return a

    def timings_total(self, reset=False):
        """(dict of per-stage ms summed over the evaluations since the last reset, number of evaluations)."""
        t = L.Timings(); n = C.c_int64(0)
        self._check(self._lib.eincm_get_timings_total(self._ctx, C.byref(t), C.byref(n), 1 if reset else 0))
        d = {name: float(t.ms[i]) for i, name in enumerate(L.STAGE_NAMES)}
        d['total'] = float(t.total_ms)
        return d, int(n.value)

    def host_profile(self, reset=False):
        """(dict of host-side microseconds per evaluation phase summed since the last reset, number of evaluations)."""
        us = (C.c_double * 4)(); n = C.c_int64(0)
        self._check(self._lib.eincm_get_host_profile(self._ctx, us, C.byref(n), 1 if reset else 0))
        return dict(zip(('begin', 'launch', 'wait', 'collect'), (float(v) for v in us))), int(n.value)

    LAUNCH_POLICY_NAMES = ('seg_gather', 'seg_splat', 'seg_gather_2dof', 'seg_splat_short', 'pitch_policy', 'span_splat', 'span_gather',
                           'span_gather_2dof', 'cap_splat', 'cap_gather', 'cap_gather_2dof', 'pitch_aligned', 'splat_short')

    def launch_policy(self):
        """Diagnostic (eincm_get_launch_policy): segment lengths and pitch regime of the staged batch, window capacities of the last
        evaluation.  No counterpart in the reference."""
        out = (C.c_double * len(self.LAUNCH_POLICY_NAMES))()
        self._check(self._lib.eincm_get_launch_policy(self._ctx, out))
        return dict(zip(self.LAUNCH_POLICY_NAMES, (float(v) for v in out)))

    def set_timed_kernels(self, splat=True, gather=True):
        """timing='dominant' contexts: which event kernels carry HIP timing events from the next evaluation on."""
        self._check(self._lib.eincm_set_timed_kernels(self._ctx, 1 if splat else 0, 1 if gather else 0))

    def set_timing_period(self, period):
        """timing='dominant' contexts: timing events on every ``period``-th evaluation only."""
        self._check(self._lib.eincm_set_timing_period(self._ctx, int(period)))

    def timings(self):
        t = L.Timings()
        self._check(self._lib.eincm_get_timings(self._ctx, C.byref(t)))
        d = {n: float(t.ms[i]) for i, n in enumerate(L.STAGE_NAMES)}
        d['total'] = float(t.total_ms)
        return d


def multi_ref_weights(n_refs):
    w = np.empty(n_refs)
    rc = L.load().eincm_multi_ref_weights(int(n_refs), _dp(w))
    if rc:
        raise EincmError(rc, 'eincm_multi_ref_weights')
    return w


def resample_matrix(n_in, n_out, method='bilinear'):
    A = np.empty((n_out, n_in))
    rc = L.load().eincm_resample_matrix(int(n_in), int(n_out), L.METHODS[method], _dp(A))
    if rc:
        raise EincmError(rc, 'eincm_resample_matrix')
    return A


class EngineGroup:
    """A batch of independent windows spread over ``n_groups`` contexts of one GPU (one HIP stream each).  ``loss_grad`` enqueues
    every group's evaluation before waiting for any, so the latency-bound small kernels of one group overlap the event kernels of
    another: on MI355X the 8-window / 10^6-event batch gains 18 % (2 groups) to 25 % (4 groups) over a single context.
    Same call shapes as ``Engine`` for set_windows / loss_grad; results are concatenated in window order."""

    def __init__(self, sensor_size, max_events_total, max_refs=8, max_windows=1, n_groups=2, device=0, timing=False):
        self.n_groups = max(1, min(int(n_groups), int(max_windows)))
        per = -(-int(max_windows) // self.n_groups)
        self.engines = [Engine(sensor_size, max_events_total, max_refs=max_refs, max_windows=per, device=device, timing=timing)
                        for _ in range(self.n_groups)]
        self.H, self.W = self.engines[0].H, self.engines[0].W
        self.B = 0
        self._slices = []

    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_windows(self, windows):
        B = len(windows)
        per = -(-B // self.n_groups)
        self._slices = [slice(i, min(i + per, B)) for i in range(0, B, per)]
        for e, sl in zip(self.engines, self._slices):
            e.set_windows(windows[sl])
        self.B = B
        self.R = self.engines[0].R

    def loss_grad(self, theta, params, want_grad=True, want_aux=False, allow_nonfinite=True):
        th = np.asarray(theta, dtype=np.float64)
        if th.ndim == 3:
            th = th[None]
        if th.shape[0] != self.B:
            raise ValueError(f'theta must be ({self.B},h,w,2), got {th.shape}')
        # Every context that was launched is waited for before anything is raised: a context left in flight would refuse
        # every later call (EINCM_ERR_STATE) and the group has no other drain.  The first error wins.
        launched, outs, first_err = [], [], None
        try:
            for e, sl in zip(self.engines, self._slices):
                e.loss_grad_async(th[sl], params, want_grad)
                launched.append(e)
        except Exception as err:          # noqa: BLE001 - re-raised below, after the drain
            first_err = err
        for e in launched:
            try:
                outs.append(e.loss_grad_wait(want_aux, allow_nonfinite))
            except Exception as err:      # noqa: BLE001
                outs.append(None)
                if first_err is None:
                    first_err = err
        if first_err is not None:
            raise first_err
        value = np.concatenate([o[0] for o in outs])
        grad = np.concatenate([o[1] for o in outs]) if want_grad else None
        aux = [a for o in outs for a in o[2]] if want_aux else None
        return value, grad, aux

    def timings(self):
        """Per-stage device times summed over the groups (they overlap on the GPU: the sum exceeds the wall time)."""
        acc = {}
        for e in self.engines[:len(self._slices)]:
            for k, v in e.timings().items():
                acc[k] = acc.get(k, 0.0) + v
        return acc
