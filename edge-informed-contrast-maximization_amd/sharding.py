"""Multi-GPU decomposition of the EINCM path.  Mode 1 (bench, BASELINE C4): independent event windows shard across ranks (one process per GPU);
the only exchange is the all-reduce of the scalar batch loss (RCCL over xGMI when the tensor is on a GPU, gloo on CPU).

The reference is single-process / single-device (no collective anywhere); windows are independent
inside loss+grad because the only coupling in the reference is the solver-level temporal prior
(src/eincm/solver.py:254-256,283-289), so no data-path collective is needed.
"""
import numpy as np


def shard_windows(n_windows, rank, world_size):
    """Contiguous, balanced slice of window indices owned by ``rank`` (first ``n % world`` ranks get one extra)."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside 0..{world_size - 1}')
    base, extra = divmod(int(n_windows), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return range(lo, hi)


def allreduce_batch_loss(local_values, device=None):
    """Sum of the per-window losses over every rank: the one collective of the sharded path.

    local_values: 1-D array of this rank's window losses.  Returns a Python float (identical on every rank).
    Uses torch.distributed's default process group (backend 'nccl' = RCCL on ROCm, or 'gloo' on CPU)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(np.sum(local_values))], dtype=torch.float64, device=device if device is not None else 'cpu')
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_window_losses(local_values, n_windows, rank, world_size, device=None):
    """All ranks' per-window losses in global window order (all_gather of equal-size padded slices)."""
    import torch
    import torch.distributed as dist
    per = -(-int(n_windows) // int(world_size))
    buf = torch.full((per,), float('nan'), dtype=torch.float64, device=device if device is not None else 'cpu')
    lv = np.asarray(local_values, dtype=np.float64)
    buf[:len(lv)] = torch.as_tensor(lv, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and world_size > 1:
        outs = [torch.empty_like(buf) for _ in range(world_size)]
        dist.all_gather(outs, buf)
    else:
        outs = [buf]
    res = np.empty(n_windows)
    for r, o in enumerate(outs):
        idx = shard_windows(n_windows, r, world_size)
        res[idx.start:idx.stop] = o.cpu().numpy()[:len(idx)]
    return res


# ---------------------------------------------------------------------------------------------------------------------
# Mode 2 (BASELINE C5): ONE window (or batch) whose EVENTS are split over the ranks.
# ---------------------------------------------------------------------------------------------------------------------
def shard_events(n_events, rank, world_size):
    """Contiguous (in time) slice of event indices owned by ``rank``."""
    return shard_windows(n_events, rank, world_size)


def rs_ag_sum_(dist, t, world):
    """In-place sum of ``t`` over the ranks as reduce_scatter + all_gather of equal chunks (padded with zeros to a multiple of the
    world size).  Integer sums, so the result equals all_reduce's whatever the chunking; on point-to-point xGMI the two steps are
    one-hop exchanges of 1/world of the buffer each."""
    import torch
    flat = t.reshape(-1)
    n = flat.numel()
    per = -(-n // world)
    if per * world != n:                                    # a copy; the usual sizes divide
        buf = torch.zeros(per * world, dtype=flat.dtype, device=flat.device)
        buf[:n] = flat
    else:
        buf = flat
    mine = torch.empty(per, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM)
    dist.all_gather_into_tensor(buf, mine)
    if buf is not flat:
        flat.copy_(buf[:n])


class ShardedEngine:
    """Event-sharded evaluation: every rank stages ITS slice of each window's events (edges replicated) in its own Engine.
    The IWE is additive over events (src/utils/event_utils.py:59 is a pure sum), so one exchange step suffices per
    evaluation: all-reduce(sum) of the (B,R,H,W) int64 fixed-point IWE accumulator between k_splat and the statistics pass
    (0.72 MB x R at 260x346, 2.5 MB x R at 480x640; integer, so the sum is exact and independent of the reduction order),
    then every rank finishes on the summed stack and the small (h,w,2) gradients are summed.
    The loss is identical on every rank.  Collectives go through torch.distributed's default group: backend 'nccl'
    (= RCCL over xGMI) reduces the engine's HBM buffer in place; 'gloo' (CPU rehearsal) bounces through host memory.
    """

    def __init__(self, engine, rank=None, world_size=None, iwe_collective='all_reduce', device_results=None):
        """iwe_collective: 'all_reduce' (the backend's default algorithm) or 'rs_ag' - reduce_scatter + all_gather of the int64
        accumulator, the one-hop form SURVEY 8(e) argues for on point-to-point xGMI (a few MB per evaluation are latency-bound, a ring
        pays 2 (n - 1) hops).  device_results: keep the gradient in HBM and all-reduce it there (default: whenever the backend is
        'nccl').  NOTE: the 'nccl' (= RCCL) branches and 'rs_ag' have never run on hardware - the builder had no multi-GPU node - and
        are covered only as far as a single process can (tests/test_gpu_sharded.py); gloo rehearsals take the host-bounced branches."""
        import torch.distributed as dist
        if iwe_collective not in ('all_reduce', 'rs_ag'):
            raise ValueError(f'iwe_collective {iwe_collective!r}: all_reduce or rs_ag')
        self.eng = engine
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.gpu_collectives = self.on and dist.get_backend() == 'nccl'
        self.iwe_collective = iwe_collective
        self.device_results = self.gpu_collectives if device_results is None else bool(device_results)
        if self.device_results:
            self.eng.set_device_results(True)

    def _allreduce_iwe_(self, t):
        """Sum of the (B,R,H,W) int64 accumulator over the ranks, in place."""
        import torch
        if not self.on or self.iwe_collective != 'rs_ag':
            self._allreduce_(t, self.dist.ReduceOp.SUM)
        elif self.gpu_collectives:
            rs_ag_sum_(self.dist, t, self.world)            # NEVER RUN ON HARDWARE (no multi-GPU box so far); arithmetic covered on gloo
            torch.cuda.current_stream().synchronize()
        else:                                               # gloo rehearsal: the same exchange on a host copy
            h = t.cpu()
            rs_ag_sum_(self.dist, h, self.world)
            t.copy_(h)
            torch.cuda.current_stream().synchronize()

    def _allreduce_(self, t, op):
        import torch
        if not self.on:
            return
        if self.gpu_collectives:
            self.dist.all_reduce(t, op=op)
            torch.cuda.current_stream().synchronize()
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)
            torch.cuda.current_stream().synchronize()

    def set_windows(self, local_windows):
        """local_windows: this rank's (xs, ys, ts, edges, edge_ts) per window — its slice of the events, the full edges."""
        D = self.dist
        self.eng.set_windows(local_windows, defer_constants=True)
        self._allreduce_(self.eng.mask_tensor(), D.ReduceOp.MAX)          # TV needs the global event mask
        self.eng.forward_iwe(None, None)                                   # theta = 0: partial IUE of this shard
        self._allreduce_iwe_(self.eng.iwe_tensor())
        self.eng.finish_constants()

    def loss_grad(self, theta, params, want_grad=True):
        """(value (B,), grad (B,h,w,2) | None): value identical on every rank, grad summed over ranks."""
        import copy
        import torch
        D = self.dist
        p = params
        if self.rank != 0:                                                  # the replicated TV gradient counts once
            from . import _lib as L
            p = copy.copy(params)
            p.flags = params.flags | L.PF_NO_TV_GRAD
        shape = self.eng.forward_iwe(theta, p, want_grad=want_grad)
        self._allreduce_iwe_(self.eng.iwe_tensor())
        if self.device_results:
            # the gradient stays in HBM: finish up to k_final, all-reduce the engine's own buffer in place, then one D2H copy
            self.eng.finish_launch()
            if want_grad and self.on:
                self._allreduce_(self.eng.grad_tensor(shape), D.ReduceOp.SUM)
            v, g, _ = self.eng.finish_collect(shape, want_grad=want_grad)
            return v, g
        v, g, _ = self.eng.finish_loss_grad(shape, want_grad=want_grad)
        if want_grad and self.on:                                           # host-side collective (gloo rehearsal)
            D.all_reduce(torch.from_numpy(g), op=D.ReduceOp.SUM)
        return v, g
