"""Multi-GPU decomposition of the EINCM path: independent event windows shard across ranks (one process per GPU);
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
