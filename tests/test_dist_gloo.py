"""The N > 1 path on CPU: two gloo ranks shard a batch of windows (sharding.shard_windows), evaluate their share and
all-reduce the scalar batch loss — the same helpers bench.py uses with RCCL.  No GPU here, so the per-window loss is
evaluated by the oracle (test infrastructure standing in for the engine; the product path is exercised by -m gpu)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _window_loss(i):
    from oracle import eincm_oracle as O
    synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
    H, W = 36, 48
    win = synth.make_window(200 + i, (H, W), 1500, 2, flow='constant', flow_mag=5.0)
    th = synth.theta_near_truth(200 + i, win, (1, 1))
    return O.loss_func(th, win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'], 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W))[0]


def _worker(rank, world, port, n_windows, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sh = importlib.import_module('edge-informed-contrast-maximization_amd.sharding')
    mine = sh.shard_windows(n_windows, rank, world)
    local = np.array([_window_loss(i) for i in mine])
    total = sh.allreduce_batch_loss(local)
    allv = sh.gather_window_losses(local, n_windows, rank, world)
    q.put((rank, list(mine), total, allv.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_windows_partition():
    sh = importlib.import_module('edge-informed-contrast-maximization_amd.sharding')
    for n in (0, 1, 5, 8, 64, 65):
        for world in (1, 2, 3, 8):
            parts = [list(sh.shard_windows(n, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        sh.shard_windows(4, 2, 2)


@pytest.mark.timeout(180)
def test_two_rank_gloo_matches_single_process():
    n_windows, world = 5, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_windows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    ref = np.array([_window_loss(i) for i in range(n_windows)])
    owned = sorted(sum((r[1] for r in res), []))
    assert owned == list(range(n_windows))                      # every window evaluated exactly once
    for _, _, total, allv in res:
        assert total == pytest.approx(ref.sum(), rel=1e-12)     # identical on every rank
        np.testing.assert_allclose(allv, ref, rtol=1e-12)


def _rs_ag_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sh = importlib.import_module('edge-informed-contrast-maximization_amd.sharding')
    out = []
    for shape in ((1, 5, 26, 34), (2, 3, 7, 9), (1, 1, 1, 1)):          # divisible by the world size, padded, smaller than it
        rng = np.random.default_rng(17 * rank + sum(shape))
        a = torch.from_numpy(rng.integers(-2**61 // world, 2**61 // world, size=shape, dtype=np.int64))
        ref = a.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        sh.rs_ag_sum_(dist, a, world)
        out.append(bool(torch.equal(a, ref)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize('world', [2, 3])
def test_reduce_scatter_all_gather_equals_all_reduce(world):
    """sharding.rs_ag_sum_ (the `iwe_collective='rs_ag'` form of the accumulator exchange) gives all_reduce's integer sums for sizes that
    divide by the world size, that need padding and that are smaller than it."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rs_ag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(all(v) for v in res.values()), res
