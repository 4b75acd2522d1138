#!/usr/bin/env python3
"""bench.py — EINCM loss+grad throughput on MI355X (contract: see the task statement / DESIGN.md "Measurement").

Two decompositions, one JSON line each:

--mode windows (default; BASELINE.json config C4, the configuration the metric "warped-events/sec/GPU ... 1e6 events @ 346x260"
    is quoted on).  A step = ONE value_and_grad(loss_func) evaluation of this rank's batch of independent event windows, inputs
    resident in HBM, theta changed every call.  Workload per GPU: 8 MVSEC-shape windows (260x346), 1e6 events each, 5 reference
    times, 2-DoF theta, full EINCM objective (contrast + edge correlation), alpha=20 beta=35.  Weak scaling: every rank holds its
    own 8 windows.  Windows are independent objects, so there is NO data-path collective; for N > 1 the timed step additionally
    all-reduces the scalar batch loss over RCCL (the one collective north_star names), issued asynchronously so that it overlaps
    the next step; the same K steps are also timed without it and reported beside (`no_collective`).
    Kernel times (`roofline`): HIP events attached to the launches on the engine's stream.  Inside the timed region only the longest
    event kernel carries them (which one that is, is measured during the spin-up), on every 4th step (a timed launch costs ~6 us of a
    240 us step: 0.2395 vs 0.2359 ms/step with every / every 4th launch timed, same box); the other one is timed in a second pass of
    the same K steps, every launch.  Marker events around every kernel would cost 10 % of a step ("cost of measuring").

--mode event-sharded (BASELINE.json config C5: 480x640, 1e7 events, R = 3, theta pyramid 1..16).  The events of ONE window are
    split over the ranks (sharding.ShardedEngine): per evaluation one all-reduce(sum) of the int64 IWE accumulator in HBM and one
    of the small gradient.  A step = one loss+grad evaluation at pyramid level (k mod 5); strong scaling (total work fixed).
    A 50-iteration BFGS solve over the pyramid is timed beside it (`solve_50_iters_s`).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
    EINCM_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --mode event-sharded   (1-GPU rehearsal)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
ROUND = 'r03'
TIMED_EVERY = int(os.environ.get('EINCM_BENCH_TIMED_EVERY', '4'))             # the dominant kernel carries HIP timing events on every 4th step of the timed region


def algorithmic_bytes(N, R, H, W, dense_theta):
    """minimum compulsory HBM traffic of one evaluation of one window (events packed 8 B, read once
    forward + once backward; IWE write/read, edge read, dL/dIWE write/read; Theta/grad for dense theta)."""
    return 2 * 8 * N + R * H * W * 4 * 5 + (H * W * 2 * 4 * 2 if dense_theta else 0)


def event_kernel_algorithmic_bytes(N, R, H, W):
    """One event kernel (k_splat: events in once, 8 B each, + the R IWE images out; k_gather: events in once + the R dL/dIWE
    images in): 8 N + 4 R H W per window."""
    return 8 * N + 4 * R * H * W


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', choices=('windows', 'event-sharded'), default='windows')
    ap.add_argument('--windows-per-gpu', type=int, default=8)
    ap.add_argument('--events', type=int, default=None, help='events per window (default 1e6; event-sharded: 1e7)')
    ap.add_argument('--refs', type=int, default=None, help='reference times (default 5; event-sharded: 3)')
    ap.add_argument('--sensor', type=str, default=None, help='HxW (default 260x346; event-sharded: 480x640)')
    ap.add_argument('--theta', type=str, default='1x1', help='hxw of theta, or "dense" (windows mode)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-latency', action='store_true')
    ap.add_argument('--solve-iters', type=int, default=50, help='event-sharded: BFGS iterations of the end-to-end leg (0 = skip)')
    ap.add_argument('--groups', type=int, default=1, help='contexts (HIP streams) the windows of a rank are spread over')
    a = ap.parse_args(argv)
    sharded = a.mode == 'event-sharded'
    if a.events is None:
        a.events = 10_000_000 if sharded else 1_000_000
    if a.refs is None:
        a.refs = 3 if sharded else 5
    if a.sensor is None:
        a.sensor = '480x640' if sharded else '260x346'
    return a


def init_dist(a):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    import torch
    import torch.distributed as dist
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev                 # == local_rank on a real node; lets 2 ranks rehearse on a 1-GPU box
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    backend = os.environ.get('EINCM_BENCH_BACKEND', 'nccl')      # 'gloo' only for the single-GPU rehearsal
    red_dev = dev if backend == 'nccl' else torch.device('cpu')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    import __graft_entry__ as ge
    if world > 1:
        # one builder per node (hipcc writes the .so in place); everyone else loads it after the barrier
        if local_rank == 0:
            ge.build()
        dist.barrier()
    ge.build()
    return rank, world, dev_index, dev, red_dev, backend


def load_traffic(workload):
    """HBM bytes per launch of the two event kernels from the committed PMC passes of this round (profiles/traffic.json)."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get(workload, {})
    except Exception:
        return {}


def kernel_roofline(name, ms, alg_bytes, traffic):
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {'kernel': name, 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
            'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': ms, 'traffic': traffic}


# =====================================================================================================================
# supplementary legs of the default run (outside the headline's timed region; each one bounded to a second or two)
# =====================================================================================================================
def _timed_evals(eng, thetas, p, n=40, spin=0.15):
    """median wall ms of eng.loss_grad over n calls (theta changes every call), after a clock spin-up."""
    t_end = time.perf_counter() + spin
    k = 0
    while time.perf_counter() < t_end:
        eng.loss_grad(thetas[k % len(thetas)], p); k += 1
    ts = []
    for k in range(n):
        t0 = time.perf_counter(); eng.loss_grad(thetas[k % len(thetas)], p); ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3)


def _event_kernel_us(engine, sensor, wins_args, R, thetas, p, n=20):
    """(k_splat us, k_gather us, step ms) of a batch with HIP events attached to both event kernels."""
    H, W = sensor
    with engine.Engine((H, W), sum(len(a[0]) for a in wins_args), max_refs=R, max_windows=len(wins_args), timing='dominant') as e:
        e.set_windows(wins_args)
        e.set_timing_period(1 << 30)                 # the step is timed without kernel events (they cost ~6 us per timed kernel) ...
        ms = _timed_evals(e, thetas, p, n=n)
        e.set_timing_period(1)                       # ... the kernels in a pass of their own, every launch
        e.timings_total(reset=True)
        for k in range(n):
            e.loss_grad(thetas[k % len(thetas)], p)
        acc, cnt = e.timings_total()
    return acc['splat'] / cnt * 1e3, acc['gather'] / cnt * 1e3, ms


def extra_legs(a, synth, engine, wins, base, dev_index, alpha, beta):
    """The configurations beside the headline that matter to a user of the path: the theta-grid batch step,
    the single-window latencies at the reference's real sizes, dense theta, the two stress thetas of SURVEY 8(d), and the C4
    end-to-end solve (lockstep batch solver against sequential solves)."""
    out = {}
    H, W = (int(v) for v in a.sensor.split('x'))
    B, N, R = a.windows_per_gpu, a.events, a.refs
    args8 = [(wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts']) for wn in wins]
    ev_bytes = B * event_kernel_algorithmic_bytes(N, R, H, W)
    try:        # ---- the same batch at a 16x16 theta (pyramid level 0: where the solver spends most of its evaluations)
        th16 = np.stack([synth.theta_near_truth(1000 + b, wn, (16, 16)) for b, wn in enumerate(wins)])
        p1 = engine.make_params(alpha, beta, 0.0, 0.0, 1)
        sp, ga, ms = _event_kernel_us(engine, (H, W), args8, R, [th16 * (1.0 + 0.01 * (k - 3)) for k in range(7)], p1)
        dom = 'k_gather' if ga >= sp else 'k_splat'
        out['ms_per_step_pyr16'] = ms
        out['pyr16'] = {'workload': f'{B}x[{H}x{W} N={N} R={R} theta=16x16]', 'k_splat_us': sp, 'k_gather_us': ga, 'dominant': dom,
                        'dominant_frac_of_hbm_peak': ev_bytes / (max(sp, ga) * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                        'warped_events_per_s': B * N * R / (ms * 1e-3)}
    except Exception as exc:          # noqa: BLE001 - a supplementary leg must not take the bench line down
        out['pyr16'] = {'error': repr(exc)[:200]}
    try:        # ---- stress thetas of SURVEY 8(d): theta = 0, and a theta that sends ~20 % of the events out of the frame
        p4 = engine.make_params(alpha, beta, 0.0, 0.0, 4)
        zero = np.zeros_like(base)
        sp0, ga0, ms0 = _event_kernel_us(engine, (H, W), args8, R, [zero], p4, n=10)
        big = base.copy()
        mag = np.linalg.norm(base.reshape(B, 2), axis=1).reshape(B, 1, 1, 1) + 1e-9
        big = base / mag * 0.45 * min(H, W)                  # |v dt| up to 0.45 min(H, W) px over the window: a fifth of the events leave
        sp1, ga1, ms1 = _event_kernel_us(engine, (H, W), args8, R, [big], p4, n=10)
        out['stress_theta'] = {'zero': {'k_splat_us': sp0, 'k_gather_us': ga0, 'ms_per_step': ms0},
                               'large_20pct_out_of_frame': {'k_splat_us': sp1, 'k_gather_us': ga1, 'ms_per_step': ms1,
                                                            'theta_px_per_window': float(0.45 * min(H, W))}}
    except Exception as exc:          # noqa: BLE001
        out['stress_theta'] = {'error': repr(exc)[:200]}
    try:        # ---- the reference's real MVSEC size: 256x336, 30 000 events, R = 5, 16x16 theta (run.sh:46,49; dataset/mvsec.yaml:1-2)
        wm = synth.make_window(77, (256, 336), 30000, 5, flow='smooth', flow_mag=8.0)
        thm = synth.theta_near_truth(77, wm, (16, 16))
        with engine.Engine((256, 336), 30000, max_refs=5, max_windows=1, device=dev_index) as e:
            e.set_window(wm['xs'], wm['ys'], wm['ts'], wm['edges'], wm['edge_ts'])
            out['eval_ms_mvsec_real_16x16'] = _timed_evals(e, [thm * (1.0 + 0.01 * k) for k in range(5)], engine.make_params(alpha, beta, 0.0, 0.0, 1))
            out['eval_ms_mvsec_real_2dof'] = _timed_evals(e, [synth.theta_near_truth(77, wm, (1, 1)) * (1.0 + 0.01 * k) for k in range(5)],
                                                          engine.make_params(alpha, beta, 0.0, 0.0, 4))
    except Exception as exc:          # noqa: BLE001
        out['eval_ms_mvsec_real_16x16'] = {'error': repr(exc)[:200]}
    try:        # ---- C3: 480x640, 10^6 events, R = 3, dense theta (the float64 theta and gradient cross PCIe: 4.9 MB each way)
        wd = synth.make_window(78, (480, 640), 1_000_000, 3, flow='smooth', flow_mag=20.0)
        thd = np.ascontiguousarray(wd['flow_gt'])
        with engine.Engine((480, 640), 1_000_000, max_refs=3, max_windows=1, device=dev_index) as e:
            e.set_window(wd['xs'], wd['ys'], wd['ts'], wd['edges'], wd['edge_ts'])
            ths = [np.ascontiguousarray(thd * (1.0 + 0.01 * k)) for k in range(3)]
            out['eval_ms_c3_dense'] = _timed_evals(e, ths, engine.make_params(alpha, beta, 0.0, 0.0, 0), n=15)
            # the same evaluations with theta and gradient resident in HBM (eincm_loss_grad_device): what an optimiser on the GPU would see
            import torch
            pd = engine.make_params(alpha, beta, 0.0, 0.0, 0)
            tds = [torch.from_numpy(t).to(f'cuda:{dev_index}') for t in ths]
            vmax = float(np.abs(thd).max() * 1.03)

            class _Dev:                      # _timed_evals calls .loss_grad(theta, p)
                @staticmethod
                def loss_grad(t, p):
                    return e.loss_grad_device(t, p, theta_abs_max=vmax)
            out['eval_ms_c3_dense_device_resident'] = _timed_evals(_Dev, tds, pd, n=15)
    except Exception as exc:          # noqa: BLE001
        out['eval_ms_c3_dense'] = {'error': repr(exc)[:200]}
    try:        # ---- C4 end to end: the 5-level solve of the 8 windows, lockstep batch solver against sequential solves
        out['c4_solve'] = c4_solve_leg(a, wins, (H, W), alpha, beta, dev_index)
        out['c4_solve_windows_per_s'] = out['c4_solve']['batched']['windows_per_s']
    except Exception as exc:          # noqa: BLE001
        out['c4_solve'] = {'error': repr(exc)[:300]}
    return out


def c4_solve_leg(a, wins, sensor, alpha, beta, dev_index, n_lvls=5, maxiter=40):
    """Pyramid 1 -> 16 with the reference's iteration budget (40, 28, 19, 11, 8; configs/main.yaml:35-50), handover off (independent
    windows): B windows by BatchedMultipleLevelEINCMSolver (one masked engine call per lockstep tick) and the same windows one after
    the other by MultipleLevelEINCMSolver (SciPy BFGS, one engine call per evaluation)."""
    from functools import partial
    sol = importlib.import_module('edge-informed-contrast-maximization_amd.solver')
    bsol = importlib.import_module('edge-informed-contrast-maximization_amd.batch_solver')
    losses = importlib.import_module('edge-informed-contrast-maximization_amd.losses')
    B = len(wins)
    args = [(wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts']) for wn in wins]
    loss = dict(alpha=alpha, beta=beta, gamma=0.0, delta=0.0, scale_to_sensor_size_method='bilinear')
    maxit = sol.growing_maxiters(n_lvls, maxiter / 5, maxiter)
    sp = {'method': 'BFGS', 'options': {'gtol': 1e-7}, 'n_extra_attempts': {'pyr_lvl_0': 1, 'pyr_lvl_1': 1}}
    runs = {}
    for n_groups in (1, 2, 4):     # > 1: several engine contexts, the lockstep pipelined (host work of one group overlaps the other's evaluation)
        bs = bsol.BatchedMultipleLevelEINCMSolver(B, sensor, n_lvls, maxit, loss, sp, pyramid_bases=[2] * (n_lvls - 1), device=dev_index,
                                                  n_groups=n_groups)
        t0 = time.perf_counter()
        bs.set_datasamples(args)
        t_st = time.perf_counter() - t0
        t0 = time.perf_counter()
        ob = bs.solve()
        t_sv = time.perf_counter() - t0
        runs[n_groups] = (t_sv, t_st, bs.n_batch_evals, bs.n_window_evals, ob)
        bs.close()
    best = min(runs, key=lambda k: runs[k][0])
    t_b, t_stage, calls_b, wins_b, out_b = runs[best]
    n_seq = min(B, 3)                                   # the sequential side on a sample of the windows (SciPy's n^3 update at 16x16 is slow)
    n_calls = [0]

    def counting(*aa, **kk):
        n_calls[0] += 1
        return losses.value_and_grad_loss_func(*aa, **kk)
    t_s, t_s2, fin_s, fin_s2, fin_b, n_calls2 = 0.0, 0.0, [], [], [], 0
    for b in range(n_seq):
        losses.engine_for(*args[b], sensor)              # staging outside the timer, like the batched side
        for upd in ('scipy', 'rank2'):                   # SciPy's BFGS as the reference runs it; the same driver with the O(n^2) update
            c0 = n_calls[0]
            s = sol.MultipleLevelEINCMSolver(n_pyr_lvls=n_lvls, theta_opt_maxiters=maxit,
                                             theta_loss_pfunc=partial(counting, n_pyr_lvls=n_lvls, sensor_size=sensor, **loss),
                                             theta_opt_solver_params={**sp, 'bfgs_update': upd}, pyramid_bases=[2] * (n_lvls - 1))
            s.set_datasample(*args[b])
            t0 = time.perf_counter()
            o = s.solve()
            dt = time.perf_counter() - t0
            fv = float(o['theta_opt_state_pyr']['pyr_lvl_0'].fun_val)
            if upd == 'scipy':
                t_s += dt; fin_s.append(fv)
            else:
                t_s2 += dt; fin_s2.append(fv); n_calls2 += n_calls[0] - c0
        fin_b.append(float(out_b[b]['theta_opt_state_pyr']['pyr_lvl_0'].fun_val))
    n_calls[0] -= n_calls2
    losses.clear_engine_cache()
    return {'workload': f'{B} independent windows, pyramid 1..16, BFGS maxiter 40/28/19/11/8 + 1 retry at levels 0, 1, handover off',
            'batched': {'seconds': t_b, 'windows_per_s': B / t_b, 'engine_calls': calls_b, 'windows_evaluated': wins_b, 'staging_s': t_stage,
                        'n_groups': best, 'seconds_by_n_groups': {str(k): v[0] for k, v in runs.items()}},
            'sequential': {'seconds_per_window': t_s / n_seq, 'windows_per_s': n_seq / t_s, 'engine_calls_per_window': n_calls[0] / n_seq,
                           'windows_timed': n_seq},
            'sequential_rank2_update': {'seconds_per_window': t_s2 / n_seq, 'windows_per_s': n_seq / t_s2,
                                        'engine_calls_per_window': n_calls2 / n_seq, 'windows_timed': n_seq},
            'speedup': (n_seq / t_s) and (B / t_b) / (n_seq / t_s),
            'speedup_over_sequential_rank2_update': (B / t_b) / (n_seq / t_s2),
            'final_loss_level0': {'sequential': fin_s, 'sequential_rank2_update': fin_s2, 'batched_same_windows': fin_b, 'batched_mean_all_windows':
                                  float(np.mean([o['theta_opt_state_pyr']['pyr_lvl_0'].fun_val for o in out_b]))},
            'final_loss_note': 'the two drivers run the same algorithm; their end points differ where a line search fails on the fp32-level noise '
                               'of the objective (BFGS status 2 at the start of a level) - which of the two then makes progress is decided by '
                               'rounding (the reference itself needs float64 for this, configs/main.yaml:34).  tests/test_gpu_batch_solver.py '
                               'compares them on windows where both converge',
            'note': 'the batched driver restates SciPy BFGS with the O(n^2) form of the inverse-Hessian update above 64 unknowns; SciPy itself '
                    'forms two n x n products per iteration (n = 512 at 16x16), which is most of the sequential time at the finest level: '
                    'sequential_rank2_update is the one-window-at-a-time driver with that cost removed (solver params bfgs_update=rank2), '
                    'i.e. what batching alone buys is speedup_over_sequential_rank2_update'}


# =====================================================================================================================
# mode: independent windows (C4)
# =====================================================================================================================
def bench_windows(a):
    import torch
    import torch.distributed as dist
    rank, world, dev_index, dev, red_dev, backend = init_dist(a)
    synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
    engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')

    H, W = (int(v) for v in a.sensor.split('x'))
    B, N, R = a.windows_per_gpu, a.events, a.refs
    dense = a.theta == 'dense'
    h, w = (H, W) if dense else (int(v) for v in a.theta.split('x'))
    alpha, beta = 20.0, 35.0            # configs/main.yaml:16-17

    # ---- synthetic windows, resident in HBM before the timed region ----
    wins = [synth.make_window(1000 * rank + b, (H, W), N, R, flow='smooth' if dense else 'constant', flow_mag=20.0)
            for b in range(B)]
    if dense:
        base = np.stack([wn['flow_gt'] for wn in wins])
    else:
        base = np.stack([synth.theta_near_truth(1000 * rank + b, wn, (h, w)) for b, wn in enumerate(wins)])
    if a.groups > 1:
        eng = engine.EngineGroup((H, W), B * N, max_refs=R, max_windows=B, n_groups=a.groups, device=dev_index, timing='dominant')
    else:
        eng = engine.Engine((H, W), B * N, max_refs=R, max_windows=B, device=dev_index, timing='dominant')
    t0 = time.perf_counter()
    eng.set_windows([(wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts']) for wn in wins])
    t_stage = time.perf_counter() - t0
    p = engine.make_params(alpha, beta, 0.0, 0.0, 4 if not dense else 0)

    def theta_at(k):                    # theta changes every call: nothing but the window constants is reusable
        return base * (1.0 + 0.01 * ((k % 7) - 3))

    def timed_region(n_steps, k0, with_allreduce):
        """n_steps evaluations between two barriers; returns (elapsed max over ranks, last values, last grads, last all-reduced loss)."""
        pending, total = [], None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if hasattr(eng, 'timings_total'):
            eng.timings_total(reset=True)     # the engine sums the per-launch HIP-event times of the timed region itself
            if hasattr(eng, 'set_timing_period'):
                eng.set_timing_period(TIMED_EVERY)     # every TIMED_EVERY-th launch of the dominant kernel carries events, the first one included
        t0 = time.perf_counter()
        for k in range(n_steps):
            v, g, _ = eng.loss_grad(theta_at(k0 + k), p)
            if with_allreduce:
                # the scalar batch loss of this step, summed over ranks; asynchronous: it overlaps the next step's kernels
                t = torch.tensor([float(v.sum())], dtype=torch.float64, device=red_dev)
                pending.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t))
                if len(pending) > 2:
                    wk, tt = pending.pop(0)
                    wk.wait()
        for wk, tt in pending:
            wk.wait()
            total = tt
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, v, g, (float(total.item()) if total is not None else float(v.sum()))

    # bring the GPU to its working clocks before anything is measured: a cold device runs the first ~100 ms about 8 % slower,
    # which a 3-step warm-up (1 ms) does not cover.  Untimed preparation, like staging; then the W warm-up steps of the contract.
    t_spin = time.perf_counter() + 0.3
    k = 0
    while time.perf_counter() < t_spin:
        eng.loss_grad(theta_at(k), p); k += 1
    # Which event kernel is the longer one is measured, not assumed: both carried timing events during the spin-up.  In the timed
    # region only that kernel does (a timed launch costs ~6 us of a step; marker events around every kernel cost 10 x that).
    dominant_stage = 'splat'
    if a.groups <= 1:
        acc0, n0 = eng.timings_total(reset=True)
        dominant_stage = 'splat' if acc0.get('splat', 0.0) >= acc0.get('gather', 0.0) else 'gather'
        eng.set_timed_kernels(splat=dominant_stage == 'splat', gather=dominant_stage == 'gather')
    for k in range(a.warmup):
        eng.loss_grad(theta_at(k), p)

    elapsed, v, g, batch_loss_all = timed_region(a.steps, a.warmup, with_allreduce=world > 1)
    stage_acc, n_timed = {}, a.steps
    if a.groups <= 1:
        stage_acc, n_timed = eng.timings_total()
        assert n_timed == -(-a.steps // TIMED_EVERY), (n_timed, a.steps)
    assert np.all(np.isfinite(v)) and np.all(np.isfinite(g)), 'non-finite loss/grad in the timed region'
    no_coll = None
    if world > 1:          # the same K steps without the collective, for comparison (not the headline)
        e2, _, _, _ = timed_region(a.steps, a.warmup, with_allreduce=False)
        no_coll = {'ms_per_step': e2 / a.steps * 1e3, 'value': world * B * N * R / (e2 / a.steps)}

    both_acc = {}
    if a.groups <= 1:                    # the same K steps once more with both event kernels timed (the other kernel's fraction)
        eng.set_timed_kernels(True, True)
        eng.set_timing_period(1)
        eng.timings_total(reset=True)
        for k in range(a.steps):
            eng.loss_grad(theta_at(a.warmup + k), p)
        both_acc, _ = eng.timings_total(reset=True)

    ms_per_step = elapsed / a.steps * 1e3
    warped = world * B * N * R           # warped events per step, all ranks
    value = warped / (elapsed / a.steps)

    # per-stage device times: a separate, untimed diagnostic pass (bracketing every kernel costs ~10 % of a step)
    diag = {}
    if rank == 0:
        eng.close()
        eng = engine.Engine((H, W), B * N, max_refs=R, max_windows=B, device=dev_index, timing=True)
        eng.set_windows([(wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts']) for wn in wins])
        nd = min(5, a.steps)
        for k in range(2 + nd):
            eng.loss_grad(theta_at(k), p)
            if k >= 2:
                for kk, vv in eng.timings().items():
                    diag[kk] = diag.get(kk, 0.0) + vv / nd

    out = None
    if rank == 0:
        wl = f'{B}x[{H}x{W} N={N} R={R} theta={a.theta}]'
        traffic = load_traffic(wl)
        ev_bytes = B * event_kernel_algorithmic_bytes(N, R, H, W)
        kern = {}
        for st in ('splat', 'gather'):
            live = st == dominant_stage
            ms = (stage_acc.get(st, 0.0) / max(n_timed, 1)) if live else (both_acc.get(st, 0.0) / a.steps)
            kern['k_' + st] = kernel_roofline('k_' + st, ms, ev_bytes, traffic.get(f'k_{st}_hbm_bytes_per_launch'))
            kern['k_' + st]['measured'] = (f'HIP events on every {TIMED_EVERY}th launch of the timed region ({n_timed} launches; a timed launch costs ~6 us of the step)' if live else
                                           'HIP events on every launch of a second pass of the same K steps (both event kernels timed)')
        dominant = kern['k_' + dominant_stage]                               # the longest kernel of the step (measured during spin-up)
        eval_bytes = B * algorithmic_bytes(N, R, H, W, dense)
        roof = dict(dominant)
        roof.update({'bound': 'hbm',
                     'bound_note': 'priced against HBM as the contract asks; the kernels are NOT HBM-bound at these sizes: both event kernels are bound by '
                                   'VALU issue (round-3 ablation: k_splat without its LDS atomics 93.1 of 94.0 us, without its tap arithmetic 83.3; '
                                   'profiles/r03/splat_bound_r03.txt, DESIGN.md section 4.2)',
                     'event_kernels': kern,
                     'lds_atomic_lane_ops_per_clk_per_cu': (9.0 * B * N * R / (kern['k_splat']['avg_launch_ms'] * 1e-3) / 256 / 2.4e9)
                     if kern['k_splat']['avg_launch_ms'] > 0 else 0.0,
                     'lds_atomic_peak_lane_ops_per_clk_per_cu': [7.2, 11.4],
                     'lds_atomic_note': '9 ds_add_u32 per warped event; peak = tools/lds_atomic_bench2.hip with the splat\'s own tap pattern at a 64-word '
                                        'row pitch: random columns, 32 distinct columns per half-wave (profiles/r03/lds_bank_pitch.txt); the kernel '
                                        'is VALU-bound, so this rate is what its arithmetic leaves room for, not what the LDS could do'})
        out = {
            'metric': 'warped-events/sec/GPU + loss+grad eval ms, 1e6 events @ 346x260',
            'value': value, 'unit': 'warped-events/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'C4 share per GPU: {wl}, EINCM contrast+edge-correlation loss+grad, alpha=20 beta=35',
                       'windows_per_gpu': B, 'events_per_window': N, 'n_refs': R, 'sensor': [H, W],
                       'theta': [h, w, 2],
                       'parallelism': f'window-parallel x{world}' + (', async RCCL all-reduce of the scalar batch loss per step' if world > 1
                                                                    else ', single GPU (no collective)')},
            'roofline': roof,
            'eval_roofline': {'achieved': eval_bytes / (ms_per_step * 1e-3) / 1e9, 'unit': 'GB/s',
                              'frac': eval_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              'algorithmic_bytes_per_step': eval_bytes,
                              'byte_model_note': 'SURVEY 8(d) prices an event at 8 B and an image pixel at 4 B; the kernels read 12 B per event '
                                                 '(u32 xy + f64 t) and flush into a u64 accumulator, so the real traffic is higher (roofline.traffic) '
                                                 'and these fractions are conservative'},
            'device_ms_per_step': round(sum(vv for k, vv in diag.items() if k != 'total'), 4),
            'device_ms_note': 'sum of the kernels\' own durations (diagnostic pass); ms_per_step minus this is host turn-around + launch gaps',
            'stage_ms_per_step': {k: round(vv, 4) for k, vv in diag.items()},
            'stage_ms_note': 'separate diagnostic pass with every kernel bracketed by HIP marker events (total includes their bubbles)',
            'set_windows_s': t_stage,
            'warped_events_per_s_per_gpu': value / world,
            'batch_loss_all_ranks': batch_loss_all,
            'deterministic': 'integer cross-workgroup accumulation; repeated evaluations are bit-identical (tests/test_gpu_determinism.py)',
        }
        if no_coll is not None:
            out['no_collective'] = no_coll
    eng.close()

    # ---- single-window latency (second half of the metric: loss+grad eval ms at 1e6 events) ----
    if rank == 0 and not a.no_latency:
        wn = wins[0]
        with engine.Engine((H, W), N, max_refs=R, max_windows=1, device=dev_index) as e1:
            e1.set_window(wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts'])
            t_spin = time.perf_counter() + 0.1            # working clocks again after the staging pause
            while time.perf_counter() < t_spin:
                e1.loss_grad(theta_at(0)[0], p)
            for k in range(3):
                e1.loss_grad(theta_at(k)[0], p)
            ts = []
            for k in range(30):
                t0 = time.perf_counter(); e1.loss_grad(theta_at(k)[0], p); ts.append(time.perf_counter() - t0)
        out['eval_ms_single_window'] = float(np.median(ts) * 1e3)
        out['warped_events_per_s_single_window'] = N * R / float(np.median(ts))

    # ---- supplementary: the same windows driven as independent solvers would drive them - several contexts per GPU, each with
    # its own free-running host thread (no join between steps).  Small kernels of one context then overlap the event kernels
    # of another.  Not the headline: per-launch kernel times are not separable under overlap, so `value`/`roofline` above stay
    # on the single-context timed region.
    if rank == 0 and not a.no_latency and B >= 4 and B % 4 == 0:
        import threading
        n_ctx, per = 4, B // 4
        engs = []
        for i in range(n_ctx):
            e = engine.Engine((H, W), per * N, max_refs=R, max_windows=per, device=dev_index)
            e.set_windows([(wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts']) for wn in wins[i * per:(i + 1) * per]])
            engs.append(e)
        bar = threading.Barrier(n_ctx + 1)
        n_free = max(a.steps, 50)

        def drive(i):
            for k in range(5):
                engs[i].loss_grad(theta_at(k)[i * per:(i + 1) * per], p)
            bar.wait()
            for k in range(n_free):
                engs[i].loss_grad(theta_at(k)[i * per:(i + 1) * per], p)
            bar.wait()

        threads = [threading.Thread(target=drive, args=(i,)) for i in range(n_ctx)]
        for t in threads:
            t.start()
        bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = (time.perf_counter() - t0) / n_free
        for t in threads:
            t.join()
        for e in engs:
            e.close()
        out['concurrent_contexts'] = {'contexts': n_ctx, 'windows_per_context': per, 'steps': n_free, 'ms_per_step': dt * 1e3,
                                      'value': B * N * R / dt, 'unit': 'warped-events/s',
                                      'note': 'same 8 windows, 4 engine contexts each driven by its own host thread without a '
                                              'join between steps; supplementary, not the headline'}

    if rank == 0 and world == 1 and not a.no_latency and not dense and a.theta == '1x1':
        out.update(extra_legs(a, synth, engine, wins, base, dev_index, alpha, beta))
    # ---- CPU baselines: ports of the reference arithmetic (the reference itself, JAX, cannot run here or on the GPU box) ----
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(wins[0], theta_at, alpha, beta, (H, W), N, R, dense)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(wn, theta_at, alpha, beta, sensor, N, R, dense):
    """CPU-C = the C / OpenMP fp64 port (oracle/eincm_ref.c) on 16 threads (the box's CPU quota; one thread per
    logical CPU only oversubscribes the quota and is no baseline: dropped in round 3); CPU-B = the torch-CPU fp64 restatement differentiated by autograd on all cores; CPU-A = the numpy oracle on
    one core.  Bounded samples (about 20 s in all)."""
    import torch
    from oracle import eincm_oracle as O
    from oracle import eincm_c_port as CP
    from oracle import eincm_torch as OT
    H, W = sensor
    cargs = (wn['xs'], wn['ys'], wn['ts'], wn['edges'], wn['edge_ts'])
    ncpu = os.cpu_count() or 1

    def time_c(threads, budget):
        CP.loss_and_grad(theta_at(0)[0], *cargs, alpha, beta, (H, W), nthreads=threads)      # warm (page-in, thread pool)
        n, t = 0, 0.0
        while t < budget and n < 64:
            t0 = time.perf_counter()
            CP.loss_and_grad(theta_at(n)[0], *cargs, alpha, beta, (H, W), nthreads=threads)
            t += time.perf_counter() - t0
            n += 1
        return n, t

    cores = min(ncpu, 16)
    n_eval, t_cpu = time_c(cores, 6.0)
    res = {'value': n_eval * N * R / t_cpu, 'unit': 'warped-events/s', 'cores': cores, 'kind': 'port',
           'sample': f'{n_eval} loss+grad evaluations of 1 window ({H}x{W}, N={N}, R={R}) by the C/OpenMP fp64 port '
                     f'(oracle/eincm_ref.c, {cores} threads), {t_cpu:.1f} s',
           'eval_ms': t_cpu / n_eval * 1e3, 'host_cores_available': ncpu}
    try:                                # CPU-B: torch fp64 forward + autograd; bounded sample: the first 1e5 events of the window
        torch.set_num_threads(cores)
        lvl = 4 if not dense else 0
        ns = min(N, 100_000)
        sargs = (wn['xs'][:ns], wn['ys'][:ns], wn['ts'][:ns], wn['edges'], wn['edge_ts'])
        n_t, t_t = 0, 0.0
        while t_t < 4.0 and n_t < 3:
            th = theta_at(n_t)[0]
            t0 = time.perf_counter()
            OT.loss_and_grad(th, *sargs, alpha, beta, 0.0, 0.0, lvl, (H, W), O.resample_matrix(th.shape[0], H, H / th.shape[0], 'bilinear'),
                             O.resample_matrix(th.shape[1], W, W / th.shape[1], 'bilinear'))
            t_t += time.perf_counter() - t0
            n_t += 1
        res['torch_cpu'] = {'value': n_t * ns * R / t_t, 'cores': cores, 'eval_ms': t_t / n_t * 1e3, 'events_in_sample': ns,
                            'sample': f'{n_t} evaluations of the first {ns} events, {t_t:.1f} s, torch {torch.__version__} fp64 forward + autograd '
                                      f'(oracle/eincm_torch.py), {cores} threads'}
    except Exception as exc:            # a baseline must not take the bench line down
        res['torch_cpu'] = {'error': repr(exc)[:200]}
    n_np, t_np = 0, 0.0
    while t_np < 3.0 and n_np < 3:
        t0 = time.perf_counter()
        O.loss_and_grad(theta_at(n_np)[0], *cargs, alpha, beta, 0.0, 0.0, 4 if not dense else 0, 5, (H, W))
        t_np += time.perf_counter() - t0
        n_np += 1
    res['numpy_1core'] = {'value': n_np * N * R / t_np, 'cores': 1, 'eval_ms': t_np / n_np * 1e3,
                          'sample': f'{n_np} evaluations, {t_np:.1f} s, oracle/eincm_oracle.py'}
    return res


# =====================================================================================================================
# mode: one window, events sharded over the ranks (C5)
# =====================================================================================================================
def bench_event_sharded(a):
    import torch
    import torch.distributed as dist
    rank, world, dev_index, dev, red_dev, backend = init_dist(a)
    synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
    engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
    sharding = importlib.import_module('edge-informed-contrast-maximization_amd.sharding')

    H, W = (int(v) for v in a.sensor.split('x'))
    N, R = a.events, a.refs
    alpha, beta, gamma = 2000.0, 4000.0, 0.0          # run.sh:104-108 (DSEC weights)
    win = synth.make_window(7, (H, W), N, R, flow='smooth', flow_mag=20.0)          # the same window on every rank
    mine = sharding.shard_events(N, rank, world)
    sl = slice(mine.start, mine.stop)
    levels = [(1, 1), (2, 2), (4, 4), (8, 8), (16, 16)]
    thetas = [synth.theta_near_truth(7, win, hw) for hw in levels]
    eng = engine.Engine((H, W), len(range(*sl.indices(N))), max_refs=R, max_windows=1, device=dev_index)
    se = sharding.ShardedEngine(eng)
    t0 = time.perf_counter()
    se.set_windows([(win['xs'][sl], win['ys'][sl], win['ts'][sl], win['edges'], win['edge_ts'])])
    t_stage = time.perf_counter() - t0

    def step(k):
        lv = k % len(levels)
        p = engine.make_params(alpha, beta, gamma, 0.0, len(levels) - 1 - lv)
        return se.loss_grad(thetas[lv] * (1.0 + 0.01 * ((k % 7) - 3)), p)

    t_spin = time.perf_counter() + 0.3
    k = 0
    while time.perf_counter() < t_spin:
        step(k); k += 1
    for k in range(a.warmup):
        step(k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        v, g = step(a.warmup + k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert np.all(np.isfinite(v)) and np.all(np.isfinite(g))

    # ---- end to end: BFGS over the pyramid on the sharded objective (every rank runs the same SciPy iterations on identical numbers) ----
    solve = None
    if a.solve_iters > 0:
        import scipy.optimize as spo
        iters = [max(1, round(a.solve_iters * f)) for f in (0.10, 0.14, 0.20, 0.26, 0.30)]      # growing with the level, summing to ~solve_iters
        theta = np.zeros((1, 1, 2))
        n_eval = 0
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for li, hw in enumerate(levels):
            if li:
                theta = np.repeat(np.repeat(theta, 2, axis=0), 2, axis=1)              # solver.py:350-352 'repeat' upscaling
            p = engine.make_params(alpha, beta, gamma, 0.0, len(levels) - 1 - li)

            def f(x, _p=p, _s=theta.shape):
                nonlocal n_eval
                vv, gg = se.loss_grad(x.reshape(_s), _p)
                n_eval += 1
                return float(vv[0]), gg[0].reshape(-1)
            r = spo.minimize(f, theta.reshape(-1), jac=True, method='BFGS', options={'maxiter': iters[li], 'gtol': 1e-7})
            theta = r.x.reshape(theta.shape)
        if world > 1:
            dist.barrier()
        solve = {'seconds': time.perf_counter() - t0, 'bfgs_iterations': int(sum(iters)), 'loss_grad_evaluations': n_eval,
                 'final_loss': float(r.fun)}

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = N * R / (elapsed / a.steps)
        iwe_bytes = R * H * W * 8
        out = {
            'metric': 'warped-events/sec/GPU + loss+grad eval ms, 1e6 events @ 346x260',
            'value': value, 'unit': 'warped-events/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'C5: 1x[{H}x{W} N={N} R={R}] theta pyramid 1..16, events sharded over {world} rank(s), '
                                   'EINCM loss+grad, alpha=2000 beta=4000',
                       'events_per_window': N, 'n_refs': R, 'sensor': [H, W], 'theta_levels': levels,
                       'parallelism': f'event-sharded x{world}: all-reduce(sum) of the int64 IWE accumulator + of the gradient per evaluation',
                       'backend': backend},
            'warped_events_per_s_per_gpu': value / world,
            'value_note': 'strong scaling: `value` is the aggregate over all ranks (N R / step time); divide by n_gpus for the per-GPU rate',
            'allreduce_bytes_per_evaluation': {'iwe_accumulator_int64': iwe_bytes, 'gradient_fp64_max': 16 * 16 * 2 * 8},
            'set_windows_s': t_stage,
            'solve_50_iters_s' if a.solve_iters == 50 else 'solve_s': solve,
            'roofline': None,
            'note': 'hardware scaling of this mode is unmeasured until an N > 1 run on a multi-GPU node exists; with EINCM_BENCH_BACKEND=gloo '
                    'the collective bounces through host memory (1-GPU rehearsal only)',
        }
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse_args()
    if a.mode == 'event-sharded':
        bench_event_sharded(a)
    else:
        bench_windows(a)


if __name__ == '__main__':
    main()
