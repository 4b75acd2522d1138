#!/usr/bin/env python3
"""SOLVE + EVAL over a sequence of event windows with the reference's configuration — the experiment-level flow of
`python -m experiments.e00` (src/experiments/e00/exp_mgr.py:615-714) on the HIP engine, with synthetic windows standing in
for the MVSEC / DSEC / ECD loaders (datasets, h5py and OpenCV are not available offline).

    python examples/run_sequence.py --config-dir /path/to/Edge-Informed-Contrast-Maximization/src/experiments/e00/configs \
        dataset=mvsec des_n_events=30000 --windows 4

Without --config-dir the defaults of configs/main.yaml are used (alpha 20, beta 35, gamma 2.5e-4, 5 pyramid levels, BFGS 40 /
L-BFGS-B 20 iterations with the growing schedule, handover solved at levels [1, 0]).
"""
import argparse
import os
import sys
import time
from functools import partial

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eincm_amd                                                         # noqa: E402
from eincm_amd import config, evaluation, losses, solver as sol, staging, synth   # noqa: E402

DEFAULTS = {'alpha': 20, 'beta': 35, 'gamma': 0.00025, 'delta': 0.0, 'n_pyr_lvls': 5, 'pyramid_bases': [2, 2, 2, 2],
            'scale_theta_to_sensor_size_method': 'bilinear', 'pyramid_downscale_method': 'lanczos3', 'pyramid_upscale_method': 'repeat',
            'des_n_events': 30000, 'use_growing_maxiters': True, 'maxiters_grow_order': 1.413,
            'solver_params': {'theta_opt': {'method': 'BFGS', 'maxiter': 40, 'miniter': 8.0, 'options': {'gtol': 1e-7},
                                            'n_extra_attempts': {'pyr_lvl_0': 1, 'pyr_lvl_1': 1}},
                              'handover_opt': {'method': 'L-BFGS-B', 'maxiter': 20, 'miniter': 4.0, 'options': {'gtol': 1e-6}}},
            'handover_settings': {'use_handover': True, 'solve_handover_for_levels': [1, 0], 'use_downscaled_finest_priors': True,
                                  'handover_limits': [0.0, 1.0], 'clip_solved_handover': False, 'clip_solved_handover_limits': [0.1, 0.9],
                                  'alpha_handover': 0.67},
            'dataset': {'sensor_size': [256, 336]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config-dir', default=None)
    ap.add_argument('--windows', type=int, default=3)
    ap.add_argument('--refs', type=int, default=5)
    ap.add_argument('--flow-mag', type=float, default=4.0)
    ap.add_argument('--sequences', type=int, default=1,
                    help='> 1: that many independent sequences side by side on the lockstep batch solver (one masked engine call per tick)')
    ap.add_argument('--rank2', action='store_true', help="sequential driver with the O(n^2) BFGS update (theta_opt_solver_params['bfgs_update'] = 'rank2')")
    ap.add_argument('overrides', nargs='*')
    a = ap.parse_args()
    cfg = config.load_config(a.config_dir, 'main', a.overrides) if a.config_dir else config._wrap(DEFAULTS)
    H, W = cfg.dataset.sensor_size
    n_lvls = cfg.n_pyr_lvls
    kw = dict(alpha=cfg.alpha, beta=cfg.beta, gamma=cfg.gamma, delta=cfg.delta, n_pyr_lvls=n_lvls, sensor_size=(H, W),
              scale_to_sensor_size_method=cfg.scale_theta_to_sensor_size_method)
    sp = cfg.solver_params
    if a.sequences > 1:
        return run_batched(a, cfg, H, W, n_lvls)
    if a.rank2:
        sp.theta_opt['bfgs_update'] = 'rank2'
    cb = sol.CollectingCallback()
    solver = sol.MultipleLevelEINCMSolver(
        n_pyr_lvls=n_lvls,
        theta_opt_maxiters=sol.growing_maxiters(n_lvls, sp.theta_opt.miniter, sp.theta_opt.maxiter, cfg.maxiters_grow_order, cfg.use_growing_maxiters),
        theta_loss_pfunc=partial(losses.value_and_grad_loss_func, **kw), theta_opt_solver_params=sp.theta_opt,
        handover_opt_maxiters=sol.growing_maxiters(n_lvls, sp.handover_opt.miniter, sp.handover_opt.maxiter, cfg.maxiters_grow_order, cfg.use_growing_maxiters),
        handover_loss_pfunc=partial(losses.value_and_grad_handover_loss_func, **kw), handover_opt_solver_params=sp.handover_opt,
        handover_settings=cfg.handover_settings, pyramid_downscale_method=cfg.pyramid_downscale_method,
        pyramid_upscale_method=cfg.pyramid_upscale_method, pyramid_bases=list(cfg.pyramid_bases), theta_solver_callback=cb)

    print(f'sensor {H}x{W}, {cfg.des_n_events} events/window, R={a.refs}, alpha={cfg.alpha} beta={cfg.beta} gamma={cfg.gamma}')
    scores = []
    for i in range(a.windows):
        # a "loader" sample: a time-sorted stream in microseconds around the evaluation window, fitted to des_n_events
        win = synth.make_window(900 + i, (H, W), int(cfg.des_n_events * 1.3), a.refs, flow='constant', flow_mag=a.flow_mag)
        t_us = 1_000_000.0 * (i + win['ts'])
        sl, deficiency = staging.select_events(t_us, 1_000_000.0 * i, 1_000_000.0 * (i + 1), cfg.des_n_events, True)
        sample = {'events': {'x': win['xs'][sl], 'y': win['ys'][sl], 't': t_us[sl]},
                  'image_ts': 1_000_000.0 * (i + win['edge_ts']), 'eval_ts_us': (1_000_000.0 * i, 1_000_000.0 * (i + 1))}
        xs, ys, ts, edges, edge_ts = staging.stage_datasample(sample, win['edges'])
        solver.set_datasample(xs, ys, ts, edges, edge_ts)
        t0 = time.perf_counter()
        out = solver.solve()
        t_solve = time.perf_counter() - t0
        theta = out['final_theta_pyr']['pyr_lvl_0']
        Theta = sol.rescale_theta(theta, (H, W), 'bilinear')
        ev, _ = evaluation.evaluate_theta_array(Theta, xs, ys, ts, edges, edge_ts, win['flow_gt'], cfg.alpha, cfg.beta, cfg.gamma, cfg.delta,
                                                (H, W), evaluation.make_event_mask(xs, ys, (H, W)))
        n_it = sum(st.iter_num for st in out['theta_opt_state_pyr'].values())
        ho = {k: round(float(v), 3) for k, v in out['final_handover_weight_pyr'].items() if k in out['ho_opt_state_pyr']}
        print(f'window {i}: {n_it} BFGS iterations in {t_solve*1e3:.1f} ms | loss {ev["loss"]:.4f} FWL {ev["fwl"]:.4f} AEE {ev["AEE"]:.3f} '
              f'A3PE {ev["A3PE"]:.1f}% | solved handover weights {ho}')
        scores.append((ev['fwl'], ev['AEE'], t_solve))
    s = np.array(scores)
    print(f'mean FWL {s[:,0].mean():.4f}  mean AEE {s[:,1].mean():.3f}  mean solve time {s[:,2].mean()*1e3:.1f} ms/window')
    losses.clear_engine_cache()


def stage_window(cfg, seed, i, refs, flow_mag, H, W):
    """One "loader" sample of a sequence, staged like exp_mgr does (time-normalised, fitted to des_n_events)."""
    win = synth.make_window(seed, (H, W), int(cfg.des_n_events * 1.3), refs, flow='constant', flow_mag=flow_mag)
    t_us = 1_000_000.0 * (i + win['ts'])
    sl, _ = staging.select_events(t_us, 1_000_000.0 * i, 1_000_000.0 * (i + 1), cfg.des_n_events, True)
    sample = {'events': {'x': win['xs'][sl], 'y': win['ys'][sl], 't': t_us[sl]},
              'image_ts': 1_000_000.0 * (i + win['edge_ts']), 'eval_ts_us': (1_000_000.0 * i, 1_000_000.0 * (i + 1))}
    return staging.stage_datasample(sample, win['edges']), win


def run_batched(a, cfg, H, W, n_lvls):
    """B sequences advance together: window i of every sequence is solved in one lockstep pyramid solve, each sequence handing over
    from its own previous window (batch_solver.BatchedMultipleLevelEINCMSolver; two engine contexts, pipelined)."""
    from eincm_amd import batch_solver as bsol
    B, sp = a.sequences, cfg.solver_params
    loss = dict(alpha=cfg.alpha, beta=cfg.beta, gamma=cfg.gamma, delta=cfg.delta, scale_to_sensor_size_method=cfg.scale_theta_to_sensor_size_method)
    bs = bsol.BatchedMultipleLevelEINCMSolver(
        B, (H, W), n_lvls,
        sol.growing_maxiters(n_lvls, sp.theta_opt.miniter, sp.theta_opt.maxiter, cfg.maxiters_grow_order, cfg.use_growing_maxiters), loss,
        dict(sp.theta_opt),
        handover_opt_maxiters=sol.growing_maxiters(n_lvls, sp.handover_opt.miniter, sp.handover_opt.maxiter, cfg.maxiters_grow_order, cfg.use_growing_maxiters),
        handover_opt_solver_params=dict(sp.handover_opt), handover_settings=dict(cfg.handover_settings),
        pyramid_downscale_method=cfg.pyramid_downscale_method, pyramid_upscale_method=cfg.pyramid_upscale_method,
        pyramid_bases=list(cfg.pyramid_bases), n_groups=min(2, B))
    print(f'{B} sequences side by side, sensor {H}x{W}, {cfg.des_n_events} events/window, R={a.refs}')
    scores = []
    for i in range(a.windows):
        staged = [stage_window(cfg, 900 + 100 * b + i, i, a.refs, a.flow_mag + 0.5 * b, H, W) for b in range(B)]
        bs.set_datasamples([st for st, _ in staged])
        t0 = time.perf_counter()
        outs = bs.solve()
        t_solve = time.perf_counter() - t0
        for b, ((xs, ys, ts, edges, edge_ts), win) in enumerate(staged):
            Theta = sol.rescale_theta(outs[b]['final_theta_pyr']['pyr_lvl_0'], (H, W), 'bilinear')
            ev, _ = evaluation.evaluate_theta_array(Theta, xs, ys, ts, edges, edge_ts, win['flow_gt'], cfg.alpha, cfg.beta, cfg.gamma, cfg.delta,
                                                    (H, W), evaluation.make_event_mask(xs, ys, (H, W)))
            scores.append((ev['fwl'], ev['AEE']))
            print(f'window {i} of sequence {b}: loss {ev["loss"]:.4f} FWL {ev["fwl"]:.4f} AEE {ev["AEE"]:.3f}')
        print(f'window {i}: {B} windows solved in {t_solve * 1e3:.1f} ms ({bs.n_batch_evals} engine calls so far)')
    bs.close()
    s = np.array(scores)
    print(f'mean FWL {s[:, 0].mean():.4f}  mean AEE {s[:, 1].mean():.3f}')
    losses.clear_engine_cache()


if __name__ == '__main__':
    main()
