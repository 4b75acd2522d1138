"""The step after the path: report metrics of a solved theta (EVAL phase of the reference).

  sparse_flow_error     restates src/evaluations/flow_eval.py:14-76 in numpy: same masks (finite and non-zero predicted
                        AND ground-truth flow, optional event mask), same keys {'errors': AEE, AREE, A{1,2,3,5,10,20}PE;
                        'counts': n_ee, n_pred, n_gt}
  evaluate_theta_array  mirrors src/evaluations/theta_eval.py:14-95: compute_loss_objectives on the evaluation events
                        (HIP engine, forward only) + flow error + the same ``evals`` keys
The masked reductions are O(H*W), run once per window, and stay on the host; every objective term comes from the GPU.
"""
import sys

import numpy as np

from . import losses

EPSN = sys.float_info.epsilon


def make_event_mask(xs, ys, sensor_size):
    """utils/event_utils.py:64-76."""
    H, W = sensor_size
    m = np.zeros((H, W), dtype=bool)
    m[np.asarray(ys).astype(np.int64), np.asarray(xs).astype(np.int64)] = True
    return m


def per_pix_theta_to_flow(theta, xs, ys, ts=None):
    """utils/theta_utils.py:40-73 (dt = 1): theta at pixels holding events, zero elsewhere."""
    theta = np.asarray(theta, dtype=np.float64)
    return theta * make_event_mask(xs, ys, theta.shape[:2])[:, :, None]


def sparse_flow_error(pred_flow, gt_flow, event_mask=None):
    """flow_eval.py:14-76."""
    pred_flow = np.asarray(pred_flow, dtype=np.float64)
    gt_flow = np.asarray(gt_flow, dtype=np.float64)
    mask_pred = (~np.isinf(pred_flow[..., 0])) & (~np.isinf(pred_flow[..., 1])) & (np.linalg.norm(pred_flow, axis=-1) > 0)
    if event_mask is not None:
        mask_pred = mask_pred & np.asarray(event_mask, dtype=bool)
    mask_gt = (~np.isinf(gt_flow[..., 0])) & (~np.isinf(gt_flow[..., 1])) & (np.linalg.norm(gt_flow, axis=-1) > 0)
    both = mask_pred & mask_gt
    pred_m, gt_m = pred_flow[both], gt_flow[both]
    ee = np.linalg.norm(pred_m - gt_m, axis=-1)
    ree = ee / (np.linalg.norm(gt_m, axis=-1) + EPSN)
    cnts = {'n_ee': int(ee.shape[0]), 'n_pred': int(mask_pred.sum()), 'n_gt': int(mask_gt.sum())}
    with np.errstate(invalid='ignore'):
        errs = {'AEE': float(ee.mean()) if ee.size else float('nan'), 'AREE': float(ree.mean()) if ee.size else float('nan')}
    for n in (1, 2, 3, 5, 10, 20):
        errs[f'A{n}PE'] = float((ee > n).sum() * 100 / (cnts['n_ee'] + EPSN))
    return {'errors': errs, 'counts': cnts}


def evaluate_theta_array(theta_array, eval_xs, eval_ys, eval_ts, edges, edge_ts, gt_flow, alpha, beta, gamma, delta,
                         sensor_size, err_eval_event_mask=None):
    """theta_eval.py:14-95 -> (evals dict, loss_obj dict).  The per-event warped coordinates (not used by the evaluation) and the IWE stay on the GPU;
    ``iwe_var`` is var(IWE at the first reference time) = flow_warp_losses[0] * var(IUE)."""
    lo = losses.compute_loss_objectives(theta_array, eval_xs, eval_ys, eval_ts, edges, edge_ts, sensor_size, warped_events=False)
    mean_rel_contrast = float(lo['rel_contrasts'].mean())
    mean_rel_corr = float(lo['rel_correlations'].mean())
    mean_rel_iwe_div = float(lo['rel_iwe_divergences'].mean())
    tot_var, theta_div = lo['theta_total_variation'], lo['theta_divergence']
    loss = alpha * (-mean_rel_contrast) + beta * (-mean_rel_corr) + gamma * tot_var + delta * mean_rel_iwe_div
    evals = {}
    if gt_flow is not None:
        fe = sparse_flow_error(per_pix_theta_to_flow(theta_array, eval_xs, eval_ys), gt_flow, err_eval_event_mask)
        evals.update(fe['errors'])
        evals.update(fe['counts'])
        evals['n_pixels'] = int(sensor_size[0] * sensor_size[1])
    evals.update({
        'loss': loss, 'iwe_var': float(lo['variances'][0]), 'mean_rel_contrast': mean_rel_contrast,
        'mean_rel_corr': mean_rel_corr, 'theta_tot_var': tot_var, 'theta_div': theta_div,
        'fwl': float(lo['flow_warp_losses'][0]), 'mean_rel_iwe_div': mean_rel_iwe_div,
        'rel_iwe_divergences': lo['rel_iwe_divergences'], 'rel_contrasts': lo['rel_contrasts'],
        'rel_correlations': lo['rel_correlations'], 'flow_warp_losses': lo['flow_warp_losses'],
        'multi_ref_weights': lo['multi_ref_weights'],
    })
    return evals, lo
