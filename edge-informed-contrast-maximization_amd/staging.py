"""The step before the path: turning a loader's raw window into the staged tuple the loss consumes.

Restates, for numpy inputs, the two rules of the reference that define the hot path's input ranges:
  * event-count fitting ``des_n_events`` (src/dataloaders/mvsec_loader.py:272-295, dsec_loader.py:294-319): the window
    [idx_start, idx_end) of a time-sorted event stream is grown symmetrically when short (ceil/floor of the deficiency,
    clamped to the stream) and cut to the latest / earliest ``des_n_events`` events when long;
  * time normalisation (src/experiments/e00/exp_mgr.py:318-327): ``(t - t0) / (t1 - t0 + eps)`` for event and image
    timestamps, so events of the evaluation window land in [0, 1) and a grown window slightly outside it.
Edge extraction itself (OpenCV chain, src/utils/img_utils.py:131-233) is out of scope; ``edges`` must arrive as the
(R, H, W) float64 stack in [0, 1] that chain produces (exp_mgr.py:343-350); ``normalize_edges`` applies the last step.
"""
import sys

import numpy as np

EPSN = sys.float_info.epsilon


def fit_event_window(n_stream, idx_start, idx_end, des_n_events, prefer_latest_events=True):
    """Return (idx_start, idx_end, n_event_deficiency) after the reference's pad / truncate rule."""
    if des_n_events is None:
        return int(idx_start), int(idx_end), 0
    deficiency = int(des_n_events) - (int(idx_end) - int(idx_start))
    if deficiency > 0:
        idx_start = max(0, int(idx_start) - int(np.ceil(deficiency / 2)))
        idx_end = min(int(idx_end) + int(np.floor(deficiency / 2)), int(n_stream))
    elif deficiency < 0:
        if prefer_latest_events:
            idx_start = int(idx_end) - int(des_n_events)
        else:
            idx_end = int(idx_start) + int(des_n_events)
    return int(idx_start), int(idx_end), deficiency


def select_events(stream_t, t_start, t_end, des_n_events=None, prefer_latest_events=True):
    """Slice of a time-sorted stream covering [t_start, t_end] (searchsorted left/right as mvsec_loader.py:272-273),
    fitted to des_n_events.  Returns (slice, n_event_deficiency)."""
    i0 = int(np.searchsorted(stream_t, t_start, side='left'))
    i1 = int(np.searchsorted(stream_t, t_end, side='right'))
    i0, i1, d = fit_event_window(len(stream_t), i0, i1, des_n_events, prefer_latest_events)
    return slice(i0, i1), d


def normalize_times(ts, image_ts, start_time, end_time, time_scaler=1.0):
    """exp_mgr.py:322-324."""
    den = (end_time - start_time) + EPSN
    ts_n = ((np.asarray(ts, dtype=np.float64) - start_time) / den) * time_scaler
    image_ts_n = ((np.asarray(image_ts, dtype=np.float64) - start_time) / den) * time_scaler
    return ts_n, image_ts_n


def normalize_edges(edge_images):
    """Min-max normalise each edge image to [0, 1] (exp_mgr.py:343-350 via img_utils.py:24-25)."""
    out = []
    for e in edge_images:
        e = np.asarray(e, dtype=np.float64)
        out.append((e - e.min()) / (e.max() - e.min() + EPSN))
    return np.stack(out)


def stage_datasample(datasample, edges):
    """The part of EINCMExperiment.stage_datasample (exp_mgr.py:278-376) that feeds the loss: returns
    (xs:int16, ys:int16, ts:float64, edges:(R,H,W) float64, edge_ts:float64) ready for solver.set_datasample."""
    ev = datasample['events']
    start_time, end_time = datasample['eval_ts_us'] if 'eval_ts_us' in datasample else datasample['eval_ts']
    ts, image_ts = normalize_times(ev['t'], datasample['image_ts'], float(start_time), float(end_time))
    from .engine import as_int16_coords       # rounds float coordinates half-to-even (event_warpers.py:29-30), range-checks
    xs = np.ascontiguousarray(as_int16_coords(ev['x'], 'x'))
    ys = np.ascontiguousarray(as_int16_coords(ev['y'], 'y'))
    return xs, ys, ts, normalize_edges(edges), image_ts
