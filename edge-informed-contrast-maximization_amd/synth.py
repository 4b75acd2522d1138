"""Deterministic synthetic event windows with the wire format the EINCM path consumes.

The reference stages a window as ``(xs:int16, ys:int16, ts:float64 in ~[0,1], edges:float64 (R,H,W) in
[0,1], edge_ts:float64 (R,))`` (/root/reference/src/experiments/e00/exp_mgr.py:278-376: time
normalisation ``(t - t0)/(t1 - t0 + eps)`` at :322-324, edge maps min-max normalised at :343-350; events
arrive time-sorted from every loader, src/dataloaders/mvsec_loader.py:272-295).  No dataset is
available offline, so this module generates windows of that exact shape: moving-edge scenes
(line segments + circles) under a ground-truth flow, events sampled on the moving edges with pixel
jitter plus uniform noise events (recipe: SURVEY.md section 8d).
"""
import numpy as np
from scipy import ndimage


def _edge_pool(rng, H, W, n_segments, n_circles, n_pool):
    """Continuous (x, y) points lying on random line segments and circles."""
    per = max(1, n_pool // (n_segments + n_circles))
    pts = []
    for _ in range(n_segments):
        x0, x1 = rng.uniform(0, W - 1, 2)
        y0, y1 = rng.uniform(0, H - 1, 2)
        s = rng.uniform(0, 1, per)
        pts.append(np.stack([x0 + (x1 - x0) * s, y0 + (y1 - y0) * s], axis=1))
    for _ in range(n_circles):
        cx, cy = rng.uniform(0.15 * W, 0.85 * W), rng.uniform(0.15 * H, 0.85 * H)
        rad = rng.uniform(0.04, 0.18) * min(H, W)
        a = rng.uniform(0, 2 * np.pi, per)
        pts.append(np.stack([cx + rad * np.cos(a), cy + rad * np.sin(a)], axis=1))
    return np.concatenate(pts, axis=0)


def _bilinear_field(grid, H, W):
    """Smooth (H,W,2) field from a coarse (g,g,2) grid (plain bilinear, align-corners)."""
    g = grid.shape[0]
    yy = np.linspace(0, g - 1, H)
    xx = np.linspace(0, g - 1, W)
    y0 = np.clip(np.floor(yy).astype(int), 0, g - 2)
    x0 = np.clip(np.floor(xx).astype(int), 0, g - 2)
    fy = (yy - y0)[:, None, None]
    fx = (xx - x0)[None, :, None]
    a = grid[y0][:, x0]
    b = grid[y0][:, x0 + 1]
    c = grid[y0 + 1][:, x0]
    d = grid[y0 + 1][:, x0 + 1]
    # C-contiguous on purpose: advanced indexing hands back a transposed memory layout that arithmetic preserves, and a
    # non-contiguous theta makes every engine call start with a 4.9 MB np.ascontiguousarray copy (1.3 ms at 480x640)
    return np.ascontiguousarray(a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)


def make_window(seed, sensor_size, n_events, n_refs, flow='constant', flow_mag=20.0, noise_frac=0.10,
                jitter_px=0.5, n_segments=64, n_circles=16):
    """Return a dict with xs, ys (int16), ts (float64, sorted), edges (R,H,W) float64 in [0,1],
    edge_ts (R,) float64 and flow_gt (H,W,2) float64 (displacement over the unit window)."""
    H, W = sensor_size
    rng = np.random.default_rng(seed)
    if flow == 'constant':
        v = rng.uniform(-flow_mag, flow_mag, 2)
        flow_gt = np.broadcast_to(v, (H, W, 2)).copy()
    elif flow == 'smooth':
        flow_gt = _bilinear_field(rng.uniform(-flow_mag, flow_mag, (16, 16, 2)), H, W)
    elif flow == 'zero':
        flow_gt = np.zeros((H, W, 2))
    else:
        raise ValueError(f'unknown flow kind {flow!r}')

    pool = _edge_pool(rng, H, W, n_segments, n_circles, max(200_000, 4 * H * W // 3))
    px = np.clip(np.rint(pool[:, 0]).astype(int), 0, W - 1)
    py = np.clip(np.rint(pool[:, 1]).astype(int), 0, H - 1)
    pool_flow = flow_gt[py, px]                                  # flow carried by each edge point

    edge_ts = np.linspace(0.0, 1.0, n_refs) if n_refs > 1 else np.array([0.0])
    edges = np.zeros((n_refs, H, W), dtype=np.float64)
    for r, tau in enumerate(edge_ts):
        ex = np.rint(pool[:, 0] + pool_flow[:, 0] * tau).astype(int)
        ey = np.rint(pool[:, 1] + pool_flow[:, 1] * tau).astype(int)
        ok = (ex >= 0) & (ex < W) & (ey >= 0) & (ey < H)
        img = np.zeros((H, W), dtype=np.float64)
        img[ey[ok], ex[ok]] = 1.0
        img = ndimage.gaussian_filter(img, sigma=1.0, mode='constant')
        lo, hi = img.min(), img.max()
        edges[r] = (img - lo) / (hi - lo + np.finfo(np.float64).eps)

    n_noise = int(round(noise_frac * n_events))
    n_sig = n_events - n_noise
    ts = np.sort(rng.uniform(0.0, 1.0, n_events))
    is_noise = np.zeros(n_events, dtype=bool)
    is_noise[rng.choice(n_events, n_noise, replace=False)] = True
    xs = np.empty(n_events, dtype=np.float64)
    ys = np.empty(n_events, dtype=np.float64)
    pick = rng.integers(0, pool.shape[0], n_sig)
    t_sig = ts[~is_noise]
    xs[~is_noise] = pool[pick, 0] + pool_flow[pick, 0] * t_sig + rng.normal(0, jitter_px, n_sig)
    ys[~is_noise] = pool[pick, 1] + pool_flow[pick, 1] * t_sig + rng.normal(0, jitter_px, n_sig)
    xs[is_noise] = rng.uniform(0, W - 1, n_noise)
    ys[is_noise] = rng.uniform(0, H - 1, n_noise)
    xs = np.clip(np.rint(xs), 0, W - 1).astype(np.int16)
    ys = np.clip(np.rint(ys), 0, H - 1).astype(np.int16)
    return {'xs': xs, 'ys': ys, 'ts': ts, 'edges': edges, 'edge_ts': edge_ts, 'flow_gt': flow_gt,
            'sensor_size': (H, W)}


def theta_near_truth(seed, window, theta_hw, scale_lo=0.5, scale_hi=1.5):
    """A (h,w,2) theta = block-mean of flow_gt times U(scale_lo, scale_hi): events land mostly in frame."""
    h, w = theta_hw
    H, W = window['sensor_size']
    rng = np.random.default_rng(seed + 7919)
    fg = window['flow_gt']
    ye = np.linspace(0, H, h + 1).astype(int)
    xe = np.linspace(0, W, w + 1).astype(int)
    th = np.zeros((h, w, 2))
    for i in range(h):
        for j in range(w):
            th[i, j] = fg[ye[i]:max(ye[i + 1], ye[i] + 1), xe[j]:max(xe[j + 1], xe[j] + 1)].mean(axis=(0, 1))
    return th * rng.uniform(scale_lo, scale_hi, (h, w, 2))
