"""Second, independent CPU witness for the EINCM loss: torch float64 forward, gradient by torch autograd.

TEST INFRASTRUCTURE ONLY (see oracle/eincm_oracle.py header; same import rules, PARITY UNPINNED).

Purpose: the reference obtains its gradient from JAX autodiff (solver.py:165-173 via jaxopt ->
jax.value_and_grad).  oracle/eincm_oracle.py uses a hand-derived reverse pass; this file restates the
*forward* only (losses.py:49-205) with torch ops written independently of the numpy oracle
(F.conv2d for the convolutions, index_put_(accumulate=True) for the scatter, amin/amax whose
backward shares the cotangent equally among ties like JAX's) and lets autograd differentiate it, so
tests can check forward-vs-forward and hand-backward-vs-autodiff.
"""
import math
import sys

import numpy as np
import torch
import torch.nn.functional as F

EPSN = sys.float_info.epsilon
_DT = torch.float64

_SX = torch.tensor([[3.0, 0.0, -3.0], [10.0, 0.0, -10.0], [3.0, 0.0, -3.0]], dtype=_DT)
_SY = torch.tensor([[3.0, 10.0, 3.0], [0.0, 0.0, 0.0], [-3.0, -10.0, -3.0]], dtype=_DT)
_DK = torch.tensor([[1 / 12, 1 / 6, 1 / 12], [1 / 6, 0.0, 1 / 6], [1 / 12, 1 / 6, 1 / 12]], dtype=_DT)


def _conv_same(img, kern):
    # true convolution = cross-correlation with the flipped kernel (img_utils.py:420-421, S6)
    k = torch.flip(kern, dims=(0, 1))[None, None]
    return F.conv2d(img[None, None], k, padding=1)[0, 0]


def _scharr(img):
    return _conv_same(img, _SX), _conv_same(img, _SY)


def _scharr_diff(img):
    """Difference-first Scharr (see oracle/eincm_oracle.py:scharr_grads): used for the TV term only, whose
    non-zero count / sign are sensitive to the summation order on locally constant flow."""
    p = F.pad(img, (1, 1, 1, 1))
    H, W = img.shape
    d, m, u = p[2:H + 2], p[1:H + 1], p[0:H]
    gx = 3.0 * (d[:, 2:] - d[:, :W]) + 10.0 * (m[:, 2:] - m[:, :W]) + 3.0 * (u[:, 2:] - u[:, :W])
    gy = 3.0 * (d[:, 2:] - u[:, 2:]) + 10.0 * (d[:, 1:W + 1] - u[:, 1:W + 1]) + 3.0 * (d[:, :W] - u[:, :W])
    return gx, gy


def _normalize(a):
    return (a - torch.amin(a)) / (torch.amax(a) - torch.amin(a) + EPSN)


def _splat(wx, wy, H, W):
    """event_utils.py:31-61 with S1 index rules."""
    rx = torch.round(wx.detach()).to(torch.int64)
    ry = torch.round(wy.detach()).to(torch.int64)
    frame = torch.zeros(H * W, dtype=_DT)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            px = rx + dx
            py = ry + dy
            qx = px.to(_DT) - wx
            qy = py.to(_DT) - wy
            k = torch.exp(-0.5 * (qx * qx + qy * qy) - math.log(2.0 * math.pi))
            px = torch.where(px < 0, px + W, px)
            py = torch.where(py < 0, py + H, py)
            ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
            idx = torch.where(ok, py * W + px, torch.zeros_like(px))
            frame = frame.index_put((idx,), torch.where(ok, k, torch.zeros_like(k)), accumulate=True)
    return frame.reshape(H, W)


def _iwe_div(n):
    gx, gy = _scharr(n)
    return torch.abs(_conv_same(gx, _DK) + _conv_same(gy, _DK)).mean()


def _weights(R):
    x = np.linspace(-1.5, 1.5, R)
    w = np.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    return w / w.sum()


def loss_from_Theta(Theta, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta, cur_pyr_lvl, contrast_kind=0):
    """losses.py:162-193 on a full-resolution Theta (H,W,2) torch tensor."""
    H, W, _ = Theta.shape
    xi = torch.as_tensor(np.asarray(xs).astype(np.int64))
    yi = torch.as_tensor(np.asarray(ys).astype(np.int64))
    t = torch.as_tensor(np.asarray(ts, dtype=np.float64))
    E = torch.as_tensor(np.asarray(edges, dtype=np.float64))
    tau = np.asarray(edge_ts, dtype=np.float64)
    R = len(tau)
    w = _weights(R)

    def contrast(img):
        if contrast_kind == 1:
            return torch.var(img, unbiased=False)
        gx, gy = _scharr(img)
        return (gx * gx + gy * gy).mean()

    I0 = _splat(xi.to(_DT), yi.to(_DT), H, W)
    n0 = _normalize(I0)
    c0 = contrast(I0)
    d0 = _iwe_div(n0)
    vx = Theta[yi, xi, 0]
    vy = Theta[yi, xi, 1]
    rel_con, rel_corr, rel_div = [], [], []
    for r in range(R):
        dts = t - float(tau[r])
        wx = xi.to(_DT) - vx * dts * 1.0
        wy = yi.to(_DT) - vy * dts * 1.0
        I = _splat(wx, wy, H, W)
        n = _normalize(I)
        corr = -((E[r] - n) ** 2).mean()
        zc = -((E[r] - n0) ** 2).mean()
        rel_corr.append(w[r] * corr / (zc + EPSN))
        rel_con.append(w[r] * contrast(I) / (c0 + EPSN))
        rel_div.append(w[r] * _iwe_div(n) / (d0 + EPSN))
    mean_rel_con = torch.stack(rel_con).mean()
    mean_rel_corr = torch.stack(rel_corr).mean()
    mean_rel_div = torch.stack(rel_div).mean()
    tv = torch.zeros((), dtype=_DT)
    if cur_pyr_lvl <= 0:
        mask = torch.zeros(H, W, dtype=_DT)
        mask[yi, xi] = 1.0
        tot = torch.zeros((), dtype=_DT)
        nz = torch.zeros(H, W, dtype=torch.bool)
        for c in (0, 1):
            gx, gy = _scharr_diff(Theta[:, :, c] * mask)
            tot = tot + (gx.abs() * 0.25 + gy.abs() * 0.25).sum()
            nz |= (gx.detach().abs() > 0) | (gy.detach().abs() > 0)
        tv = tot / (float(nz.sum()) + EPSN)
    return (alpha * (-mean_rel_con) + beta * (-mean_rel_corr)) + (gamma * tv + delta * mean_rel_div)


def loss_and_grad(theta, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta, cur_pyr_lvl, sensor_size,
                  A_H, A_W, contrast_kind=0):
    """value and autograd gradient w.r.t. the coarse theta (h,w,2); A_H (H,h), A_W (W,w) resampling matrices."""
    th = torch.tensor(np.asarray(theta, dtype=np.float64), requires_grad=True)
    AH = torch.as_tensor(np.asarray(A_H, dtype=np.float64))
    AW = torch.as_tensor(np.asarray(A_W, dtype=np.float64))
    Theta = torch.einsum('yi,xj,ijc->yxc', AH, AW, th)
    val = loss_from_Theta(Theta, xs, ys, ts, edges, edge_ts, alpha, beta, gamma, delta, cur_pyr_lvl, contrast_kind)
    val.backward()
    return float(val.detach()), th.grad.numpy().copy()
