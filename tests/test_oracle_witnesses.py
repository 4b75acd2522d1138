"""Independent witness for JAX semantic S7 (jax.image.scale_and_translate, theta_utils.py:25-35), bilinear case.

The reference cannot run here, so the oracle's weight matrices are a restatement from knowledge of JAX.  PyTorch implements the
same definition independently: half-pixel centres, triangle kernel, out-of-range taps dropped and the rest renormalised
(``F.interpolate(mode='bilinear', align_corners=False)``; with ``antialias=True`` the kernel is widened by the scale factor when
shrinking, as JAX does).  Agreement to 1e-14 pins the bilinear path of ``resample_matrix`` - the one the reference's default config
uses (configs/theta_loss_func/default.yaml: scale_to_sensor_size_method bilinear) - to a second implementation.  The cubic and
Lanczos kernels have no second implementation here (torch's bicubic uses a = -0.75, JAX a = -0.5) and stay unpinned.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eincm_oracle as O


@pytest.mark.parametrize('hw,HW', [((1, 1), (20, 30)), ((2, 2), (180, 240)), ((4, 4), (260, 346)), ((3, 5), (33, 65)),
                                   ((16, 16), (260, 346)), ((8, 8), (480, 640)), ((7, 11), (50, 50))])
def test_bilinear_upsampling_matches_torch(hw, HW):
    rng = np.random.default_rng(sum(hw) + sum(HW))
    theta = rng.normal(size=hw + (2,))
    ours = O.scale_theta_to_sensor_size(theta, HW, 'bilinear')
    t = torch.from_numpy(theta).permute(2, 0, 1)[None]                       # (1, 2, h, w) float64
    ref = F.interpolate(t, size=HW, mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(ours, ref, rtol=0, atol=1e-14)


@pytest.mark.parametrize('n_in,n_out', [(16, 8), (16, 4), (346, 16), (260, 16), (10, 3), (9, 4), (33, 7)])
def test_bilinear_antialiased_downsampling_matches_torch(n_in, n_out):
    A = O.resample_matrix(n_in, n_out, n_out / n_in, 'bilinear')            # (n_out, n_in), antialias=True
    eye = torch.eye(n_in, dtype=torch.float64)[None, None]                   # rows of the identity -> columns of the operator
    ref = F.interpolate(eye, size=(n_in, n_out), mode='bilinear', align_corners=False, antialias=True)[0, 0].numpy().T
    np.testing.assert_allclose(A, ref, rtol=0, atol=1e-14)
    assert np.allclose(A.sum(axis=1), 1.0, atol=1e-15)


@pytest.mark.parametrize('method,pil_name', [('cubic', 'BICUBIC'), ('lanczos3', 'LANCZOS'), ('bilinear', 'BILINEAR')])
def test_cubic_and_lanczos3_match_pillow(method, pil_name):
    """jax.image.scale_and_translate's 'cubic' (Keys, a = -0.5) and 'lanczos3' kernels with half-pixel centres, the support stretched
    when down-sampling and the weights renormalised at the borders are Pillow's BICUBIC / LANCZOS resize (which jax.image documents as
    its model).  Pillow is importable here: the operator matrices agree to its float32 precision, for the up-sampling of theta
    (theta_utils.py:25-35) and for the lanczos3 down-sampling of the pyramid priors (solver.py:350-377).  lanczos5 has no witness."""
    Image = pytest.importorskip('PIL.Image')
    flt = getattr(Image, pil_name)
    for n_in, n_out in ((16, 260), (16, 346), (4, 37), (346, 16), (260, 16), (16, 8), (8, 4), (33, 7), (2, 5), (1, 9)):
        A = O.resample_matrix(n_in, n_out, n_out / n_in, method)               # (n_out, n_in)
        P = np.zeros((n_out, n_in))
        for k in range(n_in):
            e = np.zeros((1, n_in), dtype=np.float32)
            e[0, k] = 1.0
            P[:, k] = np.asarray(Image.fromarray(e, mode='F').resize((n_out, 1), resample=flt), dtype=np.float64)[0]
        assert np.abs(A - P).max() <= 2e-7, (method, n_in, n_out, np.abs(A - P).max())


def test_adjoint_is_the_transpose():
    rng = np.random.default_rng(0)
    theta = rng.normal(size=(4, 6, 2)); G = rng.normal(size=(37, 53, 2))
    lhs = (O.scale_theta_to_sensor_size(theta, (37, 53), 'bilinear') * G).sum()
    rhs = (theta * O.scale_theta_adjoint(G, theta.shape, 'bilinear')).sum()
    assert lhs == pytest.approx(rhs, rel=1e-13)


# ---------------------------------------------------------------------------------------------------------------------
# Library routines the reference's call sites name, where a second implementation IS importable here (scipy):
# jax.scipy.signal.convolve / jax.scipy.stats.multivariate_normal / scipy.stats.norm are documented as drop-ins of scipy's.
# ---------------------------------------------------------------------------------------------------------------------
def test_convolution_matches_scipy_signal():
    """S6 (img_utils.py:420-421, event_collapse_objectives.py:15-18): 'same' true convolution, zero padded."""
    from scipy import signal
    rng = np.random.default_rng(1)
    img = rng.normal(size=(23, 31))
    for K in (O.SCHARR_GX, O.SCHARR_GY, O.DIV_KERN):
        np.testing.assert_allclose(O.conv2_same(img, K), signal.convolve2d(img, K, mode='same'), rtol=0, atol=1e-12)
    gx, gy = O.scharr_grads(img)                       # the difference-first form used by the oracle and the kernels
    np.testing.assert_allclose(gx, signal.convolve2d(img, O.SCHARR_GX, mode='same'), rtol=0, atol=1e-12)
    np.testing.assert_allclose(gy, signal.convolve2d(img, O.SCHARR_GY, mode='same'), rtol=0, atol=1e-12)


def test_tap_weights_match_scipy_multivariate_normal():
    """S8 (event_utils.py:52-56): the nine tap weights are multivariate_normal.pdf(q, mean 0, cov I2)."""
    from scipy.stats import multivariate_normal
    wx, wy = np.array([10.3]), np.array([7.8])
    f = O.events_to_pdf_frame(wx, wy, (20, 24))
    rx, ry = int(np.rint(wx[0])), int(np.rint(wy[0]))
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            q = np.array([rx + dx - wx[0], ry + dy - wy[0]])
            assert f[ry + dy, rx + dx] == pytest.approx(multivariate_normal.pdf(q, mean=np.zeros(2), cov=np.eye(2)), rel=1e-14)


def test_multi_reference_weights_match_scipy_norm():
    """losses.py:39-46 calls scipy.stats.norm.pdf on linspace(-1.5, 1.5, R) and normalises."""
    from scipy.stats import norm
    for R in (1, 2, 3, 5, 8):
        p = norm.pdf(np.linspace(-1.5, 1.5, R))
        np.testing.assert_allclose(O.compute_weights_for_multi_reference(R), p / p.sum(), rtol=1e-15)
