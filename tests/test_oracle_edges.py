"""Oracle for SURVEY row f-4 (oracle/edge_smoothing.py) against what can pin it here:
  * the Euclidean distance transform is unique: brute-force definition == scipy.ndimage (the routine the reference calls,
    src/utils/img_utils.py:230) == the restated RTEF_IEDT Meijster transform (img_utils.py:314-370);
  * the Gaussian flavour (OpenCV semantics, PARITY UNPINNED) only against analytic properties;
  * the tiled objectives against their untiled siblings.
"""
import numpy as np
import pytest
from scipy import ndimage

from oracle import edge_smoothing as ES
from oracle import eincm_oracle as O


def _edge_images():
    rng = np.random.default_rng(7)
    imgs = []
    for shape, p in [((9, 13), 0.15), ((24, 17), 0.05), ((16, 16), 0.5), ((20, 31), 0.01)]:
        e = (rng.random(shape) < p)
        e[rng.integers(shape[0]), rng.integers(shape[1])] = True       # at least one edge pixel
        imgs.append(e.astype(np.uint8) * 255)                          # Canny output: 0 / 255
    one = np.zeros((11, 7), np.uint8); one[0, 6] = 255                 # single corner pixel: every other row is empty
    imgs.append(one)
    rows = np.zeros((12, 10), np.uint8); rows[3, :] = 255; rows[9, 2] = 255
    imgs.append(rows)
    return imgs


@pytest.mark.parametrize('k', range(6))
def test_edt_three_ways(k):
    e = _edge_images()[k]
    brute = ES.edt_squared_bruteforce(e)
    sp = ndimage.distance_transform_edt(~e.astype(bool))
    assert np.array_equal(np.rint(sp ** 2).astype(np.int64), brute)
    assert np.array_equal(ES.rtef_edt_squared(e), brute)


def test_iedt_values_and_range():
    e = _edge_images()[0]
    out = ES.eincm_inv_exp_dist_transform(e, alpha=6)
    d = np.sqrt(ES.edt_squared_bruteforce(e).astype(np.float64))
    x = 1 - np.exp(-d / 6)
    np.testing.assert_allclose(out, 1 - x / (x.max() + ES.EPSN), rtol=0, atol=1e-15)
    assert out.max() == 1.0 and np.all(out[e > 0] == 1.0)
    assert out.min() == pytest.approx(0.0, abs=1e-15)


def test_rtef_equals_scipy_flavour_for_same_alpha():
    e = _edge_images()[1]
    a = ES.rtef_inv_exp_dist_transform(e, None, 6.0, 'exponential')
    b = ES.eincm_inv_exp_dist_transform(e, alpha=6.0)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-15)


def test_rtef_defaults_and_formulations():
    e = _edge_images()[3]
    d = np.sqrt(ES.edt_squared_bruteforce(e).astype(np.float64))
    lin = ES.rtef_inv_exp_dist_transform(e, formulation='linear')
    np.testing.assert_allclose(lin, 1 - d / (d.max() + ES.EPSN), atol=1e-15)
    lb = ES.rtef_inv_exp_dist_transform(e, 4.0, None, 'linear-bound')
    np.testing.assert_allclose(lb, 1 - np.minimum(d, 4.0) / (4.0 + ES.EPSN), atol=1e-15)
    lg = ES.rtef_inv_exp_dist_transform(e, formulation='logarithmic')
    np.testing.assert_allclose(lg, 1 - np.log(d + 1) / (np.log(d.max() + 1) + ES.EPSN), atol=1e-15)
    ex = ES.rtef_inv_exp_dist_transform(e)                             # d_sat 6 -> alpha 6/5.541 (img_utils.py:256-257)
    x = 1 - np.exp(-d / (6.0 / 5.541))
    np.testing.assert_allclose(ex, 1 - x / (x.max() + ES.EPSN), atol=1e-15)
    with pytest.raises(AssertionError):
        ES.rtef_inv_exp_dist_transform(np.zeros((4, 4), np.uint8))
    with pytest.raises(NotImplementedError):
        ES.rtef_inv_exp_dist_transform(e, formulation='quadratic')


def test_gaussian_kernel_and_blur_properties():
    k = ES.gaussian_kernel_cv(1.0)
    assert len(k) == 9 and k.sum() == pytest.approx(1.0, abs=1e-15) and np.allclose(k, k[::-1])
    assert k[4] / k[3] == pytest.approx(np.exp(0.5), rel=1e-14)
    assert len(ES.gaussian_kernel_cv(2.0)) == 17 and len(ES.gaussian_kernel_cv(0.8)) == 7
    const = np.full((12, 15), 3.25)
    np.testing.assert_allclose(ES.smoothen_edges(const), const, atol=1e-14)
    imp = np.zeros((21, 23)); imp[10, 11] = 1.0
    out = ES.smoothen_edges(imp, k_size=1, sigma=99)                   # `sigma` is ignored (module doc)
    np.testing.assert_allclose(out[6:15, 7:16], np.outer(k, k), atol=1e-17)
    # BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba): column -j mirrors to column +j, the border pixel is not repeated
    edge = np.zeros((9, 12)); edge[:, 1] = 1.0
    row = ES.smoothen_edges(edge)[4]
    assert row[0] == pytest.approx(2 * k[3], rel=1e-14)                # col 1 seen at offset +1 and mirrored at offset -1
    assert row[1] == pytest.approx(k[4] + k[2], rel=1e-14)           # itself, and column -1 mirrors back onto column 1
    with pytest.raises(ValueError):
        ES.smoothen_edges(np.zeros((3, 20)))


def test_gaussian_blur_matches_scipy_ndimage():
    """The restated cv.GaussianBlur (normalised exp taps on round(8 sigma + 1) | 1 samples, separable, BORDER_REFLECT_101) is
    scipy.ndimage.gaussian_filter with mode='mirror' and the same radius.  OpenCV itself is absent, so which kernel SIZE OpenCV picks
    for a float64 image stays the restatement's claim (cv::getGaussianKernel / createGaussianKernels); the filter given that size is
    witnessed here."""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    img = (rng.random((60, 80)) > 0.9).astype(np.float64)
    for sigma in (0.8, 1.0, 2.0, 3.3):
        r = (len(ES.gaussian_kernel_cv(sigma)) - 1) // 2
        np.testing.assert_allclose(ES.smoothen_edges(img, k_size=sigma, sigma=7), ndimage.gaussian_filter(img, sigma=sigma, mode='mirror', radius=r),
                                   rtol=0, atol=1e-15)


def test_tiled_objectives():
    rng = np.random.default_rng(3)
    a, b = rng.random((70, 90)), rng.random((70, 90))
    assert ES.extract_tiles(a, 32, 42).shape == (4, 32, 42)
    assert np.array_equal(ES.extract_tiles(a, 32, 42)[3], a[32:64, 42:84])
    # one tile covering the image == the untiled objective
    assert ES.compute_adaptive_variance(a, (70, 90)) == pytest.approx(O.compute_variance(a), rel=1e-14)
    assert ES.compute_adaptive_mean_gradient_magnitude(a, (70, 90)) == pytest.approx(O.compute_mean_gradient_magnitude(a), rel=1e-14)
    assert ES.compute_adaptive_mean_squared_error(a, b, (70, 90)) == pytest.approx(O.compute_mean_squared_error(a, b), rel=1e-14)
    # default tile size and additivity over tiles
    t = ES.extract_tiles(a, 32, 42)
    assert ES.compute_adaptive_variance(a) == pytest.approx(sum(np.var(x) for x in t), rel=1e-14)
    assert ES.compute_adaptive_variance(np.ones((64, 84))) == 0.0
    assert ES.compute_sum_squared_error(a, b) == pytest.approx(O.compute_mean_squared_error(a, b) * a.size, rel=1e-13)
    assert ES.compute_sum_hadamard_product(a, b) == pytest.approx(ES.compute_mean_hadamard_product(a, b) * a.size, rel=1e-13)
    assert ES.compute_joint_contrast(a, b) == pytest.approx(O.compute_mean_gradient_magnitude(a + b), rel=1e-14)
