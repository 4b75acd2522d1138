"""Oracle against the committed golden fixtures (tests/golden/*.npz; provenance in tests/golden/make_golden.py:
build-generated, since the reference ships none and cannot run here)."""
import numpy as np
import pytest

from oracle import eincm_oracle as O
from _golden import golden_names, load_golden


@pytest.mark.parametrize('name', golden_names())
def test_oracle_reproduces_golden(name):
    d = load_golden(name)
    kw = d['kw']
    H, W = d['sensor_size']
    val, grad, aux = O.loss_and_grad(d['theta'], d['xs'], d['ys'], d['ts'], d['edges'], d['edge_ts'], kw['alpha'],
                                     kw['beta'], kw['gamma'], kw['delta'], kw['cur_pyr_lvl'], 5, (H, W), kw['method'],
                                     contrast_kind=kw['contrast_kind'], return_intermediates=True)
    assert val == pytest.approx(float(d['value']), rel=1e-12)
    np.testing.assert_allclose(grad, d['grad'], rtol=1e-9, atol=1e-12 * np.abs(d['grad']).max())
    np.testing.assert_allclose(aux['_iwes'], d['iwes'], rtol=2e-7, atol=1e-7)          # stored as fp32
    assert aux['mean_rel_corr'] == pytest.approx(float(d['mean_rel_corr']), rel=1e-12)
    assert aux['mean_rel_contrast'] == pytest.approx(float(d['mean_rel_contrast']), rel=1e-12)
    assert aux['theta_total_variation'] == pytest.approx(float(d['theta_total_variation']), rel=1e-12)


def test_golden_set_is_present():
    assert len(golden_names()) >= 6
