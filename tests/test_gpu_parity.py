"""Parity of the HIP path (through the C-ABI of libeincm_hip.so) with the fp64 CPU oracle.

Tolerance (BASELINE.json north_star): loss and gradient within 1e-5 relative of the fp64 CPU path; the HIP path
accumulates images in fp32 and reduces in fp64.  Gradient error is measured max-norm relative,
max|g - g*| / max|g*|.  The IWE is a float image (never an integer count image), compared
with a relative max-norm bound of 1e-5 as well.

Run on the GPU box:  python -m pytest tests -m gpu
"""
import importlib

import numpy as np
import pytest

from oracle import eincm_oracle as O
from _golden import golden_names, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5

synth = importlib.import_module('edge-informed-contrast-maximization_amd.synth')
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
losses = importlib.import_module('edge-informed-contrast-maximization_amd.losses')
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def win_args(win):
    return (win['xs'], win['ys'], win['ts'], win['edges'], win['edge_ts'])


@pytest.fixture(scope='module', autouse=True)
def _lib(built_lib):
    return built_lib


# ---------------------------------------------------------------------------------------------------
# golden fixtures
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', golden_names())
def test_golden(name):
    d = load_golden(name)
    kw = d['kw']
    H, W = d['sensor_size']
    with engine.Engine((H, W), len(d['xs']), max_refs=len(d['edge_ts'])) as eng:
        eng.set_window(d['xs'], d['ys'], d['ts'], d['edges'], d['edge_ts'])
        p = engine.make_params(kw['alpha'], kw['beta'], kw['gamma'], kw['delta'], kw['cur_pyr_lvl'], kw['method'],
                               kw['contrast_kind'], full_aux=True)
        v, g, aux = eng.loss_grad(d['theta'], p, want_aux=True)
        assert abs(v[0] - d['value']) <= TOL * abs(d['value'])
        assert rel(g[0], d['grad']) <= TOL
        assert rel(eng.iwes()[0], d['iwes']) <= TOL
        assert rel(eng.zero_iwe()[0], d['zero_iwe']) <= TOL
        assert aux[0]['mean_rel_corr'] == pytest.approx(float(d['mean_rel_corr']), rel=TOL)
        assert aux[0]['mean_rel_contrast'] == pytest.approx(float(d['mean_rel_contrast']), rel=TOL)
        assert aux[0]['mean_rel_iwe_divergence'] == pytest.approx(float(d['mean_rel_iwe_divergence']), rel=TOL)
        assert aux[0]['theta_total_variation'] == pytest.approx(float(d['theta_total_variation']), rel=TOL, abs=1e-12)
        # compute_loss_objectives on the scaled theta
        Theta = O.scale_theta_to_sensor_size(d['theta'], (H, W), kw['method'])
        assert rel(eng.scaled_theta()[0], Theta) <= 1e-13
        ob = eng.objectives(Theta)[0]
        for k in ('correlations', 'zero_correlations', 'contrasts', 'iwe_divergences', 'flow_warp_losses'):
            assert rel(ob[k], d[k]) <= TOL, k
        for k in ('zero_contrast', 'zero_iwe_divergence', 'theta_divergence'):
            assert ob[k] == pytest.approx(float(d[k]), rel=TOL), k


# ---------------------------------------------------------------------------------------------------
# seeded cases against the oracle run in the test
# ---------------------------------------------------------------------------------------------------
CASES = [
    # id, (H,W), N, R, theta, flow, mag, alpha, beta, gamma, lvl, contrast_kind, method
    ('c1', (180, 240), 10000, 1, (1, 1), 'constant', 20.0, 20.0, 0.0, 0.0, 4, 1, 'bilinear'),
    ('c2', (260, 346), 100000, 5, (1, 1), 'constant', 20.0, 20.0, 35.0, 0.0, 4, 0, 'bilinear'),
    ('c2_pyr16_tv', (260, 346), 100000, 5, (16, 16), 'smooth', 20.0, 20.0, 35.0, 2.5e-4, 0, 0, 'bilinear'),
    ('mvsec_crop_30k', (256, 336), 30000, 5, (8, 8), 'smooth', 15.0, 20.0, 35.0, 0.0, 1, 0, 'bilinear'),
    ('odd_sizes', (33, 65), 3000, 2, (3, 5), 'smooth', 5.0, 60.0, 60.0, 1e-3, 0, 0, 'cubic'),
    ('dense_small', (60, 80), 20000, 3, 'dense', 'smooth', 8.0, 2000.0, 4000.0, 2.5e-3, 0, 0, 'bilinear'),
    ('bigflow', (120, 160), 30000, 3, (2, 2), 'smooth', 150.0, 20.0, 35.0, 0.0, 3, 0, 'bilinear'),
    ('lanczos', (120, 160), 30000, 3, (4, 4), 'smooth', 10.0, 20.0, 35.0, 0.0, 2, 0, 'lanczos3'),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_against_oracle(case):
    _, (H, W), N, R, hw, flow, mag, al, be, ga, lvl, ck, method = case
    win = synth.make_window(11, (H, W), N, R, flow=flow, flow_mag=mag)
    if hw == 'dense':
        th = win['flow_gt'] * np.random.default_rng(2).uniform(0.5, 1.5, (H, W, 2))
    else:
        th = synth.theta_near_truth(11, win, hw)
    v_ref, g_ref, aux = O.loss_and_grad(th, *win_args(win), al, be, ga, 0.0, lvl, 5, (H, W), method, contrast_kind=ck,
                                        return_intermediates=True)
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, _ = eng.loss_grad(th, engine.make_params(al, be, ga, 0.0, lvl, method, ck))
        assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
        assert rel(g[0], g_ref) <= TOL
        assert rel(eng.iwes()[0], aux['_iwes']) <= TOL
        assert rel(eng.image_grad()[0], aux['_G']) <= TOL
        # forward only gives the same value
        v2, g2, _ = eng.loss_grad(th, engine.make_params(al, be, ga, 0.0, lvl, method, ck), want_grad=False)
        assert g2 is None and v2[0] == pytest.approx(v[0], rel=1e-6)


@pytest.mark.parametrize('hw,ck', [((1, 1), 0), ((4, 4), 1), ('dense', 0)])
def test_delta_term_gradient(hw, ck):
    """delta != 0: loss and gradient through the IWE-divergence objective (event_collapse_objectives.py:8-20),
    including its path through the min/max of the normalisation."""
    H, W, N, R = 70, 90, 15000, 3
    win = synth.make_window(13, (H, W), N, R, flow='smooth', flow_mag=8.0)
    th = win['flow_gt'] * np.random.default_rng(2).uniform(0.5, 1.5, (H, W, 2)) if hw == 'dense' else synth.theta_near_truth(13, win, hw)
    for delta in (0.7, 25.0):
        v_ref, g_ref, aux = O.loss_and_grad(th, *win_args(win), 20.0, 35.0, 2.5e-4, delta, 0, 5, (H, W), contrast_kind=ck,
                                            return_intermediates=True)
        with engine.Engine((H, W), N, max_refs=R) as eng:
            eng.set_window(*win_args(win))
            v, g, a = eng.loss_grad(th, engine.make_params(20.0, 35.0, 2.5e-4, delta, 0, contrast_kind=ck), want_aux=True)
            assert rel(eng.image_grad()[0], aux['_G']) <= TOL
        assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
        assert rel(g[0], g_ref) <= TOL
        assert a[0]['mean_rel_iwe_divergence'] == pytest.approx(aux['mean_rel_iwe_divergence'], rel=TOL)


def test_sixteen_reference_times():
    """EINCM_MAX_REFS = 16 reference images per window."""
    H, W, N, R = 64, 96, 12000, 16
    win = synth.make_window(14, (H, W), N, R, flow='smooth', flow_mag=6.0)
    th = synth.theta_near_truth(14, win, (2, 2))
    v_ref, g_ref, _ = O.loss_and_grad(th, *win_args(win), 20.0, 35.0, 0.0, 0.0, 3, 5, (H, W))
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 3))
    assert abs(v[0] - v_ref) <= TOL * abs(v_ref) and rel(g[0], g_ref) <= TOL
    with pytest.raises(engine.EincmError):
        engine.Engine((H, W), N, max_refs=17)


@pytest.mark.parametrize('hw,method', [((59, 79), 'bilinear'), ((61, 50), 'lanczos3'), ((40, 100), 'cubic')])
def test_theta_finer_or_mixed_resolution(hw, method):
    """theta need not be coarser than the sensor: scale_and_translate down-samples with the anti-aliased (widened) kernel
    (theta_utils.py:25-35); rows have many taps and a tile covers many cells (k_project's LDS / direct paths)."""
    H, W, N, R = 60, 80, 15000, 2
    win = synth.make_window(15, (H, W), N, R, flow='smooth', flow_mag=6.0)
    th = np.random.default_rng(7).normal(0.0, 3.0, hw + (2,))
    v_ref, g_ref, aux = O.loss_and_grad(th, *win_args(win), 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W), method)
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 2.5e-4, 0.0, 0, method))
        assert rel(eng.scaled_theta()[0], aux['scaled_theta']) <= 1e-12
    assert abs(v[0] - v_ref) <= TOL * abs(v_ref) and rel(g[0], g_ref) <= TOL
    if hw == (59, 79):
        with engine.Engine((H, W), N, max_refs=R) as eng:          # more cells than pixels: refused, not mis-computed
            eng.set_window(*win_args(win))
            with pytest.raises(engine.EincmError, match='more cells than'):
                eng.loss_grad(np.zeros((90, 120, 2)), engine.make_params(20.0, 35.0, 0.0, 0.0, 0))


def test_context_lifecycle_and_coexistence():
    """Contexts can be created and destroyed repeatedly, and two can be alive (and interleaved) on one GPU."""
    H, W, N, R = 70, 90, 8000, 3
    wa = synth.make_window(16, (H, W), N, R, flow='constant', flow_mag=5.0)
    wb = synth.make_window(17, (H, W), N, R, flow='smooth', flow_mag=5.0)
    tha, thb = synth.theta_near_truth(16, wa, (1, 1)), synth.theta_near_truth(17, wb, (4, 4))
    p = engine.make_params(20.0, 35.0, 0.0, 0.0, 2)
    ref = {}
    for k, (w_, t_) in {'a': (wa, tha), 'b': (wb, thb)}.items():
        ref[k] = O.loss_and_grad(t_, *win_args(w_), 20.0, 35.0, 0.0, 0.0, 2, 5, (H, W))[:2]
    for _ in range(12):                       # create / stage / evaluate / destroy
        with engine.Engine((H, W), N, max_refs=R) as e:
            e.set_window(*win_args(wa))
            v, g, _ = e.loss_grad(tha, p)
            assert abs(v[0] - ref['a'][0]) <= TOL * abs(ref['a'][0])
    with engine.Engine((H, W), N, max_refs=R) as ea, engine.Engine((H, W), N, max_refs=R) as eb:
        ea.set_window(*win_args(wa))
        eb.set_window(*win_args(wb))
        for _ in range(3):                    # interleaved use must not cross-talk
            va, ga, _ = ea.loss_grad(tha, p)
            vb, gb, _ = eb.loss_grad(thb, p)
            assert abs(va[0] - ref['a'][0]) <= TOL * abs(ref['a'][0]) and rel(ga[0], ref['a'][1]) <= TOL
            assert abs(vb[0] - ref['b'][0]) <= TOL * abs(ref['b'][0]) and rel(gb[0], ref['b'][1]) <= TOL


def test_dense_c3_shape():
    """BASELINE config C3 shape (480x640, dense per-pixel theta, R=3) at a size the oracle finishes in seconds."""
    H, W, N, R = 480, 640, 200000, 3
    win = synth.make_window(12, (H, W), N, R, flow='smooth', flow_mag=30.0)
    th = win['flow_gt'] * np.random.default_rng(4).uniform(0.5, 1.5, (H, W, 2))
    v_ref, g_ref, _ = O.loss_and_grad(th, *win_args(win), 2000.0, 4000.0, 2.5e-4, 0.0, 0, 5, (H, W))
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, _ = eng.loss_grad(th, engine.make_params(2000.0, 4000.0, 2.5e-4, 0.0, 0))
    assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
    assert rel(g[0], g_ref) <= TOL


# ---------------------------------------------------------------------------------------------------
# size-independent properties at BASELINE sizes (the oracle is too slow to run these in a test)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def big_window():
    return synth.make_window(21, (260, 346), 1_000_000, 5, flow='constant', flow_mag=20.0)


def test_full_size_zero_theta_identity(big_window):
    """theta = 0  =>  every IWE_r equals the IUE, rel_contrast = rel_corr = 1, loss = -(alpha+beta)/R  (C.4 i)."""
    win = big_window
    H, W = win['sensor_size']
    R = len(win['edge_ts'])
    with engine.Engine((H, W), len(win['xs']), max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, aux = eng.loss_grad(np.zeros((1, 1, 2)), engine.make_params(20.0, 35.0, 0.0, 0.0, 4), want_aux=True)
        iw, z = eng.iwes()[0], eng.zero_iwe()[0]
        for r in range(R):
            assert rel(iw[r], z) <= 2e-6          # same taps, fp32 sums in a different order
        assert v[0] == pytest.approx(-(20.0 + 35.0) / R, rel=TOL)
        assert aux[0]['mean_rel_contrast'] == pytest.approx(1.0 / R, rel=TOL)
        # total mass of the IUE: every in-frame tap of every event (interior events carry 0.7794836797093877)
        assert z.sum(dtype=np.float64) == pytest.approx(0.7794836797093877 * len(win['xs']), rel=2e-3)


def test_full_size_permutation_and_shard_additivity(big_window):
    """IWE is invariant to event order and additive over event shards (the multi-GPU contract, C.4 iii): a batch of two
    half-windows sums to the full window's IWE."""
    win = big_window
    H, W = win['sensor_size']
    R = len(win['edge_ts'])
    N = len(win['xs'])
    th = synth.theta_near_truth(21, win, (1, 1))
    p = engine.make_params(20.0, 35.0, 0.0, 0.0, 4)
    with engine.Engine((H, W), N, max_refs=R, max_windows=2) as eng:
        eng.set_window(*win_args(win))
        v_full, g_full, _ = eng.loss_grad(th, p)
        iw_full = eng.iwes()[0].astype(np.float64)
        perm = np.random.default_rng(5).permutation(N)
        eng.set_window(win['xs'][perm], win['ys'][perm], win['ts'][perm], win['edges'], win['edge_ts'])
        v_perm, g_perm, _ = eng.loss_grad(th, p)
        assert v_perm[0] == pytest.approx(v_full[0], rel=2e-6)
        assert rel(g_perm[0], g_full[0]) <= 2e-5
        assert rel(eng.iwes()[0], iw_full) <= 2e-6
        a, b = perm[:N // 2], perm[N // 2:]
        halves = [(win['xs'][s], win['ys'][s], win['ts'][s], win['edges'], win['edge_ts']) for s in (a, b)]
        eng.set_windows(halves)
        eng.loss_grad(np.stack([th, th]), p, want_grad=False)
        iw = eng.iwes().astype(np.float64)
        assert rel(iw[0] + iw[1], iw_full) <= 2e-6


@pytest.mark.parametrize('shape', ['headline_2dof', 'headline_pyr16_bigflow', 'c3_dense'])
def test_count_images_bit_exact_full_size(shape):
    """Integer work is bit-exact: the histogram of ROUNDED warped coordinates (fp64 warp, half-to-even, JAX wrap/drop) equals
    the oracle's at BASELINE sizes (10^6 events), every reference time, including events leaving the frame."""
    if shape == 'c3_dense':
        H, W, N, R = 480, 640, 1_000_000, 3
        win = synth.make_window(81, (H, W), N, R, flow='smooth', flow_mag=30.0)
        th = win['flow_gt'] * np.random.default_rng(1).uniform(0.5, 1.5, (H, W, 2))
    else:
        H, W, N, R = 260, 346, 1_000_000, 5
        big = shape.endswith('bigflow')
        win = synth.make_window(82, (H, W), N, R, flow='smooth' if big else 'constant', flow_mag=120.0 if big else 20.0)
        th = synth.theta_near_truth(82, win, (16, 16) if big else (1, 1))
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 1), want_grad=False)
        cnt = eng.count_images()[0]
    Theta = O.scale_theta_to_sensor_size(th, (H, W))
    total = 0
    for r in range(R):
        wx, wy = O.per_pix_warp(Theta, win['xs'], win['ys'], win['ts'], win['edge_ts'][r])
        ref = O.rounded_count_image(wx, wy, (H, W))
        assert np.array_equal(cnt[r].astype(np.int64), ref), f'reference time {r}: {np.count_nonzero(cnt[r] != ref)} pixels differ'
        total += int(ref.sum())
    assert total <= N * R and (total < N * R) == (shape != 'headline_2dof' or total < N * R)


def test_c5_shape_ten_million_events():
    """BASELINE C5 shape: 480x640, 10^7 events, R = 3, 16x16 theta.  The oracle's float path is too slow for a test at this
    size; its integer skeleton is not: rounded-coordinate count images must match bit for bit, the theta = 0 pass must
    reproduce the IUE at every reference time, and the gradient must be finite."""
    H, W, N, R = 480, 640, 10_000_000, 3
    win = synth.make_window(91, (H, W), N, R, flow='smooth', flow_mag=30.0)
    th = synth.theta_near_truth(91, win, (16, 16))
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v0, _, _ = eng.loss_grad(np.zeros((1, 1, 2)), engine.make_params(2000.0, 4000.0, 0.0, 0.0, 4), want_grad=False)
        assert v0[0] == pytest.approx(-(2000.0 + 4000.0) / R, rel=TOL)
        iw, z = eng.iwes()[0], eng.zero_iwe()[0]
        assert all(rel(iw[r], z) <= 2e-6 for r in range(R))
        v, g, _ = eng.loss_grad(th, engine.make_params(2000.0, 4000.0, 2.5e-4, 0.0, 0))
        assert np.isfinite(v[0]) and np.all(np.isfinite(g)) and np.abs(g).max() > 0
        cnt = eng.count_images()[0]
    Theta = O.scale_theta_to_sensor_size(th, (H, W))
    for r in range(R):
        wx, wy = O.per_pix_warp(Theta, win['xs'], win['ys'], win['ts'], win['edge_ts'][r])
        assert np.array_equal(cnt[r].astype(np.int64), O.rounded_count_image(wx, wy, (H, W)))


def test_integer_shift_identity_full_size():
    """Constant integer displacement (theta = (k,0), all t - tau = 1): IWE = IUE shifted by -k columns (C.4 ii)."""
    H, W, N, k = 260, 346, 200000, 7
    rng = np.random.default_rng(8)
    xs = rng.integers(0, W, N).astype(np.int16)
    ys = rng.integers(0, H, N).astype(np.int16)
    ts = np.ones(N)
    edges = rng.uniform(0, 1, (1, H, W))
    with engine.Engine((H, W), N, max_refs=1) as eng:
        eng.set_window(xs, ys, ts, edges, np.array([0.0]))
        eng.loss_grad(np.array([[[float(k), 0.0]]]), engine.make_params(20.0, 35.0, 0.0, 0.0, 4), want_grad=False)
        I, I0 = eng.iwes()[0, 0], eng.zero_iwe()[0]
    assert rel(I[:, 1:W - k - 1], I0[:, 1 + k:W - 1]) <= 2e-6


# ---------------------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------------------
def test_border_events_wrap_and_drop():
    """Events on the sensor border warped across it: JAX scatter rules (negative index wraps, >= n drops)."""
    H, W = 40, 50
    rng = np.random.default_rng(3)
    N = 4000
    side = rng.integers(0, 4, N)
    xs = np.where(side == 0, 0, np.where(side == 1, W - 1, rng.integers(0, W, N))).astype(np.int16)
    ys = np.where(side == 2, 0, np.where(side == 3, H - 1, rng.integers(0, H, N))).astype(np.int16)
    ts = np.sort(rng.uniform(0, 1, N))
    edges = rng.uniform(0, 1, (2, H, W))
    edge_ts = np.array([0.0, 1.0])
    th = np.array([[[9.0, -7.0]]])
    args = (xs, ys, ts, edges, edge_ts)
    v_ref, g_ref, aux = O.loss_and_grad(th, *args, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W), return_intermediates=True)
    with engine.Engine((H, W), N, max_refs=2) as eng:
        eng.set_window(*args)
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4))
        assert rel(eng.iwes()[0], aux['_iwes']) <= TOL
    assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
    assert rel(g[0], g_ref) <= TOL


def test_huge_displacement_takes_the_direct_path():
    """Displacements far beyond the LDS window (and beyond the sensor): the clamped window + direct-to-HBM taps
    must give the same image as the oracle, and mostly-dropped events must not fault."""
    H, W, N, R = 96, 128, 20000, 2
    win = synth.make_window(31, (H, W), N, R, flow='smooth', flow_mag=5.0)
    th = np.zeros((2, 2, 2))
    th[0, 0] = (400.0, -90.0); th[0, 1] = (-35.0, 260.0); th[1, 0] = (60.0, 70.0); th[1, 1] = (-1e9, 3.0)
    v_ref, g_ref, aux = O.loss_and_grad(th, *win_args(win), 20.0, 35.0, 0.0, 0.0, 3, 5, (H, W), return_intermediates=True)
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 3))
        assert rel(eng.iwes()[0], aux['_iwes']) <= TOL
    assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
    assert rel(g[0], g_ref) <= 5 * TOL


@pytest.mark.parametrize('N', [1, 2, 257])
def test_tiny_windows(N):
    H, W = 35, 47
    rng = np.random.default_rng(N)
    xs = rng.integers(0, W, N).astype(np.int16)
    ys = rng.integers(0, H, N).astype(np.int16)
    ts = np.sort(rng.uniform(0, 1, N))
    edges = rng.uniform(0, 1, (3, H, W))
    edge_ts = np.array([0.0, 0.5, 1.0])
    th = np.array([[[3.3, -2.1]]])
    v_ref, g_ref, _ = O.loss_and_grad(th, xs, ys, ts, edges, edge_ts, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W))
    with engine.Engine((H, W), 1000, max_refs=3) as eng:
        eng.set_window(xs, ys, ts, edges, edge_ts)
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4))
    assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
    assert rel(g[0], g_ref) <= TOL


def test_empty_window_is_nonfinite_not_a_crash():
    """No events: IUE is all zero, zero_contrast = 0, the reference's ratios are 0/eps or NaN; must not fault."""
    H, W = 35, 47
    edges = np.random.default_rng(0).uniform(0, 1, (2, H, W))
    with engine.Engine((H, W), 16, max_refs=2) as eng:
        eng.set_window(np.zeros(0, np.int16), np.zeros(0, np.int16), np.zeros(0), edges, np.array([0.0, 1.0]))
        v, g, _ = eng.loss_grad(np.zeros((1, 1, 2)), engine.make_params(20.0, 35.0, 0.0, 0.0, 4))
        v_ref, _, _ = O.loss_and_grad(np.zeros((1, 1, 2)), np.zeros(0, np.int16), np.zeros(0, np.int16), np.zeros(0), edges,
                                      np.array([0.0, 1.0]), 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W))
        assert (np.isnan(v[0]) and np.isnan(v_ref)) or v[0] == pytest.approx(v_ref, rel=TOL)
        assert np.all(g == 0.0) or np.all(np.isnan(g))


def test_batch_equals_individual_windows():
    H, W, R = 100, 132, 3
    wins = [synth.make_window(40 + i, (H, W), n, R, flow='smooth', flow_mag=10.0) for i, n in enumerate((5000, 20000, 1234))]
    ths = np.stack([synth.theta_near_truth(40 + i, w, (4, 4)) for i, w in enumerate(wins)])
    p = engine.make_params(20.0, 35.0, 2.5e-4, 0.0, 0)
    singles = []
    for w, th in zip(wins, ths):
        with engine.Engine((H, W), 20000, max_refs=R) as eng:
            eng.set_window(*win_args(w))
            v, g, _ = eng.loss_grad(th, p)
            singles.append((v[0], g[0]))
    with engine.Engine((H, W), 30000, max_refs=R, max_windows=3) as eng:
        eng.set_windows([win_args(w) for w in wins])
        v, g, _ = eng.loss_grad(ths, p)
    for b in range(3):
        assert v[b] == pytest.approx(singles[b][0], rel=2e-6)
        assert rel(g[b], singles[b][1]) <= 2e-5
        v_ref, g_ref, _ = O.loss_and_grad(ths[b], *win_args(wins[b]), 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W))
        assert abs(v[b] - v_ref) <= TOL * abs(v_ref)
        assert rel(g[b], g_ref) <= TOL


def test_device_and_host_binning_agree(monkeypatch):
    """eincm_set_windows bins events on the GPU (eincm_binning.hip.h); EINCM_HOST_BINNING=1 selects the host counting sort.
    Same segments up to event order inside a tile -> same loss / gradient / IWE up to fp32 summation order."""
    H, W, R = 150, 200, 3
    wins = [synth.make_window(70 + i, (H, W), n, R, flow='smooth', flow_mag=12.0) for i, n in enumerate((40000, 9000, 1))]
    ths = np.stack([synth.theta_near_truth(70 + i, w, (4, 4)) for i, w in enumerate(wins)])
    p = engine.make_params(20.0, 35.0, 2.5e-4, 0.0, 0)
    res = {}
    for mode in ('device', 'host'):
        if mode == 'host':
            monkeypatch.setenv('EINCM_HOST_BINNING', '1')
        else:
            monkeypatch.delenv('EINCM_HOST_BINNING', raising=False)
        with engine.Engine((H, W), 60000, max_refs=R, max_windows=3) as eng:
            eng.set_windows([win_args(w) for w in wins])
            v, g, _ = eng.loss_grad(ths, p)
            res[mode] = (v, g, eng.iwes(), eng.zero_iwe())
    monkeypatch.delenv('EINCM_HOST_BINNING', raising=False)
    assert np.allclose(res['device'][0], res['host'][0], rtol=2e-6)
    assert rel(res['device'][1], res['host'][1]) <= 2e-5
    assert rel(res['device'][2], res['host'][2]) <= 2e-6 and rel(res['device'][3], res['host'][3]) <= 2e-6
    v_ref, g_ref, _ = O.loss_and_grad(ths[0], *win_args(wins[0]), 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W))
    assert abs(res['device'][0][0] - v_ref) <= TOL * abs(v_ref) and rel(res['device'][1][0], g_ref) <= TOL


def test_handover():
    H, W, R = 100, 132, 3
    win = synth.make_window(50, (H, W), 20000, R, flow='smooth', flow_mag=10.0)
    th = synth.theta_near_truth(50, win, (2, 2))
    prev = 0.8 * th + 0.5
    v_ref, dv_ref = O.handover_loss_and_grad(0.37, prev, th, *win_args(win), alpha=20.0, beta=35.0, gamma=0.0, delta=0.0,
                                             cur_pyr_lvl=1, n_pyr_lvls=5, sensor_size=(H, W))
    v, dv = losses.value_and_grad_handover_loss_func(0.37, prev, th, *win_args(win), 20.0, 35.0, 0.0, 0.0, 1, 5, (H, W), 'bilinear')
    assert v == pytest.approx(v_ref, rel=TOL)
    assert dv == pytest.approx(dv_ref, rel=TOL, abs=TOL * abs(v_ref))
    assert losses.handover_loss_func(0.37, prev, th, *win_args(win), 20.0, 35.0, 0.0, 0.0, 1, 5, (H, W), 'bilinear') == \
        pytest.approx(v_ref, rel=TOL)
    losses.clear_engine_cache()


def test_reference_shaped_callables():
    """losses.loss_func / value_and_grad_loss_func / compute_loss_objectives: the reference's signatures and aux keys."""
    H, W, R = 100, 132, 5
    win = synth.make_window(60, (H, W), 30000, R, flow='smooth', flow_mag=10.0)
    th = synth.theta_near_truth(60, win, (16, 16))
    a = win_args(win)
    v_ref, aux_ref = O.loss_func(th, *a, 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W), 'bilinear')
    v, aux = losses.loss_func(th, *a, 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W), 'bilinear')
    assert set(aux) == set(aux_ref)            # losses.py:195-203 keys
    assert v == pytest.approx(v_ref, rel=TOL)
    for k in ('final_loss', 'mean_rel_corr', 'mean_rel_contrast', 'mean_rel_iwe_divergence', 'theta_total_variation'):
        assert aux[k] == pytest.approx(aux_ref[k], rel=TOL), k
    assert rel(aux['scaled_theta'], aux_ref['scaled_theta']) <= 1e-13
    assert rel(aux['multi_ref_weights'], aux_ref['multi_ref_weights']) <= 1e-15
    (v2, _), g = losses.value_and_grad_loss_func(th, *a, 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W), 'bilinear')
    _, g_ref, _ = O.loss_and_grad(th, *a, 20.0, 35.0, 2.5e-4, 0.0, 0, 5, (H, W), 'bilinear')
    assert v2 == pytest.approx(v_ref, rel=TOL) and rel(g, g_ref) <= TOL
    lo_ref = O.compute_loss_objectives(aux_ref['scaled_theta'], *a, (H, W))
    lo = losses.compute_loss_objectives(aux_ref['scaled_theta'], *a, (H, W))
    assert set(k for k in lo_ref if not k.startswith('_')) <= set(lo)       # every key of losses.py:89-105
    for k, vref in lo_ref.items():
        if k.startswith('_'):
            continue
        if k in ('warped_xs', 'warped_ys'):                                 # fp64, the reference's two roundings: bit for bit
            assert lo[k].shape == vref.shape and np.array_equal(lo[k], vref), k
            continue
        assert rel(lo[k], vref) <= TOL, k
    assert len(losses._CACHE) == 1             # one staged window reused by all four calls
    losses.clear_engine_cache()


def test_engine_cache_notices_in_place_edits():
    """losses.py keeps the staged window while the same array objects are passed; numpy arrays are mutable, so a content fingerprint
    (256 strided samples per array) re-stages after an in-place edit of the events or the edges."""
    H, W, R = 64, 96, 3
    win = synth.make_window(61, (H, W), 6000, R, flow='constant', flow_mag=4.0)
    a = [np.array(x) for x in win_args(win)]
    th = synth.theta_near_truth(61, win, (1, 1))
    call = lambda: losses.value_and_grad_loss_func(th, *a, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W))
    (v0, _), g0 = call()
    (v1, _), g1 = call()
    assert v1 == v0 and np.array_equal(g0, g1) and len(losses._CACHE) == 1
    eng0 = losses._CACHE[0][2]
    a[2][:] = a[2][::-1].copy()                              # the same objects, other timestamps
    (v2, _), g2 = call()
    assert losses._CACHE[0][2] is not eng0 and v2 != v0
    v_ref, g_ref, _ = O.loss_and_grad(th, *a, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W))
    assert v2 == pytest.approx(v_ref, rel=TOL) and rel(g2, g_ref) <= TOL
    a[3] *= 0.5                                              # edges edited in place
    (v3, _), _ = call()
    v_ref3, _, _ = O.loss_and_grad(th, *a, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W))
    assert v3 == pytest.approx(v_ref3, rel=TOL) and v3 != v2
    losses.clear_engine_cache()


def test_warped_events_of_a_batch(monkeypatch):
    """eincm_get_warped_events: per window of a batch, caller's event order, 2-DoF theta (no Theta image until asked), a theta grid, and the
    host-binned staging path; bit for bit the oracle's per_pix_warp (event_warpers.py:28-37)."""
    H, W, R = 64, 96, 3
    wins = [synth.make_window(80 + b, (H, W), n, R, flow='smooth', flow_mag=6.0) for b, n in enumerate((5000, 0, 1777))]
    for host_binning in (False, True):
        if host_binning:
            monkeypatch.setenv('EINCM_HOST_BINNING', '1')
        with engine.Engine((H, W), 7000, max_refs=R, max_windows=3) as eng:
            eng.set_windows([win_args(w) for w in wins])
            for hw in ((1, 1), (4, 4)):
                th = np.stack([synth.theta_near_truth(80 + b, w, hw) for b, w in enumerate(wins)])
                eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 2))
                Theta = eng.scaled_theta()
                for b, w in enumerate(wins):
                    wx, wy = eng.warped_events(b)
                    assert wx.shape == wy.shape == (R, len(w['xs']))
                    for r in range(R):
                        rx, ry = O.per_pix_warp(Theta[b], w['xs'], w['ys'], w['ts'], w['edge_ts'][r], 1.0)
                        assert np.array_equal(wx[r], rx) and np.array_equal(wy[r], ry), (host_binning, hw, b, r)
            with pytest.raises(engine.EincmError):
                eng.warped_events(3)


# ---------------------------------------------------------------------------------------------------
# error behaviour of the boundary
# ---------------------------------------------------------------------------------------------------
def test_errors():
    H, W = 40, 50
    edges = np.zeros((1, H, W))
    with engine.Engine((H, W), 100, max_refs=1) as eng:
        with pytest.raises(engine.EincmError) as e:
            eng.B = 1
            eng.loss_grad(np.zeros((1, 1, 2)), engine.make_params(1, 1, 0, 0, 0))
        assert e.value.code == L.ERR_STATE
        with pytest.raises(engine.EincmError) as e:      # x == W is outside the sensor
            eng.set_window(np.array([W], np.int16), np.array([0], np.int16), np.array([0.5]), edges, np.array([0.0]))
        assert e.value.code == L.ERR_ARG and 'outside' in str(e.value)
        with pytest.raises(engine.EincmError) as e:      # too many events for this context
            eng.set_window(np.zeros(101, np.int16), np.zeros(101, np.int16), np.zeros(101), edges, np.array([0.0]))
        assert e.value.code == L.ERR_ARG
        with pytest.raises(engine.EincmError) as e:      # more reference times than the context was built for
            eng.set_window(np.zeros(1, np.int16), np.zeros(1, np.int16), np.zeros(1), np.zeros((2, H, W)), np.array([0.0, 1.0]))
        assert e.value.code == L.ERR_ARG
        xs = np.arange(30, dtype=np.int16); ys = np.arange(30, dtype=np.int16)
        eng.set_window(xs, ys, np.linspace(0, 1, 30), np.random.default_rng(0).uniform(0, 1, (1, H, W)), np.array([0.0]))
        th = np.array([[[np.nan, 0.0]]])
        v, g, _ = eng.loss_grad(th, engine.make_params(20, 35, 0, 0, 4))          # allow_nonfinite=True: values returned
        assert np.isnan(v[0])
        with pytest.raises(engine.NonFiniteLoss):
            eng.loss_grad(th, engine.make_params(20, 35, 0, 0, 4), allow_nonfinite=False)
        # the context is still usable afterwards
        v, _, _ = eng.loss_grad(np.zeros((1, 1, 2)), engine.make_params(20, 35, 0, 0, 4))
        assert np.isfinite(v[0])


def test_half_pixel_ties_and_near_ties():
    """Displacements of exactly k + 1/2 px (ties: half-to-even on x - v*dt, which depends on the parity of x) and just beside
    them must land on the oracle's pixels: a wrong rounding moves a whole 3x3 stamp by one pixel."""
    H, W, R = 64, 96, 2
    rng = np.random.default_rng(4)
    n = 6000
    xs = rng.integers(0, W, n).astype(np.int16); ys = rng.integers(0, H, n).astype(np.int16)
    ts = rng.choice([0.25, 0.5, 0.75, 1.0], n)                      # exact binary fractions
    order = np.argsort(ts, kind='stable'); xs, ys, ts = xs[order], ys[order], ts[order]
    edges = rng.random((R, H, W)); edge_ts = np.array([0.0, 1.0])
    with engine.Engine((H, W), n, max_refs=R) as eng:
        eng.set_window(xs, ys, ts, edges, edge_ts)
        for theta in ([2.0, -6.0], [2.0 + 3e-5, -6.0 - 3e-5], [30.0, 31.5], [33.0, -2.0]):
            th = np.array(theta).reshape(1, 1, 2)
            v_ref, g_ref, aux = O.loss_and_grad(th, xs, ys, ts, edges, edge_ts, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W), return_intermediates=True)
            v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4))
            assert abs(v[0] - v_ref) <= TOL * abs(v_ref), theta
            assert rel(g[0], g_ref) <= TOL, theta
            assert rel(eng.iwes()[0], aux['_iwes']) <= TOL, theta
            Theta = O.scale_theta_to_sensor_size(th, (H, W))
            cnt = eng.count_images()[0]
            for r in range(R):
                wx, wy = O.per_pix_warp(Theta, xs, ys, ts, edge_ts[r])
                assert np.array_equal(cnt[r], O.rounded_count_image(wx, wy, (H, W))), theta


def test_long_splat_segments_use_two_lds_windows(monkeypatch):
    """EINCM_SEG_SPLAT > EINCM_CHUNK selects k_splat's multi-chunk form (u32 chunk window + f32 segment window in LDS).  With a
    large-flow 16x16 theta the per-evaluation window capacity is raised; it must stay within the 64 KiB of dynamic LDS a launch
    gets (2 x 6912 x 4 B + the 16 KiB Theta tile did not)."""
    monkeypatch.setenv('EINCM_SEG_SPLAT', '8192')
    monkeypatch.setenv('EINCM_CHUNK', '4096')
    H, W, N, R = 130, 170, 150000, 3
    win = synth.make_window(17, (H, W), N, R, flow='smooth', flow_mag=90.0)
    th = synth.theta_near_truth(17, win, (16, 16))
    v_ref, g_ref, aux = O.loss_and_grad(th, *win_args(win), 20.0, 35.0, 0.0, 0.0, 0, 5, (H, W), return_intermediates=True)
    with engine.Engine((H, W), N, max_refs=R) as eng:
        eng.set_window(*win_args(win))
        v, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 0))
        assert abs(v[0] - v_ref) <= TOL * abs(v_ref)
        assert rel(g[0], g_ref) <= TOL
        assert rel(eng.iwes()[0], aux['_iwes']) <= TOL


def fma_tie_events(v, W, n_want, rng):
    """Events (x, t) for which fl(x - fl(v*t)) sits exactly on a half-integer while the unrounded x - v*t does not, chosen so that
    the two round to DIFFERENT pixels: what a fused multiply-add (one rounding) gets wrong against the reference's two roundings
    (/root/reference/src/eincm/event_warpers.py:34, `xs - theta * dts * delta_time`).  Exact arithmetic with fractions."""
    from fractions import Fraction
    xs, ts = [], []
    fv = Fraction(v)
    for m in (int(q) for q in rng.permutation(np.arange(8, W - 8))):
        p = float(m) + 0.5                              # the rounded product we want: a half-integer
        t0 = p / v
        for t in (t0, np.nextafter(t0, 0.0), np.nextafter(t0, 2.0)):
            if not (0.0 < t <= 1.0) or v * t != p:
                continue
            eps = fv * Fraction(t) - Fraction(p)        # exact product minus its fp64 rounding
            if eps == 0:
                continue
            for k in (2, 3, 4, 5):                      # w = x - p = k + 0.5: half-to-even gives k (k even) or k + 1 (k odd)
                x = m + k + 1                           # x - (m + 0.5) = k + 0.5
                if x >= W:
                    continue
                w_two = float(x) - v * t                # numpy / the reference: product rounded first
                exact = Fraction(x) - fv * Fraction(t)
                r_two = int(np.rint(w_two))
                r_fma = int(np.rint(float(exact)))      # float(Fraction) rounds once, like an FMA
                if w_two == k + 0.5 and r_fma != r_two:
                    xs.append(x); ts.append(t)
                    break
            break
        if len(xs) >= n_want:
            break
    return np.array(xs, dtype=np.int16), np.array(ts, dtype=np.float64)


def test_warp_rounds_the_product_first():
    """The integer-deciding code performs the reference's two roundings, w = fl(x - fl(theta*dt)), not one fused multiply-add:
    on events constructed so that the two disagree about rint(w), the rounded-coordinate count image equals the oracle's bit for
    bit (with `x - v*dt` contracted to v_fma_f64 every one of these events lands one pixel off)."""
    H, W, R = 48, 346, 1
    rng = np.random.default_rng(11)
    v = 297.3187654321                                   # px per window along x; dt = t - 0
    xs_c, ts_c = fma_tie_events(v, W, 60, rng)
    assert len(xs_c) >= 40, 'construction found too few FMA-sensitive events'
    n_bg = 4000                                          # background events so that the window is an ordinary one
    xs = np.concatenate([xs_c, rng.integers(0, W, n_bg).astype(np.int16)])
    ts = np.concatenate([ts_c, rng.random(n_bg)])
    ys = rng.integers(0, H, len(xs)).astype(np.int16)
    order = np.argsort(ts, kind='stable'); xs, ys, ts = xs[order], ys[order], ts[order]
    edges = rng.random((R, H, W)); edge_ts = np.array([0.0])
    th = np.array([v, 0.0]).reshape(1, 1, 2)
    Theta = O.scale_theta_to_sensor_size(th, (H, W))
    wx, wy = O.per_pix_warp(Theta, xs, ys, ts, edge_ts[0])
    ref = O.rounded_count_image(wx, wy, (H, W))
    with engine.Engine((H, W), len(xs), max_refs=R) as eng:
        eng.set_window(xs, ys, ts, edges, edge_ts)
        eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4), want_grad=False)
        cnt = eng.count_images()[0]
        assert np.array_equal(cnt[0].astype(np.int64), ref), f'{np.count_nonzero(cnt[0] != ref)} pixels differ'
        # the float path sees the same decisions: IWE and gradient against the oracle
        v_ref, g_ref, aux = O.loss_and_grad(th, xs, ys, ts, edges, edge_ts, 20.0, 35.0, 0.0, 0.0, 4, 5, (H, W), return_intermediates=True)
        vv, g, _ = eng.loss_grad(th, engine.make_params(20.0, 35.0, 0.0, 0.0, 4))
        assert abs(vv[0] - v_ref) <= TOL * abs(v_ref)
        assert rel(eng.iwes()[0], aux['_iwes']) <= TOL


def test_masked_evaluation_matches_the_full_one():
    """eincm_loss_grad_masked: the windows that take part get exactly the values and gradients of a full evaluation (2-DoF and a
    theta grid), the others NaN / 0, and the next full evaluation is unaffected (no accumulator is left dirty)."""
    H, W, N, R, B = 96, 128, 15000, 3, 4
    wins = [synth.make_window(120 + b, (H, W), N, R, flow='constant', flow_mag=5.0) for b in range(B)]
    with engine.Engine((H, W), B * N, max_refs=R, max_windows=B) as eng:
        eng.set_windows([win_args(w) for w in wins])
        for hw, lvl in (((1, 1), 4), ((4, 4), 1)):
            th = np.stack([synth.theta_near_truth(120 + b, w, hw) for b, w in enumerate(wins)])
            p = engine.make_params(20.0, 35.0, 0.0, 0.0, lvl)
            v0, g0, _ = eng.loss_grad(th, p)
            mask = np.array([True, False, True, False])
            v1, g1, _ = eng.loss_grad(th, p, active=mask)
            assert np.array_equal(v1[mask], v0[mask]) and np.array_equal(g1[mask], g0[mask])
            assert np.all(np.isnan(v1[~mask])) and np.all(g1[~mask] == 0.0)
            v2, g2, _ = eng.loss_grad(th, p)
            assert np.array_equal(v2, v0) and np.array_equal(g2, g0)
