"""-m gpu: distributional parity of EVERY element -> Philox-block mapping of the in-kernel Dirichlet sampler
(mfg_ac2.py:236-254: row i of P ~ Dirichlet(alpha_i. * alpha_scale), rows independent).

RNG bit parity with the reference's MT19937 stream is impossible for a parallel sampler, so the sampler is pinned
distributionally.  test_gpu_parity.py does that at d in {4, 21, 128, 130}; the kernels have more mappings than those:

  d <= 64   k_core_small     a lane owns a row; quads are 4 NEIGHBOURING columns (4q .. 4q+3) of the row
  d = 128   k_core_large<2>  quad = {(i, c), (i, c+64), (i+1, c), (i+1, c+64)}: two rows share a Philox block
  d = 192   k_core_large<3>  the same pairs + an odd last column: quad {(i, c+128), (i+1, c+128)} (2 elements)
  d = 256   k_core_large<4>  whole-row quads {(i, c), (i, c+64), (i, c+128), (i, c+192)}
  d = 320   k_core_large<5>  pairs + odd column, not a multiple of 64 per lane group ... (R = 5)

A slip in the counter mapping (two elements fed from the same bits, a Box-Muller partner reused, a row's variates
leaking into its neighbour) leaves every self-consistency test green; it shows up as (a) wrong marginals, (b) dependence
between elements that share a block.  Checked here, both precisions:
  * moments of every entry (z-score of the mean, chi-square band of the variance) and Kolmogorov-Smirnov of the Beta
    marginal of entries spread over all lane / column-group positions, in the policy's regime (shapes 1e3..1e5), in a
    small-shape regime (boosted Marsaglia-Tsang, shapes < 1) and in between;
  * the empirical covariance matrix of TWO NEIGHBOURING ROWS of P (all 2d x 2d entries, >= 20 000 draws) against the
    Dirichlet covariance: -m_j m_k / (A+1) inside a row, 0 across rows -- this covers every pair of elements that share
    a Philox block or a Box-Muller pair in any of the mappings above.
Everything is seeded; bounds are >= 6 sigma of the estimator, so the tests do not flake.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU: the HIP path has no CPU fallback')
    return torch.device('cuda:0')


def ops():
    from discrete_mean_field_game_amd import ops as _ops
    return _ops


def O():
    from oracle import mfg_oracle
    return mfg_oracle


def _draw(dev, pi1, B, theta, shift, scale, precision, seed, step=2, chunk=None):
    """[B, d, d] fp32 device tensor of actions for B copies of the state pi1 (trajectory ids 0 .. B-1)."""
    d = pi1.shape[0]
    th = torch.tensor([theta], dtype=torch.float64, device=dev)
    pi = torch.as_tensor(np.repeat(pi1[None].astype(np.float32), B, 0), device=dev)
    return ops().sample_dirichlet(pi, th, shift, scale, seed=seed, step=step, precision=precision)


# (theta, shift, alpha_scale): the reference policy (mfg_ac2.py:832; shapes ~1e3..1e5), a mid regime, and one with all
# shapes below 1 (every element takes the boosted small-shape path)
REGIMES = {'policy': (8.86349, 0.16, 12000.0), 'mid': (5.0, 0.1, 40.0), 'small': (6.0, 0.3, 2.5)}


@pytest.mark.parametrize('precision', ['mixed', 'f64'])
@pytest.mark.parametrize('regime', ['policy', 'mid', 'small'])
@pytest.mark.parametrize('d', [192, 256, 320, 448])
def test_large_d_sampler_marginals(dev, d, regime, precision):
    from scipy import stats
    theta, shift, scale = REGIMES[regime]
    rs = np.random.RandomState(1000 + d)
    pi1 = rs.dirichlet(np.ones(d) * 0.7).astype(np.float32)            # uneven state: concentrations differ across columns
    B = 3072
    P = _draw(dev, pi1, B, theta, shift, scale, precision, seed=77 + d).double()
    assert bool(torch.isfinite(P).all()) and bool((P > 0).all())
    assert float((P.sum(-1) - 1).abs().max()) < 1e-6                   # rows are stochastic (test2.py:14-32)
    al = O().calc_alpha(pi1, theta, shift) * scale                      # [d, d] shapes
    A = al.sum(-1, keepdims=True)
    if regime == 'small':
        assert al.max() < 1.0
    mean = al / A
    var = mean * (1 - mean) / (A + 1)
    m_hat = P.mean(0).cpu().numpy()
    v_hat = P.var(0, unbiased=True).cpu().numpy()
    z = (m_hat - mean) / np.sqrt(var / B)
    assert np.max(np.abs(z)) < 6.5, (np.unravel_index(np.argmax(np.abs(z)), z.shape), np.max(np.abs(z)))
    # sample variance: Var(s^2) = sigma^4 (excess kurtosis / B + 2 / (B-1))
    b_ = A - al
    kurt = 6 * ((al - b_) ** 2 * (A + 1) - al * b_ * (A + 2)) / (al * b_ * (A + 2) * (A + 3))   # Beta excess kurtosis
    tol = 6.5 * np.sqrt(np.maximum(kurt, 0) / B + 2.0 / (B - 1))
    ok = (np.abs(v_hat / var - 1) < tol) | (kurt > 1.0)                # per entry where s^2 is near normal (large shapes) ...
    assert ok.all(), (np.argwhere(~ok)[:5], (v_hat / var)[~ok][:5], tol[~ok][:5])
    # ... and in aggregate everywhere (heavy-tailed small-shape entries: d^2 ratios average to 1)
    assert abs(np.mean(v_hat / var) - 1.0) < 0.02
    # Kolmogorov-Smirnov of the Beta marginal at entries that cover every column group of a lane (c, c+64, ...), both
    # rows of a row pair, first / last lanes and the ragged tail
    cols = sorted({0, 1, 63, 64, 65, 127, 128, 129, d - 65, d - 64, d - 2, d - 1})
    rows = [0, 1, 2, d // 2 | 1, d - 2, d - 1]
    Ph = P[:, rows][:, :, cols].cpu().numpy()
    for a_, i in enumerate(rows):
        for b2, j in enumerate(cols):
            ks = stats.kstest(Ph[:, a_, b2], stats.beta(al[i, j], A[i, 0] - al[i, j]).cdf)
            assert ks.pvalue > 2e-5, (i, j, al[i, j], ks)               # 72 seeded tests per case


def _dirichlet_cov(al_row):
    A = al_row.sum()
    m = al_row / A
    return (np.diag(m) - np.outer(m, m)) / (A + 1)


@pytest.mark.parametrize('precision', ['mixed', 'f64'])
@pytest.mark.parametrize('d,rows', [(21, (0, 1)), (21, (19, 20)), (15, (6, 7)), (47, (2, 3)), (128, (0, 1)), (128, (126, 127)),
                                    (192, (4, 5)), (256, (0, 1)), (256, (200, 201)), (320, (318, 319))])
def test_no_dependence_between_elements_that_share_a_philox_block(dev, d, rows, precision):
    """Covariance of two neighbouring rows of P over >= 20 000 draws vs the product-Dirichlet covariance (6.5 sigma of the
    estimator per entry, all (2d)^2 entries): within a row -m_j m_k/(A+1), across rows 0.  A correlation rho between the
    gamma variates of two elements fed from one Philox block would move their entry by rho sigma_j sigma_k, i.e. by
    rho sqrt(B) standard errors: the test sees |rho| > ~0.05."""
    theta, shift, scale = REGIMES['policy']
    rs = np.random.RandomState(4000 + d + rows[0])
    pi1 = rs.dirichlet(np.ones(d)).astype(np.float32)
    B = 20480
    al = O().calc_alpha(pi1, theta, shift) * scale
    # draw in chunks (d = 320: 20 480 x 320 x 320 fp32 would be 8.4 GB), keep only the two rows, accumulate in fp64
    X = []
    th = torch.tensor([theta], dtype=torch.float64, device=dev)
    ch = 2048
    pi = torch.as_tensor(np.repeat(pi1[None], ch, 0), device=dev)
    for c0 in range(0, B, ch):
        P = ops().sample_dirichlet(pi, th, shift, scale, seed=991, step=5, traj_offset=c0, precision=precision)
        X.append(P[:, list(rows)].reshape(ch, 2 * d).double())
    X = torch.cat(X, 0)
    Xc = X - X.mean(0, keepdim=True)
    C_hat = (Xc.T @ Xc / (B - 1)).cpu().numpy()
    C = np.zeros((2 * d, 2 * d))
    C[:d, :d] = _dirichlet_cov(al[rows[0]])
    C[d:, d:] = _dirichlet_cov(al[rows[1]])
    v = np.diag(C)
    se = np.sqrt((np.outer(v, v) + C * C) / B)                         # normal-theory s.e. of a sample covariance
    zs = (C_hat - C) / se
    off = ~np.eye(2 * d, dtype=bool)
    worst = np.unravel_index(np.argmax(np.abs(zs) * off), zs.shape)
    assert np.abs(zs[off]).max() < 6.5, (worst, zs[worst], C_hat[worst], C[worst])
    # the z-scores of the cross-row block are N(0,1): their mean square is 1 +- a few / d
    cross = zs[:d, d:]
    assert abs((cross ** 2).mean() - 1.0) < 8.0 * np.sqrt(2.0) / d + 0.02


@pytest.mark.parametrize('d', [21, 256])
def test_quad_partners_uncorrelated_in_the_small_shape_regime(dev, d):
    """Shapes < 1 take the boosted path (Gamma(a) = Gamma(a+1) U^(1/a), extra Philox blocks keyed by the ELEMENT): rank
    correlation between quad partners of one row must be sampling noise (Spearman: the marginals are far from normal)."""
    from scipy import stats
    theta, shift, scale = REGIMES['small']
    rs = np.random.RandomState(9 + d)
    pi1 = rs.dirichlet(np.ones(d)).astype(np.float32)
    B = 16384
    P = _draw(dev, pi1, B, theta, shift, scale, 'mixed', seed=5)
    row = P[:, 1].cpu().numpy().astype(np.float64)
    pairs = [(0, 1), (0, 2), (0, 3), (2, 3), (1, 3)] if d == 21 else [(5, 69), (5, 133), (5, 197), (69, 197), (133, 197)]
    for (j, k) in pairs:
        rho = stats.spearmanr(row[:, j], row[:, k]).statistic
        # Dirichlet rows are negatively correlated by the normalisation: -sqrt(m_j m_k / ((1-m_j)(1-m_k))) ~ -1/d here
        assert abs(rho) < 6.0 / np.sqrt(B) + 2.0 / d, (j, k, rho)
