"""Large-sample check of the in-kernel Dirichlet sampler (developer tool; results in profiles/rNN_sampler_soak.txt).
The -m gpu suite pins the sampler with a few thousand draws per case; this runs 2^22 .. 2^24 draws per case so that a
distortion of the marginals at the 1e-3 level (a wrong bit field, a biased acceptance test, a truncated normal tail)
would show: KS distance of Beta marginals, z-scores of the first two moments of every entry, frequency of the rows'
largest entry.  Regimes: the reference policy (shapes 1e2 .. 1e5), a mid regime (shapes around 1 .. 40) and one with all
shapes below 1 (the boosted path); d = 2 (one Box-Muller pair per row), d = 5 (quad + single), d = 21 (the benchmark size).
usage: python tools/sampler_soak.py [log2_draws]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy import stats
from discrete_mean_field_game_amd import ops
from oracle import mfg_oracle as O

dev = torch.device('cuda:0')
LOG2 = int(sys.argv[1]) if len(sys.argv) > 1 else 22
REGIMES = {'policy': (8.86349, 0.16, 12000.0), 'mid': (5.0, 0.1, 40.0), 'small': (6.0, 0.3, 2.5)}


def run(d, regime, precision, n_total, seed_offset=0):
    theta, shift, scale = REGIMES[regime]
    rs = np.random.RandomState(100 + d)
    pi1 = rs.dirichlet(np.ones(d) * 0.8).astype(np.float32)
    al = O.calc_alpha(pi1.astype(np.float64), theta, shift) * scale
    A = al.sum(-1, keepdims=True)
    # KS entries: a row and two columns whose Beta shapes (a, A - a) are both >= 0.5 -- below that the marginal piles up within
    # fp32 resolution of 0 / 1 and the STORED value is numerically a point mass (the -m gpu suite skips such entries too)
    ok = [(i, j) for i in range(d) for j in range(d) if al[i, j] >= 0.5 and A[i, 0] - al[i, j] >= 0.5]
    rows_ok = sorted({i for i, _ in ok}, key=lambda i: -sum(1 for a_, _b in ok if a_ == i))
    ks_row = rows_ok[0] if rows_ok else 0
    ks_cols = [j for i, j in ok if i == ks_row][:2]
    chunk = min(n_total, 1 << 20)
    s1 = torch.zeros(d, d, dtype=torch.float64, device=dev)
    s2 = torch.zeros(d, d, dtype=torch.float64, device=dev)
    keep = []
    th = torch.tensor([theta], dtype=torch.float64, device=dev)
    pi = torch.as_tensor(np.repeat(pi1[None], chunk, 0), device=dev)
    done = 0
    while done < n_total:
        P = ops.sample_dirichlet(pi, th, shift, scale, seed=12345 + d + seed_offset, step=7, traj_offset=done, precision=precision).double()
        s1 += P.sum(0)
        s2 += (P * P).sum(0)
        if len(keep) < 8:
            keep.append(P[:, ks_row, ks_cols].cpu().numpy())  # two entries for the KS test (8 chunks = 2^23 draws at most)
        done += chunk
    n = float(done)
    mean = al / A
    var = mean * (1 - mean) / (A + 1)
    m_hat = (s1 / n).cpu().numpy()
    v_hat = ((s2 / n).cpu().numpy() - m_hat ** 2) * n / (n - 1)
    z = (m_hat - mean) / np.sqrt(var / n)
    b_ = A - al
    kurt = 6 * ((al - b_) ** 2 * (A + 1) - al * b_ * (A + 2)) / (al * b_ * (A + 2) * (A + 3))
    zv = (v_hat / var - 1) / np.sqrt(np.maximum(kurt, 0) / n + 2.0 / (n - 1))
    xs = np.concatenate(keep, 0)
    ks = [stats.kstest(xs[:, c], stats.beta(al[ks_row, j], A[ks_row, 0] - al[ks_row, j]).cdf) for c, j in enumerate(ks_cols)]
    zvs = ('%.2f' % np.abs(zv[kurt < 1.0]).max()) if (kurt < 1.0).any() else 'n/a (all entries heavy tailed)'
    kss = ', '.join('P[%d,%d] D = %.2e (p = %.3f)' % (ks_row, j, k.statistic, k.pvalue) for j, k in zip(ks_cols, ks)) or 'no entry with both shapes >= 0.5'
    print('d=%2d %-6s %-5s draws 2^%d: max |z| of the means %.2f, of the variances %s (over %d entries); KS over %d draws: %s'
          % (d, regime, precision, int(np.log2(n)), np.abs(z).max(), zvs, d * d, xs.shape[0], kss), flush=True)
    return np.abs(z).max(), min([k.pvalue for k in ks] + [1.0]), len(ks)


if __name__ == '__main__':
    worst_z, worst_p, nks, worst_case = 0.0, 1.0, 0, None
    for d in (2, 5, 21):
        for regime in ('policy', 'mid', 'small'):
            for precision in ('mixed', 'f64'):
                zz, pp, k = run(d, regime, precision, 1 << (LOG2 if d < 21 else LOG2 - 2))
                if pp < worst_p:
                    worst_case = (d, regime, precision, 1 << (LOG2 if d < 21 else LOG2 - 2))
                worst_z, worst_p, nks = max(worst_z, zz), min(worst_p, pp), nks + k
    print('worst |z| of a mean %.2f (%.0f entries in total: 5 sigma would be suspicious), smallest KS p-value %.4f (%d tests)'
          % (worst_z, 2 * 3 * (4 + 25 + 441), worst_p, nks))
    if worst_case is not None and worst_p < 0.02:
        # the smallest of 32 p-values is expected around 0.03; one well below that is re-drawn with three other Philox keys: a
        # distortion of the marginal reproduces (p collapses in every re-draw), a fluctuation does not
        print('re-draw of the case with the smallest p-value %s under three other seeds:' % (worst_case[:3],))
        for off in (1000, 2000, 3000):
            run(*worst_case, seed_offset=off)
