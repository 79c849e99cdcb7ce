"""Statistical comparison of Monte-Carlo dropout samples with the reference's own distribution
(tests/golden/mc_stats.npz: mean, covariance and quantiles of `monte_carlo_predictions` samples drawn by the
reference, nn_models.py:191-207).  Test infrastructure shared by the CPU (oracle) and GPU (HIP) tests.

Bounds (z = 5.5 standard errors each; ~130 comparisons per model set, false-alarm probability < 1e-5):
  mean      |m - m_ref| <= z * sqrt(var_ref / n_ref + var / n)                      per target
  std       |s / s_ref - 1| <= z * sqrt((k - 1) / 4 * (1 / n + 1 / n_ref)), k = sample kurtosis (>= 3 assumed)
  quantiles fraction of samples below the reference's q-quantile = q +- z * sqrt(q (1 - q) (1 / n + 1 / n_ref))
  correlations of target pairs within z * sqrt(1 / n + 1 / n_ref) absolute (n >= 8000; 0.07 at n = 8192)
A sampler with a wrong mask scale (no 1/(1-p)), a wrong p or a mask in the wrong place moves the means by tens of
standard errors (negative controls in the tests)."""
import numpy as np

Z = 5.5


def compare(samples, ref_mean, ref_cov, ref_quant, levels, n_ref, what=""):
    """samples [n, D] vs reference statistics; returns a list of violation strings (empty = consistent)."""
    x = np.asarray(samples, dtype=np.float64)
    n, D = x.shape
    bad = []
    m, v = x.mean(axis=0), x.var(axis=0, ddof=1)
    rv = np.diag(ref_cov)
    live = rv > 1e-24                       # constant outputs (none expected) carry no statistics
    se = np.sqrt(rv / n_ref + v / n)
    for d in np.nonzero(live)[0]:
        if abs(m[d] - ref_mean[d]) > Z * se[d]:
            bad.append(f"{what} mean[{d}]: {m[d]:.6g} vs {ref_mean[d]:.6g} ({abs(m[d] - ref_mean[d]) / se[d]:.1f} se)")
        xc = x[:, d] - m[d]
        kurt = max(3.0, float(np.mean(xc ** 4) / max(v[d] ** 2, 1e-300)))
        tol = Z * np.sqrt((kurt - 1.0) / 4.0 * (1.0 / n + 1.0 / n_ref))
        ratio = np.sqrt(v[d] / rv[d])
        if abs(ratio - 1.0) > tol:
            bad.append(f"{what} std[{d}]: ratio {ratio:.4f} (tolerance {tol:.4f})")
        for qi, q in enumerate(levels):
            frac = float(np.mean(x[:, d] < ref_quant[qi, d]))
            tolq = Z * np.sqrt(q * (1 - q) * (1.0 / n + 1.0 / n_ref))
            if abs(frac - q) > tolq:
                bad.append(f"{what} quantile {q}[{d}]: {frac:.4f} of the samples below the reference's (tolerance {tolq:.4f})")
    if n >= 8000 and live.sum() > 1:
        idx = np.nonzero(live)[0]
        c = np.corrcoef(x[:, idx], rowvar=False)
        sd = np.sqrt(rv[idx])
        rc = ref_cov[np.ix_(idx, idx)] / np.outer(sd, sd)
        worst = float(np.abs(c - rc).max())
        if worst > Z * np.sqrt(1.0 / n + 1.0 / n_ref):
            bad.append(f"{what} correlation matrix differs by {worst:.3f}")
    return bad
