"""2DSWD evaluation of generated alignments: site-frequency spectrum per alignment -> PCA(2) of real
and generated spectra together -> sliced 2-Wasserstein distance between the two point clouds.

Host-side numpy restatement of the reference's evaluation numerics
(/root/reference/src/data_processing_v2.py:50-121 `plot_sfs_pca`, /root/reference/src/sliced.py:14-130),
without scikit-allel and POT, which the reference calls but which are not needed for these numerics:
  * `allel.sfs(dac)` of derived-allele counts is `np.bincount(dac)`, zero-padded to n+1 entries
    (data_processing_v2.py:59-63);
  * `ot.emd2_1d(x, y, a, b)` (POT 0.7 default metric: squared Euclidean) is the exact 1-D optimal-transport
    cost, obtained by merging the two sorted samples' cumulative weights; for equal sizes and uniform
    weights it is the mean squared difference of the sorted samples.
The reference's own files hold one known answer for this path (sliced.py:83-84: distance of a sample to
itself is 0.0); everything else is pinned here by brute-force linear programmes on small cases
(tests/test_evaluation.py).  Plot files are not produced.
"""
from __future__ import annotations

import numpy as np


def site_frequency_spectra(sites: np.ndarray) -> np.ndarray:
    """sites [N,1,n,S] (or [N,n,S]) with entries near {0,1}: rows individuals, columns sites.
    Negative entries are set to 0 and entries rounded (data_processing_v2.py:53-55); returns the
    [N, n+1] integer spectra: entry k = number of columns whose derived-allele count is k."""
    x = np.array(sites, dtype=np.float64, copy=True)
    if x.ndim == 4:
        x = x[:, 0]
    x[x < 0] = 0
    x = np.rint(x)
    n = x.shape[1]
    counts = x.sum(axis=1).astype(np.int64)            # [N, S] derived-allele count per column
    out = np.zeros((x.shape[0], n + 1), dtype=np.int64)
    for i in range(x.shape[0]):
        b = np.bincount(counts[i], minlength=n + 1)
        out[i] = b[:n + 1] if b.shape[0] <= n + 1 else b[:n + 1]
    return out


def pca2(rows: np.ndarray) -> np.ndarray:
    """First two principal-component scores of the rows (sklearn.decomposition.PCA(n_components=2)
    .fit_transform: centre, SVD, component signs fixed by sklearn's svd_flip convention)."""
    x = np.asarray(rows, dtype=np.float64)
    xc = x - x.mean(axis=0, keepdims=True)
    u, s, vt = np.linalg.svd(xc, full_matrices=False)
    # svd_flip(u_based_decision=True): the largest-magnitude entry of every column of u is positive
    idx = np.argmax(np.abs(u), axis=0)
    signs = np.sign(u[idx, np.arange(u.shape[1])])
    signs[signs == 0] = 1.0
    u = u * signs
    return u[:, :2] * s[:2]


def get_random_projections(n_projections: int, d: int, seed=None) -> np.ndarray:
    """n_projections unit vectors, normal draws normalised (sliced.py:14-46)."""
    rs = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
    p = rs.normal(0.0, 1.0, [n_projections, d])
    return p / np.linalg.norm(p, ord=2, axis=1, keepdims=True)


def emd2_1d(x: np.ndarray, y: np.ndarray, a: np.ndarray | None = None, b: np.ndarray | None = None) -> float:
    """Exact optimal-transport cost between two weighted 1-D samples under the squared Euclidean
    ground metric (what the reference obtains from ot.emd2_1d, sliced.py:118)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    n, m = x.shape[0], y.shape[0]
    a = np.full(n, 1.0 / n) if a is None else np.asarray(a, dtype=np.float64)
    b = np.full(m, 1.0 / m) if b is None else np.asarray(b, dtype=np.float64)
    ix, iy = np.argsort(x, kind="stable"), np.argsort(y, kind="stable")
    xs, ys, aw, bw = x[ix], y[iy], a[ix], b[iy]
    if n == m and np.all(aw == aw[0]) and np.all(bw == bw[0]):
        return float(np.mean((xs - ys) ** 2) * aw[0] * n)
    # monotone coupling: walk both cumulative distributions
    ca, cb = np.cumsum(aw), np.cumsum(bw)
    cuts = np.unique(np.concatenate([ca, cb]))
    cuts = cuts[cuts <= min(ca[-1], cb[-1]) + 1e-15]
    prev = np.concatenate([[0.0], cuts[:-1]])
    mass = cuts - prev
    mid = 0.5 * (cuts + prev)
    i = np.minimum(np.searchsorted(ca, mid, side="left"), n - 1)
    j = np.minimum(np.searchsorted(cb, mid, side="left"), m - 1)
    return float(np.sum(mass * (xs[i] - ys[j]) ** 2))


def sliced_wasserstein_distance(X_s, X_t, a=None, b=None, n_projections: int = 50, seed=None, log: bool = False):
    """Monte-Carlo sliced 2-Wasserstein distance (sliced.py:49-130): mean over random unit directions of
    the 1-D transport cost of the projected samples, then the square root."""
    X_s, X_t = np.asanyarray(X_s, dtype=np.float64), np.asanyarray(X_t, dtype=np.float64)
    if X_s.shape[1] != X_t.shape[1]:
        raise ValueError("X_s and X_t must have the same number of dimensions {} and {} respectively given".format(
            X_s.shape[1], X_t.shape[1]))
    proj = get_random_projections(n_projections, X_s.shape[1], seed)
    ps, pt = proj @ X_s.T, proj @ X_t.T
    emds = np.array([emd2_1d(u, v, a, b) for u, v in zip(ps, pt)])
    res = float((emds.sum() / n_projections) ** 0.5)
    if log:
        return res, {"projections": proj, "projected_emds": emds}
    return res


def plot_sfs_pca(fake_sites, curr_epoch, in_dat, data_size, odir=None, device=None, seed=None) -> float:
    """2DSWD between the input and the generated alignments (data_processing_v2.py:50-121; same argument
    order).  The reference calls the sliced distance unseeded (:119); pass `seed` for a repeatable value."""
    sf = site_frequency_spectra(np.asarray(fake_sites))
    sr = site_frequency_spectra(np.asarray(in_dat))
    pcs = pca2(np.concatenate([sr, sf], axis=0))
    return sliced_wasserstein_distance(pcs[:sr.shape[0]], pcs[sr.shape[0]:sr.shape[0] + sf.shape[0]], seed=seed)
