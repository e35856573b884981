"""2DSWD evaluation numerics (host numpy).  The reference's only known answer for this path is the
doctest sliced_wasserstein_distance(X, X) == 0.0 (src/sliced.py:83-84); the 1-D transport cost is pinned
against brute-force linear programmes, the spectrum against a direct count, PCA against sklearn."""
import numpy as np
import pytest

import data_processing_v2 as dp


def test_distance_to_itself_is_zero():
    X = np.random.RandomState(0).normal(0.0, 1.0, (20, 5))
    assert dp.sliced_wasserstein_distance(X, X, seed=0) == 0.0


def test_projections_are_unit_vectors():
    p = dp.get_random_projections(100, 5, seed=3)
    assert np.allclose(np.sum(np.square(p), 1), 1.0)
    assert np.array_equal(p, dp.get_random_projections(100, 5, seed=np.random.RandomState(3)))


@pytest.mark.parametrize("n,m,weighted", [(6, 6, False), (5, 8, False), (7, 4, True), (1, 5, False)])
def test_emd2_1d_against_linear_programme(n, m, weighted):
    from scipy.optimize import linprog
    rs = np.random.RandomState(n * 10 + m)
    x, y = rs.normal(size=n), rs.normal(size=m) + 0.3
    a = rs.dirichlet(np.ones(n)) if weighted else np.full(n, 1.0 / n)
    b = rs.dirichlet(np.ones(m)) if weighted else np.full(m, 1.0 / m)
    C = (x[:, None] - y[None, :]) ** 2
    Aeq = np.zeros((n + m, n * m))
    for i in range(n):
        Aeq[i, i * m:(i + 1) * m] = 1
    for j in range(m):
        Aeq[n + j, j::m] = 1
    res = linprog(C.ravel(), A_eq=Aeq, b_eq=np.concatenate([a, b]), bounds=(0, None), method="highs")
    assert res.status == 0
    assert dp.emd2_1d(x, y, a, b) == pytest.approx(res.fun, rel=1e-9, abs=1e-12)


def test_site_frequency_spectra():
    rs = np.random.RandomState(5)
    sites = (rs.random_sample((7, 1, 64, 64)) < 0.3).astype(np.float32)
    sites[0, 0, :, 0] = 0          # monomorphic columns count in bin 0 / bin n
    sites[0, 0, :, 1] = 1
    noisy = sites + rs.uniform(-0.3, 0.3, sites.shape).astype(np.float32)   # rounding and the clamp of negatives
    sfs = dp.site_frequency_spectra(noisy)
    assert sfs.shape == (7, 65) and np.all(sfs.sum(1) == 64)
    for i in range(7):
        counts = sites[i, 0].sum(0).astype(int)
        for k in range(65):
            assert sfs[i, k] == int(np.sum(counts == k))
    assert sfs[0, 0] >= 1 and sfs[0, 64] >= 1


def test_pca2_matches_sklearn():
    from sklearn.decomposition import PCA
    rs = np.random.RandomState(9)
    rows = rs.poisson(3.0, size=(40, 65)).astype(float)
    want = PCA(n_components=2, svd_solver="full").fit_transform(rows)
    got = dp.pca2(rows)
    # component signs: the scikit-learn release the reference pins fixes them on the scores (largest
    # |score| of a component positive); newer releases fix them on the loadings instead
    assert np.allclose(np.abs(got), np.abs(want), atol=1e-9)
    for c in range(2):
        assert got[np.argmax(np.abs(got[:, c])), c] > 0


def test_2dswd_separates_and_is_repeatable():
    rs = np.random.RandomState(11)
    real = (rs.random_sample((60, 1, 64, 64)) < 0.2).astype(np.float32)
    same = (rs.random_sample((60, 1, 64, 64)) < 0.2).astype(np.float32)
    other = (rs.random_sample((60, 1, 64, 64)) < 0.5).astype(np.float32)
    near = dp.plot_sfs_pca(same, 1, real, 64, seed=0)
    far = dp.plot_sfs_pca(other, 1, real, 64, seed=0)
    assert near == dp.plot_sfs_pca(same, 1, real, 64, seed=0)
    assert far > 5 * near > 0
