"""ML-1M-shaped data (6040 x 3706, one cluster): every user's list against the fp64 DEFINITION with a dense M
(numpy), all candidates -- the brute-force oracle would need 2e13 multiply-adds here."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import RTOL, pkg, synth

pytestmark = pytest.mark.gpu


def definition_lists(u, i, s, lam, n_items_conf, top_n):
    uu, ui = np.unique(u, return_inverse=True)
    iu, ii = np.unique(i, return_inverse=True)
    U, I = len(uu), len(iu)
    R = sp.csr_matrix((s.astype(np.float64), (ui, ii)), shape=(U, I))
    su = np.asarray(R.sum(1)).ravel()
    T = np.floor(su).sum()
    p = np.asarray(R.sum(0)).ravel() / T
    X = sp.diags(1.0 / su) @ R
    G = (X.T @ X).toarray()
    b = np.asarray(X.sum(0)).ravel()
    M = (1 - lam) ** 2 * G + lam * (1 - lam) * np.outer(p, b)          # M[j][i]
    out = {}
    for ux in range(U):
        J = X.indices[X.indptr[ux]:X.indptr[ux + 1]]
        x = X.data[X.indptr[ux]:X.indptr[ux + 1]]
        n = len(J)
        e = (1 - lam) * (b[J] - x) + lam * (U - 1) * p[J]
        sc = (n - 1) * np.log(n_items_conf) - n * np.log(U) + np.log(M[J, :] + np.outer(e, lam * p)).sum(0)
        sc[J] = -np.inf
        out[int(uu[ux])] = (iu, sc, n)
    return out


@pytest.mark.parametrize("shape", ["ml1m"])
def test_all_users_against_definition(shape):
    P, S = pkg(), synth()
    u, i, s, facts = S.generate(shape)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    lam, N = 0.1, 50
    conf = P.Configuration()
    conf.set("lambda", repr(lam))
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", 1)
    conf.setInt("numberOfRecommendations", N)
    ctx = P.Context(0)
    rows = P.RM2Job(conf, ctx).run((u, i, s)).rows()
    ref = definition_lists(u, i, s, lam, facts["n_items"], N)
    starts = np.flatnonzero(np.r_[True, rows["user"][1:] != rows["user"][:-1]])
    worst, worst_n = 0.0, 0
    for a in starts:
        uid = int(rows["user"][a])
        iu, sc, n = ref[uid]
        k = min(N, int(np.isfinite(sc).sum()))
        gi, gs = rows["item"][a:a + k], rows["score"][a:a + k].astype(np.float64)
        want = sc[np.searchsorted(iu, gi)]
        err = np.abs(gs - want) / np.abs(want)
        if err.max() > worst:
            worst, worst_n = float(err.max()), n
        best = np.sort(sc)[::-1][:k]
        assert np.all(np.abs(gs - best) <= RTOL * np.abs(best)), uid
    print("worst relative error vs fp64 definition %.2e (user with %d ratings)" % (worst, worst_n))
    assert worst <= RTOL
    ctx.close()
