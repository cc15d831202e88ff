"""Parity evidence at BASELINE.json's full size (ML-25M shape: 162 541 x 59 047, 25.0 M ratings, one cluster).

The brute-force oracle cannot run a 162 541-user neighbourhood (U_c - 1 multiply-adds per term), so this test uses
size-independent properties plus spot checks against the DEFINITION evaluated in fp64 with scipy.sparse
(score(u,i) = pvpi + sum_j ln(sum_{v != u} c_vi c_vj), the sum over v done as exact sparse dot products):
  * every user gets min(N, I_c - n_u) rows, contiguous, non-increasing, distinct, never an item the user rated;
  * for sampled users the returned scores equal the fp64 definition within 1e-5 relative, and sampled items that
    were NOT returned score no better than the last returned one;
  * item-item similarity rows equal the fp64 cosine definition, the K-th value is the true K-th best.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from util import RTOL, pkg, synth

pytestmark = pytest.mark.gpu
LAM, TOPN = 0.1, 50


@pytest.fixture(scope="module")
def data():
    S = synth()
    u, i, s, facts = S.generate("ml25m", device="cuda:0")
    torch.cuda.synchronize()
    hu, hi, hs = u.cpu().numpy(), i.cpu().numpy(), s.cpu().numpy().astype(np.float64)
    uu, ui = np.unique(hu, return_inverse=True)
    iu, ii = np.unique(hi, return_inverse=True)
    R = sp.csr_matrix((hs, (ui, ii)), shape=(len(uu), len(iu)))
    return dict(dev=(u, i, s), facts=facts, uu=uu, iu=iu, R=R)


def test_rm2_full_size(data):
    P = pkg()
    ctx = P.Context(0)
    conf = P.Configuration()
    conf.set("lambda", repr(LAM))
    conf.setInt("numberOfItems", data["facts"]["n_items"])
    conf.setInt("numberOfClusters", 1)
    conf.setInt("numberOfRecommendations", TOPN)
    rec = P.RM2Job(conf, ctx).run(P.Ratings(ctx, *data["dev"]))
    rows, sums, st = rec.rows(), rec.sums(), rec.stats
    R, uu, iu = data["R"], data["uu"], data["iu"]
    U, I = R.shape
    n_u = np.diff(R.indptr)
    # ---- statistics: exact
    su = np.asarray(R.sum(1)).ravel()
    np.testing.assert_array_equal(sums["user_id"], uu)
    np.testing.assert_array_equal(sums["user_sum"], su)
    T = np.floor(su).sum()                                   # quirk Q1 on half-star data
    assert sums["total_sum"] == T and T < su.sum()
    p = np.asarray(R.sum(0)).ravel() / T
    np.testing.assert_allclose(sums["item_coll"], p, rtol=1e-14)
    assert st["log_terms"] == int((n_u.astype(np.int64) * (I - n_u)).sum())
    # ---- structure of the lists
    user, item, score = rows["user"], rows["item"], rows["score"]
    assert len(user) == int(np.minimum(TOPN, I - n_u).sum()) == st["recs"]
    starts = np.flatnonzero(np.r_[True, user[1:] != user[:-1]])
    assert len(starts) == U and len(np.unique(user[starts])) == U          # contiguous, every user once
    same = user[1:] == user[:-1]
    assert np.all(score[1:][same] <= score[:-1][same])                       # best first
    assert np.isfinite(score).all()
    key = user.astype(np.int64) * (int(iu.max()) + 1) + item
    assert len(np.unique(key)) == len(key)                                   # no item twice in a list
    rated = uu[np.repeat(np.arange(U), n_u)].astype(np.int64) * (int(iu.max()) + 1) + iu[R.indices]
    assert not np.isin(key, rated).any()                                     # never an item the user already rated
    # ---- spot checks against the fp64 definition
    X = sp.diags(1.0 / su) @ R
    Xc = X.tocsc()
    b = np.asarray(X.sum(0)).ravel()
    w2, w1 = (1 - LAM) ** 2, LAM * (1 - LAM)
    order = np.argsort(-n_u, kind="stable")
    rng = np.random.default_rng(1)
    picks = list(order[[0, 3]]) + list(order[[U // 2, U // 2 + 7]]) + list(order[[-1, -5]]) + list(rng.choice(U, 4, replace=False))
    pos_of_item = {int(v): k for k, v in enumerate(iu)}
    start_of_user = dict(zip(user[starts].tolist(), starts.tolist()))     # rows are in slot order, not in user-id order
    worst = 0.0
    for ux in picks:
        J = X.indices[X.indptr[ux]:X.indptr[ux + 1]]          # indices and data of the SAME matrix (X may order a row differently from R)
        x = X.data[X.indptr[ux]:X.indptr[ux + 1]]
        n = len(J)
        e = (1 - LAM) * (b[J] - x) + LAM * (U - 1) * p[J]
        pvpi = (n - 1) * np.log(data["facts"]["n_items"]) - n * np.log(U)

        def exact(ix):
            g = np.asarray((X.T @ Xc[:, ix]).todense()).ravel()[J]       # sum_v x_vi x_vj, exact sparse dot products
            return pvpi + np.log(w2 * g + w1 * p[J] * b[ix] + LAM * p[ix] * e).sum()

        a = start_of_user[int(uu[ux])]
        lst_items, lst_scores = item[a:a + TOPN], score[a:a + TOPN].astype(np.float64)
        for k in (0, 1, 2, TOPN // 2, TOPN - 1):
            ref = exact(pos_of_item[int(lst_items[k])])
            worst = max(worst, abs(lst_scores[k] - ref) / abs(ref))
        israted = np.zeros(I, bool)
        israted[J] = True
        inlist = np.zeros(I, bool)
        inlist[[pos_of_item[int(v)] for v in lst_items]] = True
        pop_rank = np.argsort(-np.diff(Xc.indptr), kind="stable")
        others = [ix for ix in list(pop_rank[:40]) + list(rng.choice(I, 20, replace=False)) if not israted[ix] and not inlist[ix]]
        for ix in others[:30]:                                               # popular non-members are the dangerous ones
            assert exact(ix) <= lst_scores[-1] + RTOL * abs(lst_scores[-1])
    assert worst <= RTOL, worst
    print("full-size worst relative error vs fp64 definition: %.2e" % worst)
    rec.close()
    ctx.close()


def test_itemsim_full_size(data):
    P = pkg()
    ctx = P.Context(0)
    res = P.RowSimilarityJob(ctx).run(P.Ratings(ctx, *data["dev"]), maxSimilaritiesPerRow=100)
    rows = res.rows()
    R, iu = data["R"], data["iu"]
    U, I = R.shape
    n_u = np.diff(R.indptr).astype(np.int64)
    assert res.stats["unordered_pairs"] == int((n_u * (n_u - 1) // 2).sum())
    Rc = R.tocsc()
    norms = np.sqrt(np.asarray(Rc.multiply(Rc).sum(0)).ravel())
    Xn = (Rc @ sp.diags(1.0 / norms)).tocsc()
    it, ot, sm = rows["item"], rows["other"], rows["sim"].astype(np.float64)
    starts = np.flatnonzero(np.r_[True, it[1:] != it[:-1]])
    assert len(np.unique(it[starts])) == len(starts)
    start_of_item = dict(zip(it[starts].tolist(), starts.tolist()))
    cnt = np.diff(Xn.indptr)
    pop = np.argsort(-cnt, kind="stable")
    for ix in [pop[0], pop[5], pop[I // 2], pop[-3], pop[I // 4]]:
        full = np.asarray((Xn.T @ Xn[:, ix]).todense()).ravel()
        full[ix] = 0.0
        k = min(100, int((full > 0).sum()))
        a = start_of_item[int(iu[ix])]
        got_o, got_s = ot[a:a + k], sm[a:a + k]
        assert (a + k == len(it)) or it[a + k] != iu[ix] or k == 100
        best = np.sort(full)[::-1][:k]
        np.testing.assert_allclose(got_s, best, rtol=2e-6)
        lookup = full[[int(np.searchsorted(iu, o)) for o in got_o]]
        np.testing.assert_allclose(got_s, lookup, rtol=2e-6)
    res.close()
    ctx.close()
