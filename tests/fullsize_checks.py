"""Size-independent parity checks at BASELINE.json's full sizes (shared by the ML-25M and Netflix-shape tests).

The brute-force oracle cannot run a 10^5-user neighbourhood (U_c - 1 multiply-adds per term), so these checks use
properties that hold at any size plus spot checks against the DEFINITION evaluated in fp64 with scipy.sparse
(score(u,i) = pvpi + sum_j ln(sum_{v != u} c_vi c_vj), AbstractRM2Reducer.java:321-371, the sum over v done as exact
sparse dot products), and -- the proof that the branch and bound never drops a list member -- an all-rows comparison
of the pruned job with the plain full pass (every log term evaluated, like the reference's loop :332-356)."""
import os

import numpy as np
import scipy.sparse as sp
import torch

from util import RTOL, pkg, synth


def load_shape(shape):
    S = synth()
    u, i, s, facts = S.generate(shape, device="cuda:0")
    torch.cuda.synchronize()
    hu, hi, hs = u.cpu().numpy(), i.cpu().numpy(), s.cpu().numpy().astype(np.float64)
    # ids are 1..n and dense in the synthetic shapes: a bincount-based unique is much cheaper than a sort of 10^8 keys
    uu = np.flatnonzero(np.bincount(hu)).astype(hu.dtype)
    iu = np.flatnonzero(np.bincount(hi)).astype(hi.dtype)
    ui = np.searchsorted(uu, hu)
    ii = np.searchsorted(iu, hi)
    R = sp.csr_matrix((hs, (ui, ii)), shape=(len(uu), len(iu)))
    return dict(dev=(u, i, s), facts=facts, uu=uu, iu=iu, R=R, shape=shape)


def run_rm2(data, top_n, lam, env=None, clusters=1):
    """One RM2 job (users hashed to `clusters` clusters; 1 = one neighbourhood) through the host mirror; `env` = tuning variables for
    this job only."""
    P = pkg()
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        ctx = P.Context(0)
        conf = P.Configuration()
        conf.set("lambda", repr(lam))
        conf.setInt("numberOfItems", data["facts"]["n_items"])
        conf.setInt("numberOfClusters", clusters)
        conf.setInt("numberOfRecommendations", top_n)
        clustering = clustering_of(data, clusters)
        rec = P.RM2Job(conf, ctx).run(P.Ratings(ctx, *data["dev"]), clustering=clustering)
        rows, sums, st = rec.rows(), rec.sums(), rec.stats
        rec.close()
        ctx.close()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return rows, sums, st


def check_rm2(data, rows, sums, st, top_n, lam, n_picks=10):
    R, uu, iu = data["R"], data["uu"], data["iu"]
    U, I = R.shape
    n_u = np.diff(R.indptr)
    # ---- statistics: exact
    su = np.asarray(R.sum(1)).ravel()
    np.testing.assert_array_equal(sums["user_id"], uu)
    np.testing.assert_array_equal(sums["user_sum"], su)
    T = np.floor(su).sum()                                   # quirk Q1 (differs from the plain total on half-star data)
    assert sums["total_sum"] == T
    if data["facts"]["half_stars"]:
        assert T < su.sum()
    p = np.asarray(R.sum(0)).ravel() / T
    np.testing.assert_allclose(sums["item_coll"], p, rtol=1e-14)
    assert st["log_terms"] == int((n_u.astype(np.int64) * (I - n_u)).sum())
    # ---- structure of the lists
    user, item, score = rows["user"], rows["item"], rows["score"]
    assert len(user) == int(np.minimum(top_n, I - n_u).sum()) == st["recs"]
    starts = np.flatnonzero(np.r_[True, user[1:] != user[:-1]])
    assert len(starts) == U and len(np.unique(user[starts])) == U          # contiguous, every user once
    same = user[1:] == user[:-1]
    assert np.all(score[1:][same] <= score[:-1][same])                       # best first
    assert np.isfinite(score).all()
    big = int(iu.max()) + 1
    key = user.astype(np.int64) * big + item
    assert len(np.unique(key)) == len(key)                                   # no item twice in a list
    rated = uu[np.repeat(np.arange(U), n_u)].astype(np.int64) * big + iu[R.indices]
    assert not np.isin(key, rated).any()                                     # never an item the user already rated
    # ---- spot checks against the fp64 definition
    X = sp.diags(1.0 / su) @ R
    Xc = X.tocsc()
    XT = X.T.tocsr()
    b = np.asarray(X.sum(0)).ravel()
    w2, w1 = (1 - lam) ** 2, lam * (1 - lam)
    order = np.argsort(-n_u, kind="stable")
    rng = np.random.default_rng(1)
    picks = list(order[[0, 3]]) + list(order[[U // 2, U // 2 + 7]]) + list(order[[-1, -5]]) + list(rng.choice(U, 4, replace=False))
    picks = picks[:n_picks]
    pos_of_item = {int(v): k for k, v in enumerate(iu)}
    start_of_user = dict(zip(user[starts].tolist(), starts.tolist()))     # rows are in slot order, not in user-id order
    pop_rank = np.argsort(-np.diff(Xc.indptr), kind="stable")
    worst = 0.0
    for ux in picks:
        J = X.indices[X.indptr[ux]:X.indptr[ux + 1]]          # indices and data of the SAME matrix
        x = X.data[X.indptr[ux]:X.indptr[ux + 1]]
        n = len(J)
        e = (1 - lam) * (b[J] - x) + lam * (U - 1) * p[J]
        pvpi = (n - 1) * np.log(data["facts"]["n_items"]) - n * np.log(U)

        def exact(ix):
            g = np.asarray((XT @ Xc[:, ix]).todense()).ravel()[J]       # sum_v x_vi x_vj, exact sparse dot products
            return pvpi + np.log(w2 * g + w1 * p[J] * b[ix] + lam * p[ix] * e).sum()

        a = start_of_user[int(uu[ux])]
        k_list = min(top_n, I - n)
        lst_items, lst_scores = item[a:a + k_list], score[a:a + k_list].astype(np.float64)
        for k in (0, 1, 2, k_list // 2, k_list - 1):
            ref = exact(pos_of_item[int(lst_items[k])])
            worst = max(worst, abs(lst_scores[k] - ref) / abs(ref))
        israted = np.zeros(I, bool)
        israted[J] = True
        inlist = np.zeros(I, bool)
        inlist[[pos_of_item[int(v)] for v in lst_items]] = True
        others = [ix for ix in list(pop_rank[:top_n + 40]) + list(rng.choice(I, 20, replace=False)) if not israted[ix] and not inlist[ix]]
        for ix in others[:24]:                                               # popular non-members are the dangerous ones
            assert exact(ix) <= lst_scores[-1] + RTOL * abs(lst_scores[-1])
    assert worst <= RTOL, worst
    return worst


def clustering_of(data, clusters):
    if clusters <= 1:
        return None
    uu = np.arange(1, data["facts"]["n_users"] + 1, dtype=np.int32)      # ids are 1..n in the synthetic shapes
    return uu, synth().hash_clustering(uu, clusters)


def all_rows_against_fp64_definition(data, rows, lam, clusters, label):
    """EVERY row of a full-size job against the fp64 definition (tests/fp64_definition.py: torch fp64 on the card, pinned to the
    reference's golden triples and to the brute-force oracle by tests/test_fp64_definition_cpu.py).  Pure relative 1e-5 -- north_star's
    criterion, no absolute term.  Prints the measured maximum and returns the report."""
    from fp64_definition import compare_with_definition, fp64_scores
    ref = fp64_scores(data["dev"], rows, lam, data["facts"]["n_items"], clustering=clustering_of(data, clusters))
    rep = compare_with_definition(rows, ref, rtol=RTOL)
    print("%s: ALL %d rows against the fp64 definition: worst relative error %.2e, %d rows over 1e-5, 99.99th percentile %.1e; worst rows (user, item, got, fp64, rel): %s"
          % (label, rep["rows"], rep["worst"], rep["n_over"], rep["p9999"], rep["worst_rows"][:3]))
    assert rep["n_over"] == 0 and rep["worst"] <= RTOL, rep
    return rep


def whole_clusters_against_the_gram_oracle(data, rows, lam, top_n, clusters, selected, label):
    """Whole clusters of a many-cluster job through oracle.rm2_gram (the CPU's fp64 Gram scorer, itself checked against the
    brute-force oracle in tests/test_oracle_golden.py): every row of those clusters, pure relative 1e-5, lists tie-tolerant."""
    import oracle
    u, i, s = (t.cpu().numpy() for t in data["dev"])
    mu, mc = clustering_of(data, clusters)
    old = os.environ.get("RM2O_GRAM_BUDGET_GB")
    os.environ["RM2O_GRAM_BUDGET_GB"] = "40"
    try:
        ref = oracle.rm2_gram(u, i, s, lam=lam, number_of_items=data["facts"]["n_items"], number_of_recommendations=top_n,
                              number_of_clusters=clusters, map_user=mu, map_cluster=mc, n_threads=min(16, os.cpu_count() or 1),
                              only_clusters=selected)
    finally:
        if old is None:
            del os.environ["RM2O_GRAM_BUDGET_GB"]
        else:
            os.environ["RM2O_GRAM_BUDGET_GB"] = old
    assert sorted(set(ref["rec_cluster"].tolist())) == sorted(selected)
    m = np.isin(rows["cluster"], np.asarray(selected))
    got = {k: rows[k][m] for k in ("user", "item", "score")}
    og = np.argsort(got["user"], kind="stable")                       # both by ascending user id, list order kept
    orf = np.argsort(ref["rec_user"], kind="stable")
    a = {k: got[k][og] for k in got}
    b = {"user": ref["rec_user"][orf], "item": ref["rec_item"][orf], "score": ref["rec_score"][orf]}
    n_diff, worst = assert_same_lists(a, b, score_rtol=RTOL, tie_rtol=RTOL)
    print("%s: clusters %s through the fp64 Gram oracle: %d rows, %d at a cut-off differ (ties), worst relative error %.2e (pure relative, no absolute slack)"
          % (label, selected, len(a["user"]), n_diff, worst))
    assert n_diff <= 1e-4 * len(a["user"])
    return len(a["user"]), n_diff, worst


def assert_same_lists(a, b, score_rtol=2e-6, tie_rtol=1e-5, score_atol=0.0):
    """All-rows comparison of two runs of the same job (pruned vs full pass): same row count per user; every (user, item)
    present in both carries the same score within `score_rtol`; a pair present in one run only must sit at that user's
    cut-off (its score within `tie_rtol` of the list's last score in the OTHER run: a tie the two passes broke differently).
    Returns (rows only in one run, worst relative score difference)."""
    ua, ia, sa = a["user"], a["item"], a["score"].astype(np.float64)
    ub, ib, sb = b["user"], b["item"], b["score"].astype(np.float64)
    assert len(ua) == len(ub)
    np.testing.assert_array_equal(ua, ub)                 # rows are grouped by user in the same (slot) order
    big = int(max(ia.max(), ib.max())) + 1
    ka = ua.astype(np.int64) * big + ia
    kb = ub.astype(np.int64) * big + ib
    oa, ob = np.argsort(ka, kind="stable"), np.argsort(kb, kind="stable")
    ka_s, kb_s = ka[oa], kb[ob]
    in_b = np.isin(ka_s, kb_s, assume_unique=True)
    in_a = np.isin(kb_s, ka_s, assume_unique=True)
    ca, cb = sa[oa][in_b], sb[ob][in_a]                  # common pairs, both in key order
    # (score_atol: for jobs whose scores nearly cancel -- pvpi > 0 against the negative log sum -- see tests/util.py ATOL)
    with np.errstate(invalid="ignore"):
        diff = np.where(ca == cb, 0.0, np.abs(ca - cb))      # (-inf in both runs is agreement, not nan)
        rel = np.where(diff == 0.0, 0.0, np.maximum(diff - score_atol, 0.0) / np.abs(cb))
    worst = float(rel.max()) if len(rel) else 0.0
    assert worst <= score_rtol, worst
    # last score of every user's list, per run
    last = np.flatnonzero(np.r_[ua[1:] != ua[:-1], True])
    first = np.r_[0, last[:-1] + 1]
    seg_of_row = np.repeat(np.arange(len(last)), last - first + 1)
    only_a = oa[~in_b]
    only_b = ob[~in_a]
    assert len(only_a) == len(only_b)
    for only, s_own, s_other in ((only_a, sa, sb), (only_b, sb, sa)):
        if len(only):
            cut = s_other[last[seg_of_row[only]]]
            with np.errstate(invalid="ignore"):
                ok = (s_own[only] == cut) | (np.abs(s_own[only] - cut) <= tie_rtol * np.abs(cut) + score_atol)
            assert np.all(ok), "a list member is missing from the other run"
    return len(only_a), worst


def check_itemsim(data, top_k=100, n_rows=5):
    P = pkg()
    ctx = P.Context(0)
    res = P.RowSimilarityJob(ctx).run(P.Ratings(ctx, *data["dev"]), maxSimilaritiesPerRow=top_k)
    rows = res.rows()
    R, iu = data["R"], data["iu"]
    U, I = R.shape
    n_u = np.diff(R.indptr).astype(np.int64)
    assert res.stats["unordered_pairs"] == int((n_u * (n_u - 1) // 2).sum())
    Rc = R.tocsc()
    norms = np.sqrt(np.asarray(Rc.multiply(Rc).sum(0)).ravel())
    Xn = (Rc @ sp.diags(1.0 / norms)).tocsc()
    XnT = Xn.T.tocsr()
    it, ot, sm = rows["item"], rows["other"], rows["sim"].astype(np.float64)
    starts = np.flatnonzero(np.r_[True, it[1:] != it[:-1]])
    assert len(np.unique(it[starts])) == len(starts)
    start_of_item = dict(zip(it[starts].tolist(), starts.tolist()))
    cnt = np.diff(Xn.indptr)
    pop = np.argsort(-cnt, kind="stable")
    for ix in [pop[0], pop[5], pop[I // 2], pop[-3], pop[I // 4]][:n_rows]:
        full = np.asarray((XnT @ Xn[:, ix]).todense()).ravel()
        full[ix] = 0.0
        k = min(top_k, int((full > 0).sum()))
        a = start_of_item[int(iu[ix])]
        got_o, got_s = ot[a:a + k], sm[a:a + k]
        assert (a + k == len(it)) or it[a + k] != iu[ix] or k == top_k
        best = np.sort(full)[::-1][:k]
        np.testing.assert_allclose(got_s, best, rtol=2e-6)
        lookup = full[[int(np.searchsorted(iu, o)) for o in got_o]]
        np.testing.assert_allclose(got_s, lookup, rtol=2e-6)
    res.close()
    ctx.close()


def compare_itemsim_builds(data, top_k=100):
    """ALL rows of the symmetric build (upper triangle by the RM2 row kernel + band sweep: what the benchmark sizes take) against
    the row-at-a-time build (fp64 accumulators in LDS, FY_ISIM_GRAM=0) on the same ratings: the same lists, similarities within
    the 2e-6 both are held to against the oracle at small sizes; positions where the item differs must be ties."""
    P = pkg()
    out = []
    for gram in ("1", "0"):
        old = os.environ.get("FY_ISIM_GRAM")
        os.environ["FY_ISIM_GRAM"] = gram
        try:
            ctx = P.Context(0)
            res = P.RowSimilarityJob(ctx).run(P.Ratings(ctx, *data["dev"]), maxSimilaritiesPerRow=top_k)
            out.append((res.rows(), res.stats))
            res.close()
            ctx.close()
        finally:
            if old is None:
                del os.environ["FY_ISIM_GRAM"]
            else:
                os.environ["FY_ISIM_GRAM"] = old
    (ra, sa), (rb, sb) = out
    assert sa["isim_candidates"] > 0 and sb["isim_candidates"] == 0
    assert len(ra["item"]) == len(rb["item"]) and np.array_equal(ra["item"], rb["item"])
    x, y = ra["sim"].astype(np.float64), rb["sim"].astype(np.float64)
    worst = float(np.max(np.abs(x - y) / y))
    assert worst <= 2e-6, worst
    differ = np.flatnonzero(ra["other"] != rb["other"])
    # a differing position is a tie: the other build holds the same value there
    assert len(differ) <= 1e-3 * len(x), len(differ)
    return len(x), len(differ), worst, sa
