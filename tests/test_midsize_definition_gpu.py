"""ML-1M-shaped data (6040 x 3706, one cluster): every user's list against the fp64 DEFINITION with a dense M
(numpy), all candidates -- the brute-force oracle would need 2e13 multiply-adds here."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from util import ATOL, RTOL, assert_topn_matches, pkg, synth

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


def test_ml1m_50_clusters_top50_against_the_oracle():
    """BASELINE.json configs[1] (C1: MovieLens-1M shape, top-50) in the reference's own regime -- 50 clusters
    (T/rmrecommender/TestRMRecommenderJob.java:49) -- against the BRUTE-FORCE oracle (oracle/rm2_oracle.c: the reference's dense
    cache and triple loop, AbstractRM2Reducer.java:185-190, 332-356), not against a re-arranged formula: every row of every user."""
    P, S = pkg(), synth()
    u, i, s, facts = S.generate("ml1m")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    lam, N, K = 0.1, 50, 50
    mu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
    mc = S.hash_clustering(mu, K)
    conf = P.Configuration()
    conf.set("lambda", repr(lam))
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", N)
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run((u, i, s), clustering=(mu, mc))
    rows = rec.rows()
    ref = oracle.rm2(u, i, s, lam=lam, number_of_items=facts["n_items"], number_of_recommendations=1 << 30, number_of_clusters=K,
                     map_user=mu, map_cluster=mc, n_threads=16)
    worst = assert_topn_matches(rows, ref, N)
    # how many comparisons needed the absolute term of tests/util.py (ATOL) on top of north_star's relative 1e-5
    want = {(int(a), int(b)): float(c) for a, b, c in zip(ref["rec_user"], ref["rec_item"], ref["rec_score"])}
    w = np.array([want[(int(a), int(b))] for a, b in zip(rows["user"], rows["item"])])
    g = rows["score"].astype(np.float64)
    fin = np.isfinite(w)
    needed_atol = int(np.sum(np.abs(g[fin] - w[fin]) > RTOL * np.abs(w[fin])))
    print("ML-1M shape, 50 clusters, top-50 vs brute-force oracle: %d rows, worst relative error %.2e, %d comparisons beyond the pure 1e-5 "
          "relative bound (absolute slack %.0e)" % (len(g), worst, needed_atol, ATOL))
    assert needed_atol == 0
    rec.close()
    ctx.close()


@pytest.fixture(scope="module")
def sampled_ml25m_cluster():
    S = synth()
    rng = np.random.Generator(np.random.PCG64(123))
    users = np.sort(rng.choice(S.SHAPES["ml25m"][0], size=400, replace=False)) + 1
    u, i, s, _ = S.generate("ml25m", users=users)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    ref = oracle.rm2_gram(u, i, s, lam=0.1, number_of_items=59047, number_of_recommendations=1 << 30, number_of_clusters=1, n_threads=16)
    return u, i, s, ref


@pytest.mark.parametrize("top_n", [300, 1000])
def test_long_lists_one_cluster_of_sampled_ml25m_users(sampled_ml25m_cluster, top_n):
    """The reference's default list length, numberOfRecommendations = 1000 (RMRecommenderDriver.java:95), on a neighbourhood with more
    items than the top-N kernel's 4096-column sample: 400 users sampled from the ML-25M shape (about 20 000 candidate items, one
    cluster).  Lists this long take the plain full pass and the one-pass top-N (k_topn_fast: the lower bound from the 4 N most
    popular columns, one stream over the row).  Checked, every row, against the Gram-restructured CPU scorer (fp64,
    oracle/rm2_oracle.c:rm2o_run_gram -- the brute-force loop would need 1e12 multiply-adds here; the two CPU scorers are held
    equal in tests/test_oracle_golden.py)."""
    P = pkg()
    u, i, s, ref = sampled_ml25m_cluster
    n_items = int(len(np.unique(i)))
    assert n_items > 4096 + top_n
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", 59047)
    conf.setInt("numberOfClusters", 1)
    conf.setInt("numberOfRecommendations", top_n)
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run((u, i, s))
    rows, st = rec.rows(), rec.stats
    worst = assert_topn_matches(rows, ref, top_n)
    print("top-%d over %d items, %d users: worst relative error %.2e; %d users through the radix-select fallback"
          % (top_n, n_items, len(np.unique(u)), worst, st["topn_select_users"]))
    assert st["blocks_total"] == 0 and st["topn_select_users"] == 0
    rec.close()
    ctx.close()


@pytest.mark.parametrize("top_n", [300, 1000])
def test_long_lists_through_the_branch_and_bound(sampled_ml25m_cluster, top_n, monkeypatch):
    """Round 4: the same long lists through the PRUNED flow (a seed of 5 N columns -- 1536 / 5120 of the 22 083 -- scored exactly,
    tau_u = its N-th best from k_topn_long's seed mode, block bounds, survivors merged by k_topn_long's merge mode); forced onto this
    400-user neighbourhood (production prunes clusters of >= 600 users), every row against the fp64 Gram scorer of the oracle."""
    monkeypatch.setenv("FY_PRUNE_MIN_USERS", "0")
    monkeypatch.setenv("FY_MAX_SURV_FRAC", "1.5")
    P = pkg()
    u, i, s, ref = sampled_ml25m_cluster
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", 59047)
    conf.setInt("numberOfClusters", 1)
    conf.setInt("numberOfRecommendations", top_n)
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run((u, i, s))
    rows, st = rec.rows(), rec.stats
    # (on a 400-user neighbourhood the bound excludes next to nothing -- all blocks survive for N = 1000: what is tested is the seed
    # threshold and the merge of a 5 N-column seed row with MANY surviving blocks, not the pruning rate)
    assert st["blocks_total"] > 0 and st["prune_fallbacks"] == 0 and 0 < st["blocks_survived"] <= st["blocks_total"]
    worst = assert_topn_matches(rows, ref, top_n)
    print("top-%d, pruned: %d of %d blocks survive, worst relative error %.2e, %d users through the radix-select fallback"
          % (top_n, st["blocks_survived"], st["blocks_total"], worst, st["topn_select_users"]))
    rec.close()
    ctx.close()
