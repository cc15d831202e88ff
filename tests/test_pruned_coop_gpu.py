"""The branch-and-bound path and the cooperative multi-rank path on data small enough for the oracle.

Both are switched on by size in production (clusters of >= 8192 items); the tuning knobs FY_PRUNE_MIN_ITEMS /
FY_M24_MIN_ITEMS / FY_SEED_CHUNKS force them onto MovieLens-100K-shaped data here, where the brute-force oracle decides.

The cooperative path (fy_collectives, include/filmyou.h) is rehearsed on ONE GPU by `world` threads of this process,
each with its own fy context, meeting in parallel.ThreadCollectives (a barrier + torch copies): the same library code
and the same callback interface as the RCCL run, another transport."""
import functools
import threading

import numpy as np
import pytest

import oracle
from util import assert_topn_matches, pkg, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced(monkeypatch):
    monkeypatch.setenv("FY_PRUNE_MIN_ITEMS", "256")
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    monkeypatch.setenv("FY_SEED_CHUNKS", "1")


@functools.lru_cache(maxsize=None)
def oracle_case(shape, K, lam):
    S = synth()
    u, i, s, facts = S.generate(shape)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    clustering = (uu, S.hash_clustering(uu, K))
    ref = oracle.rm2(u, i, s, lam=float(lam), number_of_items=facts["n_items"], number_of_recommendations=1 << 30,
                     number_of_clusters=K, map_user=clustering[0], map_cluster=clustering[1], n_threads=8)
    return (u, i, s), clustering, facts, ref


def make(shape, K, lam="0.1", top_n=20):
    P = pkg()
    (u, i, s), clustering, facts, ref = oracle_case(shape, K, lam)
    conf = P.Configuration()
    conf.set("lambda", lam)
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", top_n)
    return (u, i, s), clustering, conf, ref


def same_rows(a, b, tol=2e-6):
    ka = sorted(zip(a["user"].tolist(), a["item"].tolist()))
    kb = sorted(zip(b["user"].tolist(), b["item"].tolist()))
    assert ka == kb
    da = dict(zip(zip(a["user"].tolist(), a["item"].tolist()), a["score"].tolist()))
    db = dict(zip(zip(b["user"].tolist(), b["item"].tolist()), b["score"].tolist()))
    assert max(abs(da[k] - db[k]) / max(1e-3, abs(db[k])) for k in da) < tol


@pytest.mark.parametrize("shape,K,lam,top_n", [("ml100k", 1, "0.1", 20), ("ml100k", 3, "0.5", 100), ("tiny", 1, "0.1", 10),
                                              ("ml100k", 1, "0.0", 30)])
def test_pruned_path_vs_oracle(forced, shape, K, lam, top_n):
    P = pkg()
    data, clustering, conf, ref = make(shape, K, lam, top_n)
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    assert rec.stats["blocks_total"] > 0, "the branch and bound did not run"
    assert rec.stats["blocks_survived"] < rec.stats["blocks_total"]
    assert_topn_matches(rec.rows(), ref, top_n)
    ctx.close()


@pytest.mark.parametrize("select", [0, 1])
def test_cooperative_path_world_1(forced, monkeypatch, select):
    """identity collectives: the cooperative kernels alone (range offsets, partial bounds, packed survivors)"""
    P = pkg()
    data, clustering, conf, ref = make("ml100k", 1, "0.1", 20)
    ctx = P.Context(0)
    plain = P.RM2Job(conf, ctx).run(data, clustering=clustering).rows()
    monkeypatch.setenv("FY_COOP_FORCE", "1")
    monkeypatch.setenv("FY_TOPN_FORCE_SELECT", str(select))
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    assert rec.stats["blocks_total"] > 0
    assert_topn_matches(rec.rows(), ref, 20)
    same_rows(rec.rows(), plain)
    ctx.close()


def run_threads(world, data, clustering, conf):
    P = pkg()
    par = __import__("importlib").import_module("filmyou-core_amd.parallel")
    group = par.ThreadGroup(world)
    out, err, comms = [None] * world, [None] * world, [None] * world

    def body(rank):
        try:
            ctx = P.Context(0)
            comms[rank] = par.ThreadCollectives(group, rank, 0)
            rec = P.RM2Job(conf, ctx).run(data, clustering=clustering, rank=rank, world=world, collectives=comms[rank])
            out[rank] = (rec.rows(), dict(rec.stats))
            rec.close()
            ctx.close()
        except BaseException as e:      # a dead rank must not leave the others at the barrier
            err[rank] = e
            group.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert all(o is not None for o in out), err
    return out, comms


@pytest.mark.parametrize("world,K,select", [(2, 1, 0), (3, 1, 0), (4, 1, 1), (8, 1, 0), (2, 2, 0)])
def test_cooperative_ranks_equal_single(forced, monkeypatch, world, K, select):
    P = pkg()
    data, clustering, conf, ref = make("ml100k", K, "0.1", 20)
    ctx = P.Context(0)
    single = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    single_rows, single_stats = single.rows(), dict(single.stats)
    ctx.close()
    monkeypatch.setenv("FY_TOPN_FORCE_SELECT", str(select))
    out, comms = run_threads(world, data, clustering, conf)
    rows = {k: np.concatenate([o[0][k] for o in out]) for k in ("user", "item", "score", "cluster")}
    owners = [set(o[0]["user"].tolist()) for o in out]
    assert sum(len(o) for o in owners) == len(set().union(*owners)) == len(np.unique(single_rows["user"]))
    assert_topn_matches(rows, ref, 20)
    same_rows(rows, single_rows)
    if K == 1:
        # one all-gather for the statistics, then per cooperative cluster: seed + bounds (+ survivors) reduce-scatters
        assert all(c.calls["reduce_scatter_f32"] >= 2 for c in comms), [c.calls for c in comms]
        # every rank built only its share of the matrix rows
        assert sum(o[1]["users_scored"] for o in out) == single_stats["users_scored"]
    assert all(c.calls["all_gather"] >= 1 for c in comms)


def test_collectives_installed_but_clusters_stay_local(forced):
    """K = 12 clusters on 2 ranks: no cluster spans... at most the boundary cluster spans both ranks; every other
    cluster takes the replicated path while the statistics still travel through the callbacks"""
    data, clustering, conf, ref = make("ml100k", 12, "0.1", 20)
    out, comms = run_threads(2, data, clustering, conf)
    rows = {k: np.concatenate([o[0][k] for o in out]) for k in ("user", "item", "score", "cluster")}
    assert_topn_matches(rows, ref, 20)


def test_failing_collective_fails_the_job(forced):
    P = pkg()
    data, clustering, conf, _ = make("tiny", 1)

    class Broken:
        def all_gather(self, *a):
            raise OSError("link down")

        def reduce_scatter_f32(self, *a):
            raise OSError("link down")

    ctx = P.Context(0)
    with pytest.raises(RuntimeError, match="RM2 failed!: collective"):
        P.RM2Job(conf, ctx).run(data, clustering=clustering, rank=0, world=2, collectives=Broken())
    ctx.close()


def test_threshold_that_does_not_bite_falls_back_to_the_full_pass(forced, monkeypatch):
    """lambda = 0 is legal (the reference accepts it): a user who rated an item nobody else of the cluster rated has only
    -inf scores, tau = -inf keeps every block.  Such batches are redone with the plain full pass instead of a survivor
    pass over (almost) everything; FY_MAX_SURV_FRAC=0 forces the fallback for every batch."""
    P = pkg()
    data, clustering, conf, ref = make("ml100k", 3, "0.0", 30)
    monkeypatch.setenv("FY_MAX_SURV_FRAC", "0")
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    assert rec.stats["prune_fallbacks"] == 3 and rec.stats["blocks_total"] > 0
    assert_topn_matches(rec.rows(), ref, 30)
    ctx.close()


def test_margin_with_heavy_users_and_positive_pvpi(forced, monkeypatch):
    """The rounding of the bound scales with the sum of |log| terms, not with the net |UB|: with numberOfItems >> U_c the
    positive pvpi cancels most of the log sum (every multi-cluster job).  Heavy users, 8 clusters, numberOfItems forced
    large, pruning on vs off: identical lists.  (With 15 blocks per row many survive: the fallback to the full pass is
    switched off so that the survivor pass itself is what is compared.)"""
    import os
    monkeypatch.setenv("FY_MAX_SURV_FRAC", "1e9")
    # (the refinement pass off: this test compares the survivor pass with the full pass on the SAME arithmetic; with the pass on, a row whose
    # |score| sits at the pass's threshold may be re-scored in one of the two runs only -- their unrefined scores differ by 4e-7 -- and
    # then differs by the forced format's whole error, which on these 750-user clusters of heavy raters is ~1e-4 absolute)
    monkeypatch.setenv("FY_REFINE", "0")
    P = pkg()
    S = synth()
    u, i, s, facts = S.generate("ml1m")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    K = 8
    clustering = (uu, S.hash_clustering(uu, K))
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", 5_000_000)        # pvpi = (n - 1) ln(5e6) - n ln(U_c) > 0 and large
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 50)
    ctx = P.Context(0)
    pruned = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering)
    assert pruned.stats["blocks_total"] > 0 and pruned.stats["prune_fallbacks"] == 0
    rp = pruned.rows()
    os.environ["FY_PRUNE"] = "0"
    try:
        full = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering)
    finally:
        del os.environ["FY_PRUNE"]
    assert full.stats["blocks_total"] == 0
    from fullsize_checks import assert_same_lists
    from util import ATOL
    n_diff, worst = assert_same_lists(rp, full.rows(), score_atol=ATOL)    # two M builds (fp32 atomics in any order), scores that nearly cancel
    assert n_diff <= 4, n_diff
    ctx.close()


def test_failed_multi_cluster_job_leaves_the_context_usable(forced):
    """an error thrown while several clusters are in flight on the lane streams must drain those lanes before
    the job's buffers go back to the caching allocator.  Every k-th HBM request of a 3-cluster pruned job is made to fail
    (fy_context_inject_alloc_failure); after each failure a clean job on the SAME context must still match the oracle."""
    P = pkg()
    data, clustering, conf, ref = make("ml100k", 3, "0.5", 100)
    ctx = P.Context(0)
    ratings = P.Ratings(ctx, *data)
    failures = 0
    nth = 1
    while True:
        ctx.inject_alloc_failure(nth)
        try:
            rec = P.RM2Job(conf, ctx).run(ratings, clustering=clustering)
        except RuntimeError as e:                      # the host mirror's "RM2 failed!: ..." (RM2Job.java:145-147)
            assert "injected fault" in str(e)
            failures += 1
            ctx.inject_alloc_failure(0)
            clean = P.RM2Job(conf, ctx).run(ratings, clustering=clustering)
            assert_topn_matches(clean.rows(), ref, 100)
            clean.close()
            nth += 7
            continue
        ctx.inject_alloc_failure(0)
        assert_topn_matches(rec.rows(), ref, 100)     # nth beyond the job's last request: it ran to completion
        break
    assert failures >= 10, failures
    ctx.close()


# ---------------------------------------------------------------- column-panel mode (many clusters)
@pytest.mark.parametrize("shape,K,lam,top_n,panel_cols,max_ch", [("ml100k", 1, "0.1", 20, 256, 0), ("ml100k", 3, "0.5", 100, 512, 0),
                                                                ("ml100k", 2, "0.1", 50, 1024, 0), ("tiny", 1, "0.1", 10, 256, 0),
                                                                ("ml100k", 1, "0.0", 30, 256, 0), ("ml100k", 1, "0.1", 20, 256, 256),
                                                                ("ml100k", 3, "0.5", 100, 256, 512), ("ml100k", 2, "0.1", 50, 512, 256)])
def test_panel_mode_vs_oracle(forced, monkeypatch, shape, K, lam, top_n, panel_cols, max_ch):
    """Panel mode stores only the first FY_PANEL_COLS columns of the co-rating rows plus 64-column block maxima; surviving
    blocks behind the panel ("strays") are scored from the sparse data.  Forced onto small data (in production: >= 4 pruned
    clusters per rank) with a panel of one or two blocks, so that most survivors are strays; the oracle decides.
    max_ch > 0: rows cut into chunks of that many columns, so that the rows behind the panel are "tail rows" -- walked over
    their first chunk(s) only, their bounds behind that from the block-compressed CSR (k_tail_blocks)."""
    monkeypatch.setenv("FY_PANEL_MIN_CLUSTERS", "1")
    monkeypatch.setenv("FY_PANEL_COLS", str(panel_cols))
    if max_ch:
        monkeypatch.setenv("FY_COOC_MAX_CH", str(max_ch))
    P = pkg()
    data, clustering, conf, ref = make(shape, K, lam, top_n)
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    st = rec.stats
    assert st["panel_clusters"] == K and st["blocks_total"] > 0 and st["prune_fallbacks"] == 0
    if shape == "ml100k" and panel_cols == 256:
        assert st["stray_blocks"] > 0, "no survivor behind the panel: the stray kernel did not run"
    assert st["stray_blocks"] <= st["blocks_survived"]
    assert_topn_matches(rec.rows(), ref, top_n)
    ctx.close()


@pytest.mark.parametrize("extra", [{}, {"FY_PANEL_SYM": "0"}, {"FY_PANEL_GROUP_MB": "24"}, {"FY_PANEL_TWO_PHASE": "0"}])
def test_panel_mode_many_clusters_equals_full_pass(monkeypatch, extra):
    """The production switch: ML-1M-shaped data in 12 clusters with the production thresholds scaled down (clusters of >= 1024
    items are pruned), panel mode chosen by the cluster count alone; all rows against the plain full pass (FY_PRUNE=0).
    Default = the symmetric panel mode (head rows walked behind the diagonal over the panel's chunks only, their bounds over the tail
    columns from the stored panel's column maxima); FY_PANEL_SYM=0 = head rows over all their chunks; a group budget of 24 MB takes the
    clusters in several groups (each through all phases); FY_PANEL_TWO_PHASE=0 = every cluster start to end on its lane (round 2)."""
    import os
    for k, v in extra.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("FY_PRUNE_MIN_ITEMS", "1024")
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    monkeypatch.setenv("FY_PANEL_COLS", "1024")
    monkeypatch.setenv("FY_COOC_MAX_CH", "1024")       # tail rows: everything behind column 1024
    P = pkg()
    S = synth()
    u, i, s, facts = S.generate("ml1m")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    K = 12
    clustering = (uu, S.hash_clustering(uu, K))
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 50)
    ctx = P.Context(0)
    panel = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering)
    assert panel.stats["panel_clusters"] == K and panel.stats["blocks_total"] > 0
    assert panel.stats["bound_repairs"] > 0, "the second bound (without the user's own co-ratings) dropped nothing"
    rp = panel.rows()
    os.environ["FY_PRUNE"] = "0"
    try:
        full = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering)
    finally:
        del os.environ["FY_PRUNE"]
    assert full.stats["panel_clusters"] == 0 and full.stats["blocks_total"] == 0
    from fullsize_checks import assert_same_lists
    from util import ATOL
    n_diff, worst = assert_same_lists(rp, full.rows(), score_rtol=1e-5, score_atol=ATOL)
    assert n_diff <= 4, n_diff
    ctx.close()


def test_small_clusters_take_the_full_pass(forced, monkeypatch):
    """Clusters of a few hundred users are not pruned (their seed thresholds are too weak: DESIGN.md section 7b, cluster-count
    sweep): the user threshold FY_PRUNE_MIN_USERS (600 in production) decides per cluster; the lists are the oracle's either way."""
    P = pkg()
    data, clustering, conf, ref = make("ml100k", 3, "0.1", 20)       # 314 users per cluster
    ctx = P.Context(0)
    monkeypatch.setenv("FY_PRUNE_MIN_USERS", "400")
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    assert rec.stats["blocks_total"] == 0 and rec.stats["panel_clusters"] == 0
    assert_topn_matches(rec.rows(), ref, 20)
    monkeypatch.setenv("FY_PRUNE_MIN_USERS", "100")
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    assert rec.stats["blocks_total"] > 0
    assert_topn_matches(rec.rows(), ref, 20)
    ctx.close()


@pytest.mark.parametrize("shape,K,lam,top_n,seed_chunks,select", [("ml100k", 1, "0.1", 300, 2, 0), ("ml100k", 3, "0.5", 400, 1, 0), ("ml100k", 1, "0.0", 300, 2, 0),
                                                                    ("ml100k", 1, "0.1", 40, 5, 0), ("ml100k", 1, "0.1", 300, 2, 1), ("ml100k", 1, "0.1", 1000, 3, 0)])
def test_long_lists_take_the_branch_and_bound(monkeypatch, shape, K, lam, top_n, seed_chunks, select):
    """Round 4: lists longer than 256 items (the reference's default is numberOfRecommendations = 1000, RMRecommenderDriver.java:95) and
    seeds wider than 1024 columns go through the pruned flow too -- k_topn_long in its seed mode (tau_u = the N-th best of the seed
    columns, the list that stands unless a block survives) and its merge mode (seed + surviving blocks against the exact tau_u).
    Forced onto MovieLens-100K-shaped data, where the brute-force oracle decides every row; lambda = 0 gives rows of -inf (massive ties
    at the cut-off); select = 1 sends every merged user through the radix-select fallback."""
    monkeypatch.setenv("FY_PRUNE_MIN_ITEMS", "256")
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    monkeypatch.setenv("FY_SEED_CHUNKS", str(seed_chunks))
    monkeypatch.setenv("FY_TOPN_FORCE_SELECT", str(select))
    # (a list of 300 of 1682 items: most blocks survive; the job must not give up and take the plain full pass -- the merge of a seed
    # row with MANY surviving blocks is what this test is for)
    monkeypatch.setenv("FY_MAX_SURV_FRAC", "1.5")
    P = pkg()
    data, clustering, conf, ref = make(shape, K, lam, top_n)
    ctx = P.Context(0)
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    st = rec.stats
    assert st["blocks_total"] > 0, "the branch and bound did not run"
    assert st["prune_fallbacks"] == 0
    # (a pruned cluster stores 24-bit rows here -- FY_M24_MIN_ITEMS=0 -- whose scores carry the format's rounding: the absolute slack of
    # tests/util.py is the reference's own criterion halved, for lists whose scores cross zero)
    from util import ATOL
    assert_topn_matches(rec.rows(), ref, top_n, atol=ATOL)
    monkeypatch.setenv("FY_PRUNE", "0")
    full = P.RM2Job(conf, ctx).run(data, clustering=clustering)
    assert full.stats["blocks_total"] == 0
    from fullsize_checks import assert_same_lists

    def by_user(r):
        o = np.argsort(r["user"], kind="stable")
        return {k: r[k][o] for k in ("user", "item", "score")}
    n_diff, worst = assert_same_lists(by_user(rec.rows()), by_user(full.rows()), score_rtol=1e-5, score_atol=ATOL)
    assert n_diff <= 4, n_diff
    ctx.close()


def test_refinement_of_rows_that_nearly_cancel(monkeypatch):
    """Round 4: list rows with |score| < 2 sqrt(n) whose item lies in the first 256 columns are scored AGAIN in fp64 from the unrounded fp32
    values of the head rows (k_refine_rows).  The packed 24-bit matrix forced onto MovieLens-100K-shaped data in three clusters with a
    large numberOfItems (pvpi > 0: scores cross zero): with the pass switched off the worst relative error against the brute-force
    oracle is that of the format (~1e-5 and more on the rows that cancel); with it the same rows come out at fp32-input precision."""
    monkeypatch.setenv("FY_PRUNE_MIN_ITEMS", "256")
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    monkeypatch.setenv("FY_SEED_CHUNKS", "1")
    P, S = pkg(), synth()
    u, i, s, facts = S.generate("ml100k")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    clustering = (uu, S.hash_clustering(uu, 3))
    M = 40_000
    ref = oracle.rm2(u, i, s, lam=0.1, number_of_items=M, number_of_recommendations=1 << 30, number_of_clusters=3,
                     map_user=clustering[0], map_cluster=clustering[1], n_threads=8)
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", M)
    conf.setInt("numberOfClusters", 3)
    conf.setInt("numberOfRecommendations", 30)
    from util import ATOL, full_ranking
    ranking = full_ranking(ref)
    # column of an item inside its cluster = popularity rank (most rated first, ties by ascending raw id): the pass re-scores columns < 256
    cl_of_user = dict(zip(clustering[0].tolist(), clustering[1].tolist()))
    keep = s > 0
    col = {}
    for c in range(3):
        m = keep & np.array([cl_of_user[x] == c for x in u.tolist()])
        ids, cnt = np.unique(i[m], return_counts=True)
        order = np.lexsort((ids, -cnt))
        col[c] = {int(ids[k]): r for r, k in enumerate(order)}
    n_of_user = dict(zip(*np.unique(u[keep], return_counts=True)))

    def worst_of(rows):
        w_all, a_in, r_in, where = 0.0, 0.0, 0.0, None
        for uid, it, sc, c in zip(rows["user"].tolist(), rows["item"].tolist(), rows["score"].tolist(), rows["cluster"].tolist()):
            items, scores, _ = ranking[uid]
            want = float(scores[np.flatnonzero(items == it)[0]])
            rel = abs(sc - want) / abs(want)
            w_all = max(w_all, rel)
            if col[c][it] < 256 and abs(want) < 1.9 * np.sqrt(n_of_user[uid]):      # (inside the pass's criterion with a margin)
                if abs(sc - want) > a_in:
                    a_in, where = abs(sc - want), (uid, it, c, col[c][it], int(n_of_user[uid]), sc, want)
                if abs(want) >= 0.05:
                    r_in = max(r_in, rel)
        print("   worst row inside the criterion (user, item, cluster, column, ratings, got, oracle):", where)
        return w_all, a_in, r_in
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("FY_REFINE", flag)
        ctx = P.Context(0)
        rec = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering)
        rows, st = rec.rows(), rec.stats
        assert_topn_matches(rows, ref, 30, atol=ATOL)
        out[flag] = worst_of(rows) + (st["rows_refined"],)
        ctx.close()
    print("24-bit matrix forced, scores that cross zero, rows inside the criterion, without / with the refinement pass: worst ABSOLUTE error %.2e / %.2e, "
          "worst relative error of those with |score| >= 0.05: %.2e / %.2e (all rows, relative: %.2e / %.2e; %d rows re-scored)"
          % (out["0"][1], out["1"][1], out["0"][2], out["1"][2], out["0"][0], out["1"][0], out["1"][3]))
    assert out["0"][3] == 0 and out["1"][3] > 0
    # what is left in a re-scored row is the fp32 rounding of the walk's WEIGHTS (x_vi / s_v enters the fixed-point sums as an fp32
    # product with the rating): ~3e-8 per term relative -- measured 2.8e-8 absolute on a score of 0.0015 made of 26 terms of ~4
    # (and grows with the list length like everything else: ~1e-6 absolute on a 300-rating user's score of 1)
    assert out["1"][1] <= 2e-6 and out["1"][1] < 0.1 * out["0"][1]
    assert out["1"][2] <= 2e-6
