"""Parity of the HIP RM2 path (through the C ABI) with the CPU oracle and the reference's golden vectors.

Reads like the reference's src/test/java/.../rm/TestHDFSRM2.java: build a Configuration, run the job on the
fixture, compare userSum, itemColl and the recommendations -- with the reference's 1e-4 absolute bound AND the
1e-5 relative bound of north_star.
"""
import os

import numpy as np
import pytest

import oracle
from util import RTOL, assert_topn_matches, pkg, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pkg().Context(0)
    yield c
    c.close()


def build_conf(g=None, lam=0.5, n_items=None, n_clusters=None, top_n=1000):
    P = pkg()
    conf = P.Configuration()
    conf.setInt("numberOfRecommendations", top_n)          # HadoopIntegrationTest.buildConf :84
    conf.setFloat("lambda", lam)                           # :95
    conf.setInt("clusterSplit", 5)                         # :96
    conf.setInt("splitSize", 3)                            # :97
    conf.setInt("numberOfItems", n_items if n_items is not None else g["numberOfItems"])
    conf.setInt("numberOfClusters", n_clusters if n_clusters is not None else g["numberOfClusters"])
    return conf


def test_hdfs_rm2_fixture(ctx, rm_golden):
    g = rm_golden
    rec = pkg().RM2Job(build_conf(g), ctx).run(g["coo"], clustering=(g["map_user"], g["map_cluster"]),
                                              clustering_count=g["clusteringCount"])
    sums = rec.sums()
    # compareIntDoubleData(userSum), compareMapIntDoubleData(itemColl)
    assert list(sums["user_id"]) == list(range(1, 31))
    np.testing.assert_array_equal(sums["user_sum"], np.asarray(g["userSum"]))
    assert sums["total_sum"] == g["totalSum"]
    np.testing.assert_allclose(sums["item_coll"], np.asarray(g["itemColl"]), rtol=1e-15)
    # compareIntPairFloatData(recommendations): exact row count, every pair within tolerance
    rows = rec.rows()
    exp = np.asarray(g["recommendations"])
    assert rec.size == len(exp) == 507
    got = {(int(u), int(i)): float(s) for u, i, s in zip(rows["user"], rows["item"], rows["score"])}
    assert len(got) == 507
    worst = 0.0
    for u, i, s in exp:
        v = got[(int(u), int(i))]
        assert abs(v - s) <= g["params"]["reference_tolerance_abs"]
        worst = max(worst, abs(v - s) / abs(s))
    assert worst <= RTOL
    cl = np.asarray(g["clustering"])
    assert all(cl[u - 1] == c for u, c in zip(rows["user"], rows["cluster"]))
    st = rec.stats
    assert st["nnz"] == int((g["coo"][2] > 0).sum()) and st["users_scored"] == 30 and st["recs"] == 507


def test_fixture_with_the_packed_matrix_format_forced(rm_golden, monkeypatch):
    """The production matrix format of the big clusters (24-bit e7m17, scaled per cluster) FORCED onto the reference's fixture, whose
    clusters (7 users, ~100 items) would keep exact fp32 rows by default: the measured worst absolute and relative error against the 507
    golden triples is printed -- the number DESIGN.md section 2 quotes.  Asserted: north_star's relative 1e-5.  The reference's own
    criterion is ABSOLUTE 1e-4 (T/util/HadoopIntegrationTest.java:53): on scores of -458 that is 2e-7 relative, which a format with a
    17-bit mantissa cannot promise -- which is exactly why clusters below 4096 items never get it."""
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    g = rm_golden
    c = pkg().Context(0)
    rec = pkg().RM2Job(build_conf(g), c).run(g["coo"], clustering=(g["map_user"], g["map_cluster"]), clustering_count=g["clusteringCount"])
    rows = rec.rows()
    exp = np.asarray(g["recommendations"])
    assert rec.size == len(exp) == 507
    got = {(int(u), int(i)): float(s) for u, i, s in zip(rows["user"], rows["item"], rows["score"])}
    worst_abs = max(abs(got[(int(u), int(i))] - s) for u, i, s in exp)
    worst_rel = max(abs(got[(int(u), int(i))] - s) / abs(s) for u, i, s in exp)
    print("fixture with the 24-bit matrix forced: worst absolute error %.2e (the reference asserts 1e-4), worst relative error %.2e" % (worst_abs, worst_rel))
    assert worst_rel <= RTOL
    c.close()


def oracle_full(user, item, score, lam, n_items, K, mu=None, mc=None):
    return oracle.rm2(user, item, score, lam=lam, number_of_items=n_items, number_of_recommendations=1 << 30,
                      number_of_clusters=K, map_user=mu, map_cluster=mc, n_threads=8)


@pytest.mark.parametrize("top_n", [1, 7, 1000])
def test_fixture_top_n_truncation(ctx, rm_golden, top_n):
    g = rm_golden
    user, item, score = g["coo"]
    rec = pkg().RM2Job(build_conf(g, top_n=top_n), ctx).run(g["coo"], clustering=(g["map_user"], g["map_cluster"]))
    ref = oracle_full(user, item, score, 0.5, 100, 10, g["map_user"], g["map_cluster"])
    assert_topn_matches(rec.rows(), ref, top_n)


def test_fixture_single_cluster_and_filter_users(ctx, rm_golden):
    g = rm_golden
    user, item, score = g["coo"]
    conf = build_conf(g, lam=0.1, n_clusters=1, top_n=10)
    conf.setInt("filterUsers", 12)
    rec = pkg().RM2Job(conf, ctx).run(g["coo"])
    ref = oracle.rm2(user, item, score, lam=float(conf.get("lambda")), number_of_items=100,
                     number_of_recommendations=1 << 30, number_of_clusters=1, filter_users=12)
    rows = rec.rows()
    assert rows["user"].min() == 12
    assert_topn_matches(rows, ref, 10)


@pytest.mark.parametrize("shape,K,lam,top_n", [("tiny", 1, 0.1, 50), ("tiny", 7, 0.3, 20), ("ml100k", 1, 0.1, 100),
                                              ("ml100k", 20, 0.1, 50)])
def test_synthetic_vs_oracle(ctx, shape, K, lam, top_n):
    S = synth()
    u, i, s, facts = S.generate(shape)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    mc = S.hash_clustering(uu, K)
    conf = build_conf(lam=lam, n_items=facts["n_items"], n_clusters=K, top_n=top_n)
    rec = pkg().RM2Job(conf, ctx).run((u, i, s), clustering=(uu, mc))
    ref = oracle_full(u, i, s, float(conf.get("lambda")), facts["n_items"], K, uu, mc)
    worst = assert_topn_matches(rec.rows(), ref, top_n)
    sums = rec.sums()
    np.testing.assert_array_equal(sums["user_sum"], ref["user_sum"])
    assert sums["total_sum"] == ref["total_sum"]                       # Q1 on half-star data for "tiny"
    np.testing.assert_allclose(sums["item_coll"], ref["item_coll"], rtol=1e-14)
    assert rec.stats["log_terms"] == ref["log_terms"]
    print("worst relative error", shape, K, worst)


@pytest.mark.parametrize("env,flat", [({"FY_FLAT": "0"}, False), ({}, True), ({"FY_FLAT_BUDGET_MB": "16"}, True),
                                      ({"FY_M24_MIN_ITEMS": "64"}, True), ({"FY_M24_MIN_ITEMS": "64", "FY_FLAT_BUDGET_MB": "16"}, True),
                                      ({"FY_M24_MIN_ITEMS": "64", "FY_FLAT": "0"}, False)])
def test_many_small_clusters_in_one_launch_per_kernel(ctx, monkeypatch, env, flat):
    """Flat batch (fy_rm2_kernels.hpp: FlatDesc): the unpruned clusters of a multi-cluster job run every kernel of their chain once for
    all of them.  Against the oracle, with fp32 and with 24-bit matrix rows, in one batch and in several (a budget of 16 MB holds three or
    four clusters), and the round-2 path (one cluster after the other on the lanes) beside it."""
    S = synth()
    u, i, s, facts = S.generate("ml100k", seed_offset=3)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    K = 20
    mc = S.hash_clustering(uu, K)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    conf = build_conf(lam=0.2, n_items=facts["n_items"], n_clusters=K, top_n=30)
    rec = pkg().RM2Job(conf, ctx).run((u, i, s), clustering=(uu, mc))
    ref = oracle_full(u, i, s, 0.2, facts["n_items"], K, uu, mc)
    assert_topn_matches(rec.rows(), ref, 30)
    launches = rec.stats["score_launches"]
    if flat:
        assert launches < K, launches                      # one scoring launch per batch
        if "FY_FLAT_BUDGET_MB" in env:
            assert launches > 1, launches                  # several batches
        else:
            assert launches == 1, launches
    else:
        assert launches == K, launches


def test_edge_cases_match_oracle(ctx):
    P = pkg()
    # a 1-user cluster whose items are all rated (no list), a 1-user cluster next to a 2-user one, an unmapped user
    user = np.array([1, 1, 2, 2, 3, 3, 4, 4, 4, 9], dtype=np.int32)
    item = np.array([1, 2, 2, 3, 1, 3, 1, 2, 3, 5], dtype=np.int32)
    score = np.array([5, 3, 4, 1, 2, 2, 0.5, 0, -1, 3], dtype=np.float32)   # 0 and -1 are dropped by score > 0
    mu, mc = np.array([1, 2, 3, 4], dtype=np.int32), np.array([1, 2, 2, 0], dtype=np.int32)   # user 9 unmapped -> 0
    conf = build_conf(lam=0.5, n_items=5, n_clusters=3, top_n=10)
    rec = P.RM2Job(conf, ctx).run((user, item, score), clustering=(mu, mc))
    ref = oracle_full(user, item, score, 0.5, 5, 3, mu, mc)
    assert_topn_matches(rec.rows(), ref, 10)
    # -inf scores (U_c = 1 with an unrated cluster item cannot happen; U_c = 2 where the neighbour lacks the item can't
    # give 0 either because of smoothing) -- force lambda = 0 so a neighbourless product is exactly zero
    conf0 = build_conf(lam=0.0, n_items=5, n_clusters=3, top_n=10)
    rec0 = P.RM2Job(conf0, ctx).run((user, item, score), clustering=(mu, mc))
    ref0 = oracle_full(user, item, score, 0.0, 5, 3, mu, mc)
    assert_topn_matches(rec0.rows(), ref0, 10)


def test_errors_mirror_the_reference(ctx, rm_golden):
    P = pkg()
    g = rm_golden
    bad = np.array(g["clusteringCount"]).copy()
    bad[0] += 1
    with pytest.raises(RuntimeError, match="RM2 failed!"):
        P.RM2Job(build_conf(g), ctx).run(g["coo"], clustering=(g["map_user"], g["map_cluster"]), clustering_count=bad)
    with pytest.raises(RuntimeError, match=r"RM2 failed!.*routed to a cluster outside"):      # cluster id outside [0, numberOfClusters)
        P.RM2Job(build_conf(g, n_clusters=3), ctx).run(g["coo"], clustering=(g["map_user"], g["map_cluster"]))
    u, i, s = g["coo"]
    # duplicate (user, item): the copy of the FIRST rating at the END of the input, the user's other ratings in between -- the check sits
    # behind the (cluster, item) sort, where the two are neighbours whatever the input order; a copy right behind the original too
    with pytest.raises(RuntimeError, match=r"RM2 failed!.*share one \(user, item\) key"):
        P.RM2Job(build_conf(g), ctx).run((np.r_[u, u[:1]], np.r_[i, i[:1]], np.r_[s, s[:1]] + 1))
    with pytest.raises(RuntimeError, match=r"RM2 failed!.*share one \(user, item\) key"):
        P.RM2Job(build_conf(g), ctx).run((np.r_[u[:5], u[4:5], u[5:]], np.r_[i[:5], i[4:5], i[5:]], np.r_[s[:5], s[4:5], s[5:]]))
    with pytest.raises(ValueError):
        P.RM2Job(P.Configuration(), ctx).run(g["coo"])
    # empty input: no rows, no failure
    rec = P.RM2Job(build_conf(g), ctx).run((np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)))
    assert rec.size == 0


def test_one_call_host_entry_point(rm_golden):
    """fy_rm2_run: the single call a JNI shim makes, host buffers in, host rows out."""
    import ctypes as C
    P = pkg()
    L = P._native.load()
    g = rm_golden
    u, i, s = g["coo"]
    p = P._native.RM2Params(0.5, 100, 1000, 0, 10, 0, 1, 0, 0)
    mu, mc = g["map_user"], g["map_cluster"]
    out = C.c_void_p()
    rc = L.fy_rm2_run(C.byref(p), len(u), u.ctypes.data, i.ctypes.data, s.ctypes.data, len(mu), mu.ctypes.data,
                      mc.ctypes.data, None, C.byref(out))
    assert rc == 0, L.fy_last_error()
    assert L.fy_result_size(out) == 507
    P._native.load().fy_result_free(out)


def test_job_from_the_references_file_layout(rm_golden, tmp_path):
    """RM2Job.run as the reference's test drives it (T/rm/TestHDFSRM2.java:39-75): ratings / clustering / clusteringCount
    written as SequenceFiles the way DataInitialization does, the job run from the Configuration's paths, rm2/userSum,
    rm2/itemColl and the recommendations read back from the files and compared with the golden vectors (:70-72)."""
    import importlib
    P = pkg()
    sf = importlib.import_module("filmyou-core_amd.seqfile")
    base = str(tmp_path / "recommendation")
    user, item, score = rm_golden["coo"]
    keep = score > 0                                             # createIntPairFloatFile writes only data[i][j] > 0
    sf.write_intpair_float(str(tmp_path / "input" / "ratings" / "data"), user[keep], item[keep], score[keep])
    sf.write_int_int(os.path.join(base, "clustering", "data"), rm_golden["map_user"], rm_golden["map_cluster"])
    cc = rm_golden["cluster_count"]
    sf.write_int_int(os.path.join(base, "clusteringCount", "data"), np.arange(len(cc), dtype=np.int32), cc)
    conf = P.Configuration()
    conf.setFloat("lambda", 0.5)
    conf.setInt("numberOfItems", rm_golden["numberOfItems"])
    conf.setInt("numberOfClusters", rm_golden["numberOfClusters"])
    conf.setInt("numberOfRecommendations", 1000)
    conf.set("directory", base)
    conf.set("mapred.input.dir", str(tmp_path / "input" / "ratings"))
    conf.set("mapred.output.dir", str(tmp_path / "output"))
    ctx = P.Context(0)
    os.makedirs(os.path.join(base, "rm2", "stale"))              # RM2Job.run wipes <directory>/rm2 first (RM2Job.java:84)
    assert P.RM2Job(conf, ctx).run_from_files() == 0
    assert not os.path.exists(os.path.join(base, "rm2", "stale"))
    ku, vu = sf.read_int_double(os.path.join(base, "rm2", "userSum"))
    np.testing.assert_array_equal(ku, np.arange(1, 31))
    np.testing.assert_array_equal(vu, np.asarray(rm_golden["userSum"]))
    ki, vi = sf.read_int_double(os.path.join(base, "rm2", "itemColl"))
    np.testing.assert_array_equal(ki, np.arange(1, 101))
    np.testing.assert_allclose(vi, np.asarray(rm_golden["itemColl"]), rtol=1e-15)
    ru, ri, rs = sf.read_intpair_float(str(tmp_path / "output"))
    exp = np.asarray(rm_golden["recommendations"])
    assert len(ru) == len(exp) == 507                             # exact row count (HadoopIntegrationTest.java:434-436)
    got = {(int(a), int(b)): float(c) for a, b, c in zip(ru, ri, rs)}
    for a, b, c in exp:
        assert abs(got[(int(a), int(b))] - c) <= 1e-4             # the reference's own tolerance (:53)
        assert abs(got[(int(a), int(b))] - c) <= 1e-5 * abs(c)
    ctx.close()


def test_two_runs_give_bit_identical_rows():
    """The packed walk accumulates in fixed point (ds_add_u64): integer sums do not depend on the order in which the waves'
    atomics land, so the co-rating matrix -- and with it every score -- is bit-reproducible from run to run."""
    P = pkg()
    S = synth()
    u, i, s, facts = S.generate("ml100k", seed_offset=5)
    u, i, s = u.numpy(), i.numpy(), (np.round(s.numpy() * 2) / 2).astype(np.float32)       # half stars: the packed walk
    uu = np.unique(u)
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", 4)
    conf.setInt("numberOfRecommendations", 25)
    ctx = P.Context(0)
    clustering = (uu, S.hash_clustering(uu, 4))
    a = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering).rows()
    for _ in range(3):
        b = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering).rows()
        for k in ("user", "item", "cluster"):
            np.testing.assert_array_equal(a[k], b[k])
        assert a["score"].tobytes() == b["score"].tobytes()
    ctx.close()


@pytest.mark.parametrize("K", [1, 6])
def test_prep_with_the_ratings_inside_or_beside_the_sort_keys(monkeypatch, K):
    """fy_prep's packed mode (scores exactly representable in fp16 ride in the low 16 bits of the three sort keys, the sorts move keys
    alone) against the general mode (rating as the sort's value): same structure, so bit-identical rows and statistics; a data set
    with ONE score that is no half takes the general mode by itself and still agrees with the oracle."""
    P = pkg()
    S = synth()
    u, i, s, facts = S.generate("ml100k", seed_offset=9)
    u, i, s = u.numpy(), i.numpy(), (np.round(s.numpy() * 2) / 2).astype(np.float32)
    uu = np.unique(u)
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 25)
    clustering = (uu, S.hash_clustering(uu, K))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FY_PREP_PACKED", mode)
        ctx = P.Context(0)
        rec = P.RM2Job(conf, ctx).run((u, i, s), clustering=clustering)
        out[mode] = (rec.rows(), rec.sums())
        ctx.close()
    for k in ("user", "item", "cluster"):
        np.testing.assert_array_equal(out["1"][0][k], out["0"][0][k])
    assert out["1"][0]["score"].tobytes() == out["0"][0]["score"].tobytes()
    for k in ("user_id", "user_sum", "item_id", "item_coll"):
        np.testing.assert_array_equal(out["1"][1][k], out["0"][1][k])
    monkeypatch.delenv("FY_PREP_PACKED")
    s2 = s.copy()
    s2[7] = np.float32(3.3)          # not a half: the whole data set takes the general mode
    ctx = P.Context(0)
    rows = P.RM2Job(conf, ctx).run((u, i, s2), clustering=clustering).rows()
    ctx.close()
    ref = oracle.rm2(u, i, s2, lam=0.1, number_of_items=facts["n_items"], number_of_recommendations=1 << 30, number_of_clusters=K,
                     map_user=clustering[0], map_cluster=clustering[1], n_threads=8)
    assert_topn_matches(rows, ref, 25)


def test_two_ranks_from_the_references_file_layout(tmp_path, rm_golden):
    """RM2Job.run_from_files with world = 2 (two threads, one context each, ThreadCollectives): every rank writes ITS part file into the
    shared output directory and nobody deletes it, rank 0 alone wipes <directory>/rm2 and writes the global statistics once; the
    directory read back holds the 507 golden rows exactly once.  An existing output directory fails the job like Hadoop's
    FileOutputFormat.checkOutputSpecs does (the reference never deletes mapred.output.dir, RM2Job.java:84 only <directory>/rm2)."""
    import importlib
    import threading
    P = pkg()
    sf = importlib.import_module("filmyou-core_amd.seqfile")
    par = importlib.import_module("filmyou-core_amd.parallel")
    user, item, score = rm_golden["coo"]
    keep = score > 0
    base = str(tmp_path / "recommendation")
    sf.write_intpair_float(str(tmp_path / "input" / "ratings" / "data"), user[keep], item[keep], score[keep])
    sf.write_int_int(os.path.join(base, "clustering", "data"), rm_golden["map_user"], rm_golden["map_cluster"])
    cc = rm_golden["cluster_count"]
    sf.write_int_int(os.path.join(base, "clusteringCount", "data"), np.arange(len(cc), dtype=np.int32), cc)

    def conf_for():
        conf = P.Configuration()
        conf.setFloat("lambda", 0.5)
        conf.setInt("numberOfItems", rm_golden["numberOfItems"])
        conf.setInt("numberOfClusters", rm_golden["numberOfClusters"])
        conf.setInt("numberOfRecommendations", 1000)
        conf.set("directory", base)
        conf.set("mapred.input.dir", str(tmp_path / "input" / "ratings"))
        conf.set("mapred.output.dir", str(tmp_path / "output"))
        return conf

    os.makedirs(os.path.join(base, "rm2", "stale"))
    world = 2
    group = par.ThreadGroup(world)
    err = [None] * world

    def body(rank):
        try:
            ctx = P.Context(0)
            assert P.RM2Job(conf_for(), ctx).run_from_files(rank=rank, world=world, collectives=par.ThreadCollectives(group, rank, 0)) == 0
            ctx.close()
        except BaseException as e:
            err[rank] = e
            group.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    assert not os.path.exists(os.path.join(base, "rm2", "stale"))
    assert sorted(os.listdir(str(tmp_path / "output"))) == ["part-r-00000", "part-r-00001"]
    assert os.listdir(os.path.join(base, "rm2", "userSum")) == ["part-r-00000"]
    ku, vu = sf.read_int_double(os.path.join(base, "rm2", "userSum"))
    np.testing.assert_array_equal(ku, np.arange(1, 31))                      # every key once
    np.testing.assert_array_equal(vu, np.asarray(rm_golden["userSum"]))
    ru, ri, rs = sf.read_intpair_float(str(tmp_path / "output"))
    exp = np.asarray(rm_golden["recommendations"])
    assert len(ru) == len(exp) == 507 and len({(int(a), int(b)) for a, b in zip(ru, ri)}) == 507
    got = {(int(a), int(b)): float(c) for a, b, c in zip(ru, ri, rs)}
    for a, b, c in exp:
        assert abs(got[(int(a), int(b))] - c) <= 1e-4
    ctx = P.Context(0)
    with pytest.raises(RuntimeError, match="RM2 failed!: output directory .* already exists"):
        P.RM2Job(conf_for(), ctx).run_from_files()
    ctx.close()


def test_warm_jobs_reuse_the_static_structures_and_give_identical_rows():
    """A Ratings object keeps what a job built from the ratings and the clustering alone (fy_rm2.hip: RM2Static / TableCache).  Warm
    jobs -- also with another lambda and list length, which none of that depends on -- must give bit-identical rows to cold jobs
    (cache=False = FY_RM2_NO_CACHE), and another clustering must not hit the cache."""
    P, S = pkg(), synth()
    u, i, s, facts = S.generate("ml100k", seed_offset=9)
    u, i, s = u.numpy(), i.numpy(), (np.round(s.numpy() * 2) / 2).astype(np.float32)
    uu = np.unique(u)
    ctx = P.Context(0)
    ratings = P.Ratings(ctx, u, i, s)

    def run(lam, top_n, K, cache, seed=0):
        conf = P.Configuration()
        conf.set("lambda", repr(lam))
        conf.setInt("numberOfItems", facts["n_items"])
        conf.setInt("numberOfClusters", K)
        conf.setInt("numberOfRecommendations", top_n)
        clustering = (uu, ((S.hash_clustering(uu, K) + seed) % K).astype(np.int32)) if K > 1 else None
        rec = P.RM2Job(conf, ctx).run(ratings, clustering=clustering, cache=cache)
        rows, st = rec.rows(), dict(rec.stats)
        rec.close()
        return rows, st

    same = lambda a, b: all(np.array_equal(a[k], b[k]) for k in ("user", "item", "score", "cluster"))
    for K in (1, 4):
        cold, st0 = run(0.1, 20, K, False)
        assert st0["prepared_from_cache"] == 0 and st0["tables_from_cache"] == 0
        first, st1 = run(0.1, 20, K, True)                   # builds and keeps
        assert st1["prepared_from_cache"] == 0 and st1["tables_from_cache"] == 0
        warm, st2 = run(0.1, 20, K, True)
        assert st2["prepared_from_cache"] == 1 and st2["tables_from_cache"] == 1
        assert same(cold, first) and same(cold, warm)
        other, st3 = run(0.4, 35, K, True)                   # lambda and N do not enter the cached structures
        assert st3["prepared_from_cache"] == 1
        cold_other, _ = run(0.4, 35, K, False)
        assert same(other, cold_other)
    moved, st4 = run(0.1, 20, 4, True, seed=1)               # another clustering: rebuilt
    assert st4["prepared_from_cache"] == 0
    cold_moved, _ = run(0.1, 20, 4, False, seed=1)
    assert same(moved, cold_moved)
    ratings.close()
    ctx.close()
