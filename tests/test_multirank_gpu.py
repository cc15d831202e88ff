"""The user-sharded (multi-GPU) RM2 path, rehearsed on ONE GPU: every rank's stage 1 runs in turn, the all-gather of
the partial item statistics is played by hand (concatenation in rank order -- exactly the layout RCCL's all-gather
produces), every rank's stage 2 runs, and the union of the ranks' rows must equal the single-rank result."""
import os

import numpy as np
import pytest
import torch

import oracle
from util import assert_topn_matches, pkg, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pkg().Context(0)
    yield c
    c.close()


def device_doubles(ptr, n):
    par = __import__("importlib").import_module("filmyou-core_amd.parallel")
    return torch.as_tensor(par._DevicePointer(ptr, n, "<f8"), device="cuda:0")


@pytest.mark.parametrize("shape,K,world", [("tiny", 1, 2), ("tiny", 5, 3), ("ml100k", 1, 4), ("ml100k", 12, 8), ("ml100k", 8, 8), ("ml100k", 50, 4)])
def test_sharded_equals_single(ctx, shape, K, world):
    sharded_equals_single(ctx, shape, K, world)


def test_whole_clusters_with_the_replicated_prep(monkeypatch):
    """FY_SHARD_PREP=0: every rank preps all ratings and owns a run of whole clusters of the common slot order (round 4's first flow)."""
    monkeypatch.setenv("FY_SHARD_PREP", "0")
    c = pkg().Context(0)
    try:
        sharded_equals_single(c, "ml100k", 12, 8)
    finally:
        c.close()


def test_a_failure_only_one_rank_can_see_fails_every_rank(ctx):
    """Sharded prep: a duplicate (user, item) rating sits in ONE rank's clusters.  That rank still hands out its exchange buffer (with the
    failure flag at its end), and fy_rm2_set_global_stats fails on every rank with the same code -- nobody waits in a collective for a
    rank that has gone."""
    P, S = pkg(), synth()
    u, i, s, facts = S.generate("ml100k")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    K, world = 8, 4
    clustering = (uu, S.hash_clustering(uu, K))
    u2, i2, s2 = np.append(u, u[0]), np.append(i, i[0]), np.append(s, s[0])
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 20)
    ratings = P.Ratings(ctx, u2, i2, s2)
    job = P.RM2Job(conf, ctx)
    prepared = [job.prepare(ratings, clustering=clustering, rank=r, world=world) for r in range(world)]     # no rank fails here
    parts = []
    for pr in prepared:
        ptr, n = pr.partial_stats()
        parts.append(device_doubles(ptr, n).clone())
    torch.cuda.synchronize()
    assert sorted(float(p[-1]) for p in parts) == [0.0] * (world - 1) + [7.0]        # -FY_ERR_DUPLICATE_RATING, on the owner alone
    gathered = torch.cat(parts).contiguous()
    for pr in prepared:
        with pytest.raises(P.FilmYouError) as e:
            pr.set_global_stats(gathered.data_ptr())
        assert e.value.code == -7
        pr.close()
    ratings.close()


def test_sharded_prep_with_ragged_input(ctx):
    """Sharded prep on input that is not tidy: users the clustering map does not name (cluster 0, quirk Q2), map entries of users who
    rated nothing, a repeated map entry (the later one wins), clusters nobody is routed to, ratings <= 0 (dropped), scores that are no
    halves (the prep's general mode) -- the ranks' rows together are the one-rank rows, bit for bit, and the clusteringCount is checked
    by the rank that owns the cluster."""
    P, S = pkg(), synth()
    u, i, s, facts = S.generate("ml100k", seed_offset=3)
    u, i, s = u.numpy().copy(), i.numpy().copy(), s.numpy().copy()
    s[::17] = 0.0
    s[5::23] = -1.0
    s[3] = np.float32(2.7)
    uu = np.unique(u)
    K, world = 11, 4
    named = uu[uu % 5 != 0]                                  # a fifth of the users is not in the map
    cl = S.hash_clustering(named, K)
    cl[cl == 7] = 2                                          # nobody in cluster 7
    cl[cl == 9] = 3                                          # nor in 9
    map_user = np.concatenate([named, [int(uu.max()) + 50, int(uu.max()) + 51], named[:3]]).astype(np.int32)
    map_cluster = np.concatenate([cl, [7, 4], [(int(cl[0]) + 1) % 7, int(cl[1]), int(cl[2])]]).astype(np.int32)
    clustering = (map_user, map_cluster)
    conf = P.Configuration()
    conf.set("lambda", "0.2")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 15)
    ratings = P.Ratings(ctx, u, i, s)
    job = P.RM2Job(conf, ctx)
    single = job.run(ratings, clustering=clustering)
    # a valid clusteringCount file: the rated users (some rating > 0) per cluster, by the map's own rules
    rows1 = single.rows()
    route = {}
    for mu, mc in zip(map_user.tolist(), map_cluster.tolist()):
        route[mu] = mc                                       # a later entry of a user wins
    count = np.zeros(K, dtype=np.int32)
    for usr in np.unique(u[s > 0]).tolist():
        count[route.get(usr, 0)] += 1
    assert count[7] == 0 and count[9] == 0 and (count > 0).sum() >= world
    prepared = [job.prepare(ratings, clustering=clustering, clustering_count=count, rank=r, world=world) for r in range(world)]
    assert all(pr.stats_layout()[1] > 0 for pr in prepared)      # sharded
    parts = []
    for pr in prepared:
        ptr, n = pr.partial_stats()
        parts.append(device_doubles(ptr, n).clone())
    torch.cuda.synchronize()
    gathered = torch.cat(parts).contiguous()
    results = []
    for pr in prepared:
        pr.set_global_stats(gathered.data_ptr())
        results.append(pr.score())
        pr.close()
    rows = {k: np.concatenate([r.rows()[k] for r in results]) for k in ("user", "item", "score", "cluster")}
    order = np.lexsort((rows["item"], rows["user"]))
    order1 = np.lexsort((rows1["item"], rows1["user"]))
    for k in ("user", "item", "cluster"):
        np.testing.assert_array_equal(rows[k][order], rows1[k][order1])
    # (one score is no half, so the item sums are fp64 sums of inexact terms: summed per rank and then over ranks they may differ from the
    # one-rank sum in the last bit, and so may a score -- as in any exchange of partial sums; with half stars everything is bit-identical,
    # test_sharded_equals_single)
    a_, b_ = rows["score"][order].astype(np.float64), rows1["score"][order1].astype(np.float64)
    assert np.max(np.abs(a_ - b_) / np.abs(b_)) < 1e-6
    for r in results:
        if r.size:
            for k2 in ("user_id", "item_id"):
                np.testing.assert_array_equal(r.sums()[k2], single.sums()[k2])
            for k2 in ("user_sum", "item_coll", "total_sum"):
                np.testing.assert_allclose(r.sums()[k2], single.sums()[k2], rtol=1e-14, atol=0)
    # a clusteringCount that is wrong for ONE cluster: only its owner can see it, every rank fails with FY_ERR_CLUSTER_COUNT
    bad = count.copy()
    bad[int(np.flatnonzero(count)[-1])] += 1
    prepared = [job.prepare(ratings, clustering=clustering, clustering_count=bad, rank=r, world=world, cache=False) for r in range(world)]
    parts = []
    for pr in prepared:
        ptr, n = pr.partial_stats()
        parts.append(device_doubles(ptr, n).clone())
    torch.cuda.synchronize()
    assert sorted(float(p[-1]) for p in parts) == [0.0] * (world - 1) + [6.0]
    gathered = torch.cat(parts).contiguous()
    for pr in prepared:
        with pytest.raises(P.FilmYouError) as e:
            pr.set_global_stats(gathered.data_ptr())
        assert e.value.code == -6
        pr.close()
    ratings.close()


def test_sharded_panel_mode(ctx, monkeypatch):
    """The many-cluster (column-panel) path with the users of every cluster split over three ranks: each rank builds the
    cluster's panel and bounds for its own users (forced onto small data like tests/test_pruned_coop_gpu.py does)."""
    for k, v in (("FY_PRUNE_MIN_ITEMS", "256"), ("FY_M24_MIN_ITEMS", "0"), ("FY_SEED_CHUNKS", "1"), ("FY_PANEL_MIN_CLUSTERS", "1"),
                 ("FY_PANEL_COLS", "256"), ("FY_COOC_MAX_CH", "256")):
        monkeypatch.setenv(k, v)
    st = sharded_equals_single(ctx, "ml100k", 2, 3)
    assert all(x["panel_clusters"] > 0 for x in st)


def sharded_equals_single(ctx, shape, K, world):
    P, S = pkg(), synth()
    u, i, s, facts = S.generate(shape)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.unique(u)
    clustering = (uu, S.hash_clustering(uu, K))
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 20)
    ratings = P.Ratings(ctx, u, i, s)
    job = P.RM2Job(conf, ctx)
    single = job.run(ratings, clustering=clustering)

    prepared = [job.prepare(ratings, clustering=clustering, rank=r, world=world) for r in range(world)]
    parts = []
    for pr in prepared:
        ptr, n = pr.partial_stats()
        parts.append(device_doubles(ptr, n).clone())
    torch.cuda.synchronize()
    gathered = torch.cat(parts).contiguous()                       # rank-major, like all_gather_into_tensor
    # the partials really are partial: each is a strict part of the total, and they add up to the item sums
    total = torch.stack(parts).sum(0).cpu().numpy()
    sums = single.sums()
    layouts = {pr.stats_layout() for pr in prepared}
    assert len(layouts) == 1
    n_item_slots, n_user_slots = layouts.pop()
    n_nonempty = len(np.unique(clustering[1][np.isin(clustering[0], uu)]))
    # at least as many clusters as ranks: sharded prep (a rank preps its own clusters' ratings alone), the buffer is by raw id and
    # carries the user sums and a failure flag; otherwise one slot per rated item
    assert (n_user_slots > 0) == (world > 1 and n_nonempty >= world and os.environ.get("FY_SHARD_PREP", "1") != "0")
    assert len(total) == n_item_slots + 1 + (n_user_slots + 1 if n_user_slots else 0)
    item_sums, floor_sum = total[:n_item_slots], total[n_item_slots]
    if n_user_slots:
        assert n_item_slots == int(i.max()) + 1 and n_user_slots == int(u.max()) + 1 and total[-1] == 0.0
        user_sums = total[n_item_slots + 1:-1]
        np.testing.assert_array_equal(np.nonzero(user_sums)[0], sums["user_id"])
        np.testing.assert_array_equal(user_sums[user_sums > 0], sums["user_sum"])
        np.testing.assert_array_equal(np.nonzero(item_sums)[0], sums["item_id"])
        item_sums = item_sums[item_sums > 0]
    np.testing.assert_array_equal(item_sums / (floor_sum / 100.0), sums["item_coll"])
    assert floor_sum / 100.0 == sums["total_sum"]
    if world > 1:
        assert all(float(p[:n_item_slots].sum()) < float(total[:n_item_slots].sum()) for p in parts)
    results = []
    for pr in prepared:
        pr.set_global_stats(gathered.data_ptr())
        results.append(pr.score())
        pr.close()
    for r in results:       # every rank returns the GLOBAL side outputs (rm2/userSum, rm2/itemColl), whatever it prepared
        if r.size:
            for k2 in ("user_id", "user_sum", "item_id", "item_coll", "total_sum"):
                np.testing.assert_array_equal(r.sums()[k2], sums[k2])
    rows = {k: np.concatenate([r.rows()[k] for r in results]) for k in ("user", "item", "score", "cluster")}
    # ranks own disjoint users, together all of them
    owners = [set(r.rows()["user"].tolist()) for r in results]
    assert sum(len(o) for o in owners) == len(set().union(*owners)) == len(np.unique(single.rows()["user"]))
    ref = oracle.rm2(u, i, s, lam=0.1, number_of_items=facts["n_items"], number_of_recommendations=1 << 30,
                     number_of_clusters=K, map_user=clustering[0], map_cluster=clustering[1], n_threads=8)
    assert_topn_matches(rows, ref, 20)
    # and the shards agree with the one-GPU run row for row (same kernels, same statistics)
    key = lambda r: sorted(zip(r["user"].tolist(), r["item"].tolist()))
    assert key(rows) == key(single.rows())
    a = dict(zip(zip(rows["user"].tolist(), rows["item"].tolist()), rows["score"].tolist()))
    b = dict(zip(zip(single.rows()["user"].tolist(), single.rows()["item"].tolist()), single.rows()["score"].tolist()))
    assert max(abs(a[k] - b[k]) / abs(b[k]) for k in a) < 1e-6
    # at least as many clusters as ranks: WHOLE clusters per rank (SURVEY.md 8e, RM2Job.java:251 -- one reduce group per cluster): no
    # cluster's co-rating matrix is built on two ranks
    n_nonempty = len(np.unique(single.rows()["cluster"]))
    if n_nonempty >= world:
        held = [set(np.unique(r.rows()["cluster"]).tolist()) for r in results]
        assert sum(len(h) for h in held) == len(set().union(*held)) == n_nonempty, held
    st = [r.stats for r in results]
    assert sum(x["users_scored"] for x in st) == single.stats["users_scored"]
    assert sum(x["log_terms"] for x in st) == single.stats["log_terms"]
    ratings.close()
    return st


def test_stats_exchange_wraps_the_library_buffer(ctx, rm_golden):
    """parallel.StatsExchange with world == 1 degenerates to a copy; the pointer round-trips through torch."""
    P = pkg()
    par = __import__("importlib").import_module("filmyou-core_amd.parallel")
    conf = P.Configuration()
    conf.setFloat("lambda", 0.5)
    conf.setInt("numberOfItems", 100)
    conf.setInt("numberOfClusters", 10)
    pr = P.RM2Job(conf, ctx).prepare(rm_golden["coo"], clustering=(rm_golden["map_user"], rm_golden["map_cluster"]))
    ptr, n = pr.partial_stats()
    assert n == 101
    ex = par.StatsExchange(0)
    g = ex(ptr, n)
    t = device_doubles(g, n).cpu().numpy()
    np.testing.assert_array_equal(t[:-1], np.asarray(rm_golden["itemSum"]))
    assert t[-1] == rm_golden["totalSum"] * 100
    pr.set_global_stats(g)
    assert pr.score().size == 507
    pr.close()
