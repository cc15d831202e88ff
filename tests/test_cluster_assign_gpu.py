"""Cluster assignment (fy_cluster_assign) against the reference's own vectors (ClusteringTestData / SubClusteringTestData,
asserted by the reference in T/nmf/clustering/TestClusterAssignment.java) and against the oracle on random matrices; then the
whole chain H -> clustering -> RM2 job reproduces the reference's 507 recommendations."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from util import pkg

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clustering_test_data.json")


@pytest.fixture(scope="module")
def ctx():
    c = pkg().Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def test_reference_clustering(ctx, golden):
    users, clusters, counts = pkg().ClusterAssignmentJob(ctx).run(np.array(golden["H"]), first_user=1)
    assert users.tolist() == list(range(1, 31))
    assert clusters.tolist() == golden["clustering"]
    assert counts.tolist() == golden["clusteringCount"]


def test_reference_sub_clustering(ctx, golden):
    s = golden["sub"]
    parts = [(0, np.array(s["H0"]), s["H0_first_user"]), (1, np.array(s["H1"]), s["H1_first_user"])]
    users, clusters, counts = pkg().ClusterAssignmentJob(ctx).run_sub(parts, s["numberOfUsers"], s["numberOfClusters"])
    assert users.tolist() == list(range(1, 31))
    assert clusters.tolist() == s["clustering"]
    assert counts.sum() == 30 and len(counts) == 2 * 15


@pytest.mark.parametrize("n,k,device", [(1000, 5, False), (4097, 50, True), (300, 200, False), (1, 1, False), (513, 64, True), (777, 65, True)])
def test_random_vs_oracle(ctx, n, k, device):
    rng = np.random.default_rng(n * 131 + k)
    H = rng.random((n, k))
    H[rng.integers(0, n, n // 10), rng.integers(0, k, n // 10)] = np.nan          # NaN never wins
    if n > 10:
        H[3, :] = H[3, 0]                                                          # a row of ties: first index
        H[5, :] = -np.inf                                                          # nothing exceeds -inf: -1
    want_u, want_c = oracle.cluster_assign(H, first_user=7)
    job = pkg().ClusterAssignmentJob(ctx)
    Hd = torch.from_numpy(H).cuda() if device else H
    users, clusters = job._assign(Hd, 7, 0, np.zeros(0, np.int32))
    assert np.array_equal(users, want_u) and np.array_equal(clusters, want_c)
    if n > 10:
        with pytest.raises(pkg().FilmYouError):      # the -1 row cannot be counted
            job.run(Hd, first_user=7)


def test_chain_into_rm2(ctx, golden, rm_golden):
    """H -> (clustering, clusteringCount) on the GPU -> RM2 job: the reference's 507 triples (tolerance of its own test)"""
    P = pkg()
    users, clusters, counts = P.ClusterAssignmentJob(ctx).run(np.array(golden["H"]), first_user=1)
    g = rm_golden
    conf = P.Configuration()
    conf.set("lambda", "0.5")
    conf.setInt("numberOfItems", 100)
    conf.setInt("numberOfClusters", 5)
    conf.setInt("numberOfRecommendations", 1000)
    rows = P.RM2Job(conf, ctx).run(g["coo"], clustering=(users, clusters), clustering_count=counts).rows()
    want = {(int(u), int(i)): s for u, i, s in g["recommendations"]}
    assert len(rows["user"]) == 507
    for u, i, s in zip(rows["user"].tolist(), rows["item"].tolist(), rows["score"].tolist()):
        assert abs(s - want[(u, i)]) < 1e-4
