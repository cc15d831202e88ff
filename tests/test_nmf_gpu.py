"""NMF / PPC factorisation (fy_nmf_factorize) against the reference's own vectors (NMFTestData / PPCTestData: one and ten
iterations from W_init / H_init, asserted by the reference with accuracy 1e-4) and against the oracle on random sparse data,
including the L1-normalisation branch the reference's vectors never reach; then the whole chain factorise -> assign."""
import json
import os

import numpy as np
import pytest

import oracle
from util import pkg

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "factorization_test_data.json")


@pytest.fixture(scope="module")
def ctx():
    c = pkg().Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def coo(A):
    A = np.array(A)
    i, u = np.nonzero(A > 0)
    return (u + 1).astype(np.int32), (i + 1).astype(np.int32), A[i, u].astype(np.float32)


def conf_for(P, n_users, n_items, k, iterations, norm=None):
    conf = P.Configuration()
    conf.setInt("numberOfUsers", n_users)
    conf.setInt("numberOfItems", n_items)
    conf.setInt("numberOfClusters", k)
    conf.setInt("numberOfIterations", iterations)
    if norm is not None:
        conf.setInt("normalizationFrequency", norm)
    return conf


@pytest.mark.parametrize("name,ppc", [("nmf", False), ("ppc", True)])
@pytest.mark.parametrize("iterations,suffix", [(1, "one"), (10, "ten")])
def test_reference_vectors(ctx, golden, name, ppc, iterations, suffix):
    P = pkg()
    d = golden[name]
    H, W = P.NMFDriver(conf_for(P, 30, 100, 10, iterations, norm=12), ctx, ppc=ppc).run(coo(d["A"]), d["H_init"], d["W_init"])
    # the reference asserts 1e-4 (HadoopIntegrationTest.accuracy); fp64 on both sides agrees far better
    np.testing.assert_allclose(H, np.array(d["H_" + suffix]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(W, np.array(d["W_" + suffix]), rtol=0, atol=1e-9)


def test_ppc_toy(ctx, golden):
    P = pkg()
    d = golden["ppc"]
    H, _ = P.NMFDriver(conf_for(P, 7, 5, 2, 1, norm=12), ctx, ppc=True).run(coo(d["Ap"]), d["h0p"], d["w0p"])
    np.testing.assert_allclose(H, np.array(d["h1p"]), rtol=0, atol=1e-7)     # Ap goes through FloatWritable: 3 decimals as fp32


@pytest.mark.parametrize("n_users,n_items,k,ppc,norm,iters", [(300, 200, 7, False, 0, 3), (500, 120, 50, True, -1, 4),
                                                              (257, 333, 65, True, 2, 4), (64, 64, 200, False, 0, 2)])
def test_random_vs_oracle(ctx, n_users, n_items, k, ppc, norm, iters):
    P = pkg()
    rng = np.random.default_rng(n_users + 7 * k)
    A = (rng.random((n_items, n_users)) < 0.15) * rng.integers(1, 6, (n_items, n_users))
    A[rng.integers(0, n_items, n_users), np.arange(n_users)] = 3          # every user rates something
    A[np.arange(n_items), rng.integers(0, n_users, n_items)] = 4          # every item is rated
    u, i, s = coo(A)
    s = np.r_[s, [0.0, -1.0]].astype(np.float32)                            # dropped by score > 0
    u, i = np.r_[u, [1, 2]].astype(np.int32), np.r_[i, [1, 1]].astype(np.int32)
    H0, W0 = rng.random((n_users, k)) + 0.01, rng.random((n_items, k)) + 0.01
    H, W = P.NMFDriver(conf_for(P, n_users, n_items, k, iters, norm=norm), ctx, ppc=ppc).run((u, i, s), H0, W0)
    Ho, Wo = oracle.nmf(u, i, s, H0, W0, iterations=iters, ppc=ppc, normalization_frequency=norm)
    np.testing.assert_allclose(H, Ho, rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(W, Wo, rtol=1e-10, atol=1e-300)
    if ppc and norm == -1:
        np.testing.assert_allclose(np.abs(H).sum(1), 1.0, rtol=1e-12)       # normalised every iteration


def test_errors_mirror_the_reference(ctx):
    P = pkg()
    u, i, s = np.array([1, 2], np.int32), np.array([1, 1], np.int32), np.array([3, 4], np.float32)
    with pytest.raises(RuntimeError, match="User 3 has not rated any item"):
        P.NMFDriver(conf_for(P, 3, 1, 2, 1), ctx).run((u, i, s), np.ones((3, 2)), np.ones((1, 2)))
    with pytest.raises(RuntimeError, match="Item 2 has not been rated by anybody"):
        P.NMFDriver(conf_for(P, 2, 2, 2, 1), ctx).run((u, i, s), np.ones((2, 2)), np.ones((2, 2)))
    with pytest.raises(RuntimeError, match="outside"):
        P.NMFDriver(conf_for(P, 1, 1, 2, 1), ctx).run((u, i, s), np.ones((1, 2)), np.ones((1, 2)))


def test_chain_factorise_then_assign(ctx, golden):
    """PPC ten iterations -> cluster assignment: the argmax of the reference's H_ten, row by row"""
    P = pkg()
    d = golden["ppc"]
    H, _ = P.NMFDriver(conf_for(P, 30, 100, 10, 10, norm=12), ctx, ppc=True).run(coo(d["A"]), d["H_init"], d["W_init"])
    users, clusters, counts = P.ClusterAssignmentJob(ctx).run(H, first_user=1)
    assert clusters.tolist() == np.array(d["H_ten"]).argmax(1).tolist()
    assert counts.sum() == 30
