"""Item-based CF recommendation phases (partialMultiply + aggregateAndRecommend) on the GPU vs the CPU oracle.

PARITY UNPINNED against the reference: the baselinerecommender package has no test and is excluded from compilation
(pom.xml:81-83); the reducer's arithmetic is followed from the tree (BaselineAggregateAndRecommendReducer.java:97-161,
195-235), the Mahout mappers in front of it are restated from the published algorithm (oracle/itemcf_oracle.c)."""
import numpy as np
import pytest

import oracle
from util import pkg, synth

pytestmark = pytest.mark.gpu
RTOL = 2e-6


@pytest.fixture(scope="module")
def ctx():
    c = pkg().Context(0)
    yield c
    c.close()


def check(rows, ref_full, N):
    """Tie-tolerant: predictions of item-based CF tie often (all contributing preferences equal)."""
    full, got = {}, {}
    for u, i, s in zip(ref_full["user"], ref_full["item"], ref_full["score"]):
        full.setdefault(int(u), []).append((int(i), float(s)))
    for u, i, s in zip(rows["user"], rows["item"], rows["score"]):
        got.setdefault(int(u), []).append((int(i), float(s)))
    assert set(got) == {u for u, v in full.items() if v}
    for u, lst in got.items():
        ref = full[u]
        k = min(N, len(ref))
        assert len(lst) == k, (u, len(lst), k)
        lookup = dict(ref)
        sc = np.array([s for _, s in lst])
        want = np.array([lookup[i] for i, _ in lst])           # KeyError: an item the oracle would never recommend
        np.testing.assert_allclose(sc, want, rtol=RTOL)
        assert len({i for i, _ in lst}) == k
        assert np.all(sc[:-1] >= sc[1:])
        np.testing.assert_allclose(sc, np.array([s for _, s in ref[:k]]), rtol=RTOL)   # nothing better left out


def run_both(ctx, u, i, s, N, max_prefs, K, boolean=False):
    P = pkg()
    rec, sims = P.BaselineRecommenderJob(ctx).run((u, i, s), numRecommendations=N, maxPrefsPerUser=max_prefs,
                                                  maxSimilaritiesPerItem=K, booleanData=boolean)
    srows = sims.rows()
    # the oracle consumes the GPU's own similarity rows (float32 values): this test isolates phases 3-4
    ref = oracle.itemcf(u, i, s, srows["item"], srows["other"], srows["sim"].astype(np.float64), num_recommendations=1 << 30,
                        max_prefs_per_user=max_prefs, boolean_data=boolean)
    return rec.rows(), ref


@pytest.mark.parametrize("N,max_prefs,K,boolean", [(10, 50, 10, False), (5, 8, 20, False), (10, 50, 10, True)])
def test_reference_matrix(ctx, rm_golden, N, max_prefs, K, boolean):
    u, i, s = rm_golden["coo"]
    keep = s > 0
    rows, ref = run_both(ctx, u[keep], i[keep], s[keep], N, max_prefs, K, boolean)
    check(rows, ref, N)
    # items of the considered preferences never come back
    rated = set(zip(u[keep].tolist(), i[keep].tolist()))
    if max_prefs >= 100:
        assert not rated & set(zip(rows["user"].tolist(), rows["item"].tolist()))


@pytest.mark.parametrize("shape,N,max_prefs,K", [("tiny", 10, 10, 15), ("ml100k", 20, 50, 100)])
def test_synthetic(ctx, shape, N, max_prefs, K):
    u, i, s, _ = synth().generate(shape)
    rows, ref = run_both(ctx, u.numpy(), i.numpy(), s.numpy(), N, max_prefs, K)
    check(rows, ref, N)


def test_user_shards_partition_the_result(ctx):
    P = pkg()
    u, i, s, _ = synth().generate("tiny")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    job = P.BaselineRecommenderJob(ctx)
    whole, sims = job.run((u, i, s), numRecommendations=7, maxSimilaritiesPerItem=12)
    parts = [job.run((u, i, s), numRecommendations=7, rank=r, world=3, similarities=sims)[0].rows() for r in range(3)]
    key = lambda rows: sorted(zip(rows["user"].tolist(), rows["item"].tolist(), rows["score"].tolist()))
    merged = {k: np.concatenate([p[k] for p in parts]) for k in ("user", "item", "score")}
    assert key(merged) == key(whole.rows())


def test_zero_numerators_are_not_candidates(ctx):
    """The reducer iterates numerators.nonZeroes() / recommendationVector.nonZeroes()
    (BaselineAggregateAndRecommendReducer.java:148, 195): a candidate whose numerator is exactly 0 -- here: every contributing
    preference has the value 0.0 -- never reaches the top-N queue.  A third of the ratings are set to 0."""
    u, i, s, _ = synth().generate("tiny")
    u, i, s = u.numpy(), i.numpy(), s.numpy().copy()
    s[::3] = 0.0
    rows, ref = run_both(ctx, u, i, s, 10, 10, 15)
    check(rows, ref, 10)
    assert np.all(rows["score"] != 0.0) and np.all(ref["score"] != 0.0)
    # and they really occur in this data: with the old rule (count > 1 only) the oracle had more candidates
    P = pkg()
    rec, sims = P.BaselineRecommenderJob(ctx).run((u, i, s), numRecommendations=300, maxPrefsPerUser=10, maxSimilaritiesPerItem=15)
    n_all = len(rec.rows()["user"])
    s2 = np.where(s == 0.0, 1e-30, s).astype(np.float32)                 # the same structure without exact zeros
    rec2, _ = P.BaselineRecommenderJob(ctx).run((u, i, s2), numRecommendations=300, maxPrefsPerUser=10, maxSimilaritiesPerItem=15)
    assert len(rec2.rows()["user"]) > n_all
