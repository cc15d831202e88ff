"""Item-item similarity build on the GPU vs the CPU oracle.

PARITY UNPINNED against the reference (no golden vector exists for this path: its arithmetic is Mahout 0.8's, see
oracle/itemsim_oracle.c); the HIP path and the oracle implement the same published definition and must agree to
float32 rounding of the similarity (the oracle works in fp64, the library accumulates in fp64 and emits float32).
"""
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


def check(rows, ref_full, K, exclude_self=True):
    full = {}
    for a, b, s in zip(ref_full["item"], ref_full["other"], ref_full["sim"]):
        full.setdefault(int(a), []).append((int(b), float(s)))
    got = {}
    for a, b, s in zip(rows["item"], rows["other"], rows["sim"]):
        got.setdefault(int(a), []).append((int(b), float(s)))
    assert set(got) == {a for a, v in full.items() if v}
    for a, lst in got.items():
        ref = full[a]
        k = min(K, len(ref))
        assert len(lst) == k, (a, len(lst), k)
        lookup = dict(ref)
        sims = np.array([s for _, s in lst])
        want = np.array([lookup[b] for b, _ in lst])
        assert len({b for b, _ in lst}) == k and (not exclude_self or all(b != a for b, _ in lst))
        np.testing.assert_allclose(sims, want, rtol=RTOL)
        assert np.all(sims[:-1] >= sims[1:])
        best = np.array([s for _, s in ref[:k]])
        np.testing.assert_allclose(sims, best, rtol=RTOL)          # nothing better was left out


@pytest.mark.parametrize("similarity,K,threshold", [("SIMILARITY_COSINE", 10, None), ("SIMILARITY_COOCCURRENCE", 5, None),
                                                     ("SIMILARITY_COSINE", 100, 0.15)])
def test_reference_matrix(ctx, rm_golden, similarity, K, threshold):
    user, item, score = rm_golden["coo"]
    keep = score > 0
    user, item, score = user[keep], item[keep], score[keep]
    res = pkg().RowSimilarityJob(ctx).run((user, item, score), similarityClassname=similarity, maxSimilaritiesPerRow=K,
                                          threshold=threshold)
    sim_id = oracle.COSINE if similarity == "SIMILARITY_COSINE" else oracle.COOCCURRENCE
    ref = oracle.itemsim(user, item, score, similarity=sim_id, max_similarities_per_item=1 << 30, threshold=threshold)
    check(res.rows(), ref, K)
    n = np.bincount(user)
    assert res.stats["unordered_pairs"] == int((n * (n - 1) // 2).sum()) == ref["pairs"]


@pytest.mark.parametrize("shape,K", [("tiny", 20), ("ml100k", 100)])
def test_synthetic(ctx, shape, K):
    u, i, s, _ = synth().generate(shape)
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    res = pkg().RowSimilarityJob(ctx).run((u, i, s), maxSimilaritiesPerRow=K)
    ref = oracle.itemsim(u, i, s, max_similarities_per_item=1 << 30, n_threads=8)
    check(res.rows(), ref, K)


@pytest.mark.parametrize("env", [{"FY_COOC_MAX_CH": "256", "FY_ISIM_HEAVY": "8"},      # column chunks, heavy rows split by chunk + merge
                                 {"FY_COOC_MAX_CH": "256", "FY_ISIM_HEAVY": "1000000"},  # column chunks, threshold carried along
                                 {"FY_COOC_PK": "0"}])                                   # 8-byte CSR entries, weights pre-divided
def test_synthetic_forced_paths(ctx, monkeypatch, env):
    """the paths that production sizes switch on (several column chunks, split rows) and off (unpacked CSR), on ML-100K shape"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    u, i, s, _ = synth().generate("ml100k")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    res = pkg().RowSimilarityJob(ctx).run((u, i, s), maxSimilaritiesPerRow=30)
    ref = oracle.itemsim(u, i, s, max_similarities_per_item=1 << 30, n_threads=8)
    check(res.rows(), ref, 30)


SYMMETRIC = [{"FY_ISIM_GRAM": "1"},                                                        # one chunk of the walk, one piece per band
             {"FY_ISIM_GRAM": "1", "FY_COOC_MAX_CH": "256", "FY_ISIM_PIECE": "128"},      # several chunks, several pieces per band
             {"FY_ISIM_GRAM": "1", "FY_ISIM_CAPG": "40"},                                 # candidate lists overflow: rows redone exactly
             {"FY_ISIM_GRAM": "1", "FY_ISIM_ACC32": "0", "FY_COOC_MAX_CH": "512"}]        # 64-bit fixed-point accumulators (32-bit is the default where exact)


@pytest.mark.parametrize("env", SYMMETRIC)
@pytest.mark.parametrize("exclude_self,threshold", [(True, None), (False, 0.2)])
def test_symmetric_build_forced(ctx, monkeypatch, env, exclude_self, threshold):
    """the build the benchmark sizes take (upper triangle by the RM2 row kernel + band sweep, fy_itemsim.hip), forced onto
    ML-100K shape: must give what the row-at-a-time build and the oracle give"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    u, i, s, _ = synth().generate("ml100k")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    res = pkg().RowSimilarityJob(ctx).run((u, i, s), maxSimilaritiesPerRow=30, excludeSelfSimilarity=exclude_self, threshold=threshold)
    ref = oracle.itemsim(u, i, s, max_similarities_per_item=1 << 30, n_threads=8, exclude_self=exclude_self, threshold=threshold)
    check(res.rows(), ref, 30, exclude_self=exclude_self)
    assert res.stats["cooc_launches"] == 1 and res.stats["isim_candidates"] > 0
    assert (res.stats["isim_redone_rows"] > 0) == ("FY_ISIM_CAPG" in env)


def test_symmetric_build_reference_matrix(ctx, rm_golden, monkeypatch):
    monkeypatch.setenv("FY_ISIM_GRAM", "1")
    user, item, score = rm_golden["coo"]
    keep = score > 0
    user, item, score = user[keep], item[keep], score[keep]
    res = pkg().RowSimilarityJob(ctx).run((user, item, score), maxSimilaritiesPerRow=10)
    ref = oracle.itemsim(user, item, score, max_similarities_per_item=1 << 30)
    check(res.rows(), ref, 10)


def test_item_row_shards_partition_the_result(ctx):
    u, i, s, _ = synth().generate("tiny")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    job = pkg().RowSimilarityJob(ctx)
    whole = job.run((u, i, s), maxSimilaritiesPerRow=10).rows()
    parts = [job.run((u, i, s), maxSimilaritiesPerRow=10, rank=r, world=3).rows() for r in range(3)]
    key = lambda rows: sorted(zip(rows["item"].tolist(), rows["other"].tolist(), rows["sim"].tolist()))
    merged = {k: np.concatenate([p[k] for p in parts]) for k in ("item", "other", "sim")}
    assert key(merged) == key(whole)
    assert len({int(x) for p in parts for x in np.unique(p["item"])}) == len(np.unique(whole["item"]))


def test_bad_arguments(ctx):
    job = pkg().RowSimilarityJob(ctx)
    u = np.array([1, 2], dtype=np.int32)
    with pytest.raises(ValueError):
        job.run((u, u, u.astype(np.float32)), similarityClassname="class org.apache.mahout...CooccurrenceCountSimilarity")
    with pytest.raises(RuntimeError, match="RowSimilarityJob failed!"):
        job.run((u, u, u.astype(np.float32)), maxSimilaritiesPerRow=0)


@pytest.mark.parametrize("min_prefs,max_prefs", [(60, None), (1, 40), (80, 120)])
def test_input_preparation_options(ctx, min_prefs, max_prefs):
    """minPrefsPerUser (users below it are dropped) and the cap of maxPrefsPerUserInItemSimilarity
    (BaselinePreparePreferenceMatrixJob.java:104, 126-129).  The cap is a DETERMINISTIC systematic sample here (Mahout samples at
    random: no parity with a particular Mahout run is claimed), the same rule in the library and in the oracle."""
    u, i, s, _ = synth().generate("ml100k")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    res = pkg().RowSimilarityJob(ctx).run((u, i, s), maxSimilaritiesPerRow=25, minPrefsPerUser=min_prefs, maxPrefsPerUser=max_prefs)
    ref = oracle.itemsim(u, i, s, max_similarities_per_item=1 << 30, n_threads=8, min_prefs_per_user=min_prefs,
                         max_prefs_per_user=max_prefs or 0)
    check(res.rows(), ref, 25)
    deg = np.bincount(u)
    deg = deg[deg > 0]
    kept = deg[deg >= min_prefs]
    if max_prefs:
        kept = np.minimum(kept, max_prefs)
    assert res.stats["nnz"] == int(kept.sum()) < len(u)               # the options really removed preferences
    assert res.stats["unordered_pairs"] == int((kept.astype(np.int64) * (kept - 1) // 2).sum()) == ref["pairs"]
