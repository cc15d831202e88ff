"""Item-item similarity oracle vs an independent scipy.sparse statement of the same definition.

PARITY UNPINNED: the reference has no test, fixture or golden vector for this path (its arithmetic is Mahout 0.8's
RowSimilarityJob, called at baselinerecommender/BaselineRecommenderJob.java:241-253); two independent statements of
the published algorithm are cross-checked instead.
"""
import numpy as np
import scipy.sparse as sp

import oracle


def scipy_itemsim(user, item, score, cosine=True, k=100, threshold=None):
    uu, ui = np.unique(user, return_inverse=True)
    iu, ii = np.unique(item, return_inverse=True)
    X = sp.csr_matrix((score.astype(np.float64), (ui, ii)), shape=(len(uu), len(iu)))
    if cosine:
        norms = np.sqrt(np.asarray(X.multiply(X).sum(axis=0)).ravel())
        X = X @ sp.diags(1.0 / norms)
    else:
        X.data[:] = 1.0
    S = (X.T @ X).toarray()
    np.fill_diagonal(S, 0.0)
    rows = []
    for a in range(len(iu)):
        js = np.nonzero(S[a] > 0 if threshold is None else S[a] >= threshold)[0]
        js = js[js != a]
        order = sorted(js, key=lambda j: (-S[a, j], iu[j]))[:k]
        rows += [(iu[a], iu[j], S[a, j]) for j in order]
    return rows


def test_cosine_and_cooccurrence_on_reference_matrix(rm_golden):
    user, item, score = rm_golden["coo"]
    keep = score > 0
    user, item, score = user[keep], item[keep], score[keep]
    for sim, cosine in ((oracle.COSINE, True), (oracle.COOCCURRENCE, False)):
        r = oracle.itemsim(user, item, score, similarity=sim, max_similarities_per_item=10)
        exp = scipy_itemsim(user, item, score, cosine=cosine, k=10)
        assert len(exp) == len(r["item"]) == 1000
        for (a, b, s), ga, gb, gs in zip(exp, r["item"], r["other"], r["sim"]):
            assert a == ga
            assert abs(s - gs) <= 1e-12 * max(1.0, abs(s))
            if b != gb:   # only on an exact tie may the order differ
                assert abs(s - gs) < 1e-12
    n = np.bincount(user)
    assert r["pairs"] == int((n * (n - 1) // 2).sum())


def test_threshold_and_sparse_random():
    rng = np.random.default_rng(7)
    U, I = 200, 80
    mask = rng.random((U, I)) < 0.08
    u, i = np.nonzero(mask)
    s = rng.integers(1, 11, size=len(u)).astype(np.float32) / 2
    r = oracle.itemsim(u + 1, i + 1, s, max_similarities_per_item=5, threshold=0.2, n_threads=3)
    exp = scipy_itemsim(u + 1, i + 1, s, cosine=True, k=5, threshold=0.2)
    assert len(exp) == len(r["item"])
    assert np.all(r["sim"] >= 0.2)
    np.testing.assert_allclose(r["sim"], [e[2] for e in exp], rtol=1e-12)


def test_input_preparation_of_the_oracle():
    """minPrefsPerUser drops users, the cap keeps exactly max preferences of a heavy user, evenly spread over its items."""
    user = np.array([1] * 2 + [2] * 10 + [3] * 5, dtype=np.int32)
    item = np.concatenate([np.arange(1, 3), np.arange(1, 11), np.arange(3, 8)]).astype(np.int32)
    score = np.ones(len(user), dtype=np.float32)
    full = oracle.itemsim(user, item, score, similarity=oracle.COOCCURRENCE, max_similarities_per_item=100)
    assert full["pairs"] == 1 + 45 + 10
    no_small = oracle.itemsim(user, item, score, similarity=oracle.COOCCURRENCE, max_similarities_per_item=100, min_prefs_per_user=3)
    assert no_small["pairs"] == 45 + 10                                  # user 1 (2 preferences) is gone
    capped = oracle.itemsim(user, item, score, similarity=oracle.COOCCURRENCE, max_similarities_per_item=100, max_prefs_per_user=5)
    assert capped["pairs"] == 1 + 10 + 10                                # user 2 keeps 5 of its 10: items 2, 4, 6, 8, 10
    pairs = {(int(a), int(b)): float(c) for a, b, c in zip(capped["item"], capped["other"], capped["sim"])}
    assert pairs[(2, 4)] == 1.0 and (1, 3) not in pairs and pairs[(4, 6)] == 2.0      # (4, 6): users 2 and 3
