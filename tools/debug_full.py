import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, scipy.sparse as sp
from fullsize_checks import load_shape, run_rm2
LAM, TOPN = 0.1, 50
data = load_shape("ml25m")
env2 = {"FY_PRUNE": "0"}
if len(sys.argv) > 1: env2 = dict(kv.split("=") for kv in sys.argv[1].split(","))
a, _, sta = run_rm2(data, TOPN, LAM)
b, _, stb = run_rm2(data, TOPN, LAM, env=env2)
print("run a stats", {k: sta[k] for k in ("ms_cooc", "ms_score", "blocks_survived")}, "run b env", env2, {k: stb[k] for k in ("ms_cooc", "ms_score")})
assert np.array_equal(a["user"], b["user"])
big = int(max(a["item"].max(), b["item"].max())) + 1
ka = a["user"].astype(np.int64) * big + a["item"]; kb = b["user"].astype(np.int64) * big + b["item"]
oa, ob = np.argsort(ka), np.argsort(kb)
common_a = np.isin(ka[oa], kb[ob], assume_unique=True); common_b = np.isin(kb[ob], ka[oa], assume_unique=True)
ia, ib = oa[common_a], ob[common_b]
sa, sb = a["score"][ia].astype(np.float64), b["score"][ib].astype(np.float64)
rel = np.abs(sa - sb) / np.abs(sb)
print("common", len(ia), "only-a", (~common_a).sum(), "rel diff quantiles", np.quantile(rel, [0.5, 0.9, 0.99, 0.999, 0.9999, 1.0]))
worst = np.argsort(-rel)[:12]
R, uu, iu = data["R"], data["uu"], data["iu"]
U, I = R.shape
su = np.asarray(R.sum(1)).ravel(); T = np.floor(su).sum(); p = np.asarray(R.sum(0)).ravel() / T
X = sp.diags(1.0 / su) @ R; Xc = X.tocsc(); XT = X.T.tocsr(); bvec = np.asarray(X.sum(0)).ravel()
w2, w1 = (1 - LAM) ** 2, LAM * (1 - LAM)
cnt = np.diff(Xc.indptr); rank_of = np.empty(I, int); rank_of[np.argsort(-cnt, kind="stable")] = np.arange(I)
for w in worst:
    u_raw, i_raw = int(a["user"][ia[w]]), int(a["item"][ia[w]])
    ux, ix = int(np.searchsorted(uu, u_raw)), int(np.searchsorted(iu, i_raw))
    J = X.indices[X.indptr[ux]:X.indptr[ux + 1]]; x = X.data[X.indptr[ux]:X.indptr[ux + 1]]; n = len(J)
    e = (1 - LAM) * (bvec[J] - x) + LAM * (U - 1) * p[J]
    pvpi = (n - 1) * np.log(data["facts"]["n_items"]) - n * np.log(U)
    g = np.asarray((XT @ Xc[:, ix]).todense()).ravel()[J]
    ex = pvpi + np.log(w2 * g + w1 * p[J] * bvec[ix] + LAM * p[ix] * e).sum()
    print("user %d (n=%d) item %d (pop rank %d, raters %d): a=%.6f b=%.6f exact=%.6f  err a %.2e  err b %.2e" % (
        u_raw, n, i_raw, rank_of[ix], cnt[ix], sa[w], sb[w], ex, abs(sa[w] - ex) / abs(ex), abs(sb[w] - ex) / abs(ex)))
