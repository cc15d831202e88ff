"""The RM2 definition in fp64 for EVERY emitted row of a full-size job (test infrastructure: torch fp64 on the GPU box's card,
nothing of the product library is involved).

    score(u, i) = (n-1) ln M - n ln U_c + sum_{j in rated(u)} ln( sum_{v in V_c, v != u} c_vi c_vj )        AbstractRM2Reducer.java:321-371
    c_vi        = (1-l) r_vi / s_v + l p_i                                                                    :384-389

The sum over the neighbours is evaluated through the exact algebraic identity (i is not rated by u)

    sum_{v != u} c_vi c_vj = (1-l)^2 (X^T X)_ji + l (1-l) p_j b_i + l p_i e_uj ,   e_uj = (1-l)(b_j - x_uj) + l (U_c - 1) p_j

with X^T X from fp64 sparse x dense products (torch.sparse.mm), every quantity fp64: against the reference's own loop nest the
difference is the rounding of fp64 sums (~1e-14 relative; tests/test_fp64_definition_gpu.py pins this file to the brute-force
oracle on the reference's fixture and to oracle.rm2_gram on whole clusters of the full-size job).  What it is for: the
all-rows measurement of the production precision (24-bit matrix, fp32 logs) against the definition instead of an fp32 twin."""
import numpy as np
import torch


def _dense_index(ids):
    uniq, inv = torch.unique(ids, return_inverse=True)
    return uniq, inv


def fp64_scores(dev_triples, rows, lam, n_items_global, clustering=None, elem_budget=1 << 26, col_chunk=2048, device="cuda:0"):
    """dev_triples = (user, item, score) torch tensors (raw ids); rows = dict(user, item, score, cluster) numpy arrays of a job;
    clustering = (user ids, cluster ids) or None (everybody in cluster 0; unmapped users -> 0, quirk Q2).
    Returns float64 numpy array: the definition's score of every row."""
    dev = torch.device(device)
    f64 = torch.float64
    u, i, s = (t.to(dev) for t in dev_triples)
    keep = s > 0                                                    # SimpleScoreByUserHDFSMapper.java:37-40
    u, i, s = u[keep].long(), i[keep].long(), s[keep].to(f64)
    uid, ux = _dense_index(u)
    iid, ix = _dense_index(i)
    U, I = len(uid), len(iid)
    su = torch.zeros(U, dtype=f64, device=dev).index_add_(0, ux, s)
    T = torch.floor(su).sum()                                       # quirk Q1, DoubleSumAndCountReducer.java:41
    p = torch.zeros(I, dtype=f64, device=dev).index_add_(0, ix, s) / T
    x = s / su[ux]
    cl_u = torch.zeros(U, dtype=torch.long, device=dev)
    if clustering is not None:
        mu = torch.as_tensor(np.asarray(clustering[0]), device=dev).long()
        mc = torch.as_tensor(np.asarray(clustering[1]), device=dev).long()
        pos = torch.searchsorted(uid, mu).clamp_(max=U - 1)
        ok = uid[pos] == mu
        cl_u[pos[ok]] = mc[ok]
    # CSR by user
    order = torch.argsort(ux, stable=True)
    csr_i, csr_x = ix[order], x[order]
    n_u = torch.bincount(ux, minlength=U)
    rowptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(n_u, 0)
    del order

    r_user = torch.as_tensor(rows["user"], device=dev).long()
    r_item = torch.as_tensor(rows["item"], device=dev).long()
    r_ux = torch.searchsorted(uid, r_user)
    r_ix = torch.searchsorted(iid, r_item)
    assert bool((uid[r_ux] == r_user).all()) and bool((iid[r_ix] == r_item).all())
    r_cl = cl_u[r_ux]
    if "cluster" in rows:
        assert bool((torch.as_tensor(rows["cluster"], device=dev).long() == r_cl).all()), "cluster column"
    ref = torch.empty(len(r_user), dtype=f64, device=dev)
    lnM = float(np.log(float(n_items_global)))
    w2, w1 = (1 - lam) ** 2, lam * (1 - lam)

    cl_of_rating = cl_u[ux]
    for c in torch.unique(r_cl).tolist():
        members = torch.nonzero(cl_u == c).ravel()
        Uc = len(members)
        loc = torch.full((U,), -1, dtype=torch.long, device=dev)
        loc[members] = torch.arange(Uc, device=dev)
        m = cl_of_rating == c
        cu, ci, cx = loc[ux[m]], ix[m], x[m]
        b = torch.zeros(I, dtype=f64, device=dev).index_add_(0, ci, cx)
        XT = torch.sparse_coo_tensor(torch.stack([ci, cu]), cx, size=(I, Uc)).coalesce()
        rsel = torch.nonzero(r_cl == c).ravel()
        cols, col_of_row = torch.unique(r_ix[rsel], return_inverse=True)        # the candidate columns the rows name
        n_row = n_u[r_ux[rsel]]
        # rows in groups of columns: G[:, chunk] = X^T X[:, chunk]
        for c0 in range(0, len(cols), col_chunk):
            cc = cols[c0:c0 + col_chunk]
            w = len(cc)
            colpos = torch.full((I,), -1, dtype=torch.long, device=dev)
            colpos[cc] = torch.arange(w, device=dev)
            inchunk = colpos[ci] >= 0
            Xd = torch.zeros(Uc, w, dtype=f64, device=dev)
            Xd[cu[inchunk], colpos[ci[inchunk]]] = cx[inchunk]
            G = torch.sparse.mm(XT, Xd)                                             # I x w, fp64
            del Xd
            rr = torch.nonzero((col_of_row >= c0) & (col_of_row < c0 + w)).ravel()
            if len(rr) == 0:
                continue
            cum = torch.cumsum(n_row[rr], 0)
            a = 0
            while a < len(rr):
                base = int(cum[a - 1]) if a else 0
                z = int(torch.searchsorted(cum, torch.tensor(base + elem_budget, device=dev), right=True))
                z = max(z, a + 1)
                part = rr[a:z]
                g_rows = rsel[part]
                uu_ = r_ux[g_rows]
                cnt = n_row[part]
                rep = torch.repeat_interleave(torch.arange(len(part), device=dev), cnt)
                starts = torch.cumsum(cnt, 0) - cnt
                off = torch.arange(len(rep), device=dev) - starts[rep]
                e_pos = rowptr[uu_][rep] + off
                j = csr_i[e_pos]
                xj = csr_x[e_pos]
                icol = r_ix[g_rows][rep]
                g = G[j, (col_of_row[part] - c0)[rep]]
                e = (1 - lam) * (b[j] - xj) + lam * (Uc - 1) * p[j]
                term = w2 * g + w1 * p[j] * b[icol] + lam * p[icol] * e
                logsum = torch.zeros(len(part), dtype=f64, device=dev).index_add_(0, rep, torch.log(term))
                nn = cnt.to(f64)
                ref[g_rows] = (nn - 1) * lnM - nn * float(np.log(float(Uc))) + logsum
                a = z
            del G
        del XT
    return ref.cpu().numpy()


def compare_with_definition(rows, ref, rtol=1e-5):
    """The reference emits (float) score (RM2HDFSReducer.java:48): the GPU's float32 rows against float32(definition).
    Returns dict(worst, n_over, worst_rows)."""
    got = rows["score"].astype(np.float64)
    want = ref.astype(np.float32).astype(np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.array_equal(got[~fin], want[~fin])
    rel = np.zeros(len(got))
    rel[fin] = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    order = np.argsort(-rel)[:5]
    return {"worst": float(rel.max()) if len(rel) else 0.0, "n_over": int((rel > rtol).sum()), "rows": int(len(rel)),
            "p9999": float(np.quantile(rel, 0.9999)) if len(rel) else 0.0,
            "worst_rows": [(int(rows["user"][k]), int(rows["item"][k]), float(got[k]), float(ref[k]), float(rel[k])) for k in order]}
