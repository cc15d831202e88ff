"""Deterministic MovieLens / Netflix-shaped synthetic ratings (there is no network for the real files).

The same integer hash decides every (user, item) cell on the CPU and on the GPU, so every rank of a multi-GPU run
and every test sees identical data:
  * user degrees ~ log-normal(sigma = 1.1), clipped to [20, 0.5 * I], rescaled to the target nnz;
  * item popularity Zipf(alpha = 1.0) over a fixed pseudo-random permutation of the item ids;
  * user u rates item i iff hash32(seed, u, i) < floor(min(P_CAP, t_u * w_i) * 2^32), t_u solved so that the expected
    row length is the user's degree (Poisson sampling without replacement) -- nnz lands within ~0.1 % of the target;
    P_CAP = 0.6 keeps the head realistic (the most popular MovieLens-25M title is rated by about half of the users);
  * values: categorical with the marginals of SURVEY.md section 8d (integers 1..5, or half stars 0.5..5.0).
Ids are 1-based and the triples are returned in a hash-shuffled order (COO, like a table scan).
"""
import math

import numpy as np
import torch

SHAPES = {
    # name: (users, items, nnz, half_stars)   -- BASELINE.json configs C0..C4
    "ml100k": (943, 1682, 100_000, False),
    "ml1m": (6040, 3706, 1_000_209, False),
    "ml25m": (162_541, 59_047, 25_000_095, True),
    "netflix": (480_189, 17_770, 100_480_507, False),
    "tiny": (200, 300, 6_000, True),
}
SEED0 = 20260104
P_CAP = 0.6

_M1 = -4658895280553007687      # 0xbf58476d1ce4e5b9 as int64
_M2 = -7723592293110705685      # 0x94d049bb133111eb
_G = -7046029254386353131       # 0x9E3779B97F4A7C15
_H = -3335678366873096957       # 0xD1B54A32D192ED03


def _lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix(x):
    x = x ^ _lsr(x, 30)
    x = x * _M1
    x = x ^ _lsr(x, 27)
    x = x * _M2
    x = x ^ _lsr(x, 31)
    return x


def _hash32(seed, u, i, stream):
    """u, i int64 tensors (broadcastable) -> int64 tensor in [0, 2^32)."""
    a = _mix(u * _G + seed)
    b = _mix(a ^ (i * _H + stream))
    return _lsr(b, 32)


def _degrees(n_users, n_items, nnz, rng):
    d = np.exp(rng.normal(0.0, 1.1, size=n_users))
    lo, hi = 20.0, 0.5 * n_items
    scale = nnz / d.sum()
    for _ in range(60):   # rescale under the clip until the total matches
        dd = np.clip(d * scale, lo, hi)
        scale *= nnz / dd.sum()
    return np.clip(d * scale, lo, hi)


def _solve_t(deg, w_sorted_desc):
    """t with sum_i min(P_CAP, t * w_i) = deg, per user (bisection on the prefix sums of the sorted weights)."""
    n = len(w_sorted_desc)
    suffix = np.concatenate([np.cumsum(w_sorted_desc[::-1])[::-1], [0.0]])   # suffix[k] = sum_{i >= k} w_i

    def total(t):
        k = np.searchsorted(-w_sorted_desc, -P_CAP / t, side="right")        # items with t*w >= P_CAP are saturated
        return P_CAP * k + t * suffix[k]

    lo = np.full(len(deg), 1e-6)
    hi = np.full(len(deg), 1.0 / w_sorted_desc[-1] * 2)
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        big = total(mid) > deg
        hi = np.where(big, mid, hi)
        lo = np.where(big, lo, mid)
    assert n > 0
    return 0.5 * (lo + hi)


def generate(shape="ml100k", seed_offset=0, device="cpu", users=None, batch_cells=1 << 27):
    """Returns (user, item, score) int32/int32/float32 torch tensors on `device`, plus a dict of shape facts.
    `users` (optional 1-based id array) restricts generation to those users (same cells as the full data set)."""
    n_users, n_items, nnz, half = SHAPES[shape] if isinstance(shape, str) else shape
    seed = SEED0 + seed_offset
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = _degrees(n_users, n_items, nnz, rng)
    perm = rng.permutation(n_items)
    w = np.empty(n_items)
    w[perm] = 1.0 / np.arange(1, n_items + 1)
    w /= w.sum()
    t = _solve_t(deg, np.sort(w)[::-1])
    if half:
        vals = np.arange(1, 11) * 0.5
        pr = np.array([1.6, 3.1, 1.6, 6.6, 5.0, 19.6, 12.7, 26.6, 8.8, 14.4])
    else:
        vals = np.arange(1, 6) * 1.0
        pr = np.array([6.0, 11.0, 27.0, 35.0, 21.0])
    cdf = np.floor(np.cumsum(pr / pr.sum()) * 4294967296.0)
    cdf[-1] = 4294967296.0

    dev = torch.device(device)
    w_t = torch.from_numpy(w).to(dev)
    t_all = torch.from_numpy(t).to(dev)
    cdf_t = torch.from_numpy(cdf).to(dev)
    vals_t = torch.from_numpy(vals.astype(np.float32)).to(dev)
    items = torch.arange(1, n_items + 1, dtype=torch.int64, device=dev)
    uids = torch.arange(1, n_users + 1, dtype=torch.int64, device=dev) if users is None else \
        torch.as_tensor(np.asarray(users), dtype=torch.int64, device=dev)
    rows = max(1, batch_cells // n_items)
    out_u, out_i, out_s = [], [], []
    for a in range(0, len(uids), rows):
        u = uids[a:a + rows]
        thr = torch.clamp(t_all[u - 1][:, None] * w_t[None, :], max=P_CAP) * 4294967296.0
        h = _hash32(seed, u[:, None], items[None, :], 1)
        keep = h.to(torch.float64) < torch.floor(thr)
        ru, ri = torch.nonzero(keep, as_tuple=True)
        uu, ii = u[ru], items[ri]
        hv = _hash32(seed, uu, ii, 2).to(torch.float64)
        cat = torch.searchsorted(cdf_t, hv, right=True).clamp_(max=len(vals) - 1)
        out_u.append(uu.to(torch.int32))
        out_i.append(ii.to(torch.int32))
        out_s.append(vals_t[cat])
    user = torch.cat(out_u)
    item = torch.cat(out_i)
    score = torch.cat(out_s)
    # table-scan order: shuffle by a hash of the cell
    order = torch.argsort(_hash32(seed, user.to(torch.int64), item.to(torch.int64), 3) * 65536
                          + (user.to(torch.int64) & 0xFFFF))
    facts = {"shape": shape if isinstance(shape, str) else "custom", "n_users": n_users, "n_items": n_items,
             "target_nnz": nnz, "nnz": int(user.numel()), "half_stars": bool(half), "seed": seed}
    return user[order].contiguous(), item[order].contiguous(), score[order].contiguous(), facts


def hash_clustering(user_ids, n_clusters, seed_offset=0):
    """Users assigned to clusters by a fixed hash (stand-in for the PPC clustering stage the job consumes)."""
    u = torch.as_tensor(np.asarray(user_ids), dtype=torch.int64)
    c = _hash32(SEED0 + seed_offset, u, torch.zeros_like(u), 7) % n_clusters
    return c.to(torch.int32).numpy()


def term_count(user, item, n_clusters=1, cluster_of_user=None):
    """Exact number of RM2 log terms sum_u n_u * (I_c - n_u) for a COO data set (numpy, host)."""
    user = np.asarray(user)
    item = np.asarray(item)
    uu, ui = np.unique(user, return_inverse=True)
    n_u = np.bincount(ui)
    if cluster_of_user is None:
        return int((n_u.astype(np.int64) * (len(np.unique(item)) - n_u)).sum())
    cl = np.asarray(cluster_of_user)[ui]
    total = 0
    for c in range(n_clusters):
        m = cl == c
        if not m.any():
            continue
        Ic = len(np.unique(item[m]))
        nu = np.bincount(ui[m], minlength=len(uu))
        nu = nu[nu > 0].astype(np.int64)
        total += int((nu * (Ic - nu)).sum())
    return total


__all__ = ["SHAPES", "generate", "hash_clustering", "term_count", "math"]
