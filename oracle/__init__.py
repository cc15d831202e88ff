"""ctypes binding of the CPU oracles (oracle/rm2_oracle.c, oracle/itemsim_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product path (filmyou-core_amd) never imports this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FY_ORACLE_LIB: another build of the same sources (tests/test_sanitizers_cpu.py loads the ASan + UBSan build)
_SO = os.environ.get("FY_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile liboracle.so with gcc (seconds).  Safe to call repeatedly."""
    srcs = [os.path.join(_HERE, f) for f in ("rm2_oracle.c", "itemsim_oracle.c", "itemcf_oracle.c", "cluster_oracle.c", "nmf_oracle.c", "oracle.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale and not os.environ.get("FY_ORACLE_LIB"):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


class _RM2Params(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("number_of_items", C.c_int32), ("number_of_recommendations", C.c_int32),
                ("filter_users", C.c_int32), ("number_of_clusters", C.c_int32), ("n_threads", C.c_int32)]


class _ICFParams(C.Structure):
    _fields_ = [("num_recommendations", C.c_int32), ("max_prefs_per_user", C.c_int32), ("boolean_data", C.c_int32)]


class _ISimParams(C.Structure):
    _fields_ = [("similarity", C.c_int32), ("max_similarities_per_item", C.c_int32), ("exclude_self", C.c_int32),
                ("has_threshold", C.c_int32), ("threshold", C.c_double), ("n_threads", C.c_int32),
                ("min_prefs_per_user", C.c_int32), ("max_prefs_per_user", C.c_int32)]


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, i64 = C.c_void_p, C.c_int64
    L.rm2o_run.argtypes = [C.POINTER(_RM2Params), i64, vp, vp, vp, i64, vp, vp, vp, C.POINTER(vp)]
    L.rm2o_run.restype = C.c_int
    L.rm2o_run_gram.argtypes = L.rm2o_run.argtypes
    L.rm2o_run_gram.restype = C.c_int
    L.rm2o_last_error.restype = C.c_char_p
    L.rm2o_select_clusters.argtypes = [C.c_int32, vp]
    L.rm2o_select_clusters.restype = None
    L.rm2o_free.argtypes = [vp]
    for name, rt in (("n_recs", i64), ("rec_user", vp), ("rec_item", vp), ("rec_cluster", vp), ("rec_score", vp),
                     ("n_users", i64), ("user_id", vp), ("user_sum", vp), ("n_items", i64), ("item_id", vp),
                     ("item_coll", vp), ("item_sum", vp), ("total_sum", C.c_double), ("log_terms", i64),
                     ("fma_terms", i64), ("seconds_scoring", C.c_double)):
        f = getattr(L, "rm2o_" + name)
        f.argtypes = [vp]
        f.restype = rt
    L.isimo_run.argtypes = [C.POINTER(_ISimParams), i64, vp, vp, vp, C.POINTER(vp)]
    L.isimo_run.restype = C.c_int
    L.isimo_free.argtypes = [vp]
    for name, rt in (("n", i64), ("item", vp), ("other", vp), ("sim", vp), ("pairs", i64), ("seconds", C.c_double)):
        f = getattr(L, "isimo_" + name)
        f.argtypes = [vp]
        f.restype = rt
    L.icfo_run.argtypes = [C.POINTER(_ICFParams), i64, vp, vp, vp, i64, vp, vp, vp, C.POINTER(vp)]
    L.icfo_run.restype = C.c_int
    L.icfo_free.argtypes = [vp]
    for name, rt in (("n", i64), ("user", vp), ("item", vp), ("score", vp)):
        f = getattr(L, "icfo_" + name)
        f.argtypes = [vp]
        f.restype = rt
    L.nmfo_run.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i64, vp, vp, vp, vp, vp]
    L.nmfo_run.restype = C.c_int
    L.clo_assign.argtypes = [C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp]
    L.clo_assign.restype = C.c_int
    L.clo_count.argtypes = [i64, vp, C.c_int32, vp]
    L.clo_count.restype = C.c_int
    _lib = L
    return L


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def rm2_gram(*args, **kw):
    """The Gram-restructured CPU scorer (bench.py's cpu_baseline_gram; same interface as rm2)."""
    return rm2(*args, _gram=True, **kw)


def rm2(user, item, score, *, lam, number_of_items, number_of_recommendations, number_of_clusters,
        map_user=None, map_cluster=None, cluster_count=None, filter_users=0, n_threads=1, _gram=False, only_clusters=None):
    """Run the RM2 oracle.  Returns a dict of numpy arrays (see oracle.h for the ordering).
    only_clusters: run job RM2-3 for these reduce groups only (statistics stay global)."""
    L = _load()
    sel = _i32(only_clusters if only_clusters is not None else [])
    L.rm2o_select_clusters(len(sel), sel.ctypes.data)
    user, item = _i32(user), _i32(item)
    score = np.ascontiguousarray(score, dtype=np.float32)
    mu = _i32(map_user if map_user is not None else [])
    mc = _i32(map_cluster if map_cluster is not None else [])
    assert len(mu) == len(mc) and len(user) == len(item) == len(score)
    cc = _i32(cluster_count) if cluster_count is not None else None
    if cc is not None:
        assert len(cc) >= number_of_clusters
    P = _RM2Params(float(lam), int(number_of_items), int(number_of_recommendations), int(filter_users),
                   int(number_of_clusters), int(n_threads))
    out = C.c_void_p()
    rc = (L.rm2o_run_gram if _gram else L.rm2o_run)(C.byref(P), len(user), user.ctypes.data, item.ctypes.data, score.ctypes.data, len(mu),
                    mu.ctypes.data, mc.ctypes.data, cc.ctypes.data if cc is not None else None, C.byref(out))
    L.rm2o_select_clusters(0, None)
    if rc != 0:
        raise RuntimeError("rm2 oracle failed (%d): %s" % (rc, L.rm2o_last_error().decode()))
    h = out.value
    try:
        n, nu, ni = L.rm2o_n_recs(h), L.rm2o_n_users(h), L.rm2o_n_items(h)
        return {
            "rec_user": _arr(L.rm2o_rec_user(h), n, np.int32), "rec_item": _arr(L.rm2o_rec_item(h), n, np.int32),
            "rec_cluster": _arr(L.rm2o_rec_cluster(h), n, np.int32),
            "rec_score": _arr(L.rm2o_rec_score(h), n, np.float32),
            "user_id": _arr(L.rm2o_user_id(h), nu, np.int32), "user_sum": _arr(L.rm2o_user_sum(h), nu, np.float64),
            "item_id": _arr(L.rm2o_item_id(h), ni, np.int32), "item_coll": _arr(L.rm2o_item_coll(h), ni, np.float64),
            "item_sum": _arr(L.rm2o_item_sum(h), ni, np.float64), "total_sum": L.rm2o_total_sum(h),
            "log_terms": L.rm2o_log_terms(h), "fma_terms": L.rm2o_fma_terms(h),
            "seconds_scoring": L.rm2o_seconds_scoring(h),
        }
    finally:
        L.rm2o_free(h)


COSINE, COOCCURRENCE = 0, 1


def itemsim(user, item, score, *, similarity=COSINE, max_similarities_per_item=100, exclude_self=True,
            threshold=None, n_threads=1, min_prefs_per_user=1, max_prefs_per_user=0):
    """Run the item-item similarity oracle (parity unpinned, see itemsim_oracle.c)."""
    L = _load()
    user, item = _i32(user), _i32(item)
    score = np.ascontiguousarray(score, dtype=np.float32)
    P = _ISimParams(int(similarity), int(max_similarities_per_item), int(bool(exclude_self)),
                    0 if threshold is None else 1, 0.0 if threshold is None else float(threshold), int(n_threads),
                    int(min_prefs_per_user), int(max_prefs_per_user))
    out = C.c_void_p()
    rc = L.isimo_run(C.byref(P), len(user), user.ctypes.data, item.ctypes.data, score.ctypes.data, C.byref(out))
    if rc != 0:
        raise RuntimeError("itemsim oracle failed (%d)" % rc)
    h = out.value
    try:
        n = L.isimo_n(h)
        return {"item": _arr(L.isimo_item(h), n, np.int32), "other": _arr(L.isimo_other(h), n, np.int32),
                "sim": _arr(L.isimo_sim(h), n, np.float64), "pairs": L.isimo_pairs(h), "seconds": L.isimo_seconds(h)}
    finally:
        L.isimo_free(h)


def itemcf(user, item, score, sim_item, sim_other, sim_value, *, num_recommendations=100, max_prefs_per_user=50,
           boolean_data=False):
    """Item-based CF recommendation phases on top of a similarity matrix (parity unpinned, see itemcf_oracle.c)."""
    L = _load()
    user, item = _i32(user), _i32(item)
    score = np.ascontiguousarray(score, dtype=np.float32)
    si, so = _i32(sim_item), _i32(sim_other)
    sv = np.ascontiguousarray(sim_value, dtype=np.float64)
    P = _ICFParams(int(num_recommendations), int(max_prefs_per_user), int(bool(boolean_data)))
    out = C.c_void_p()
    rc = L.icfo_run(C.byref(P), len(user), user.ctypes.data, item.ctypes.data, score.ctypes.data, len(si), si.ctypes.data,
                    so.ctypes.data, sv.ctypes.data, C.byref(out))
    if rc != 0:
        raise RuntimeError("itemcf oracle failed (%d)" % rc)
    h = out.value
    try:
        n = L.icfo_n(h)
        return {"user": _arr(L.icfo_user(h), n, np.int32), "item": _arr(L.icfo_item(h), n, np.int32),
                "score": _arr(L.icfo_score(h), n, np.float32)}
    finally:
        L.icfo_free(h)


def cluster_assign(H, first_user=1, cluster_offset=0, n_clusters=None):
    """Cluster-assignment oracle: (users, clusters[, counts]) from the rows of H (see cluster_oracle.c)."""
    L = _load()
    H = np.ascontiguousarray(H, dtype=np.float64)
    n, k = H.shape
    user, cluster = np.zeros(n, np.int32), np.zeros(n, np.int32)
    if L.clo_assign(n, k, H.ctypes.data, int(first_user), int(cluster_offset), user.ctypes.data, cluster.ctypes.data) != 0:
        raise RuntimeError("cluster oracle failed")
    if n_clusters is None:
        return user, cluster
    count = np.zeros(int(n_clusters), np.int32)
    if L.clo_count(n, cluster.ctypes.data, int(n_clusters), count.ctypes.data) != 0:
        raise RuntimeError("cluster id outside [0, n_clusters)")
    return user, cluster, count


def nmf(user, item, score, H, W, *, iterations, ppc=False, normalization_frequency=0):
    """NMF / PPC factorisation oracle (nmf_oracle.c): returns the updated (H, W) copies.  ids are 1-based."""
    L = _load()
    user, item = _i32(user), _i32(item)
    score = np.ascontiguousarray(score, dtype=np.float32)
    H = np.array(H, dtype=np.float64, order="C")
    W = np.array(W, dtype=np.float64, order="C")
    rc = L.nmfo_run(H.shape[0], W.shape[0], H.shape[1], int(iterations), int(bool(ppc)), int(normalization_frequency), len(user),
                    user.ctypes.data, item.ctypes.data, score.ctypes.data, H.ctypes.data, W.ctypes.data)
    if rc != 0:
        raise RuntimeError("nmf oracle failed (%d)" % rc)
    return H, W
