"""ctypes declarations of include/filmyou.h and the in-tree build of libfilmyou_hip.so.

There is no CPU fallback: if the shared library is missing or does not load, importing callers get a loud
ImportError/OSError; if no gfx950 device is present, fy_context_create returns FY_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libfilmyou_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "filmyou.h")
SOURCES = ["fy_api.hip", "fy_prep.hip", "fy_rm2.hip", "fy_itemsim.hip", "fy_itemcf.hip", "fy_cluster.hip", "fy_nmf.hip", "fy_rccl.hip", "fy_seqfile.cpp"]
HEADERS = ["fy_common.hpp", "fy_prep.hpp", "fy_cooc.hpp", "fy_rm2.hpp", "fy_rm2_kernels.hpp", "fy_rm2_coop.hpp"]  # fy_itemcf.hip uses fy_prep.hpp / fy_rm2.hpp
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-Wall",
               "-Wno-unused-result", "-ldl"]

# every symbol include/filmyou.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "fy_abi_version", "fy_last_error", "fy_context_create", "fy_context_destroy", "fy_context_synchronize", "fy_context_reload_tuning", "fy_context_inject_alloc_failure",
    "fy_context_stream", "fy_ratings_create", "fy_ratings_destroy", "fy_ratings_nnz", "fy_ratings_drop_cache", "fy_rm2_prepare",
    "fy_rm2_partial_stats", "fy_rm2_stats_layout", "fy_rm2_set_global_stats", "fy_rm2_set_collectives", "fy_rccl_unique_id", "fy_rccl_create", "fy_rccl_collectives", "fy_rccl_counters", "fy_rccl_destroy", "fy_rccl_detach_context", "fy_rm2_score", "fy_rm2_job_destroy", "fy_rm2_run",
    "fy_itemsim_build", "fy_itemsim_run", "fy_itemcf_recommend", "fy_cluster_assign", "fy_nmf_factorize", "fy_result_size", "fy_result_key0", "fy_result_key1", "fy_result_value",
    "fy_result_aux", "fy_result_n_users", "fy_result_user_id", "fy_result_user_sum", "fy_result_n_items",
    "fy_result_item_id", "fy_result_item_coll", "fy_result_total_sum", "fy_result_free", "fy_result_stats",
    "fy_seqfile_read_int_int", "fy_seqfile_read_int_double", "fy_seqfile_read_intpair_float", "fy_seqfile_write_int_int",
    "fy_seqfile_write_int_double", "fy_seqfile_write_intpair_float", "fy_mapfile_write_int_double", "fy_buffer_free",
]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [INCLUDE]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU: one object per source (in parallel, only the stale ones), then one link.
    Returns the .so path."""
    if not force and not _stale():
        build_host_driver()
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cflags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-ldl")]
    common = max(os.path.getmtime(d) for d in [os.path.join(CSRC, h) for h in HEADERS] + [INCLUDE])
    # objects are only as good as the command that made them: another compiler or other flags (HIPCC, HIPCC_FLAGS edited) rebuild
    # everything -- staleness by mtime alone would link objects of mixed flags
    import hashlib
    stamp_path = os.path.join(obj_dir, "flags.stamp")
    stamp = hashlib.sha256(("\0".join([os.path.realpath(hipcc)] + cflags)).encode()).hexdigest()
    try:
        with open(stamp_path) as f:
            same_flags = f.read().strip() == stamp
    except OSError:
        same_flags = False
    if not same_flags:
        force = True

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(obj_dir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(common, os.path.getmtime(path)):
            return obj
        cmd = [hipcc] + cflags + ["-c", "-o", obj, path]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    with open(stamp_path, "w") as f:
        f.write(stamp + "\n")
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    build_host_driver()
    return LIB_PATH


HOST_DIR = os.path.join(_HERE, "host")
HOST_DRIVER = os.path.join(HOST_DIR, "rm2_main")


def build_host_driver():
    """g++ build of the C++ host-side mirror's driver (filmyou-core_amd/host/rm2_main.cpp) against the library."""
    src = os.path.join(HOST_DIR, "rm2_main.cpp")
    hdr = os.path.join(HOST_DIR, "filmyou_job.hpp")
    if os.path.exists(HOST_DRIVER) and all(os.path.getmtime(HOST_DRIVER) >= os.path.getmtime(p) for p in (src, hdr, LIB_PATH)):
        return HOST_DRIVER
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-o", HOST_DRIVER, src, "-L" + LIB_DIR, "-lfilmyou_hip",
                    "-Wl,-rpath,$ORIGIN/../lib"], check=True)
    return HOST_DRIVER


class RM2Params(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("number_of_items", C.c_int32), ("number_of_recommendations", C.c_int32),
                ("filter_users", C.c_int32), ("number_of_clusters", C.c_int32), ("rank", C.c_int32),
                ("world", C.c_int32), ("flags", C.c_uint32), ("workspace_bytes", C.c_int64)]


# fy_collectives: RCCL-shaped callbacks (device pointers, hipStream_t), see include/filmyou.h
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
REDUCE_SCATTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class Collectives(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("reduce_scatter_f32", REDUCE_SCATTER_FN)]


class NMFParams(C.Structure):
    _fields_ = [("number_of_users", C.c_int32), ("number_of_items", C.c_int32), ("number_of_clusters", C.c_int32),
                ("number_of_iterations", C.c_int32), ("ppc", C.c_int32), ("normalization_frequency", C.c_int32)]


class ItemSimParams(C.Structure):
    _fields_ = [("similarity", C.c_int32), ("max_similarities_per_item", C.c_int32), ("exclude_self", C.c_int32),
                ("has_threshold", C.c_int32), ("threshold", C.c_double), ("rank", C.c_int32), ("world", C.c_int32),
                ("flags", C.c_uint32), ("min_prefs_per_user", C.c_int32), ("max_prefs_per_user", C.c_int32)]


class ItemCFParams(C.Structure):
    _fields_ = [("num_recommendations", C.c_int32), ("max_prefs_per_user", C.c_int32), ("boolean_data", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("n_users", C.c_int64), ("n_items", C.c_int64),
                ("n_clusters_nonempty", C.c_int64), ("users_scored", C.c_int64), ("recs", C.c_int64),
                ("log_terms", C.c_int64), ("pair_contribs", C.c_int64), ("unordered_pairs", C.c_int64),
                ("ms_prepare", C.c_double), ("ms_cooc", C.c_double), ("ms_score", C.c_double),
                ("ms_topn", C.c_double), ("ms_total", C.c_double), ("score_launches", C.c_int64),
                ("cooc_launches", C.c_int64), ("blocks_total", C.c_int64), ("blocks_survived", C.c_int64),
                ("log_terms_evaluated", C.c_int64), ("prune_fallbacks", C.c_int64), ("ms_tables", C.c_double), ("ms_mirror", C.c_double), ("topn_select_users", C.c_int64),
                ("panel_clusters", C.c_int64), ("stray_blocks", C.c_int64), ("bound_repairs", C.c_int64),
                ("isim_candidates", C.c_int64), ("isim_redone_rows", C.c_int64),
                ("prepared_from_cache", C.c_int64), ("tables_from_cache", C.c_int64),
                ("cooc_segments", C.c_int64), ("cooc_matrix_bytes", C.c_int64), ("rows_refined", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """dlopen the in-tree library and declare argument types.  Raises OSError when it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(the HIP extension is required; there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so under the same
    # SONAMEs as /opt/rocm's.  Whichever is loaded first serves both torch and this library; if this library came first,
    # torch would later run on a runtime it was not built with (observed: "No HIP GPUs are available" from torch.cuda.init()).
    # The Python host layer hands torch tensors / streams to the library anyway, so torch goes first.  A host without torch
    # (the JNI shim, the C++ driver) simply uses the system ROCm.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, pvp = C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)
    L.fy_abi_version.restype = C.c_int
    L.fy_last_error.restype = C.c_char_p
    L.fy_context_create.argtypes = [C.c_int, pvp]
    L.fy_context_destroy.argtypes = [vp]
    L.fy_context_destroy.restype = None
    L.fy_context_synchronize.argtypes = [vp]
    L.fy_context_inject_alloc_failure.argtypes = [vp, i64]
    L.fy_context_reload_tuning.argtypes = [vp]
    L.fy_context_stream.argtypes = [vp]
    L.fy_context_stream.restype = vp
    L.fy_ratings_create.argtypes = [vp, i64, vp, vp, vp, C.c_int, pvp]
    L.fy_ratings_destroy.argtypes = [vp]
    L.fy_ratings_destroy.restype = None
    L.fy_ratings_nnz.argtypes = [vp]
    L.fy_ratings_nnz.restype = i64
    L.fy_ratings_drop_cache.argtypes = [vp]
    L.fy_ratings_drop_cache.restype = None
    L.fy_rm2_prepare.argtypes = [vp, C.POINTER(RM2Params), vp, i64, vp, vp, vp, pvp]
    L.fy_rm2_partial_stats.argtypes = [vp, pvp, C.POINTER(i64)]
    L.fy_rm2_set_global_stats.argtypes = [vp, vp, i32]
    L.fy_rm2_stats_layout.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.fy_nmf_factorize.argtypes = [vp, C.POINTER(NMFParams), vp, vp, vp, C.POINTER(Stats)]
    L.fy_cluster_assign.argtypes = [vp, i32, i32, vp, C.c_int, i32, i32, i32, vp, vp, vp]
    L.fy_rm2_set_collectives.argtypes = [vp, C.POINTER(Collectives)]
    L.fy_rccl_unique_id.argtypes = [vp]
    L.fy_rccl_create.argtypes = [vp, C.c_int, C.c_int, vp, pvp]
    L.fy_rccl_collectives.argtypes = [vp, C.POINTER(Collectives)]
    L.fy_rccl_counters.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.fy_rccl_destroy.argtypes = [vp]
    L.fy_rccl_destroy.restype = None
    L.fy_rccl_detach_context.argtypes = [vp]
    L.fy_rccl_detach_context.restype = None
    L.fy_rm2_score.argtypes = [vp, pvp]
    L.fy_rm2_job_destroy.argtypes = [vp]
    L.fy_rm2_job_destroy.restype = None
    L.fy_rm2_run.argtypes = [C.POINTER(RM2Params), i64, vp, vp, vp, i64, vp, vp, vp, pvp]
    L.fy_itemsim_build.argtypes = [vp, C.POINTER(ItemSimParams), vp, pvp]
    L.fy_itemsim_run.argtypes = [C.POINTER(ItemSimParams), i64, vp, vp, vp, pvp]
    L.fy_itemcf_recommend.argtypes = [vp, C.POINTER(ItemCFParams), vp, vp, pvp]
    for name, rt in (("size", i64), ("key0", vp), ("key1", vp), ("value", vp), ("aux", vp), ("n_users", i64),
                     ("user_id", vp), ("user_sum", vp), ("n_items", i64), ("item_id", vp), ("item_coll", vp),
                     ("total_sum", C.c_double)):
        f = getattr(L, "fy_result_" + name)
        f.argtypes = [vp]
        f.restype = rt
    L.fy_result_free.argtypes = [vp]
    L.fy_result_free.restype = None
    L.fy_result_stats.argtypes = [vp, C.POINTER(Stats)]
    cp, pi64 = C.c_char_p, C.POINTER(i64)
    L.fy_seqfile_read_int_int.argtypes = [cp, pi64, pvp, pvp]
    L.fy_seqfile_read_int_double.argtypes = [cp, pi64, pvp, pvp]
    L.fy_seqfile_read_intpair_float.argtypes = [cp, pi64, pvp, pvp, pvp]
    L.fy_seqfile_write_int_int.argtypes = [cp, i64, vp, vp]
    L.fy_seqfile_write_int_double.argtypes = [cp, i64, vp, vp]
    L.fy_seqfile_write_intpair_float.argtypes = [cp, i64, vp, vp, vp]
    L.fy_mapfile_write_int_double.argtypes = [cp, i64, vp, vp]
    L.fy_buffer_free.argtypes = [vp]
    L.fy_buffer_free.restype = None
    _lib = L
    return L
