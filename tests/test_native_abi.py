"""CPU-side checks of the drop-in boundary: the library builds, loads and exports what include/filmyou.h declares."""
import ctypes as C
import os
import re

import pytest

from util import pkg


def test_library_exports_every_declared_symbol():
    P = pkg()
    P.build()
    lib = P._native.load()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "filmyou.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fy_[a-z0-9_]+)\s*\(", header))
    assert declared == set(P._native.SYMBOLS), declared ^ set(P._native.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fy_abi_version() == 5


def test_struct_layouts_match_the_header():
    P = pkg()
    assert C.sizeof(P._native.RM2Params) == 48
    assert C.sizeof(P._native.ItemSimParams) == 48
    assert C.sizeof(P._native.ItemCFParams) == 24
    assert C.sizeof(P._native.Stats) == 33 * 8


def test_no_gpu_means_a_loud_failure():
    """The product path has no CPU fallback: without a device the context cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    P = pkg()
    with pytest.raises(P.FilmYouError) as e:
        P.Context(0)
    assert e.value.code == -2


def test_argument_validation_without_a_device():
    P = pkg()
    lib = P._native.load()
    out = C.c_void_p()
    assert lib.fy_context_create(0, None) == -1
    assert lib.fy_ratings_create(None, 0, None, None, None, 0, C.byref(out)) == -1
    assert b"NULL" in lib.fy_last_error()
    assert lib.fy_result_size(None) == 0


def test_configuration_mirrors_hadoop_semantics():
    P = pkg()
    conf = P.Configuration()
    conf.setFloat("lambda", 0.5)
    assert conf.get("lambda") == "0.5"                  # Float.toString(0.5f), parsed back by Double.valueOf (Q3)
    assert conf.getInt("clusterSplit", -1) == 400       # RMRecommenderDriver defaults
    assert conf.getInt("numberOfClusters", -1) == -1
    conf.setBoolean("useCassandraInput", False)
    assert conf.getBoolean("useCassandraInput", True) is False
