"""Host-side logic that needs no GPU: the Configuration mirror (option names and defaults of RMRecommenderDriver), argument
checks that fire before any device call, and the loud failure without a device."""
import importlib

import numpy as np
import pytest
import torch

P = importlib.import_module("filmyou-core_amd")


def test_configuration_mirrors_the_driver_defaults_and_string_semantics():
    conf = P.Configuration()
    # M/rmrecommender/RMRecommenderDriver.java:89-120
    assert conf.get("lambda") == "0.1" and conf.getInt("numberOfRecommendations", -1) == 1000
    assert conf.getInt("filterUsers", -1) == 0 and conf.get("clusteringCount") == "clusteringCount"
    assert conf.getInt("numberOfItems", -1) == -1                      # no default: the jobs insist on it
    conf.setFloat("lambda", 0.5)                                       # quirk Q3: Float.toString(0.5f) -> "0.5"
    assert conf.get("lambda") == "0.5" and conf.getDouble("lambda", 0.0) == 0.5
    conf.setInt("numberOfClusters", 7)
    conf.setBoolean("useCassandraInput", False)
    assert conf.get("numberOfClusters") == "7" and conf.get("useCassandraInput") == "false"


class _NoDevice:       # stands in for a Context: nothing below may reach the device
    _h = None


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU container check")
def test_every_entry_point_fails_loudly_without_a_device():
    with pytest.raises(P.FilmYouError) as e:
        P.Context(0)
    assert e.value.code == -2                                            # FY_ERR_NO_DEVICE: there is no CPU fallback


def test_argument_checks_fire_before_the_device_is_touched():
    conf = P.Configuration()
    conf.setInt("numberOfItems", 10)
    conf.setInt("numberOfClusters", 1)
    u = np.array([1], np.int32)
    with pytest.raises(ValueError, match="world > 1 needs collectives"):
        P.RM2Job(conf, _NoDevice()).run((u, u, np.ones(1, np.float32)), world=2, rank=0)
    fconf = P.Configuration()
    for k, v in (("numberOfUsers", 3), ("numberOfItems", 2), ("numberOfClusters", 2), ("numberOfIterations", 1)):
        fconf.setInt(k, v)
    with pytest.raises(ValueError, match="numberOfUsers x numberOfClusters"):
        P.NMFDriver(fconf, _NoDevice()).run((u, u, np.ones(1, np.float32)), np.ones((4, 2)), np.ones((2, 2)))
    with pytest.raises(ValueError, match="users x clusters"):
        P.ClusterAssignmentJob(_NoDevice()).run(np.ones(5))


def test_bench_refuses_a_mislabelled_gpu_count():
    """bench.py --gpus N must run N ranks or fail: with WORLD_SIZE unset it starts the ranks itself (each of them then fails
    loudly here, there is no GPU); with a WORLD_SIZE that disagrees with --gpus it refuses to print a line."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--shape", "ml100k"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in bad.stderr and "{" not in bad.stdout
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--shape", "ml100k"],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "{" not in out.stdout
        assert out.stderr.count("needs an MI355X") >= 1          # the child ranks were started and refused to fall back


def test_ranks_refuse_an_output_directory_that_holds_any_part_file(tmp_path):
    """a rerun with world = 2 into the output directory of a job with a larger world (a foreign part-r-00003) must fail on
    EVERY rank before its first collective -- Hadoop's FileOutputFormat.checkOutputSpecs refuses any existing output directory; with
    several ranks the directory is shared, so it may exist but must hold no part file.  (The check needs no device: it runs here.)"""
    import os
    P = importlib.import_module("filmyou-core_amd")
    sf = importlib.import_module("filmyou-core_amd.seqfile")
    base = str(tmp_path / "recommendation")
    u = np.array([1, 1, 2, 2], np.int32)
    i = np.array([1, 2, 1, 3], np.int32)
    sf.write_intpair_float(str(tmp_path / "input" / "ratings" / "data"), u, i, np.array([1, 2, 3, 4], np.float32))
    sf.write_int_int(os.path.join(base, "clustering", "data"), np.array([1, 2], np.int32), np.array([0, 0], np.int32))
    sf.write_int_int(os.path.join(base, "clusteringCount", "data"), np.array([0], np.int32), np.array([2], np.int32))
    out = tmp_path / "output"
    sf.write_intpair_float(str(out / "part-r-00003"), u, i, np.array([1, 1, 1, 1], np.float32))
    os.makedirs(os.path.join(base, "rm2", "userSum"))
    conf = P.Configuration()
    conf.setFloat("lambda", 0.5)
    conf.setInt("numberOfItems", 3)
    conf.setInt("numberOfClusters", 1)
    conf.set("directory", base)
    conf.set("mapred.input.dir", str(tmp_path / "input" / "ratings"))
    conf.set("mapred.output.dir", str(out))
    for rank in (1, 0):
        with pytest.raises(RuntimeError, match="RM2 failed!: output directory .* already exists and holds part-r-00003"):
            P.RM2Job(conf).run_from_files(rank=rank, world=2, collectives=object())
    assert sorted(os.listdir(str(out))) == [".part-r-00003.crc", "part-r-00003"] or sorted(os.listdir(str(out))) == ["part-r-00003"]
    # an EMPTY shared directory is fine for several ranks (rank 0 may have created it), an existing one is not for a single rank
    with pytest.raises(RuntimeError, match="RM2 failed!: output directory .* already exists"):
        P.RM2Job(conf).run_from_files()
