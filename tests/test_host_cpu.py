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
