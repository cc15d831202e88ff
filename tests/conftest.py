import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def coo_from_items_by_users(A):
    """A[i][j] = rating of user j+1 for item i+1 (DataInitialization.createIntPairFloatFile, :164-172).
    Zeros are written too, exactly like the reference's fixture writer: the score > 0 filter is the job's."""
    A = np.asarray(A, dtype=np.float64)
    n_items, n_users = A.shape
    item, user = np.meshgrid(np.arange(1, n_items + 1), np.arange(1, n_users + 1), indexing="ij")
    return user.ravel().astype(np.int32), item.ravel().astype(np.int32), A.ravel().astype(np.float32)


@pytest.fixture(scope="session")
def rm_golden():
    with open(os.path.join(GOLDEN, "rm_test_data.json")) as f:
        d = json.load(f)
    user, item, score = coo_from_items_by_users(d["A_items_by_users"])
    d["coo"] = (user, item, score)
    # clustering[u-1] = cluster of user u (written with start index 1: TestHDFSRM2.java:50-51)
    d["map_user"] = np.arange(1, len(d["clustering"]) + 1, dtype=np.int32)
    d["map_cluster"] = np.asarray(d["clustering"], dtype=np.int32)
    cc = np.zeros(d["numberOfClusters"], dtype=np.int32)
    cc[: len(d["clusteringCount"])] = d["clusteringCount"]
    d["cluster_count"] = cc
    return d


@pytest.fixture(scope="session")
def rm_golden2():
    with open(os.path.join(GOLDEN, "rm_test_data2.json")) as f:
        d = json.load(f)
    d["coo"] = coo_from_items_by_users(d["A_items_by_users"])
    return d


def pytest_terminal_summary(terminalreporter):
    """How often the oracle comparisons of this run needed an absolute term on top of north_star's relative 1e-5 (tests/util.py)."""
    try:
        from util import SLACK_TALLY as T
    except Exception:
        return
    if T["calls"]:
        terminalreporter.write_line("oracle comparisons: %d calls, %d scores, %d needed an absolute slack, worst relative error %.2e"
                                    % (T["calls"], T["comparisons"], T["needed_atol"], T["worst_rel"]))
