"""CPU checks of the all-rows list comparator used by the full-size GPU tests (tests/fullsize_checks.py)."""
import sys

import numpy as np
import pytest

sys.modules.setdefault("torch", __import__("torch"))
from fullsize_checks import assert_same_lists


def _rows(lists):
    u = np.concatenate([[k] * len(v) for k, v in lists]).astype(np.int32)
    i = np.concatenate([[x[0] for x in v] for _, v in lists]).astype(np.int32)
    s = np.concatenate([[x[1] for x in v] for _, v in lists]).astype(np.float32)
    return {"user": u, "item": i, "score": s}


def test_identical_and_tie_swap():
    a = _rows([(7, [(1, -10.0), (2, -11.0), (3, -12.0)]), (9, [(5, -1.0), (6, -2.0)])])
    assert assert_same_lists(a, a) == (0, 0.0)
    # user 7's last place is a tie the other run resolved with item 4
    b = _rows([(7, [(1, -10.0), (2, -11.0), (4, -12.00001)]), (9, [(5, -1.0), (6, -2.0)])])
    n, worst = assert_same_lists(a, b)
    assert n == 1 and worst == 0.0


def test_missing_member_and_score_drift_are_caught():
    a = _rows([(7, [(1, -10.0), (2, -11.0), (3, -12.0)])])
    b = _rows([(7, [(1, -10.0), (4, -11.0), (3, -12.0)])])       # item 2 (not at the cut-off) is missing
    with pytest.raises(AssertionError):
        assert_same_lists(a, b)
    c = _rows([(7, [(1, -10.0), (2, -11.001), (3, -12.0)])])
    with pytest.raises(AssertionError):
        assert_same_lists(a, c)
