#!/usr/bin/env python3
"""Worst rows of the packed-matrix job against the fp32-matrix job (FY_M24=0), ML-25M shape: who they are."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize_checks import load_shape, run_rm2  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
    clusters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    top_n = 100 if shape == "netflix" else 50
    data = load_shape(shape)
    a, _, sta = run_rm2(data, top_n, 0.1, clusters=clusters)
    b, _, stb = run_rm2(data, top_n, 0.1, env={"FY_M24": "0"}, clusters=clusters)
    big = int(max(a["item"].max(), b["item"].max())) + 1
    ka = a["user"].astype(np.int64) * big + a["item"]
    kb = b["user"].astype(np.int64) * big + b["item"]
    oa, ob = np.argsort(ka), np.argsort(kb)
    common_a = np.isin(ka[oa], kb[ob], assume_unique=True)
    common_b = np.isin(kb[ob], ka[oa], assume_unique=True)
    sa, sb = a["score"][oa][common_a].astype(np.float64), b["score"][ob][common_b].astype(np.float64)
    ua = a["user"][oa][common_a]
    rel = np.abs(sa - sb) / np.abs(sb)
    deg = np.bincount(data["dev"][0].cpu().numpy())
    order = np.argsort(-rel)[:12]
    print("rows %d common %d worst %.3e mean %.3e p99.9 %.3e" % (len(ka), len(sa), rel.max(), rel.mean(), np.quantile(rel, 0.999)))
    for x in order:
        print("user %d n=%d packed %.7f fp32 %.7f rel %.2e" % (ua[x], deg[ua[x]], sa[x], sb[x], rel[x]))
    for lo, hi in ((0, 25), (25, 50), (50, 100), (100, 400), (400, 10**9)):
        m = (deg[ua] >= lo) & (deg[ua] < hi)
        if m.any():
            print("n in [%d, %d): %d rows, worst %.2e, |score| median %.1f" % (lo, hi, m.sum(), rel[m].max(), np.median(np.abs(sb[m]))))


if __name__ == "__main__":
    main()
