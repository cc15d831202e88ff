"""ALL-ROWS precision of the production RM2 job against the fp64 definition (tests/fp64_definition.py) at the benchmark sizes.

    python tools/precision_all_rows.py [ml25m|netflix] [clusters ...]         -> gpurun_out/precision_<shape>.json

Not part of the product; the same measurement runs inside `pytest -m gpu` (tests/test_full_size_gpu.py, tests/test_netflix_gpu.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fp64_definition import compare_with_definition, fp64_scores  # noqa: E402
from fullsize_checks import load_shape, run_rm2  # noqa: E402
from util import synth  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
    ks = [int(a) for a in sys.argv[2:]] or [1, 50]
    top_n = 100 if shape == "netflix" else 50
    lam = 0.1
    data = load_shape(shape)
    out = {}
    for k in ks:
        env = {}
        for kv in os.environ.get("PREC_ENV", "").split(","):
            if "=" in kv:
                a, b = kv.split("=")
                env[a] = b
        t0 = time.time()
        rows, sums, st = run_rm2(data, top_n, lam, clusters=k, env=env)
        t1 = time.time()
        clustering = None
        if k > 1:
            uu = np.arange(1, data["facts"]["n_users"] + 1, dtype=np.int32)
            clustering = (uu, synth().hash_clustering(uu, k))
        ref = fp64_scores(data["dev"], rows, lam, data["facts"]["n_items"], clustering=clustering)
        t2 = time.time()
        rep = compare_with_definition(rows, ref)
        n_u = np.diff(data["R"].indptr)
        rep["worst_rows_n_u"] = [int(n_u[u - 1]) for u, *_ in rep["worst_rows"]]
        rep["seconds_job"] = t1 - t0
        rep["seconds_fp64"] = t2 - t1
        # the distribution of the error against list length and |score|
        got = rows["score"].astype(np.float64)
        rel = np.abs(got - ref) / np.abs(ref)
        absd = np.abs(got - ref)
        rep["max_abs"] = float(absd.max())
        rep["rows_over_5e-6"] = int((rel > 5e-6).sum())
        rep["min_abs_score"] = float(np.abs(ref).min())
        out["clusters_%d" % k] = rep
        print(shape, k, json.dumps(rep), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "precision_%s.json" % shape), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
