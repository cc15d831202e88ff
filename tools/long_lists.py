"""Long lists (numberOfRecommendations = 1000, the reference's default) at full size: the pruned flow with a 5 N-column seed against
the plain full pass -- time per job and, on request, all-rows equality.

    python tools/long_lists.py <shape> <top_n> <clusters> [check] [env K=V,...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fullsize_checks import assert_same_lists, load_shape, run_rm2  # noqa: E402

KEYS = ("ms_prepare", "ms_tables", "ms_cooc", "ms_mirror", "ms_score", "ms_topn", "ms_total", "blocks_total", "blocks_survived", "log_terms_evaluated",
        "log_terms", "prune_fallbacks", "topn_select_users", "panel_clusters", "stray_blocks", "score_launches")


def main():
    shape, top_n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    check = "check" in sys.argv[4:]
    env = {}
    for a in sys.argv[4:]:
        if "=" in a:
            for kv in a.split(","):
                x, y = kv.split("=")
                env[x] = y
    data = load_shape(shape)
    out = {}
    res = {}
    for name, e in (("pruned", dict(env)), ("full", dict(env, FY_PRUNE="0"))):
        if name == "full" and not check and "both" not in sys.argv[4:]:
            continue
        best = None
        for rep in range(3):
            t0 = time.time()
            rows, _, st = run_rm2(data, top_n, 0.1, env=e, clusters=k)
            wall = time.time() - t0
            job = st["ms_prepare"] + st["ms_total"]
            if best is None or job < best[0]:
                best = (job, {kk: st[kk] for kk in KEYS}, wall)
        out[name] = {"ms_job": best[0], "wall_s": best[2], **best[1]}
        res[name] = rows
        print(shape, top_n, k, name, json.dumps(out[name]), flush=True)
    if check:
        n_diff, worst = assert_same_lists(res["pruned"], res["full"], score_rtol=1e-5)
        print("pruned vs full pass: %d rows, %d differ (ties at a cut-off), worst score difference %.2e" % (len(res["pruned"]["user"]), n_diff, worst), flush=True)
        out["check"] = {"rows": int(len(res["pruned"]["user"])), "differ": int(n_diff), "worst": worst}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "long_lists_%s_%d_%d.json" % (shape, top_n, k)), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
