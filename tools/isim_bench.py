#!/usr/bin/env python3
"""Times the item-item similarity build alone (ML-25M shape by default): `python tools/isim_bench.py [shape] [reps]`.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split (row kernel / band sweep / finish)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "ml25m"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import torch
    P = importlib.import_module("filmyou-core_amd")
    S = importlib.import_module("filmyou-core_amd.synth")
    dev = torch.device("cuda", 0)
    user, item, score, facts = S.generate(shape, device=dev)
    ctx = P.Context(0)
    ratings = P.Ratings(ctx, user, item, score)
    job = P.RowSimilarityJob(ctx)
    out = []
    for r in range(reps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        res = job.run(ratings, maxSimilaritiesPerRow=100)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = res.stats
        res.close()
        if r:
            out.append({"ms_wall": 1e3 * dt, "ms_cooc": st["ms_cooc"], "ms_prepare": st["ms_prepare"], "ms_total": st["ms_total"],
                        "pairs": st["unordered_pairs"], "rows": st["recs"], "ms_tables": st["ms_tables"],
                        "candidates": st["isim_candidates"], "redone_rows": st["isim_redone_rows"]})
    best = min(out, key=lambda x: x["ms_cooc"])
    best["GBps_8B_per_pair"] = 8.0 * best["pairs"] / (best["ms_cooc"] * 1e-3) / 1e9
    best["env"] = {k: v for k, v in os.environ.items() if k.startswith("FY_")}
    print(json.dumps(best))


if __name__ == "__main__":
    main()
