"""Where a cold fy_rm2_prepare spends its time: `python tools/prep_probe.py <clusters> <world> [rank] [shape]` prints ms_prepare of a few cold
prepares of one rank (ML-25M shape; run it under `rocprofv3 --kernel-trace --stats` for the kernels behind the number)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    K, world = int(sys.argv[1]), int(sys.argv[2])
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    shape = sys.argv[4] if len(sys.argv) > 4 else "ml25m"
    P = importlib.import_module("filmyou-core_amd")
    S = importlib.import_module("filmyou-core_amd.synth")
    user, item, score, facts = S.generate(shape, device=torch.device("cuda", 0))
    uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
    clustering = (uu, S.hash_clustering(uu, K)) if K > 1 else None
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 50)
    ctx = P.Context(0)
    r = P.Ratings(ctx, user, item, score)
    job = P.RM2Job(conf, ctx)
    out = []
    for rep in range(6):
        ctx.synchronize()
        t0 = time.perf_counter()
        pr = job.prepare(r, clustering=clustering, rank=rank, world=world, cache=False)
        ctx.synchronize()
        out.append(1e3 * (time.perf_counter() - t0))
        pr.close()
    print("clusters %d world %d rank %d: cold prepare wall ms %s" % (K, world, rank, " ".join("%.2f" % x for x in out)), flush=True)


if __name__ == "__main__":
    main()
