"""Extra CPU-baseline legs of bench.py's JSON line (rank 0, N = 1 only; bounded: about half a minute of CPU work).

  itemsim_cpu_baseline     oracle/itemsim_oracle.c (the restated RowSimilarityJob, parity unpinned) on a user subsample of
                           the bench data set, all host cores
  like_for_like_ml1m_k50   ONE configuration that both sides run IN FULL: ML-1M shape, 50 hashed clusters, top-50 --
                           the reference's own regime (numberOfClusters = 50, T/rmrecommender/TestRMRecommenderJob.java:49):
                           the GPU job, the faithful oracle (all cores, and its serial time scaled from a 1-core run of
                           one cluster), and the Gram-restructured CPU scorer
Test infrastructure: imports oracle/ as bench.py's cpu_baseline leg is allowed to."""
import os
import time

import numpy as np


def extra_legs(P, S, ctx, lam, device="cpu"):
    import oracle
    try:
        cores = max(1, min(16, len(os.sched_getaffinity(0))))      # the box's CPU share for one GPU (see bench.py:host_cores)
    except AttributeError:
        cores = max(1, min(16, os.cpu_count() or 1))
    out = {}
    # ---- item-sim on a subsample of the ML-25M-shaped users
    rng = np.random.Generator(np.random.PCG64(9))
    n_users = S.SHAPES["ml25m"][0]
    users = np.sort(rng.choice(n_users, size=6000, replace=False)) + 1
    u, i, s, _ = S.generate("ml25m", users=users, device=device)      # the same cells on any device
    u, i, s = u.cpu().numpy(), i.cpu().numpy(), s.cpu().numpy()
    t0 = time.time()
    r = oracle.itemsim(u, i, s, max_similarities_per_item=100, n_threads=cores)
    dt = time.time() - t0
    out["itemsim_cpu_baseline"] = {"value": r["pairs"] / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "seconds": dt,
                                   "sample": "6000 users sampled from the ml25m-shaped data (%d ratings, %.3g unordered co-rating pairs), "
                                             "oracle/itemsim_oracle.c (cosine, top-100), %d threads" % (len(u), r["pairs"], cores)}
    # ---- one configuration both sides run in full
    u, i, s, facts = S.generate("ml1m", device="cpu")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
    K = 50
    mc = S.hash_clustering(uu, K)
    kw = dict(lam=lam, number_of_items=facts["n_items"], number_of_recommendations=50, number_of_clusters=K, map_user=uu, map_cluster=mc)
    t0 = time.time()
    ref = oracle.rm2(u, i, s, n_threads=cores, **kw)
    t_all = time.time() - t0
    t0 = time.time()
    gram = oracle.rm2_gram(u, i, s, n_threads=cores, **kw)
    t_gram = time.time() - t0
    # serial reducer (LocalJobRunner): the users of one of the 50 clusters on one core, scaled by the exact multiply-add count
    keep = np.isin(u, uu[mc < 1])
    t0 = time.time()
    part = oracle.rm2(u[keep], i[keep], s[keep], n_threads=1, **kw)
    t_1 = (time.time() - t0) * ref["fma_terms"] / max(1, part["fma_terms"])
    conf = P.Configuration()
    conf.set("lambda", repr(lam))
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", 50)
    job = P.RM2Job(conf, ctx)
    ratings = P.Ratings(ctx, u, i, s)
    job.run(ratings, clustering=(uu, mc)).close()
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        rec = job.run(ratings, clustering=(uu, mc))
        n_rows = rec.size
        rec.close()
    ctx.synchronize()
    t_gpu = (time.perf_counter() - t0) / reps
    ratings.close()
    n = len(ref["rec_user"])
    assert n_rows == n == len(gram["rec_user"])
    out["like_for_like_ml1m_k50"] = {
        "workload": "ml1m-shaped synthetic (6040 x 3706, %d ratings), 50 hashed clusters, top-50, lambda %g: %d rows, %.3g log terms, "
                    "%.3g multiply-adds in the reference's loop" % (len(u), lam, n, ref["log_terms"], ref["fma_terms"]),
        "gpu_recs_per_s": n / t_gpu, "gpu_ms": 1e3 * t_gpu,
        "cpu_faithful_recs_per_s": n / t_all, "cpu_faithful_seconds": t_all, "cpu_faithful_cores": cores,
        "cpu_faithful_1core_recs_per_s": n / t_1, "cpu_faithful_1core_seconds_scaled": t_1,
        "cpu_faithful_1core_note": "one core on the first cluster, scaled by the exact multiply-add count (x%.1f)" % (ref["fma_terms"] / max(1, part["fma_terms"])),
        "cpu_gram_recs_per_s": n / t_gram, "cpu_gram_seconds": t_gram, "cpu_gram_cores": cores,
        "gpu_over_cpu_faithful": t_all / t_gpu, "gpu_over_cpu_gram": t_gram / t_gpu, "host_nproc": os.cpu_count()}
    return out
