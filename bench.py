#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the RM2 top-N scorer (and the item-item similarity build) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU over RCCL.
One "step" = one complete RM2 job over the ML-25M-shaped synthetic ratings, which are resident in HBM when the timed
region starts: COO -> CSR/CSC, statistics, (all-gather), per-cluster co-rating matrix, scoring, top-N.
Rank 0 prints ONE JSON line.  `value` = top-N recommendation rows produced by all ranks per second.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="ml25m", help="ml100k | ml1m | ml25m | netflix")
    ap.add_argument("--clusters", type=int, default=1, help="numberOfClusters (users hashed to clusters); 1 = one neighbourhood")
    ap.add_argument("--top-n", type=int, default=None)
    ap.add_argument("--lam", type=float, default=0.1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-itemsim", action="store_true", help="skip the item-item similarity leg")
    ap.add_argument("--no-factorization", action="store_true", help="skip the PPC factorisation + cluster assignment leg")
    ap.add_argument("--cpu-users", type=int, default=0, help="users in the CPU sample cluster (0 = auto)")
    return ap.parse_args()


def cpu_baseline(S, shape, facts, lam, top_n, n_users_sample):
    """The faithful CPU oracle (kind "port": this image has no JVM for the reference itself) timed on a bounded
    sample: one cluster of `n_users_sample` users drawn from the same synthetic data set, scored with the reference's
    brute-force loop nest on the host cores."""
    import oracle
    cores = max(1, min(16, os.cpu_count() or 1))
    rng = np.random.Generator(np.random.PCG64(7))
    users = np.sort(rng.choice(facts["n_users"], size=min(n_users_sample, facts["n_users"]), replace=False)) + 1
    u, i, s, _ = S.generate(shape, users=users, device="cpu")
    u, i, s = u.numpy(), i.numpy(), s.numpy()
    t0 = time.time()
    ref = oracle.rm2(u, i, s, lam=lam, number_of_items=facts["n_items"], number_of_recommendations=top_n,
                     number_of_clusters=1, n_threads=cores)
    dt = time.time() - t0
    recs = len(ref["rec_user"])
    return {"value": recs / dt, "unit": "recs/s", "cores": cores, "kind": "port",
            "sample": "one %d-user cluster sampled from the same %s-shaped data (%d ratings, %d candidate items, "
                      "%.3g log-terms x %d neighbours), oracle/rm2_oracle.c with %d OpenMP threads, %.1f s; the "
                      "reference's cost per term grows with the cluster size, the GPU path's does not"
                      % (len(users), shape, len(u), len(ref["item_id"]), ref["log_terms"], len(users) - 1, cores, dt),
            "seconds": dt, "log_terms_per_s": ref["log_terms"] / dt}


def main():
    a = parse()
    P = importlib.import_module("filmyou-core_amd")
    S = importlib.import_module("filmyou-core_amd.synth")
    par = importlib.import_module("filmyou-core_amd.parallel")
    # FY_BENCH_REHEARSAL=1: several ranks share GPU 0 over gloo -- only to rehearse the multi-rank control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("FY_BENCH_REHEARSAL") == "1"
    rank, local_rank, world = par.init_distributed(backend="gloo" if rehearsal else None)
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    top_n = a.top_n if a.top_n is not None else (100 if a.shape == "netflix" else 50)

    # ---- synthetic ratings, generated on the GPU with the device-independent hash (identical on every rank)
    t0 = time.time()
    user, item, score, facts = S.generate(a.shape, device=dev)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    K = a.clusters
    clustering = None
    if K > 1:
        uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
        clustering = (uu, S.hash_clustering(uu, K))

    conf = P.Configuration()
    conf.set("lambda", repr(a.lam))
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", K)
    conf.setInt("numberOfRecommendations", top_n)
    ctx = P.Context(local_rank)
    ratings = P.Ratings(ctx, user, item, score)       # resident in HBM before the timed region
    # world > 1: RCCL collectives through torch.distributed (fy_collectives): the item statistics are all-gathered and
    # a cluster that spans all ranks is scored cooperatively (row-sharded matrix build, reduce-scattered partial sums);
    # FY_BENCH_EXCHANGE_ONLY=1 keeps the round-1 flow (statistics exchange only, matrix replicated) for comparison
    exchange = collectives = None
    if world > 1:
        if os.environ.get("FY_BENCH_EXCHANGE_ONLY") == "1":
            exchange = par.StatsExchange(local_rank)
        else:
            collectives = par.TorchCollectives(local_rank)
    job = P.RM2Job(conf, ctx)

    def step():
        rec = job.run(ratings, clustering=clustering, rank=rank, world=world, exchange=exchange, collectives=collectives)
        st = rec.stats
        rec.close()
        return st

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def reduce_(t, op):
        if world > 1:
            if rehearsal:      # gloo: reduce on the host
                h = t.cpu()
                dist.all_reduce(h, op=op)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op)

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(a.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    recs = torch.tensor([float(stats[-1]["recs"]), float(stats[-1]["log_terms"]), float(stats[-1]["users_scored"])],
                        dtype=torch.float64, device=dev)
    reduce_(tt, dist.ReduceOp.MAX)
    reduce_(recs, dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    total_recs, total_terms, total_users = (float(x) for x in recs.tolist())
    ms_per_step = 1e3 * elapsed / a.steps

    # ---- roofline of the dominant kernel, from HIP events the library records on ITS stream around every launch.
    # Candidates: the co-rating row kernel (k_cooc_rm2, builds M) and the scoring kernels (k_score*, one family).
    # Algorithmic bytes are SURVEY.md 8d's: 8 B per unordered co-rating pair contribution / 4 B per evaluated log-term.
    st = stats[-1]
    ms_score = float(np.mean([s["ms_score"] for s in stats]))
    ms_cooc = float(np.mean([s["ms_cooc"] for s in stats]))
    terms_eval = st["log_terms_evaluated"] if st["log_terms_evaluated"] > 0 else st["log_terms"]
    unordered_pairs = (st["pair_contribs"] - st["nnz"]) // 2
    cand = {
        "k_cooc_rm2": {"ms": ms_cooc, "bytes": 8.0 * unordered_pairs, "launches": st["cooc_launches"]},
        "k_score": {"ms": ms_score, "bytes": 4.0 * terms_eval, "launches": st["score_launches"]},
    }
    dom = max(cand, key=lambda k: cand[k]["ms"])

    def roof(name):
        c = cand[name]
        ach = c["bytes"] / (c["ms"] * 1e-3) / 1e9 if c["ms"] > 0 else 0.0
        return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "launches_per_step": c["launches"],
                "avg_launch_ms": c["ms"] / max(1, c["launches"]),
                "algorithmic_bytes_per_launch": c["bytes"] / max(1, c["launches"])}

    roofline = roof(dom)
    if a.shape == "ml25m" and K == 1:
        # HBM-side bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command
        # (tools/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes; committed under profiles/
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
        if cands:
            with open(cands[-1]) as f:
                tj = json.load(f)
            if dom in tj:
                roofline["traffic"] = tj[dom]["hbm_bytes_per_launch"]
    other = roof("k_score" if dom != "k_score" else "k_cooc_rm2")
    other["log_terms_evaluated"] = terms_eval
    other["log_terms_reference"] = st["log_terms"]
    other["blocks_survived_frac"] = (st["blocks_survived"] / st["blocks_total"]) if st["blocks_total"] else None
    other["note"] = ("scoring family (seed pass, block-maximum bound pass, survivor pass); frac > 1 would mean the column "
                     "panels are served from L2 / Infinity Cache")

    out = {
        "metric": "top-N recs/sec (RM2), %s shape" % a.shape, "value": total_recs / (elapsed / a.steps), "unit": "recs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic ratings (%d users x %d items, %d nnz), RM2 top-%d, lambda %g, "
                               "numberOfClusters %d, users range-sharded over %d GPU(s)"
                               % (a.shape, facts["n_users"], facts["n_items"], facts["nnz"], top_n, a.lam, K, world),
                   "shape": a.shape, "top_n": top_n, "clusters": K, "lambda": a.lam, "nnz": facts["nnz"]},
        "lists_per_s": total_users / (elapsed / a.steps), "log_terms_per_step": total_terms,
        "phase_ms_rank0": {k: float(np.mean([s[k] for s in stats])) for k in ("ms_prepare", "ms_cooc", "ms_score", "ms_topn", "ms_total")},
        "datagen_s": gen_s, "roofline": roofline, "roofline_other_kernel": other,
    }

    if world > 1:
        n_runs = a.steps + a.warmup
        out["multi_gpu"] = {"mode": "statistics exchange only (matrix replicated per rank)" if collectives is None else
                            "fy_collectives over torch.distributed/%s: statistics all-gather + cooperative scoring of clusters "
                            "that span all ranks" % dist.get_backend(),
                            "collective_calls_per_step": None if collectives is None else
                            {k: v / n_runs for k, v in collectives.calls.items() if k != "bytes"},
                            "payload_bytes_per_rank_per_step": None if collectives is None else collectives.calls["bytes"] / n_runs}

    # ---- item-item similarity build on the same ratings (second headline unit: pairs/s)
    if not a.no_itemsim:
        try:
            sim_job = P.RowSimilarityJob(ctx)
            sim_job.run(ratings, maxSimilaritiesPerRow=100, rank=rank, world=world).close()
            fence()
            t0 = time.perf_counter()
            res = sim_job.run(ratings, maxSimilaritiesPerRow=100, rank=rank, world=world)
            fence()
            dt = time.perf_counter() - t0
            sst = res.stats
            res.close()
            # unordered_pairs is the data set's total (every rank reports the same number); the item rows are sharded
            pp = torch.tensor([float(sst["unordered_pairs"]), dt], dtype=torch.float64, device=dev)
            if world > 1:
                tmax = pp[1:2].clone()
                reduce_(tmax, dist.ReduceOp.MAX)
                dt = float(tmax.item())
            out["itemsim"] = {"metric": "item-sim pairs/sec (cosine, top-100)", "value": float(pp[0].item()) / dt,
                              "unit": "pairs/s", "seconds": dt, "ms_kernel_rank0": sst["ms_cooc"],
                              "roofline": {"bound": "hbm", "kernel": "k_cooc_itemsim",
                                           "achieved": 8.0 * sst["unordered_pairs"] / world / (sst["ms_cooc"] * 1e-3) / 1e9 if sst["ms_cooc"] > 0 else 0.0,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s"}}
            out["itemsim"]["roofline"]["frac"] = out["itemsim"]["roofline"]["achieved"] / HBM_PEAK_GBS
        except RuntimeError as e:
            out["itemsim"] = {"error": str(e)}

    # ---- the stage in front of the hot path: PPC factorisation (k = 50) + cluster assignment on the same
    # ratings.  Algorithmic bytes per iteration: both SpMMs read 12 B per rating and gather one k-row of the dense factor
    # (8 k B) per rating; the updates stream H, W, X once.
    if world == 1 and not a.no_factorization:
        try:
            kf, iters = 50, 4
            rng = np.random.Generator(np.random.PCG64(11))
            H0 = rng.random((facts["n_users"], kf)) + 0.01
            W0 = rng.random((facts["n_items"], kf)) + 0.01
            fconf = P.Configuration()
            for kk, vv in (("numberOfUsers", facts["n_users"]), ("numberOfItems", facts["n_items"]), ("numberOfClusters", kf),
                           ("numberOfIterations", iters), ("normalizationFrequency", 12)):
                fconf.setInt(kk, vv)
            drv = P.NMFDriver(fconf, ctx, ppc=True)
            drv.run(ratings, H0, W0)
            fence()
            t0 = time.perf_counter()
            Hn, _ = drv.run(ratings, H0, W0)
            fence()
            dt_f = time.perf_counter() - t0
            Hd = torch.from_numpy(Hn).to(dev)
            job_c = P.ClusterAssignmentJob(ctx)
            job_c.run(Hd, first_user=1)
            fence()
            t0 = time.perf_counter()
            _, _, counts = job_c.run(Hd, first_user=1)
            fence()
            dt_c = time.perf_counter() - t0
            gpu_ms = drv.stats["ms_total"] - drv.stats["ms_prepare"]
            bytes_it = 2.0 * facts["nnz"] * (12 + 8 * kf) + 3.0 * 8 * kf * (facts["n_users"] + facts["n_items"])
            out["factorization"] = {"what": "PPC, k = %d, %d iterations (host H/W in and out) + cluster assignment" % (kf, iters),
                                    "seconds": dt_f, "ms_per_iteration_gpu": gpu_ms / iters, "ms_prepare": drv.stats["ms_prepare"],
                                    "roofline": {"bound": "hbm", "achieved": bytes_it / (gpu_ms / iters * 1e-3) / 1e9,
                                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": bytes_it / (gpu_ms / iters * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "note": "ms_per_iteration_gpu still contains the H/W transfers of the call"},
                                    "cluster_assign_ms": 1e3 * dt_c, "users_per_s_cluster_assign": facts["n_users"] / dt_c,
                                    "largest_cluster": int(counts.max())}
        except RuntimeError as e:
            out["factorization"] = {"error": str(e)}

    if world == 1:
        # the boundary also takes host buffers (fy_ratings_create FY_HOST + result download): PCIe-inclusive rate, reported
        # beside the headline value, never as it
        hu, hi_, hs = user.cpu().numpy(), item.cpu().numpy(), score.cpu().numpy()
        t0 = time.perf_counter()
        rec = job.run((hu, hi_, hs), clustering=clustering)
        n_rows = len(rec.rows()["user"])
        dt = time.perf_counter() - t0
        rec.close()
        out["pcie_inclusive"] = {"value": n_rows / dt, "unit": "recs/s", "ms": 1e3 * dt,
                                 "note": "host COO -> HBM -> job -> rows back in host memory, one run"}
    if rank == 0 and world == 1 and not a.no_cpu:
        n_cpu = a.cpu_users or {"ml25m": 270, "netflix": 200, "ml1m": 800, "ml100k": 943}.get(a.shape, 200)
        out["cpu_baseline"] = cpu_baseline(S, a.shape, facts, a.lam, top_n, n_cpu)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    ratings.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
