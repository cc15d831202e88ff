#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the RM2 top-N scorer (and the item-item similarity build) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 one rank per GPU over
RCCL: either the driver launches this file under `python -m torch.distributed.run --nproc-per-node N ...` (RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or -- WORLD_SIZE unset -- this script starts the N ranks itself, as child
processes, BEFORE anything touches a GPU, and exits with their code.
One "step" = one complete RM2 job over the ML-25M-shaped synthetic ratings, which are resident in HBM when the timed
region starts: COO -> CSR/CSC, statistics, (all-gather), per-cluster co-rating matrix, scoring, top-N.
Rank 0 prints ONE JSON line.  `value` = top-N recommendation rows produced by all ranks per second.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md, "L2 (per XCD)": ~34.5 TB/s aggregate over the 8 XCDs
NUM_CUS, CLOCK_HZ = 256, 2.4e9   # MI355X_MICROARCH.md
DS_ADD_U64_CYCLES = 10.8         # profiles/r2/micro_lds_atomic_rate.txt (tools/micro/lds_atomic_rate.hip): cycles per ds_add_u64 wave instruction, random addresses, every CU
V_LOG_F32_PEAK = NUM_CUS * 4 * 16 / 4 * CLOCK_HZ   # transcendental (quarter) rate: 4 SIMDs x 16 lanes / 4 per clock per CU = 9.8e12 v_log_f32 per second
DTYPE = "f32 (scores and logs fp32 / v_log_f32, folded in fp64; co-rating sums 64-bit fixed point in LDS (fp64 fallback); G stored as 24-bit e7m17 floats scaled per cluster above 4096 items; ratings fp16 in the row kernel when exactly representable)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="ml25m", help="ml100k | ml1m | ml25m | netflix")
    ap.add_argument("--clusters", type=int, default=1, help="numberOfClusters (users hashed to clusters); 1 = one neighbourhood")
    ap.add_argument("--top-n", type=int, default=None)
    ap.add_argument("--lam", type=float, default=0.1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-itemsim", action="store_true", help="skip the item-item similarity leg")
    ap.add_argument("--no-factorization", action="store_true", help="skip the PPC factorisation + cluster assignment leg")
    ap.add_argument("--no-regime", action="store_true", help="skip the reference-regime legs (50 clusters; top-1000)")
    ap.add_argument("--cpu-users", type=int, default=0, help="users in the CPU sample cluster (0 = auto)")
    return ap.parse_args()


def launch_ranks(n):
    """WORLD_SIZE unset and --gpus N > 1: start the N ranks ourselves (torch.distributed.run, rendezvous on 127.0.0.1).
    Nothing in this process has touched a GPU yet; the ranks are fresh child processes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def host_cores():
    """Threads for the CPU legs: the cores this process may run on, at most 16 -- a one-GPU box of the pool is a 1/8 share
    (16 cores) of a 256-thread host, and the OpenMP loops of the oracles get slower, not faster, when they are spread over
    all 256 hardware threads (measured: 540 recs/s with 256 threads against 1013 with 16 on the same sample).  The JSON line
    states both numbers (cores, host_nproc)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    return max(1, min(16, allowed))


def source_rev():
    """Hash of the kernel sources the roofline numbers belong to (profiles/rN/traffic.json is stamped with it)."""
    h = hashlib.sha256()
    for f in ("fy_cooc.hpp", "fy_rm2.hip", "fy_rm2_kernels.hpp"):
        with open(os.path.join(ROOT, "filmyou-core_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_legs(S, shape, facts, lam, top_n, n_users_sample, np):
    """CPU baselines (kind "port": this image has no JVM for the reference itself), all on bounded samples of the same
    synthetic data set, on the host cores of the GPU box:
      cpu_baseline        the faithful oracle (the reference's brute-force loop nest), OpenMP over the target users, all cores
      cpu_baseline_1core  the same on one core = what Hadoop's LocalJobRunner (a serial reducer) would do
      cpu_baseline_gram   a CPU scorer restructured like the GPU path (cluster Gram + per-user correction): the "best CPU" line
    """
    import oracle
    cores = host_cores()
    rng = np.random.Generator(np.random.PCG64(7))
    out = {}

    def sample(n):
        users = np.sort(rng.choice(facts["n_users"], size=min(n, facts["n_users"]), replace=False)) + 1
        u, i, s, _ = S.generate(shape, users=users, device="cpu")
        return users, u.numpy(), i.numpy(), s.numpy()

    def leg(kind, n, threads, fn):
        users, u, i, s = sample(n)
        t0 = time.time()
        ref = fn(u, i, s, threads)
        dt = time.time() - t0
        recs = len(ref["rec_user"])
        return {"value": recs / dt, "unit": "recs/s", "cores": threads, "kind": "port", "host_nproc": os.cpu_count(),
                "sample": "one %d-user cluster sampled from the same %s-shaped data (%d ratings, %d candidate items, %.3g log-terms), "
                          "%s, %d thread(s), %.1f s" % (len(users), shape, len(u), len(ref["item_id"]), ref["log_terms"], kind, threads, dt),
                "seconds": dt, "log_terms_per_s": ref["log_terms"] / dt}

    brute = lambda u, i, s, th: oracle.rm2(u, i, s, lam=lam, number_of_items=facts["n_items"], number_of_recommendations=top_n,
                                           number_of_clusters=1, n_threads=th)
    out["cpu_baseline"] = leg("oracle/rm2_oracle.c (the reference's loop nest: U_c - 1 multiply-adds per log term)", n_users_sample, cores, brute)
    out["cpu_baseline"]["note"] = ("the reference's cost per term grows with the cluster size, the GPU path's does not: "
                                   "a GPU/CPU ratio taken from this line mostly measures that O(U_c) loop, not kernel quality")
    n1 = max(40, int(n_users_sample / max(1.0, cores ** (1.0 / 3.0)) / 1.6))
    out["cpu_baseline_1core"] = leg("oracle/rm2_oracle.c, serial (LocalJobRunner runs one reducer thread)", n1, 1, brute)
    if hasattr(oracle, "rm2_gram"):
        gram = lambda u, i, s, th: oracle.rm2_gram(u, i, s, lam=lam, number_of_items=facts["n_items"], number_of_recommendations=top_n,
                                                   number_of_clusters=1, n_threads=th)
        out["cpu_baseline_gram"] = leg("oracle/rm2_oracle.c:rm2o_run_gram (Gram-restructured CPU scorer, the GPU path's identity in fp64)",
                                       {"ml25m": 400, "netflix": 600}.get(shape, 800), cores, gram)
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    P = importlib.import_module("filmyou-core_amd")
    S = importlib.import_module("filmyou-core_amd.synth")
    par = importlib.import_module("filmyou-core_amd.parallel")
    # FY_BENCH_REHEARSAL=1: several ranks share GPU 0 over gloo -- only to rehearse the multi-rank control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("FY_BENCH_REHEARSAL") == "1"
    rank, local_rank, world = par.init_distributed(backend="gloo" if rehearsal else None)
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would be labelled with the wrong GPU count" % (a.gpus, world))
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

    ctx = P.Context(local_rank)
    ratings = P.Ratings(ctx, user, item, score)       # resident in HBM before the timed region
    # world > 1: RCCL collectives (fy_collectives): the item statistics are all-gathered and a cluster that spans all
    # ranks is scored cooperatively (row-sharded matrix build, reduce-scattered partial sums).  Transport: the compiled
    # RCCL binding of the library when it is present, else torch.distributed on the library's stream;
    # FY_BENCH_EXCHANGE_ONLY=1 keeps the round-1 flow (statistics exchange only, matrix replicated) for comparison
    exchange = collectives = None
    transport = None
    if world > 1:
        if os.environ.get("FY_BENCH_EXCHANGE_ONLY") == "1":
            exchange = par.StatsExchange(local_rank)
            transport = "statistics all-gather only (torch.distributed/%s)" % dist.get_backend()
        elif not rehearsal and hasattr(par, "RcclCollectives") and os.environ.get("FY_BENCH_TORCH_COLLECTIVES") != "1":
            # the compiled transport, or -- decided by ALL ranks together, so that no rank waits in a communicator the
            # others never joined -- torch.distributed's RCCL group on the job's stream (the same library underneath)
            why = ""
            try:
                collectives = par.RcclCollectives(ctx, rank, world)
            except RuntimeError as e:
                why = str(e)
            ok = torch.tensor([0.0 if why else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) >= 1.0:
                transport = "librccl.so through the library's compiled fy_collectives (ncclAllGather / ncclReduceScatter on the job's stream)"
            else:
                if collectives is not None:
                    collectives.close()
                collectives = par.TorchCollectives(local_rank)
                transport = ("torch.distributed/%s on the job's stream (the compiled fy_rccl transport did not start on every rank%s)"
                             % (dist.get_backend(), ": " + why if why else ""))
        else:
            collectives = par.TorchCollectives(local_rank)
            transport = "torch.distributed/%s on the job's stream" % dist.get_backend()

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

    def time_job(K, n_rec, steps, warmup, cache=False):
        """W untimed + K timed complete jobs between fences; returns (per-step stats, elapsed seconds = max over ranks, totals).
        cache=False: every job is COLD (FY_RM2_NO_CACHE: the ratings are sorted into CSR / CSC and the row kernel's tables are built
        inside the timed step, nothing is kept) -- what the headline value is measured on.  cache=True: the jobs after the first
        find those structures on the resident ratings object (same clustering): the WARM job, reported beside the headline."""
        clustering = None
        if K > 1:
            uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
            clustering = (uu, S.hash_clustering(uu, K))
        conf = P.Configuration()
        conf.set("lambda", repr(a.lam))
        conf.setInt("numberOfItems", facts["n_items"])
        conf.setInt("numberOfClusters", K)
        conf.setInt("numberOfRecommendations", n_rec)
        job = P.RM2Job(conf, ctx)

        def step():
            rec = job.run(ratings, clustering=clustering, rank=rank, world=world, exchange=exchange, collectives=collectives, cache=cache)
            st = rec.stats
            rec.close()
            return st

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        stats = [step() for _ in range(steps)]
        fence()
        elapsed = time.perf_counter() - t0
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tot = torch.tensor([float(stats[-1]["recs"]), float(stats[-1]["log_terms"]), float(stats[-1]["users_scored"])],
                           dtype=torch.float64, device=dev)
        reduce_(tt, dist.ReduceOp.MAX)
        reduce_(tot, dist.ReduceOp.SUM)
        return stats, float(tt.item()), [float(x) for x in tot.tolist()], (job, clustering)

    K = a.clusters
    stats, elapsed, (total_recs, total_terms, total_users), (job, clustering) = time_job(K, top_n, a.steps, a.warmup)
    ms_per_step = 1e3 * elapsed / a.steps

    # ---- roofline of the dominant kernel, from HIP events the library records on ITS stream around every launch.
    # Candidates: the co-rating row kernel (k_cooc_rm2, builds M) and the scoring kernels (k_score*, one family).
    # Algorithmic bytes are SURVEY.md 8d's: 8 B per unordered co-rating pair contribution / 4 B per evaluated log-term.
    st = stats[-1]
    mean = lambda k: float(np.mean([s[k] for s in stats]))
    ms_score, ms_cooc = mean("ms_score"), mean("ms_cooc")
    terms_eval = st["log_terms_evaluated"] if st["log_terms_evaluated"] > 0 else st["log_terms"]
    unordered_pairs = (st["pair_contribs"] - st["nnz"]) // 2
    rev = source_rev()
    # HBM-side bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command
    # (tools/profile_round.sh), corrected as MI355X_MICROARCH.md prescribes; committed under profiles/ and stamped with
    # the hash of the kernel sources it was measured on -- a stale file gives null, not an old number
    traffic, traffic_note = {}, "no profiles/r*/traffic.json"
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if cands and a.shape == "ml25m" and K == 1:
        with open(cands[-1]) as f:
            tj = json.load(f)
        if tj.get("source_rev") == rev:
            traffic, traffic_note = tj, "%s (rocprofv3 --pmc, same kernel sources %s)" % (os.path.relpath(cands[-1], ROOT), rev)
        else:
            traffic_note = "%s was measured on kernel sources %s, this build is %s: not reported" % (os.path.relpath(cands[-1], ROOT), tj.get("source_rev"), rev)

    roofline = {"bound": "hbm", "kernel": "k_cooc_rm2", "achieved": 8.0 * unordered_pairs / (ms_cooc * 1e-3) / 1e9 if ms_cooc > 0 else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic.get("k_cooc_rm2", {}).get("hbm_bytes_per_launch"),
                "traffic_source": traffic_note, "launches_per_step": st["cooc_launches"],
                "avg_launch_ms": ms_cooc / max(1, st["cooc_launches"]),
                "algorithmic_bytes_per_launch": 8.0 * unordered_pairs / max(1, st["cooc_launches"]),
                "share_of_step": ms_cooc / ms_per_step if ms_per_step > 0 else None}
    roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
    roofline["unit_note"] = ("achieved / frac use SURVEY 8d's unit, 8 B per unordered co-rating pair; the packed symmetric walk reads 4 B per pair, "
                             "so this is the survey's model, not the kernel's own bytes: see frac_own_bytes and lds_atomic_floor_ms")
    if roofline["frac"] > 1.0:
        roofline["note"] = ("SURVEY 8d's unit (8 B per unordered co-rating pair) overstates this launch: the packed symmetric walk reads 4 B per unordered "
                            "pair and the column chunks stay in L2 / Infinity Cache; the fraction is the model's, not measured HBM traffic")
    # The kernel's OWN minimum traffic: one 4-byte packed entry per unordered pair, 12 B of descriptor per <= 64-entry segment, and the
    # matrix / panel / bound rows it stores (fy_stats::cooc_segments, cooc_matrix_bytes) -- against the same 8 TB/s
    own_bytes = 4.0 * unordered_pairs + 12.0 * st.get("cooc_segments", 0) + float(st.get("cooc_matrix_bytes", 0))
    roofline["own_bytes_per_step"] = own_bytes
    roofline["frac_own_bytes"] = own_bytes / (ms_cooc * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_cooc > 0 else None
    # ... and what actually bounds it on chip: every segment is one ds_add_u64 wave instruction on random LDS columns
    # (10.8 cycles each, measured with every CU busy): the LDS-atomic floor of the launch, whatever the memory system delivers
    floor_ms = st.get("cooc_segments", 0) * DS_ADD_U64_CYCLES / (NUM_CUS * CLOCK_HZ) * 1e3
    roofline["lds_atomic_floor_ms"] = floor_ms
    roofline["frac_of_lds_atomic_floor"] = floor_ms / ms_cooc if ms_cooc > 0 else None
    # the symmetric walk leaves the lower triangle to the mirror pass (k_mirror_tiles / k_mirror_diag): the matrix build as a whole
    ms_mirror = mean("ms_mirror")
    roofline["with_mirror_pass"] = {"ms": ms_cooc + ms_mirror, "achieved": 8.0 * unordered_pairs / ((ms_cooc + ms_mirror) * 1e-3) / 1e9 if ms_cooc > 0 else 0.0,
                                    "frac": 8.0 * unordered_pairs / ((ms_cooc + ms_mirror) * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_cooc > 0 else 0.0,
                                    # (the mirror pass is two launches per job: bytes per launch x launches / jobs profiled)
                                    "traffic": ((traffic["k_cooc_rm2"]["hbm_bytes_per_launch"] + traffic["k_mirror"]["hbm_bytes_per_launch"]
                                                 * traffic["k_mirror"]["launches"] / max(1, traffic["k_cooc_rm2"]["launches"]))
                                                if "k_mirror" in traffic and "k_cooc_rm2" in traffic else None)}
    # The scoring family is NOT priced against HBM: the branch and bound evaluates ~1.4 % of the reference's log terms and
    # the column panels it reads (seed columns, block maxima) live in L2 / Infinity Cache, so an HBM fraction means nothing
    # there (round 1 printed 1.09).  Its bound is the L2: 4 B per EVALUATED log term against the aggregate L2 bandwidth.
    other = {"bound": "l2", "kernel": "k_score family (seed + bound pass, survivor pass)",
             "achieved": 4.0 * terms_eval / (ms_score * 1e-3) / 1e9 if ms_score > 0 else 0.0, "peak": L2_PEAK_GBS, "unit": "GB/s",
             "log_terms_evaluated": terms_eval, "log_terms_reference": st["log_terms"],
             "reference_terms_per_s": st["log_terms"] / (ms_score * 1e-3) if ms_score > 0 else None,
             "blocks_survived_frac": (st["blocks_survived"] / st["blocks_total"]) if st["blocks_total"] else None,
             "traffic": traffic.get("k_score", {}).get("hbm_bytes_per_launch"), "ms": ms_score,
             "note": "4 B per evaluated log term against the L2; SURVEY 8d's unit (4 B x the reference's terms) does not apply: "
                     "98.6 % of those terms are excluded by the exact bound, never read"}
    other["frac"] = other["achieved"] / L2_PEAK_GBS
    # CU-side bytes (L2 -> CU) of the family from the TCP_TCC_READ_REQ pass of tools/profile_round.sh, when the stamped traffic.json
    # holds it: measured bytes / ms against the aggregate L2 bandwidth (requests priced at the bytes per request calibrated on the
    # same run's k_isim_sweep, whose 14 GB streaming read is known)
    l2 = traffic.get("k_score", {}).get("l2_read_bytes_per_step")
    other["l2_to_cu_bytes_per_step"] = l2
    other["frac_measured_l2_bytes"] = (l2 / (ms_score * 1e-3) / 1e9 / L2_PEAK_GBS) if (l2 and ms_score > 0) else None
    phases = {k: mean(k) for k in ("ms_prepare", "ms_tables", "ms_cooc", "ms_mirror", "ms_score", "ms_topn", "ms_total")}
    phases["ms_job"] = phases["ms_prepare"] + phases["ms_total"]            # ms_total = everything after prepare (HIP events)
    phases["ms_other_in_job"] = phases["ms_total"] - phases["ms_tables"] - phases["ms_cooc"] - phases["ms_mirror"] - phases["ms_score"] - phases["ms_topn"]
    phases["ms_host_gaps"] = ms_per_step - phases["ms_job"]                  # wall clock outside the two event spans
    # The phases no kernel roofline covers, each against HBM with a stated byte model (what the phase must move at the least):
    #   prepare  read the COO (12 B per rating), write the CSR (8 B) and the CSC (12 B per rating: slot, rating, pair)
    #   tables   read CSR + CSC once (20 B per rating), write the packed CSR (4 B per rating) and the segment table (12 B per segment)
    #   mirror   read and write the lower triangle of the stored matrix (half of what the row kernels stored, twice)
    nnz = float(st["nnz"])
    pm = {"prepare": 32.0 * nnz, "tables": 24.0 * nnz + 12.0 * st.get("cooc_segments", 0), "mirror": float(st.get("cooc_matrix_bytes", 0))}
    roofline_phases = {}
    for name, b in pm.items():
        ms = phases["ms_" + name]
        roofline_phases[name] = {"bound": "hbm", "model_bytes": b, "ms": ms, "achieved": b / (ms * 1e-3) / 1e9 if ms > 0 else None, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
    out = {
        "metric": "top-N recs/sec (RM2), %s shape" % a.shape, "value": total_recs / (elapsed / a.steps), "unit": "recs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic ratings (%d users x %d items, %d nnz), RM2 top-%d, lambda %g, "
                               "numberOfClusters %d, users range-sharded over %d GPU(s)"
                               % (a.shape, facts["n_users"], facts["n_items"], facts["nnz"], top_n, a.lam, K, world),
                   "shape": a.shape, "top_n": top_n, "clusters": K, "lambda": a.lam, "nnz": facts["nnz"],
                   "headline": "one neighbourhood (numberOfClusters 1) is the hardest case and one the reference cannot run "
                               "(its dense cache would need 77 TB); the reference's own regime is in reference_regime"},
        "cold_or_warm": "cold: every timed job sorts the resident COO ratings into CSR / CSC and builds its tables (FY_RM2_NO_CACHE); see `warm`",
        "lists_per_s": total_users / (elapsed / a.steps), "log_terms_per_step": total_terms,
        "phase_ms_rank0": phases, "datagen_s": gen_s,
        "job_stats_rank0": {k: int(st[k]) for k in ("n_clusters_nonempty", "cooc_launches", "score_launches", "panel_clusters", "blocks_total",
                                                     "blocks_survived", "stray_blocks", "bound_repairs", "prune_fallbacks", "topn_select_users", "rows_refined")},
        "roofline": roofline, "roofline_other_kernel": other, "roofline_phases": roofline_phases,
        "kernel_source_rev": rev,
    }

    if world > 1:
        n_runs = a.steps + a.warmup
        calls = getattr(collectives, "calls", None)
        out["multi_gpu"] = {"rccl_ranks": 0 if rehearsal else world, "transport": transport,
                            "collective_calls_per_step": None if not calls else {k: v / n_runs for k, v in calls.items() if k != "bytes"},
                            "payload_bytes_per_rank_per_step": None if not calls else calls["bytes"] / n_runs}
        out["cpu_baseline"] = None
        out["cpu_baseline_note"] = "timed on rank 0 of the N = 1 run only (see that line)"
        # the reference's own regime on N ranks: 50 clusters, WHOLE clusters per rank (no Gram is built twice, only the item statistics
        # are exchanged: RM2Job.java:251 runs one reduce group per cluster) -- the case SURVEY.md 8e expects to scale best
        if not a.no_regime and a.shape == "ml25m" and K == 1:
            try:
                s50, el50, (r50, t50, u50), _ = time_job(50, top_n, 2, 1)
                out["multi_gpu"]["clusters_50_top_%d" % top_n] = {
                    "value": r50 / (el50 / 2), "unit": "recs/s", "ms_per_step": 1e3 * el50 / 2,
                    "phase_ms_rank0": {k: float(np.mean([x[k] for x in s50])) for k in ("ms_prepare", "ms_tables", "ms_cooc", "ms_mirror", "ms_score", "ms_topn", "ms_total")},
                    "clusters_nonempty": int(s50[-1]["n_clusters_nonempty"]), "panel_clusters_rank0": int(s50[-1]["panel_clusters"]),
                    "note": "numberOfClusters 50, users hashed to clusters; every rank scores whole clusters (contiguous runs of the cluster order, balanced by log terms)"}
            except RuntimeError as e:
                out["multi_gpu"]["clusters_50_top_%d" % top_n] = {"error": str(e)}

    # ---- the warm job: same ratings object, same clustering -- the CSR / CSC, the per-item statistics and the row kernel's tables
    # of the previous job are found on the ratings object (fy_stats.prepared_from_cache); never the headline value
    if world == 1:
        try:
            sw, elw, (rw, _, _), _ = time_job(K, top_n, max(3, min(a.steps, 10)), 2, cache=True)
            nw = len(sw)
            out["warm"] = {"value": rw / (elw / nw), "unit": "recs/s", "ms_per_step": 1e3 * elw / nw,
                           "prepared_from_cache": int(sw[-1]["prepared_from_cache"]), "tables_from_cache": int(sw[-1]["tables_from_cache"]),
                           "phase_ms": {k: float(np.mean([x[k] for x in sw])) for k in ("ms_prepare", "ms_tables", "ms_cooc", "ms_mirror", "ms_score", "ms_topn", "ms_total")},
                           "note": "jobs after the first over the same resident ratings and clustering: nothing is sorted, no table is rebuilt"}
        except RuntimeError as e:
            out["warm"] = {"error": str(e)}

    # ---- the reference's own operating regime, beside the headline: many clusters (numberOfClusters is a required option,
    # 50 in T/rmrecommender/TestRMRecommenderJob.java:49) and the default list length 1000 (RMRecommenderDriver.java:95)
    if world == 1 and not a.no_regime and a.shape == "ml25m" and K == 1:
        reg = {}
        for name, kk, nn in (("clusters_50_top_50", 50, top_n), ("clusters_50_top_1000", 50, 1000), ("clusters_1_top_1000", 1, 1000)):
            try:
                s2, el2, (r2, t2, u2), _ = time_job(kk, nn, 2, 1)
                reg[name] = {"value": r2 / (el2 / 2), "unit": "recs/s", "ms_per_step": 1e3 * el2 / 2, "lists_per_s": u2 / (el2 / 2),
                             "log_terms_per_s": t2 / (el2 / 2),
                             "phase_ms": {k: float(np.mean([x[k] for x in s2])) for k in ("ms_prepare", "ms_tables", "ms_cooc", "ms_mirror", "ms_score", "ms_topn", "ms_total")},
                             "pruned": bool(s2[-1]["blocks_total"] > 0), "panel_clusters": int(s2[-1]["panel_clusters"]),
                             # a job that takes the plain full pass evaluates every log term: its ceiling is the v_log_f32 issue rate
                             "full_pass_frac_of_v_log_f32_peak": (t2 / (float(np.mean([x["ms_score"] for x in s2])) * 1e-3) / V_LOG_F32_PEAK)
                                                                 if (s2[-1]["blocks_total"] == 0 and np.mean([x["ms_score"] for x in s2]) > 0) else None,
                             # ... and by SURVEY 8d's byte model (4 B per log term against HBM) over the whole job; measured beside it
                             # (profiles/r4/k50_n1000_pmc_*.txt): 2.8 B of fabric reads per term, 10.9 VALU instructions per term, 66 - 84 % VALU-busy
                             "full_pass_frac_of_hbm_by_survey_unit": (4.0 * t2 / (el2 / 2) / 1e9 / HBM_PEAK_GBS) if s2[-1]["blocks_total"] == 0 else None,
                             "log_terms_evaluated": int(s2[-1]["log_terms_evaluated"]) if s2[-1]["blocks_total"] else int(s2[-1]["log_terms"]),
                             "blocks_survived_frac": (s2[-1]["blocks_survived"] / s2[-1]["blocks_total"]) if s2[-1]["blocks_total"] else None,
                             "stray_blocks": int(s2[-1]["stray_blocks"]), "bound_repairs": int(s2[-1]["bound_repairs"]), "rows_refined": int(s2[-1]["rows_refined"])}
            except RuntimeError as e:
                reg[name] = {"error": str(e)}
        out["reference_regime"] = reg

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
                              "ms_prepare": sst["ms_prepare"], "ms_tables": sst["ms_tables"], "ms_total": sst["ms_total"],
                              "build": "symmetric (upper triangle by the RM2 row kernel + band sweep)" if sst["isim_candidates"] else "row at a time",
                              "sweep_candidates": sst["isim_candidates"], "rows_redone_exactly": sst["isim_redone_rows"],
                              "roofline": {"bound": "hbm", "kernel": "k_cooc_rm2 + k_isim_sweep + k_isim_finish" if sst["isim_candidates"] else "k_cooc_itemsim",
                                           "achieved": 8.0 * sst["unordered_pairs"] / world / (sst["ms_cooc"] * 1e-3) / 1e9 if sst["ms_cooc"] > 0 else 0.0,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s"}}
            out["itemsim"]["roofline"]["frac"] = out["itemsim"]["roofline"]["achieved"] / HBM_PEAK_GBS
            if traffic and sst["isim_candidates"]:
                fam = [traffic.get(k, {}).get("hbm_bytes_per_launch") for k in ("itemsim_walk", "k_isim_sweep", "k_isim_finish")]
                out["itemsim"]["roofline"]["traffic"] = sum(fam) if all(x is not None for x in fam) else None
                out["itemsim"]["roofline"]["algorithmic_bytes"] = 8.0 * sst["unordered_pairs"]
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
            gpu_ms = drv.stats["ms_cooc"]            # the iterations alone (HIP events around the loop: H / W resident in HBM)
            bytes_it = 2.0 * facts["nnz"] * (12 + 8 * kf) + 3.0 * 8 * kf * (facts["n_users"] + facts["n_items"])
            out["factorization"] = {"what": "PPC, k = %d, %d iterations (host H/W in and out) + cluster assignment" % (kf, iters),
                                    "seconds": dt_f, "ms_per_iteration_gpu": gpu_ms / iters, "ms_prepare": drv.stats["ms_prepare"],
                                    "ms_call_with_host_transfers": drv.stats["ms_total"],
                                    "roofline": {"bound": "hbm", "achieved": bytes_it / (gpu_ms / iters * 1e-3) / 1e9,
                                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": bytes_it / (gpu_ms / iters * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "note": "iterations alone (H / W in HBM); algorithmic bytes: two SpMMs of 12 B + one k-row of the dense factor per rating, "
                                                         "the updates stream H, W, X once"},
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
        out.update(cpu_legs(S, a.shape, facts, a.lam, top_n, n_cpu, np))
        bl = importlib.import_module("tools.cpu_baselines") if os.path.exists(os.path.join(ROOT, "tools", "cpu_baselines.py")) else None
        if bl is not None:
            out.update(bl.extra_legs(P, S, ctx, a.lam, device=dev))
    elif rank == 0 and world == 1:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    ratings.close()
    if collectives is not None and hasattr(collectives, "close"):
        collectives.close()          # before the context: the communicator drains the context's stream
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
