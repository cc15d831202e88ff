"""Rehearsal of the cooperative multi-rank RM2 job on ONE GPU at full size.

`--world W` threads of this process play the ranks (parallel.ThreadCollectives, serialised: one rank computes at a
time), so that (a) the W-rank result can be compared row for row with the single-GPU result at ML-25M shape, and
(b) every rank's compute critical path -- the time it would need on a GPU of its own, communication excluded -- is
measured.  The communication volume per rank is reported for an xGMI estimate.  Not a benchmark: bench.py is."""
import argparse
import importlib
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml25m")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--top-n", type=int, default=50)
    ap.add_argument("--clusters", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    P = importlib.import_module("filmyou-core_amd")
    S = importlib.import_module("filmyou-core_amd.synth")
    par = importlib.import_module("filmyou-core_amd.parallel")
    dev = torch.device("cuda", 0)
    user, item, score, facts = S.generate(a.shape, device=dev)
    clustering = None
    if a.clusters > 1:
        uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
        clustering = (uu, S.hash_clustering(uu, a.clusters))
    conf = P.Configuration()
    conf.set("lambda", "0.1")
    conf.setInt("numberOfItems", facts["n_items"])
    conf.setInt("numberOfClusters", a.clusters)
    conf.setInt("numberOfRecommendations", a.top_n)

    single_rows = None
    single_ms = None
    if not a.no_check:
        ctx = P.Context(0)
        r = P.Ratings(ctx, user, item, score)
        job = P.RM2Job(conf, ctx)
        job.run(r, clustering=clustering).close()
        ctx.synchronize()
        t0 = time.perf_counter()
        rec = job.run(r, clustering=clustering)
        ctx.synchronize()
        single_ms = 1e3 * (time.perf_counter() - t0)
        single_rows = rec.rows()
        rec.close()
        r.close()
        ctx.close()

    W = a.world
    out = [None] * W
    err = [None] * W
    group = par.ThreadGroup(W, serialize=True)
    comms = [par.ThreadCollectives(group, k, 0) for k in range(W)]
    rep_gate = threading.Barrier(W)
    report = [None] * a.reps

    def body(rank):
        try:
            ctx = P.Context(0)          # the context (and its caching allocator) lives across the repetitions, like in bench.py
            r = P.Ratings(ctx, user, item, score)
            job = P.RM2Job(conf, ctx)
            for rep in range(a.reps):
                ctx.synchronize()
                rep_gate.wait()
                if rank == 0:
                    group.busy_s = [0.0] * W
                    group.busy_log = [[] for _ in range(W)]
                    for c in comms:
                        c.calls = {"all_gather": 0, "reduce_scatter_f32": 0, "bytes": 0}
                rep_gate.wait()
                comms[rank].enter()
                rec = job.run(r, clustering=clustering, rank=rank, world=W, collectives=comms[rank])
                ctx.synchronize()
                comms[rank].leave("tail")
                out[rank] = (rec.rows() if rep == a.reps - 1 and not a.no_check else None, dict(rec.stats))
                rec.close()
                rep_gate.wait()
                if rank == 0:
                    report[rep] = {"busy_ms": [round(1e3 * x, 2) for x in group.busy_s],
                                   "ms_cooc": [round(o[1]["ms_cooc"], 2) for o in out],
                                   "ms_prepare": [round(o[1]["ms_prepare"], 2) for o in out],
                                   "prepared_from_cache": [int(o[1]["prepared_from_cache"]) for o in out],   # rep 0 = cold jobs, later reps warm
                                   "comm_MB_sent_per_rank": round(comms[0].calls["bytes"] / 1e6, 1), "calls": dict(comms[0].calls),
                                   "phases_rank0": [(w, round(1e3 * d, 2)) for w, d in group.busy_log[0]],
                                   "phases_last_rank": [(w, round(1e3 * d, 2)) for w, d in group.busy_log[W - 1]]}
            r.close()
            ctx.close()
        except BaseException as e:
            err[rank] = e
            group.barrier.abort()
            rep_gate.abort()
            try:
                group.lock.release()
            except RuntimeError:
                pass

    th = [threading.Thread(target=body, args=(k,)) for k in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    res = {"shape": a.shape, "world": W, "single_gpu_ms": single_ms, "reps": report}
    if single_rows is not None:
        rows = {k: np.concatenate([o[0][k] for o in out]) for k in ("user", "item", "score")}
        ka = np.lexsort((rows["item"], rows["user"]))
        kb = np.lexsort((single_rows["item"], single_rows["user"]))
        same_pairs = (len(ka) == len(kb) and np.array_equal(rows["user"][ka], single_rows["user"][kb])
                      and np.array_equal(rows["item"][ka], single_rows["item"][kb]))
        res["rows"] = int(len(ka))
        res["same_user_item_pairs"] = bool(same_pairs)
        if same_pairs:
            x, y = rows["score"][ka].astype(np.float64), single_rows["score"][kb].astype(np.float64)
            res["max_rel_score_diff"] = float(np.max(np.abs(x - y) / np.maximum(1e-3, np.abs(y))))
        else:
            # near-ties at the cut-off may swap when the summation order changes: count the differing pairs
            sa = set(zip(rows["user"].tolist(), rows["item"].tolist()))
            sb = set(zip(single_rows["user"].tolist(), single_rows["item"].tolist()))
            res["pairs_only_in_sharded"] = len(sa - sb)
            res["pairs_only_in_single"] = len(sb - sa)
            # every such pair must sit AT its user's cut-off: its score equals the other run's last kept score within rounding
            def cutoffs(r, users):
                out = {}
                for uu in users:
                    m = r["user"] == uu
                    out[uu] = float(r["score"][m].min())
                return out
            users = sorted({p[0] for p in (sa - sb) | (sb - sa)})
            cs, cm = cutoffs(single_rows, users), cutoffs(rows, users)
            score_of = lambda r, pr: float(r["score"][(r["user"] == pr[0]) & (r["item"] == pr[1])][0])
            gaps = [abs(score_of(rows, pr) - cs[pr[0]]) / abs(cs[pr[0]]) for pr in sa - sb] + \
                   [abs(score_of(single_rows, pr) - cm[pr[0]]) / abs(cm[pr[0]]) for pr in sb - sa]
            res["max_rel_gap_of_swapped_pairs_to_cutoff"] = max(gaps) if gaps else 0.0
    print(json.dumps(res))


if __name__ == "__main__":
    main()
