#!/usr/bin/env python3
"""BASELINE.md's results table, generated from the recorded default `python bench.py` lines of every round
(profiles/rN/bench_default_run.json) and the driver's own records (BENCH_rNN.json): one column per round, FINAL numbers only.

    python tools/baseline_table.py > /tmp/table.md
"""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def fmt(x, unit="", nd=1):
    if x is None:
        return "—"
    return ("%." + str(nd) + "f%s") % (x, unit)


def main():
    rounds = []
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*"))):
        f = os.path.join(d, "bench_default_run.json")
        if os.path.exists(f):
            rounds.append((os.path.basename(d), load(f)))
    driver = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json"))):
        r = int(re.search(r"r(\d+)", os.path.basename(f)).group(1))
        p = json.load(open(f)).get("parsed") or {}
        if p.get("value"):
            driver["r%d" % r] = p
    rows = []

    def row(name, fn):
        rows.append("| " + name + " | " + " | ".join(fn(r, p) for r, p in rounds) + " |")

    print("| quantity (one MI355X, `python bench.py`, synthetic ML-25M shape unless said otherwise) | " + " | ".join("round %s" % r[1:] for r, _ in rounds) + " |")
    print("|---|" + "---|" * len(rounds))
    row("**headline: RM2 top-50, ONE cluster (162 541-user neighbourhood), cold job** — M recs/s (ms per job), builder's record",
        lambda r, p: "**%s** (%s ms)" % (fmt(p["value"] / 1e6), fmt(p["ms_per_step"], nd=2)))
    row("the same, the DRIVER's end-of-round run (`BENCH_rNN.json`)",
        lambda r, p: ("%s (%s ms)" % (fmt(driver[r]["value"] / 1e6), fmt(driver[r]["ms_per_step"], nd=2))) if r in driver else "—")
    row("warm job (structures of the previous job found on the ratings object), ms", lambda r, p: fmt((p.get("warm") or {}).get("ms_per_step"), nd=1))
    row("phases of the cold job, ms: prepare / tables / row kernel / mirror / scoring / top-N",
        lambda r, p: " / ".join(fmt(p["phase_ms_rank0"].get(k), nd=1) for k in ("ms_prepare", "ms_tables", "ms_cooc", "ms_mirror", "ms_score", "ms_topn")))
    row("row kernel: fraction of HBM peak by SURVEY 8d's 8 B per pair / by its own bytes / share of its LDS-atomic floor",
        lambda r, p: "%s / %s / %s" % (fmt(p["roofline"].get("frac"), nd=2), fmt(p["roofline"].get("frac_own_bytes"), nd=2), fmt(p["roofline"].get("frac_of_lds_atomic_floor"), nd=2)))
    for key, label in (("clusters_50_top_50", "50 clusters (the reference's regime), top-50, ms per job"),
                       ("clusters_50_top_1000", "50 clusters, top-1000 (the reference's defaults), ms per job"),
                       ("clusters_1_top_1000", "one cluster, top-1000, ms per job")):
        row(label, lambda r, p, key=key: fmt(((p.get("reference_regime") or {}).get(key) or {}).get("ms_per_step"), nd=1))
    row("item-item cosine build, top-100: pairs/s (whole build ms; kernels' fraction of HBM peak by 8 B per pair)",
        lambda r, p: ("%.2e (%s ms; %s)" % (p["itemsim"]["value"], fmt(1e3 * p["itemsim"]["seconds"], nd=1), fmt((p["itemsim"].get("roofline") or {}).get("frac"), nd=2)))
        if p.get("itemsim") and "value" in p["itemsim"] else "—")
    row("like for like (ML-1M shape, 50 clusters, top-50), GPU ms / faithful CPU oracle s on 16 cores / Gram CPU s",
        lambda r, p: ("%s / %s / %s" % (fmt(p["like_for_like_ml1m_k50"].get("gpu_ms"), nd=1), fmt(p["like_for_like_ml1m_k50"].get("cpu_faithful_seconds"), nd=1),
                                        fmt(p["like_for_like_ml1m_k50"].get("cpu_gram_seconds"), nd=1))) if p.get("like_for_like_ml1m_k50") else "—")
    row("CPU baselines (`kind: port`): faithful oracle 16 cores / 1 core / Gram-restructured 16 cores, recs/s",
        lambda r, p: " / ".join(fmt((p.get(k) or {}).get("value"), nd=0) for k in ("cpu_baseline", "cpu_baseline_1core", "cpu_baseline_gram")))
    row("PPC factorisation k = 50, ms per iteration (round 4: iterations alone)", lambda r, p: fmt((p.get("factorization") or {}).get("ms_per_iteration_gpu"), nd=1))
    row("PCIe-inclusive job (host COO in, rows back), ms", lambda r, p: fmt((p.get("pcie_inclusive") or {}).get("ms"), nd=1))
    print("\n".join(rows))


if __name__ == "__main__":
    main()
