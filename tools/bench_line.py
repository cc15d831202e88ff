#!/usr/bin/env python3
"""Prints the fields of a bench.py line (stdin) that are compared between experiments: tools/bench_line.py <label>"""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ph = {k: round(v, 2) for k, v in d["phase_ms_rank0"].items()}
warm = d.get("warm", {}).get("ms_per_step")
print(label, "value %.4g" % d["value"], "ms %.2f" % d["ms_per_step"], "warm %s" % (("%.2f" % warm) if warm else None), ph,
      "frac %.3f" % d["roofline"]["frac"])
if "job_stats_rank0" in d:
    print(label, d["job_stats_rank0"])
if "itemsim" in d and d["itemsim"]:
    i = d["itemsim"]
    print(label, "itemsim %.4g" % i["value"], "kernel ms %.2f" % i.get("ms_kernel_rank0", 0), "tables %.2f" % i.get("ms_tables", 0))
