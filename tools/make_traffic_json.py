#!/usr/bin/env python3
"""profiles/rN/summary_pmc_{fetch,write}.txt -> profiles/rN/traffic_k_score.json (HBM-side bytes per launch of the
dominant kernel, gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md: the counter reads half of a wide coalesced read)."""
import json
import re
import sys

d = sys.argv[1]
f = open(d + "/summary_pmc_fetch.txt").read()
w = open(d + "/summary_pmc_write.txt").read()
m = re.search(r"(k_score<[^>]*>)\s+FETCH_SIZE=([0-9.e+]+) \(n=(\d+)\)", f)
name, fs, n = m.group(1), float(m.group(2)), int(m.group(3))
ws = float(re.search(re.escape(name) + r"\s+WRITE_SIZE=([0-9.e+]+)", w).group(1))
out = {"kernel": name, "launches": n, "fetch_size_kb_sum": fs, "write_size_kb_sum": ws,
       "hbm_bytes_per_launch": (2 * fs + ws) * 1024 / n,
       "note": "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md; summed over the launches of two jobs and divided by "
               "their number; config: ml25m shape, numberOfClusters 1, top-50 (python3 bench.py --steps 1 --warmup 1 --no-cpu)"}
json.dump(out, open(d + "/traffic_k_score.json", "w"), indent=1)
print(json.dumps(out))
