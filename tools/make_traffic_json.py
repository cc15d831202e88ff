#!/usr/bin/env python3
"""profiles/rN/summary_pmc_{fetch,write}.txt -> profiles/rN/traffic.json: HBM-side bytes per launch of the two kernel
families bench.py prices (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md: the counter reads half of a wide
coalesced read, so it is doubled; WRITE_SIZE is exact).  Usage: make_traffic_json.py profiles/rN"""
import hashlib
import json
import os
import re
import sys

d = sys.argv[1]
f = open(d + "/summary_pmc_fetch.txt").read()
w = open(d + "/summary_pmc_write.txt").read()


def family(prefix):
    fs = ws = 0.0
    n = 0
    prefix = re.escape(prefix)
    for m in re.finditer(r"^(?:void )?fy::(%s[^\s]*(?: [^\s=]+)*?)\s+FETCH_SIZE=([0-9.e+]+) \(n=(\d+)\)" % prefix, f, re.M):
        fs += float(m.group(2))
        n += int(m.group(3))
    for m in re.finditer(r"^(?:void )?fy::(%s[^\s]*(?: [^\s=]+)*?)\s+WRITE_SIZE=([0-9.e+]+)" % prefix, w, re.M):
        ws += float(m.group(2))
    return {"launches": n, "fetch_size_kb_sum": fs, "write_size_kb_sum": ws,
            "hbm_bytes_per_launch": (2 * fs + ws) * 1024 / max(n, 1)}


def source_rev():
    """the same hash bench.py computes: the file is only reported for the kernel sources it was measured on"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ("fy_cooc.hpp", "fy_rm2.hip", "fy_rm2_kernels.hpp"):
        with open(os.path.join(root, "filmyou-core_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def l2_requests(prefix):
    """TCP_TCC_READ_REQ_sum over the launches of a kernel family (summary_pmc_l2.txt), or None"""
    path = d + "/summary_pmc_l2.txt"
    if not os.path.exists(path):
        return None
    t = open(path).read()
    tot, n = 0.0, 0
    for m in re.finditer(r"^(?:void )?fy::(%s[^\s]*(?: [^\s=]+)*?)\s+.*?TCP_TCC_READ_REQ_sum=([0-9.e+]+) \(n=(\d+)\)" % re.escape(prefix), t, re.M):
        tot += float(m.group(2))
        n += int(m.group(3))
    return (tot, n) if n else None


out = {"source_rev": sys.argv[2] if len(sys.argv) > 2 else source_rev(),
       # the RM2 matrix build (64-bit fixed point) and the item-similarity walk (32-bit) are two instantiations of one kernel
       "k_cooc_rm2": family("k_cooc_rm2<true, unsigned long long"), "k_mirror": family("k_mirror"), "k_score": family("k_score"),
       "itemsim_walk": family("k_cooc_rm2<true, unsigned int"), "k_isim_sweep": family("k_isim_sweep"), "k_isim_finish": family("k_isim_finish"),
       "note": "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md; summed over the launches of the profiled jobs and divided by "
               "their number; config: ml25m shape, numberOfClusters 1, top-50 (python3 bench.py --steps 1 --warmup 1 --no-cpu)"}
# CU-side bytes of the scoring family: L1 -> L2 read requests, priced at the bytes per request calibrated on the SAME run's
# k_isim_sweep (I x ldm x 4 bytes per launch are known, see below) -- the counter's unit
# is not documented for gfx950 in MI355X_MICROARCH.md, so it is never used uncalibrated.  (The sweep reads every element of the upper
# triangle twice, once along its row and once down its column, by different workgroups: I x ldm x 4 bytes cross L1 -> L2 per launch.)
cal = l2_requests("k_isim_sweep")
fam = l2_requests("k_score")
if cal and fam:
    n_items, ldm = 59047, 59136                      # ML-25M shape (the only shape profiled)
    sweep_bytes = n_items * ldm * 4.0
    bytes_per_req = sweep_bytes / (cal[0] / cal[1])
    jobs = max(1, out["k_cooc_rm2"]["launches"])     # one row-kernel launch per profiled job
    out["k_score"]["l2_read_requests_per_step"] = fam[0] / jobs
    out["k_score"]["l2_bytes_per_request_calibrated"] = bytes_per_req
    out["k_score"]["l2_read_bytes_per_step"] = fam[0] / jobs * bytes_per_req
    out["l2_calibration"] = {"kernel": "k_isim_sweep", "known_bytes_per_launch": sweep_bytes, "requests_per_launch": cal[0] / cal[1]}
json.dump(out, open(d + "/traffic.json", "w"), indent=1)
print(json.dumps(out))
