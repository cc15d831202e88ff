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


def merge(*fams):
    """several kernels that together are one pass of a job (the lazy mirror: tiles + block maxima + flags): bytes per job"""
    n = max(1, max(f["launches"] for f in fams))
    fs = sum(f["fetch_size_kb_sum"] for f in fams)
    ws = sum(f["write_size_kb_sum"] for f in fams)
    return {"launches": n, "fetch_size_kb_sum": fs, "write_size_kb_sum": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024 / n,
            "note": "sum over the kernels of the pass; `launches` = launches of its most frequent kernel"}


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
       "k_cooc_rm2": family("k_cooc_rm2<true, unsigned long long"), "k_mirror": merge(family("k_mirror"), family("k_colmax_upper"), family("k_flag_surviving")), "k_score": family("k_score"),
       "itemsim_walk": family("k_cooc_rm2<true, unsigned int"), "k_isim_sweep": family("k_isim_sweep"), "k_isim_finish": family("k_isim_finish"),
       "note": "FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md; summed over the launches of the profiled jobs and divided by "
               "their number; config: ml25m shape, numberOfClusters 1, top-50 (python3 bench.py --steps 1 --warmup 1 --no-cpu)"}
# CU-side bytes of the scoring family: L1 -> L2 read requests (TCP_TCC_READ_REQ_sum).  The counter's unit is not documented for gfx950 in
# MI355X_MICROARCH.md, so it is calibrated on the SAME run: k_colmax_upper streams whole 768-byte row segments of known count
# (sum_{B >= seed blocks} (tiles behind B) x 256 rows x 768 B: 26 335 tiles = 5.178 GB at ML-25M shape with a one-block seed) -- round 4
# measured 5.178e9 / 4.048e7 = 127.9 bytes per request, i.e. one 128-byte line.  (k_isim_sweep, whose 14 GB are also known, gives 97.5:
# its 256-byte column segments start at arbitrary 4-byte offsets and touch three lines.)
cal = l2_requests("k_colmax_upper")
fam = [(k, l2_requests(k)) for k in ("k_score_sup", "k_score<", "k_score_blocks", "k_topn_seed")]
if cal and any(v for _, v in fam):
    n_items, nblk, seed_blocks = 59047, 231, 1          # ML-25M shape (the only shape profiled), top-50: one seed block
    tiles = sum((nblk - 1) - B for B in range(seed_blocks, nblk - 1))
    colmax_bytes = tiles * 256 * 768.0
    bytes_per_req = colmax_bytes / (cal[0] / cal[1])
    jobs = max(1, out["k_cooc_rm2"]["launches"])     # one row-kernel launch per profiled job
    req = sum(v[0] for _, v in fam if v)
    out["k_score"]["l2_read_requests_per_step"] = req / jobs
    out["k_score"]["l2_bytes_per_request_calibrated"] = bytes_per_req
    out["k_score"]["l2_read_bytes_per_step"] = req / jobs * bytes_per_req
    out["k_score"]["l2_family"] = {k: (v[0] / jobs if v else None) for k, v in fam}
    out["l2_calibration"] = {"kernel": "k_colmax_upper", "known_bytes_per_launch": colmax_bytes, "requests_per_launch": cal[0] / cal[1]}
json.dump(out, open(d + "/traffic.json", "w"), indent=1)
print(json.dumps(out))
