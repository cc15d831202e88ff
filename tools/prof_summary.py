#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC) per kernel name.  Usage: prof_summary.py <dir> [substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0])
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0]
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                agg[name][0] += 1
                agg[name][1] += dur
        tot = sum(v[1] for v in agg.values())
        print("# %s  total kernel ms %.3f" % (os.path.relpath(path, d), tot))
        print("%-60s %8s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
        for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('PROF_TOP', '25'))]:
            if want in name:
                print("%-60s %8d %12.3f %12.4f %6.1f" % (name[:60], n, ms, ms / n, 100 * ms / tot))
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0]
                agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(name, r["Counter_Name"])] += 1
        print("# %s" % os.path.relpath(path, d))
        for name, cs in sorted(agg.items()):
            if want in name:
                print("%-50s " % name[:50] + "  ".join("%s=%.6g (n=%d)" % (k, v, cnt[(name, k)]) for k, v in sorted(cs.items())))


if __name__ == "__main__":
    main()
