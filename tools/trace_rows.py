#!/usr/bin/env python3
"""Per-call rows of a rocprofv3 kernel trace: tools/trace_rows.py <dir> <substr> [max] -> start offset (ms), duration (us), name."""
import csv
import glob
import os
import sys

d, want = sys.argv[1], sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(path)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    n = 0
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        if want in r["Kernel_Name"]:
            print("%10.3f ms  %9.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60]))
            n += 1
            if n >= limit:
                break
