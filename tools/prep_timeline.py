"""Prints the kernel timeline of the LAST cold prepare in a rocprofv3 --kernel-trace database (tools/prep_probe.py under the profiler):
    python tools/prep_timeline.py <results.db> [first-kernel-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "k_user_item_keys"
    rows = list(db.execute("select name, start, end from kernels order by start"))
    idx = [k for k, r in enumerate(rows) if first in r[0]]
    a = idx[-1]
    t0 = rows[a][1]
    busy = 0
    for r in rows[a:]:
        busy += r[2] - r[1]
        print("%8.1f us  +%7.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:110]))
    print("kernels: %d, busy %.1f us of %.1f us" % (len(rows) - a, busy / 1e3, (rows[-1][2] - t0) / 1e3))


if __name__ == "__main__":
    main()
