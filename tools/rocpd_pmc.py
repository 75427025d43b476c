#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 (rocpd sqlite) --pmc run.
Usage: python tools/rocpd_pmc.py <results.db>"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                       "from counters_collection group by kernel_name, counter_name order by avg(value) desc").fetchall()
    print(f"# rocprofv3 --pmc summary of {path} (value unit as the counter defines it: FETCH_SIZE/WRITE_SIZE = KiB)")
    print(f"{'kernel':<100} {'counter':<12} {'n':>4} {'avg':>16} {'min':>16} {'max':>16}")
    for k, c, n, a, mn, mx in rows:
        k = k if len(k) <= 99 else k[:96] + "..."
        print(f"{k:<100} {c:<12} {n:>4} {a:>16.1f} {mn:>16.1f} {mx:>16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
