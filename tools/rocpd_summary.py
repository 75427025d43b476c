#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
Usage: python tools/rocpd_summary.py gpurun_out/prof_r01/r01_results.db > profiles/<name>.txt"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':<86} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} "
          f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scr':>5}")
    for name, calls, tot, avg, mn, mx, vg, ag, sg, lds, scr in rows:
        short = name if len(name) <= 85 else name[:82] + "..."
        print(f"{short:<86} {calls:>6} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} "
              f"{100.0 * tot / total:>6.2f} {vg or 0:>5} {ag or 0:>5} {sg or 0:>5} {lds or 0:>7} {scr or 0:>5}")


if __name__ == "__main__":
    main(sys.argv[1])
