#!/usr/bin/env python3
"""GPU busy analysis of a rocprofv3 (rocpd sqlite) kernel trace: union of kernel intervals vs wall span, overall and
for the window between the first and last launch of kernels matching a substring.
Usage: python tools/rocpd_busy.py trace_results.db [substring]"""
import sqlite3
import sys


def union(iv):
    iv.sort()
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def main(path, sub=None):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_time"
    e_col = "end" if "end" in cols else "end_time"
    rows = con.execute(f"select name, {s_col}, {e_col} from kernels").fetchall()
    if sub:
        sel = [(s, e) for n, s, e in rows if sub in n]
        lo, hi = min(s for s, _ in sel), max(e for _, e in sel)
        rows = [(n, s, e) for n, s, e in rows if s >= lo and e <= hi]
    iv = [(s, e) for _, s, e in rows]
    span = max(e for _, e in iv) - min(s for s, _ in iv)
    print(f"kernels {len(iv)}  span {span / 1e6:.3f} ms  busy(union) {union(list(iv)) / 1e6:.3f} ms  "
          f"sum {sum(e - s for s, e in iv) / 1e6:.3f} ms  avg concurrency {sum(e - s for s, e in iv) / max(union(list(iv)), 1):.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
