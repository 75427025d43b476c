#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/{trace,fetch,write}_results.db into the committed summaries:
  profiles/<tag>_kernel_stats.txt   per-kernel calls / avg / min / max (rocprofv3 --kernel-trace --stats)
  profiles/<tag>_pmc.txt            per-kernel FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
  profiles/stack_pmc.json           HBM bytes per launch of the stacking kernel (bench.py's roofline.traffic)
FETCH_SIZE on gfx950 counts 128-byte reads as 64 (MI355X_MICROARCH.md, HBM section); calibrated on
this access pattern with the loads-only stage of tools/stack_ablate (4 GiB read -> counter 2 GiB),
so the read side is doubled.  WRITE_SIZE is exact for dword stores (fill kernel: 64 MiB -> 65536 KiB)."""
import contextlib
import io
import json
import os
import sqlite3
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import rocpd_pmc  # noqa: E402
import rocpd_summary  # noqa: E402


def capture(fn, path):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(path)
    return buf.getvalue()


def counter_avg(db, like):
    """Per-launch counter value of one stack operation = sum over the distinct kernels it launches (fast pass + general
    pass since the two-pass mode) of each kernel's average."""
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, avg(value) from counters_collection where kernel_name like ? group by kernel_name",
                       (like,)).fetchall()
    return sum(r[1] for r in rows), [r[0] for r in rows]


def main(tag, outdir=None, frames=64, pixels=4096 * 4096):
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    pdir = outdir or os.path.join(ROOT, "profiles")
    os.makedirs(pdir, exist_ok=True)
    with open(os.path.join(pdir, f"{tag}_kernel_stats.txt"), "w") as f:
        f.write(capture(rocpd_summary.main, os.path.join(d, "trace_results.db")))
    txt = capture(rocpd_pmc.main, os.path.join(d, "fetch_results.db")) + capture(rocpd_pmc.main, os.path.join(d, "write_results.db"))
    with open(os.path.join(pdir, f"{tag}_pmc.txt"), "w") as f:
        f.write(txt)
    fk, names = counter_avg(os.path.join(d, "fetch_results.db"), "%stack_sigma_clip_kernel<64%")
    wk, _ = counter_avg(os.path.join(d, "write_results.db"), "%stack_sigma_clip_kernel<64%")
    out = {"source": f"profiles/{tag}_pmc.txt", "kernels_per_launch": len(names), "frames": frames, "pixels": pixels,
           "FETCH_SIZE_KiB_avg": fk, "WRITE_SIZE_KiB_avg": wk, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024),
           "algorithmic_bytes": 4 * frames * pixels + 4 * pixels}
    out["traffic_over_algorithmic"] = round(out["hbm_bytes_per_launch"] / out["algorithmic_bytes"], 4)
    # the AVERAGE duration of one stack launch under `rocprofv3 --kernel-trace --stats` (fast pass + general pass): the figure a
    # reader of profiles/<tag>_kernel_stats.txt recomputes; bench.py prints it beside its own live measurements
    con = sqlite3.connect(os.path.join(d, "trace_results.db"))
    rows = con.execute("select name, count(*), avg(duration) from kernels where name like ? group by name", ("%stack_sigma_clip_kernel<64%",)).fetchall()
    out["profile_avg_ms"] = round(sum(r[2] for r in rows) / 1e6, 4)
    out["profile_kernels"] = [{"name": r[0][:90], "calls": r[1], "avg_us": round(r[2] / 1e3, 2)} for r in rows]
    out["profile_source"] = f"profiles/{tag}_kernel_stats.txt"
    # ties the counters to the kernel source they were taken from: bench.py reports `traffic` only while this still matches
    import hashlib
    h = hashlib.sha256()
    for fn in ("stack_sigma_clip.hip", "sortnet_gen.hpp", "sort_ops.hpp"):
        with open(os.path.join(ROOT, "astroburst_amd", "csrc", fn), "rb") as fh:
            h.update(fh.read())
    out["kernel_source_sha256"] = h.hexdigest()
    with open(os.path.join(pdir, "stack_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01", sys.argv[2] if len(sys.argv) > 2 else None)
