"""Developer tool: aggregate a rocprofv3 PC-sampling CSV (--pc-sampling-beta-enabled) by instruction.

    python tools/pcs_summary.py <dir with *pc_sampling*.csv> [kernel-name substring]   ->  samples per instruction, in address order
"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)


def main():
    d = sys.argv[1]
    files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
    print("files:", files)
    for f in files:
        rows = csv.DictReader(open(f))
        print(f, rows.fieldnames)
        by = collections.Counter()
        stall = collections.Counter()
        n = 0
        for r in rows:
            n += 1
            key = (r.get("Instruction_Comment", ""), r.get("Instruction", ""))
            by[key] += 1
            for k in ("Stall_Reason", "Wave_Issued", "Instruction_Type"):
                if k in r:
                    stall[(k, r[k])] += 1
        print("samples", n)
        for k, v in stall.most_common(40):
            print(f"  {v:9d} {k}")
        ops = collections.Counter()
        for (com, ins), v in by.items():
            ops[ins.split(" ")[0]] += v
        print("by opcode:")
        for k, v in ops.most_common(40):
            print(f"  {v:9d} {100.0 * v / max(n, 1):5.1f}%  {k}")
        print("top instructions:")
        for (com, ins), v in by.most_common(60):
            print(f"  {v:9d}  {ins:60s} {com}")


main()
