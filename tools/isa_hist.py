"""Developer tool: opcode histogram of one kernel in a --save-temps .s file (static count of the whole body, cold paths included).

    python tools/isa_hist.py <file.s> <regex on the mangled kernel name> [--cost]
"""
import collections
import re
import sys

# cycles per wave64 instruction at >= 2 waves / SIMD (tools/valu_rate.hip, gpurun_out/valu_rate_r03.txt)
FULL = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_mov_b32", "v_bitop3_b32", "v_xor_b32", "v_and_b32", "v_or_b32",
        "v_fmac_f32", "v_cndmask_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*:", l) and re.search(pat, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    c = collections.Counter()
    for l in lines[start + 1:end]:
        l = l.strip()
        if not l or l[0] in ".;" or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"{lines[start].split(':')[0]}\n  {tot} instructions, {valu} VALU, {sum(v for k, v in c.items() if k.startswith('s_'))} SALU/branch, "
          f"{sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'ds_', 'scratch_', 'flat_')))} memory")
    for k, v in c.most_common(50):
        print(f"{v:6d} {k}")
    for l in lines[end:end + 400]:
        if re.search(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):", l) or "; Occupancy" in l or "; NumVgprs" in l or "; NumAgprs" in l or "; ScratchSize" in l:
            print(l.strip())
        if l.startswith("_Z") and l.endswith(":"):
            break


main()
