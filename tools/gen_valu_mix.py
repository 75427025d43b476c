#!/usr/bin/env python3
"""Generate tools/valu_mix.hip: how a SORTING NETWORK's instruction stream issues on a gfx950 SIMD (developer tool).

tools/valu_rate.hip times one opcode at a time.  A compare-exchange network is a dependent stream of half-rate (v_min / v_max /
v_med3: 16 lanes per cycle) and -- with the xor form of the exchange -- full-rate (v_bitop3 / v_xor: 32 lanes per cycle)
instructions on registers the allocator picked, and what it costs depends on things the single-opcode numbers do not show:
how far apart a min and the bitop3 that consumes it are, whether the three VGPR sources of a VOP3 sit in different register
banks (bank = register number mod 4), and whether a half-rate instruction blocks the full-rate ones behind it.

Every kernel here is ONE asm statement with hard-coded VGPRs (the loop included), so the register numbers are exactly what
is written.  The network is Batcher's odd-even merge sort on 16 wires, renamed through a small pool of spare registers the
way the compiler renames it (the minimum of an exchange lands in a free register, the register it frees becomes the next
temporary).  Output: cycles per compare-exchange for each form.

    python tools/gen_valu_mix.py && hipcc --offload-arch=gfx950 -O3 tools/valu_mix.hip -o build/valu_mix && build/valu_mix
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_sortnet import batcher  # noqa: E402

WIRES = 16
BASE = 32          # v32 .. : the wires and the spare pool
REPS = 4           # networks per loop iteration
ITER = 4096


def network():
    return batcher(WIRES) * REPS


def emit(form, spares, bank_aware=False, distance=1, seed=1):
    """form: 'minmax' | 'bitop3' | 'xor2' | 'med3s' (min through v_med3 with an SGPR -inf + bitop3).
    distance: how many exchanges are in flight (1 = the consumer right behind its min).  Returns (instructions, wire -> register)."""
    rnd = random.Random(seed)
    regs = list(range(BASE, BASE + WIRES + spares))
    rnd.shuffle(regs)
    wire = {i: regs[i] for i in range(WIRES)}
    free = regs[WIRES:]
    out = []
    pending = []  # (instructions, registers written, registers read) of exchanges whose first half has been issued

    def flush(n_keep):
        while len(pending) > n_keep:
            out.extend(pending.pop(0)[0])

    for (i, j) in network():
        x, y = wire[i], wire[j]
        if bank_aware:
            good = [r for r in free if r % 4 not in (x % 4, y % 4)]
            t = good[0] if good else free[0]
        else:
            t = free[0]
        free.remove(t)
        # dependences on exchanges still in flight: their second halves write y' (read by us?) or read the register we overwrite
        now = {t, y} if form == "xor2" else {t}  # registers this exchange overwrites right away
        if any(({x, y} & w) or (now & r) or (now & w) for _, w, r in pending):
            flush(0)
        if form == "minmax":
            out.append(f"v_min_f32 v{t}, v{x}, v{y}")
            pending.append(([f"v_max_f32 v{y}, v{x}, v{y}"], {y}, {x, y}))
        elif form == "bitop3":
            out.append(f"v_min_f32 v{t}, v{x}, v{y}")
            pending.append(([f"v_bitop3_b32 v{y}, v{x}, v{y}, v{t} bitop3:0x96"], {y}, {x, y, t}))
        elif form == "med3s":
            out.append(f"v_med3_f32 v{t}, v{x}, v{y}, s21")
            pending.append(([f"v_bitop3_b32 v{y}, v{x}, v{y}, v{t} bitop3:0x96"], {y}, {x, y, t}))
        elif form == "xor2":
            out.append(f"v_min_f32 v{t}, v{x}, v{y}")
            out.append(f"v_xor_b32 v{y}, v{x}, v{y}")
            pending.append(([f"v_xor_b32 v{y}, v{y}, v{t}"], {y}, {y, t}))
        else:
            raise ValueError(form)
        flush(distance - 1)
        wire[i] = t     # the minimum lives in t now
        free.append(x)  # x is dead once the consumer in flight has read it (checked above before it is overwritten)
    flush(0)
    return out, wire


def simulate(ins, wire):
    """run the stream on random bit patterns (as Python ints) and check that the wires end up sorted"""
    rnd = random.Random(7)
    for _ in range(20):
        vals = {r: rnd.randrange(1, 1 << 30) for r in range(BASE, BASE + WIRES + 16)}
        want = sorted(vals[r] for r in list(range(BASE, BASE + WIRES + 16)) if r in set(emit_initial_wires))
        for s in ins:
            op, rest = s.split(" ", 1)
            a = [t.strip() for t in rest.replace(" bitop3:0x96", "").split(",")]
            d = int(a[0][1:])
            src = [vals[int(t[1:])] if t.startswith("v") else -1 for t in a[1:]]
            if op == "v_min_f32":
                vals[d] = min(src[0], src[1])
            elif op == "v_max_f32":
                vals[d] = max(src[0], src[1])
            elif op == "v_med3_f32":
                vals[d] = sorted(src)[1]
            elif op == "v_bitop3_b32":
                vals[d] = src[0] ^ src[1] ^ src[2]
            elif op == "v_xor_b32":
                vals[d] = src[0] ^ src[1]
        got = [vals[wire[i]] for i in range(WIRES)]
        assert got == want, (got, want)


KERNEL = """
__global__ __launch_bounds__(256) void k_%(name)s(float *out) {
    float r;
    asm volatile(
        "v_cvt_f32_u32 v31, %%1\\n"
%(init)s
        "s_mov_b32 s21, 0xff800000\\n"
        "s_mov_b32 s20, %(iter)d\\n"
        "1:\\n"
%(body)s
        "s_sub_u32 s20, s20, 1\\n"
        "s_cmp_lg_u32 s20, 0\\n"
        "s_cbranch_scc1 1b\\n"
%(fold)s
        "v_mov_b32 %%0, v31\\n"
        : "=v"(r)
        : "v"(threadIdx.x)
        : %(clob)s, "s20", "s21", "scc", "vcc");
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
"""


def kernel(name, ins, nregs):
    regs = list(range(BASE, BASE + nregs))
    init = "\n".join(f'        "v_mul_f32 v{r}, {1.0 + 0.37 * ((r * 7) % 11):.2f}, v31\\n"' for r in regs)
    body = "\n".join(f'        "{s}\\n"' for s in ins)
    fold = "\n".join(f'        "v_add_f32 v31, v31, v{r}\\n"' for r in regs)
    clob = ", ".join(f'"v{r}"' for r in [31] + regs)
    return KERNEL % dict(name=name, init=init, body=body, fold=fold, clob=clob, iter=ITER)


def checked(form, spares, **kw):
    """emit + simulate: the renamed stream sorts, whatever the interleaving"""
    global emit_initial_wires
    rnd = random.Random(kw.get("seed", 1))
    regs = list(range(BASE, BASE + WIRES + spares))
    rnd.shuffle(regs)
    emit_initial_wires = regs[:WIRES]
    ins, wire = emit(form, spares, **kw)
    simulate(ins, wire)
    return ins


emit_initial_wires = []


def simple(name, pattern, n):
    """n copies of a list of independent instructions (explicit registers)"""
    return name, pattern * n


def main():
    variants = []
    for form in ("minmax", "bitop3", "med3s", "xor2"):
        for distance in (1, 2, 4):
            variants.append((f"{form}_d{distance}", checked(form, 6, distance=distance), WIRES + 6, len(network())))
    variants.append(("bitop3_d1_banks", checked("bitop3", 6, bank_aware=True, distance=1), WIRES + 6, len(network())))
    variants.append(("bitop3_d2_banks", checked("bitop3", 6, bank_aware=True, distance=2), WIRES + 6, len(network())))
    variants.append(("med3s_d2_banks", checked("med3s", 6, bank_aware=True, distance=2), WIRES + 6, len(network())))
    # register-bank probes: independent instructions, three VGPR sources in one bank / in three banks
    same = [f"v_bitop3_b32 v{32 + 4 * k}, v{36 + 4 * ((k + 1) % 4)}, v{36 + 4 * ((k + 2) % 4)}, v{36 + 4 * ((k + 3) % 4)} bitop3:0x96" for k in range(4)]
    diff = [f"v_bitop3_b32 v{32 + k}, v{36 + (k + 1) % 4}, v{40 + (k + 2) % 4}, v{44 + (k + 3) % 4} bitop3:0x96" for k in range(4)]
    variants.append(("probe_bitop3_same_bank", same * 64, 20, 256))
    variants.append(("probe_bitop3_three_banks", diff * 64, 20, 256))
    same2 = [f"v_min_f32 v{32 + 4 * k}, v{36 + 4 * ((k + 1) % 4)}, v{36 + 4 * ((k + 2) % 4)}" for k in range(4)]
    diff2 = [f"v_min_f32 v{32 + k}, v{36 + (k + 1) % 4}, v{40 + (k + 2) % 4}" for k in range(4)]
    variants.append(("probe_min_same_bank", same2 * 64, 20, 256))
    variants.append(("probe_min_two_banks", diff2 * 64, 20, 256))
    # does a half-rate instruction block the full-rate ones behind it?  independent min / xor streams, 1:1 and 1:2
    alt11 = [f"v_min_f32 v{32 + k}, v{36 + k}, v{40 + k}" if k % 2 == 0 else f"v_xor_b32 v{32 + k}, v{36 + k}, v{40 + k}" for k in range(4)]
    alt12 = [f"v_min_f32 v32, v36, v41", "v_xor_b32 v33, v37, v42", "v_xor_b32 v34, v38, v43", "v_min_f32 v35, v39, v40", "v_xor_b32 v44, v45, v46", "v_xor_b32 v47, v48, v49"]
    variants.append(("probe_min_xor_1to1", alt11 * 64, 20, 256))
    variants.append(("probe_min_xor_1to2", alt12 * 42, 20, 252))
    alt_b3 = [f"v_min_f32 v32, v36, v41", "v_bitop3_b32 v33, v37, v42, v47 bitop3:0x96", "v_min_f32 v34, v38, v43", "v_bitop3_b32 v35, v39, v40, v45 bitop3:0x96"]
    variants.append(("probe_min_bitop3_1to1_indep", alt_b3 * 64, 20, 256))

    src = ["// GENERATED by tools/gen_valu_mix.py -- do not edit.  See that file for what this measures.",
           "#include <hip/hip_runtime.h>", "#include <cstdio>"]
    for name, ins, nregs, _ in variants:
        src.append(kernel(name, ins, nregs))
    src.append("""
template <typename K>
static void run(const char *name, K kern, float *out, int cus, double clk_ghz, double instr, double units, const char *unit, int waves) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<cus * waves, 256>>>(out);
    hipEventRecord(e0);
    kern<<<cus * waves, 256>>>(out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cycles = ms * 1e-3 * clk_ghz * 1e9 / waves;  // per wave (each SIMD runs `waves` of them)
    printf("%-30s %d waves/SIMD  %7.3f ms  %6.2f cycles/instr  %7.2f cycles/%s\\n", name, waves, ms, cycles / instr, cycles / units, unit);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e-6;
    float *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    printf("%s: %d CUs, clock %.2f GHz; network = Batcher odd-even merge sort, 16 wires (63 exchanges)\\n", p.gcnArchName, cus, clk);
    for (int waves : {3, 4, 1}) {""")
    for name, ins, nregs, units in variants:
        unit = "CE" if not name.startswith("probe") else "instr"
        src.append(f'        run("{name}", k_{name}, out, cus, clk, {len(ins)}.0 * {ITER}, {units}.0 * {ITER}, "{unit}", waves);')
    src.append("    }\n    return 0;\n}")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_mix.hip")
    with open(path, "w") as f:
        f.write("\n".join(src) + "\n")
    print("wrote", path, "with", len(variants), "kernels")


if __name__ == "__main__":
    main()
