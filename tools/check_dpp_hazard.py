#!/usr/bin/env python3
"""Build-time check: no DPP instruction reads a VGPR that a VALU instruction wrote fewer than two wait states before.

gfx9 / CDNA hazard: "VALU writes VGPR, followed by a VALU DPP read of that VGPR: 2 wait states".  The compiler's hazard recogniser
inserts the s_nop itself -- except when the write sits inside an inline-asm statement, which it does not see.  csrc/stack_pair.hip's
sorting network is inline asm (v_min_f32 / v_max_f32 / v_med3_f32 on hand-picked registers) followed by DPP exchanges, and the source
places a volatile `s_nop 1` between them by hand.  Whether that fence survives a compiler upgrade cannot be seen from
the source: this script reads the LISTING.

    python tools/check_dpp_hazard.py [file.hip ...]      # default: every csrc/*.hip that contains "dpp"
    python tools/check_dpp_hazard.py --listing file.s    # an existing --save-temps listing

For every `*_dpp` instruction (and DPP-modified VALU op) it walks back over the straight-line predecessors and counts wait states
(every instruction = 1, `s_nop N` = N + 1) until two have passed; a VALU write of the DPP source register inside that window is a
violation.  A label resets the window conservatively to "unknown predecessors": the first two instructions after a label must not be
DPP reads of a register (reported as a violation unless preceded by an s_nop in the same block), which the kernels here satisfy.
Exit status 1 and a list of the offending lines on failure.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "astroburst_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wno-unused-function", "-w"]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(line):
    """-> (mnemonic, [operand strings]) or None for directives / comments / blank lines"""
    l = line.split(";")[0].strip()
    if not l or l.startswith(".") or l.startswith("//"):
        return None
    if l.endswith(":"):
        return ("<label>", [])
    parts = l.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return (parts[0], ops)


def is_valu(mn):
    return mn.startswith("v_") and not mn.startswith("v_cmpx") or mn.startswith("v_cmpx")


def check_listing(path):
    lines = open(path).read().split("\n")
    viol, ndpp = [], 0
    hist = []  # (wait states this instruction provides, set of VGPRs it writes as a VALU op, line no, text); reset at labels
    for no, raw in enumerate(lines, 1):
        p = parse(raw)
        if p is None:
            continue
        mn, ops = p
        if mn == "<label>":
            hist = [(0, None, no, raw)]  # unknown predecessors: None = "may have written anything"
            continue
        dpp = mn.endswith("_dpp") or any("quad_perm" in o or "row_shr" in o or "row_shl" in o or "row_ror" in o or "row_bcast" in o or
                                         "wave_sh" in o or "wave_ro" in o or "row_mirror" in o or "row_half_mirror" in o or "row_newbcast" in o
                                         for o in ops)
        if dpp:
            ndpp += 1
            src = regs_of(ops[1].split()[0]) if len(ops) > 1 else set()   # the DPP-permuted operand is src0
            waited = 0
            for (ws, writes, hno, htxt) in reversed(hist):
                if waited >= 2:
                    break
                if writes is None:
                    viol.append((no, raw.strip(), hno, "label (unknown predecessor) within two wait states"))
                    break
                if writes & src:
                    viol.append((no, raw.strip(), hno, htxt.strip()))
                    break
                waited += ws
        ws = 1
        writes = set()
        if mn == "s_nop":
            ws = int(ops[0], 0) + 1 if ops else 1
        elif is_valu(mn) and ops:
            writes = regs_of(ops[0])      # vdst (a VALU op writing only SGPRs / VCC contributes nothing)
        hist.append((ws, writes, no, raw))
        if len(hist) > 8:
            hist = hist[-8:]
    return ndpp, viol


def compile_listing(src, tmp, extra=()):
    out = os.path.join(tmp, os.path.basename(src) + ".o")
    subprocess.run([HIPCC, *FLAGS, *extra, "--save-temps", "-c", src, "-o", out], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    stem = os.path.splitext(os.path.basename(src))[0]
    return os.path.join(tmp, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")


def main(argv):
    if argv and argv[0] == "--listing":
        targets = [(p, p) for p in argv[1:]]
        tmp = None
    else:
        srcs = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and "dpp" in open(os.path.join(CSRC, f)).read())
        tmp = tempfile.mkdtemp(prefix="ab_dpp_")
        targets = [(s, compile_listing(s, tmp)) for s in srcs]
    bad = 0
    for src, lst in targets:
        ndpp, viol = check_listing(lst)
        print(f"{os.path.basename(src)}: {ndpp} DPP instructions, {len(viol)} hazard violations")
        for (no, txt, hno, why) in viol[:20]:
            print(f"  line {no}: {txt}\n      <- line {hno}: {why}")
        bad += len(viol)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
