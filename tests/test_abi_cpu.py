"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares,
the build recipe works, the synthetic generator is deterministic."""
import ctypes
import os

import numpy as np


def test_library_exports_every_declared_symbol():
    from astroburst_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    syms = _lib.declared_symbols()
    assert len(syms) >= 26
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert b"gfx950" in ctypes.c_char_p(ctypes.cast(L.ab_version, ctypes.c_void_p).value and
                                         ctypes.CFUNCTYPE(ctypes.c_char_p)(("ab_version", L))()).value


def test_no_device_is_loud_not_silent():
    """Without an MI355X the product must refuse (no CPU fallback)."""
    import torch
    import astroburst_amd as ab
    if torch.cuda.is_available():
        return
    try:
        ab.Context(0)
    except ab.AstroBurstError as e:
        assert e.code in (ab._lib.AB_ERR_NO_DEVICE, ab._lib.AB_ERR_HIP)
    else:
        raise AssertionError("Context() succeeded without a GPU")


def test_auto_stf_is_host_math_and_matches_oracle(oracle):
    """ab_auto_stf is scalar host code in the library (stf.rs:13-39): callable without a GPU."""
    from astroburst_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        mn = float(rng.uniform(0, 100))
        mx = mn + float(rng.uniform(1e-3, 1e5))
        med = float(rng.uniform(mn, mx))
        sig = float(rng.uniform(0, (mx - mn)))
        st = oracle.ImageStats(mn, mx, med, sig / 1.4826, sig, med, 1000)
        want = oracle.auto_stf(st, 0.25, -2.8)
        s = _lib.ImageStatsC(st.min, st.max, st.median, st.mad, st.sigma, st.mean, st.valid_count)
        cfg = _lib.AutoStfConfigC(0.25, -2.8)
        p = _lib.StfParamsC()
        assert L.ab_auto_stf(ctypes.byref(s), ctypes.byref(cfg), ctypes.byref(p)) == 0
        assert (p.shadow, p.midtone, p.highlight) == (want.shadow, want.midtone, want.highlight)
    empty = _lib.ImageStatsC()
    p = _lib.StfParamsC()
    L.ab_auto_stf(ctypes.byref(empty), ctypes.byref(_lib.AutoStfConfigC(0.25, -2.8)), ctypes.byref(p))
    assert (p.shadow, p.midtone, p.highlight) == (0.0, 0.5, 1.0)


def test_synth_is_deterministic():
    from astroburst_amd import synth
    a = synth.make_stack(3, 64, 80)
    b = synth.make_stack(3, 64, 80)
    for x, y in zip(a, b):
        assert np.array_equal(x.numpy(), y.numpy(), equal_nan=True)
    assert a[0].shape == (64, 80) and np.isfinite(a[0].numpy()).mean() > 0.9


def test_oracle_order_modes_agree(oracle):
    """Summation order of iterations >= 1 is unspecified in the reference; the two pinned orders
    must agree far inside the 1e-5 contract (they differ by <= 1 ulp(f64) before the f32 cast)."""
    from astroburst_amd import synth
    frames = [f.numpy() for f in synth.make_stack(20, 96, 128)]
    a, ra = oracle.stack_images(frames, order=oracle.ORDER_ASCENDING)
    b, rb = oracle.stack_images(frames, order=oracle.ORDER_SELECT)
    assert ra == rb
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=0)
    assert (a != b).mean() < 1e-3


# ---- boundary hardening (round 2) -----------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rust_sys_bindings_are_complete_and_current():
    """bindings/sys.rs (generated from the header) declares every exported symbol exactly once and is not stale"""
    import re
    from astroburst_amd import _lib
    gen = _tool("gen_rust_sys")
    text, structs, funcs = gen.generate()
    assert open(os.path.join(ROOT, "bindings", "sys.rs")).read() == text, "run python tools/gen_rust_sys.py"
    declared = re.findall(r"pub fn (ab_\w+)\(", text)
    assert sorted(declared) == sorted(_lib.declared_symbols()) and len(set(declared)) == len(declared)
    assert len(structs) >= 29


def test_rust_repr_c_layout_equals_the_c_header(tmp_path):
    """every header struct: sizeof / alignof / each field's offset under #[repr(C)] rules (computed from the generated Rust
    types) == what gcc reports for include/astroburst_hip.h"""
    import subprocess
    gen = _tool("gen_rust_sys")
    structs, enums, funcs, defines, cb = gen.parse(open(gen.HEADER).read())
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "astroburst_hip.h"', 'int main(void) {']
    for name, fields in structs.items():
        lines.append(f'    printf("{name} %zu %zu", sizeof({name}), _Alignof({name}));')
        for fname, _, _, _ in fields:
            lines.append(f'    printf(" %zu", offsetof({name}, {fname}));')
        lines.append('    printf("\\n");')
    lines.append('    return 0;\n}')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == len(structs)
    for line in out:
        toks = line.split()
        name, nums = toks[0], [int(t) for t in toks[1:]]
        size, align, lay = gen.layout(structs, name)
        assert (size, align) == (nums[0], nums[1]), name
        assert [off for _, off, _ in lay] == nums[2:], name


def test_every_status_returning_entry_point_has_the_exception_barrier():
    """no C++ exception may unwind through extern "C" (the host is built panic = "abort"): each int-returning entry point is
    a function-try-block ending in AB_CATCH / AB_CATCH_NOCTX (tools/add_exception_barrier.py)"""
    import glob
    import re
    from astroburst_amd import _lib
    barrier = _tool("add_exception_barrier")
    src = "".join(open(p).read() for p in sorted(glob.glob(os.path.join(ROOT, "astroburst_amd", "csrc", "*.hip"))))
    wrapped = set(re.findall(r'^(?:extern "C" )?(?:int|uint64_t)\s+(ab_[a-z0-9_]+)\s*\([^;{}]*?\)\s*try\s*\{', src, re.M))
    expected = set(_lib.declared_symbols()) - barrier.SKIP
    assert expected - wrapped == set(), sorted(expected - wrapped)
    assert src.count("AB_CATCH(ctx)") + src.count("AB_CATCH_NOCTX") >= len(expected)


def test_allocation_failure_is_a_status_code_not_an_abort():
    """ab_affine_from_stars is host-only: an absurd star count makes std::vector throw (length_error / bad_alloc) before any
    input is read; the barrier turns it into AB_ERR_INVALID / AB_ERR_NOMEM and the process lives on"""
    from astroburst_amd import _lib
    L = _lib.lib()
    xy = (ctypes.c_double * 8)(*range(8))
    res, found = _lib.AffineAlignResultC(), ctypes.c_int(0)
    rc = L.ab_affine_from_stars(xy, ctypes.c_size_t(1 << 61), xy, 4, 100, 100, 8, ctypes.byref(res), ctypes.byref(found))
    assert rc == _lib.AB_ERR_INVALID          # std::length_error
    rc = L.ab_affine_from_stars(xy, ctypes.c_size_t(1 << 44), xy, 4, 100, 100, 8, ctypes.byref(res), ctypes.byref(found))
    assert rc == _lib.AB_ERR_NOMEM            # std::bad_alloc: 2^48 bytes exceed the address space
    rc = L.ab_affine_from_stars(xy, 4, xy, 4, 100, 100, 8, ctypes.byref(res), ctypes.byref(found))
    assert rc == _lib.AB_OK                   # and the library still works


def test_sorting_networks_are_current_and_sort():
    """astroburst_amd/csrc/sortnet_gen.hpp is what tools/gen_sortnet.py writes (the generator checks every network with the
    0-1 principle before it emits it), and its AB_SORT4 base case -- sort three with min3 / med3 / max3, then place the fourth
    with min, med3, med3, max -- sorts every arrangement of four values, ties and infinities included."""
    import importlib.util
    import itertools
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_sortnet", os.path.join(root, "tools", "gen_sortnet.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    # (the header is a deterministic function of tools/sortnet_choice.json -- which exchanges absorb which values -- and render()
    # re-verifies every rewritten network against the plain one on every 0-1 state of every merge level; QUICK: of the 17.9 M
    # states of the 256-wire network's top level it takes 2 M here, `python tools/gen_sortnet.py` runs them all)
    gen.QUICK = True
    assert open(gen.PATH).read() == gen.render()

    def med3(a, b, c):
        return sorted((a, b, c))[1]

    def sort4(x0, x1, x2, x3):  # the macro in stack_sigma_clip.hip / batch_pipeline.hip, operation for operation
        s0, s1, s2 = min(x0, x1, x2), med3(x0, x1, x2), max(x0, x1, x2)
        return [med3(float("-inf"), s0, x3), med3(s0, s1, x3), med3(s1, s2, x3), med3(float("inf"), s2, x3)]

    vals = [0.0, 1.0, 1.0, 2.5, float("inf"), float("-inf")]
    for combo in itertools.product(vals, repeat=4):
        assert sort4(*combo) == sorted(combo)


def test_rust_safe_wrappers_cover_the_header():
    """bindings/mod.rs + device.rs (the Rust side of the boundary: drop-ins for the reference's core::* functions) call EVERY entry
    point include/astroburst_hip.h declares, except the three test / bench hooks named here -- a new C function without a Rust
    wrapper, or a wrapper calling a function the header no longer has, fails this test."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "astroburst_hip.h")).read(), flags=re.S)
    declared = re.findall(r"AB_API\s+[^;(]*?\b(ab_\w+)\s*\(", header)
    assert len(declared) > 100
    src = open(os.path.join(root, "bindings", "mod.rs")).read() + open(os.path.join(root, "bindings", "device.rs")).read()
    called = set(re.findall(r"sys::(ab_\w+)\s*\(", src))
    hooks = {"ab_correlate_single",       # parity-test hook: the correlation surface of one <= 512^2 pair
             "ab_background_tile_stats",  # parity-test hook: the tile map behind estimate_background
             "ab_bench_copy"}             # bench.py's streaming probe
    missing = [n for n in declared if n not in called and n not in hooks]
    assert not missing, f"no Rust wrapper calls {missing}"
    stale = sorted(called - set(declared))
    assert not stale, f"bindings call functions the header does not declare: {stale}"
    sys_rs = open(os.path.join(root, "bindings", "sys.rs")).read()
    for n in called:
        assert re.search(r"pub fn %s\s*\(" % n, sys_rs), f"{n} is not in bindings/sys.rs"
    # every (a)-row function of SURVEY 8 has a drop-in of the reference's own name
    for name in ("stack_images", "shift_image_subpixel", "warp_image", "align_channel_affine", "detect_stars", "estimate_background",
                 "phase_correlate", "compute_image_stats", "auto_stf", "apply_stf", "apply_stf_f32", "extract_background", "masked_stretch",
                 "masked_stretch_rgb_shared", "generate_star_mask", "apply_scnr_inplace", "apply_curve", "apply_levels", "blend_channels",
                 "process_rgb", "spcc_calibrate_rgb", "arcsinh_stretch", "calibrate_image", "median_combine", "run_batch_pipeline",
                 "analyze_subframe", "align_pair", "resample_image", "apply_lrgb", "calibrate_channel"):
        assert re.search(r"pub fn %s\b" % name, src), f"no drop-in named {name}"


def test_no_dpp_reads_a_register_inside_its_write_hazard_window(tmp_path):
    """csrc/stack_pair.hip's correctness leans on a hand-placed `s_nop 1` between the inline-asm sorting network
    (VALU writes the hazard recogniser cannot see) and the DPP exchanges that read the same registers (gfx9: two wait states).
    tools/check_dpp_hazard.py compiles the file for gfx950 and walks the LISTING: every DPP source register must be two wait states
    clear of its last VALU write.  The checker is itself checked: the same listing with every s_nop removed must fail."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(root, "tools", "check_dpp_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    lst = chk.compile_listing(os.path.join(chk.CSRC, "stack_pair.hip"), str(tmp_path))
    ndpp, viol = chk.check_listing(lst)
    assert ndpp > 500, "the listing holds no DPP instructions: wrong file or the kernel changed shape"
    assert not viol, viol[:5]
    stripped = os.path.join(tmp_path, "nofence.s")
    with open(stripped, "w") as f:
        f.write("\n".join(l for l in open(lst).read().split("\n") if "s_nop" not in l))
    assert chk.check_listing(stripped)[1], "the checker does not see a hazard when the fences are gone"
    # round 6: the fast two- and four-lane kernels (csrc/stack_duo.hip, stack_quad.hip) have the same network + DPP structure; one
    # frame-count class of each is compiled
    for name, flag in (("stack_duo.hip", "-DAB_DUO_ONE_CLASS"), ("stack_quad.hip", "-DAB_QUAD_ONE_CLASS")):
        lst = chk.compile_listing(os.path.join(chk.CSRC, name), str(tmp_path), extra=(flag,))
        ndpp, viol = chk.check_listing(lst)
        assert ndpp > 500 and not viol, (name, ndpp, viol[:5])


def test_the_deepest_sort_network_stays_in_registers(tmp_path):
    """Round 6: wave_sort.hpp's bitonic network over 64 registers per lane (2049 .. 4096 frames) is ~20 000 instructions fully unrolled --
    above the default size limit of `#pragma unroll`.  Left to that limit it stays a loop, indexes its registers at run time, and the two
    sort arrays of stack_wide.hip's kernels live in scratch memory (2100 x 1024^2: 590 ms instead of 89).  The Makefile raises the limit
    for that file; this test compiles it with the Makefile's flags and reads the kernels' metadata: no scratch, no spills."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "astroburst_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    flags = re.search(r"^FLAGS_stack_wide\s*:=\s*(.+)$", mk, flags=re.M).group(1).split()
    assert "-pragma-unroll-threshold=1000000" in flags, flags
    base = re.search(r"^CXXFLAGS\s*\?=\s*(.+)$", mk, flags=re.M).group(1).split()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *base, "-w", *flags, "--save-temps", "-c", os.path.join(csrc, "stack_wide.hip"), "-o",
                    os.path.join(tmp_path, "stack_wide.o")], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lst = open(os.path.join(tmp_path, "stack_wide-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)", lst, re.S):
        if "stack_wide" not in m.group(1):
            continue
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", m.group(2)).group(1))
        assert scratch == 0 and int(m.group(3)) == 0, (m.group(1), scratch, m.group(3))
    assert seen >= 10, seen


def test_the_release_library_reads_the_documented_environment_variables_only():
    """`ship one code path`.  (i) no raw getenv in csrc/ outside the two helpers of ab_common.hpp; (ii) the names
    given to ab_env() -- the release library's -- are exactly the eight the header's "Environment" section and INTEGRATION.md list;
    (iii) none of the developer names (ab_dev_env / AB_DEV_NAME: superseded forms, sweep knobs, stage cuts, fault injection) is
    present in the built release library, not even as a string."""
    import glob
    import re
    from astroburst_amd import _lib
    rel, dev = set(), set()
    for path in glob.glob(os.path.join(_lib.CSRC, "*.h*")):
        if path.endswith("sortnet_gen.hpp"):
            continue
        text = open(path).read()
        code = re.sub(r"//[^\n]*", "", text)
        raw = re.findall(r"\bgetenv\s*\(", code)
        assert len(raw) == (2 if path.endswith("ab_common.hpp") else 0), f"{path}: raw getenv"
        rel |= set(re.findall(r'\bab_env\("([A-Z0-9_]+)"\)', code))
        dev |= set(re.findall(r'\bab_dev_env\("([A-Z0-9_]+)"\)', code)) | set(re.findall(r'AB_DEV_NAME\("([A-Z0-9_]+)"\)', code))
    header = open(_lib.HEADER_PATH).read()
    section = header[header.index(" * Environment"):header.index("#ifndef ASTROBURST_HIP_H")]
    documented = set(re.findall(r"^ \*    (AB_[A-Z0-9_]+)=", section, flags=re.M))
    assert len(documented) == 8 and rel == documented, (sorted(rel), sorted(documented))
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in documented:
        assert name in integration, f"{name} is not in INTEGRATION.md"
    assert len(dev) >= 25 and not (dev & rel)
    if not os.path.exists(_lib.LIB_PATH) or _lib.LIB_PATH.endswith("_dev.so"):
        return
    blob = open(_lib.LIB_PATH, "rb").read()
    present = sorted(n for n in dev if n.encode() in blob)
    assert not present, f"developer switches reached the release library: {present}"
    for n in documented:
        assert n.encode() in blob, n
