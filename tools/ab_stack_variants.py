"""Developer tool (GPU box): same-box A/B of libastroburst_hip.so VARIANTS on the bench's registered frames.

    python tools/ab_stack_variants.py [--rounds R] [--clean] <lib_a.so> <lib_b.so> ...     ("default" = the in-tree library)

The parent generates the bench's 64 registered 4096^2 frames ONCE with the in-tree library (as tools/time_stack_bench_data.py
does), parks them in /dev/shm, and runs every variant in its own process (AB_LIB_PATH), R rounds in A B C A B C order so that
clock / thermal drift shows up as spread instead of bias.  Every child reports the library's own kernel events
(ab_stack_last_kernel_ms: fast pass + general pass), the back-to-back wall time, the rejected count and how many output pixels
differ from the FIRST library's output (bitwise and beyond 1e-5 relative).  --clean: also i.i.d. Gaussian frames + 1e-4 cosmic rays.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil  # noqa: E402

SHM = os.environ.get("AB_VARIANTS_DIR") or ("/dev/shm/ab_variants" if shutil.disk_usage("/dev/shm").free > 12 << 30 else "/tmp/ab_variants")
R = C = 4096
N = 64


def child(tag, kind):
    import numpy as np
    import torch

    import astroburst_amd as ab
    ctx = ab.Context(0)
    ctx.use_torch_stream()
    mm = np.memmap(os.path.join(SHM, f"{kind}.f32"), dtype=np.float32, mode="r", shape=(N, R, C))
    frames = [torch.from_numpy(np.ascontiguousarray(mm[k])).cuda() for k in range(N)]
    out = torch.empty((R, C), device="cuda")
    for _ in range(3):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
    torch.cuda.synchronize()
    ks = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
    e1.record()
    torch.cuda.synchronize()
    for _ in range(10):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
        torch.cuda.synchronize()
        ks.append(ctx.stack_last_kernel_ms())
    rej = ctx.last_rejected()
    got = out.cpu().numpy()
    refp = os.path.join(SHM, f"{kind}.ref.f32")
    if not os.path.exists(refp):
        got.tofile(refp)
        diff = "reference"
    else:
        ref = np.fromfile(refp, dtype=np.float32).reshape(R, C)
        nb = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref.astype(np.float64)), 1e-30)
        diff = f"bit-diff {nb}  >1e-5 {int((rel > 1e-5).sum())}"
    ks.sort()
    print(f"{tag:28s} {kind:6s} back-to-back {e0.elapsed_time(e1) / reps:7.4f} ms   kernels min {ks[0]:.4f} med {ks[len(ks) // 2]:.4f} ms   "
          f"roofline {4362076160 / (ks[len(ks) // 2] * 1e-3) / 8e12:.4f}   rejected {rej}   {diff}", flush=True)


def make_frames(kinds):
    import numpy as np
    import torch

    import astroburst_amd as ab
    from astroburst_amd import synth
    from bench import rigid_transforms
    os.makedirs(SHM, exist_ok=True)
    if "bench" in kinds and not os.path.exists(os.path.join(SHM, "bench.f32")):
        ctx = ab.Context(0)
        ctx.use_torch_stream()
        P = R * C
        cy, cx, cf = synth.star_catalog(R, C, int(360.0 * P / 1e6))
        cf = cf * 25.0
        T = rigid_transforms(N, R, C)
        raw = []
        for k in range(N):
            a_, b_, tx_, c_, d_, ty_ = T[k]
            truth = torch.full((R, C), 200.0, device="cuda") + synth.render_stars(R, C, (c_ * cx + d_ * cy + ty_, a_ * cx + b_ * cy + tx_, cf), device="cuda")
            raw.append(synth.make_frame(R, C, k, device="cuda", truth=truth, border=16 if k % 10 == 9 else 0))
        warped = [raw[0]] + [torch.empty_like(raw[0]) for _ in range(1, N)]
        ctx.align_pairs_affine(raw[0], raw[1:], warped[1:], 8)
        torch.cuda.synchronize()
        mm = np.memmap(os.path.join(SHM, "bench.f32"), dtype=np.float32, mode="w+", shape=(N, R, C))
        for k in range(N):
            mm[k] = warped[k].cpu().numpy()
        mm.flush()
        del mm, raw, warped
    if "clean" in kinds and not os.path.exists(os.path.join(SHM, "clean.f32")):
        g = torch.Generator(device="cuda").manual_seed(1)
        mm = np.memmap(os.path.join(SHM, "clean.f32"), dtype=np.float32, mode="w+", shape=(N, R, C))
        for k in range(N):
            f = 1200.0 + 12.7 * torch.randn((R, C), device="cuda", generator=g)
            f = torch.where(torch.rand((R, C), device="cuda", generator=g) < 1e-4, f * 30.0, f)
            mm[k] = f.cpu().numpy()
        mm.flush()
        del mm


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], args[2])
        return
    rounds = 2
    kinds = ["bench"]
    libs = []
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1])
            i += 2
        elif args[i] == "--clean":
            kinds.append("clean")
            i += 1
        else:
            libs.append(args[i])
            i += 1
    make_frames(kinds)
    for kind in kinds:
        for r in range(rounds):
            for lib in libs:
                env = dict(os.environ)
                env["AB_VARIANTS_DIR"] = SHM
                if lib != "default":
                    env["AB_LIB_PATH"] = os.path.abspath(lib)
                tag = "default" if lib == "default" else os.path.basename(lib).replace("libab_", "").replace(".so", "")
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag, kind], env=env, check=False)


if __name__ == "__main__":
    main()
