"""Developer tool (GPU box): what one ab_stack_sigma_clip call costs the STREAM beyond its two kernels.

    python tools/stack_gaps.py run                      # 12 back-to-back 64 x 4096^2 stacks (run it under rocprofv3 --kernel-trace)
    python tools/stack_gaps.py report <kernel_trace.csv> # per launch: the commands on the stream, their durations and the idle gaps

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d D -o t -- python $REPO/tools/stack_gaps.py run && python $REPO/tools/stack_gaps.py report D/*/t_kernel_trace.csv
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch

    import astroburst_amd as ab
    R = C = 4096
    N = 64
    ctx = ab.Context(0)
    ctx.use_torch_stream()
    g = torch.Generator(device="cuda").manual_seed(1)
    frames = []
    for _ in range(N):
        f = 1200.0 + 12.7 * torch.randn((R, C), device="cuda", generator=g)
        frames.append(torch.where(torch.rand((R, C), device="cuda", generator=g) < 1e-4, f * 30.0, f))
    out = torch.empty((R, C), device="cuda")
    for _ in range(3):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"back-to-back {e0.elapsed_time(e1) / 12:.4f} ms per call; library events of the last call {ctx.stack_last_kernel_ms():.4f} ms")


def report(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "stack_sigma_clip_kernel" in r["Kernel_Name"] and "Li1ELi0E" in r["Kernel_Name"]]  # fast passes
    if len(idx) < 6:
        print("fewer than 6 fast-pass launches in the trace")
        return
    first = idx[-6]
    prev_end = None
    t_first = int(rows[first]["Start_Timestamp"])
    for r in rows[first - 2:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        short = "fast pass" if ("stack_sigma_clip_kernel" in name and "Li1ELi0E" in name) else ("general pass" if "stack_sigma_clip_kernel" in name else name[:40])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t_first) / 1e3:10.1f} us  gap {gap:7.1f} us  dur {(e - s) / 1e3:8.1f} us  {short}")
        prev_end = e
    starts = [int(rows[i]["Start_Timestamp"]) for i in idx[-6:]]
    per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
    print("launch-to-launch (us):", " ".join(f"{p:.1f}" for p in per))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        run()
