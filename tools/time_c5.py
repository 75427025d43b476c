"""Developer tool: timings of the C5 rows (8192^2 planes): extract_background, masked stretch, SCNR, SPCC, process_rgb."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from astroburst_amd import Context  # noqa: E402
from test_gpu_full_size import star_field_gpu  # noqa: E402


def timed(name, fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name:44s} {(time.perf_counter() - t0) / reps * 1e3:9.2f} ms")


ctx = Context(0)
r = c = 8192
red, green, blue = star_field_gpu(r, c, 20000, 5, gains=(1.0, 0.8, 1.25))
timed("extract_background 8192^2", lambda: ctx.extract_background(green))
timed("masked_stretch 8192^2", lambda: ctx.masked_stretch(green))
timed("masked_stretch_rgb_shared 3 x 8192^2", lambda: ctx.masked_stretch_rgb_shared(red, green, blue))
a, b2, c2 = red.clone(), green.clone(), blue.clone()
timed("apply_scnr_inplace 3 x 8192^2", lambda: ctx.apply_scnr_inplace(a, b2, c2, "average", 1.0, True))
timed("spcc_calibrate_rgb 3 x 8192^2", lambda: ctx.spcc_calibrate_rgb(red, green, blue, 0.3))
timed("process_rgb 3 x 8192^2 (auto WB, linked STF off)", lambda: ctx.process_rgb(red, green, blue))
timed("compute_image_stats 8192^2", lambda: ctx.compute_image_stats(green))
timed("detect_stars 8192^2 (5 sigma)", lambda: ctx.detect_stars(green, 5.0))
