"""Per-tile parity of the background tile statistics (csrc/tile_bucket.hpp) against sigma_clipped_stats of the oracle
(math/sigma_clip.rs:4-34 on every tile of star_detection.rs:36-68) -- EVERY tile compared bit for bit, not just the
median of the tile medians.  The tiles are built to hit every branch of the one-histogram algorithm: flat tiles (one
bucket of equal keys: the bisection fallbacks), two-valued and heavily quantised tiles (ties at the median, MAD = 0),
ramps (every key distinct), tiles spanning thirty binades (coarse buckets), tiles with 7 / 8 / 9 valid pixels, NaN / inf
/ negative / sub-threshold pixels, even and odd counts, outliers that the clipping removes and windows that empty out.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_tiles(oracle, img, tile):
    step = max(tile, 16)
    rows, cols = img.shape
    nty, ntx = -(-rows // step), -(-cols // step)
    med, sig, val = np.zeros((nty, ntx)), np.zeros((nty, ntx)), np.zeros((nty, ntx), bool)
    for ty in range(nty):
        for tx in range(ntx):
            t = img[ty * step:(ty + 1) * step, tx * step:(tx + 1) * step].ravel()
            v = t[np.isfinite(t) & (t > np.float32(1e-7))]
            if v.size >= 8:
                val[ty, tx] = True
                med[ty, tx], sig[ty, tx] = oracle.sigma_clipped_stats(v, 3.0, 2)
            else:
                med[ty, tx], sig[ty, tx] = 0.0, 1.0
    return med, sig, val


def check(ctx, oracle, img, tile):
    gm, gs, gv = ctx.background_tile_stats(img, tile)
    wm, ws, wv = oracle_tiles(oracle, img, tile)
    assert np.array_equal(gv, wv)
    bad = np.argwhere(~((gm == wm) & (gs == ws)) & wv)
    assert bad.size == 0, [(tuple(b), gm[tuple(b)], wm[tuple(b)], gs[tuple(b)], ws[tuple(b)]) for b in bad[:5]]


def adversarial_image(tile, seed):
    """a 6 x 8 grid of tiles, each with its own pathology"""
    rng = np.random.default_rng(seed)
    T = tile
    img = np.zeros((6 * T, 8 * T), np.float32)
    mk = []
    n = T * T
    mk.append(lambda: np.full(n, 100.0))                                                   # flat
    mk.append(lambda: np.where(rng.random(n) < 0.5, 10.0, 11.0))                           # two values
    mk.append(lambda: np.round(rng.normal(1000, 30, n) / 16) * 16)                         # heavy ties
    mk.append(lambda: np.linspace(1.0, 2.0, n))                                            # ramp, all distinct
    mk.append(lambda: rng.normal(1000, 30, n))                                             # sky
    mk.append(lambda: np.where(rng.random(n) < 0.02, rng.uniform(5e3, 6e4, n), rng.normal(1000, 30, n)))   # sky + stars
    mk.append(lambda: 10.0 ** rng.uniform(-6.5, 4.5, n))                                   # thirty binades
    mk.append(lambda: np.abs(rng.normal(0, 1e-3, n)))                                      # half-normal near the threshold
    mk.append(lambda: np.where(rng.random(n) < 0.3, np.nan, rng.normal(50, 5, n)))         # 30 % NaN
    mk.append(lambda: np.where(rng.random(n) < 0.5, -rng.random(n), rng.normal(5, 1, n)))  # half negative
    mk.append(lambda: rng.uniform(0, 1, n))                                                # normalised-frame-like
    mk.append(lambda: np.clip(rng.normal(0.2, 0.2, n), 0, 1))                              # clamped at 0 and 1 (ties at both ends)
    mk.append(lambda: np.concatenate([np.full(n - n // 3, 7.0), rng.normal(7, 1e-3, n // 3)]))   # MAD = 0 with a spread tail
    mk.append(lambda: np.concatenate([np.full(n // 2, 5.0), np.full(n - n // 2, 5.0 + 1e-6)]))   # adjacent floats
    mk.append(lambda: rng.normal(1000, 30, n).astype(np.float32).astype(np.float64) * 1e30)      # huge values
    mk.append(lambda: rng.normal(1000, 30, n) * 1e-36)                                     # near the subnormal range (below 1e-7: invalid)
    mk.append(lambda: rng.standard_cauchy(n) * 10 + 500)                                   # fat tails: the window really clips
    mk.append(lambda: np.exp(rng.normal(0, 2, n)))                                         # log-normal
    for k, few in enumerate((7, 8, 9, 10, 11)):                                            # 7 .. 11 valid pixels
        def f(few=few):
            a = np.full(n, np.nan)
            a[rng.choice(n, few, replace=False)] = rng.normal(100, 10, few)
            return a
        mk.append(f)
    mk.append(lambda: np.where(np.arange(n) % 2 == 0, 1.0, 1e6))                           # lo > hi after the first clip?  (bimodal)
    mk.append(lambda: np.concatenate([[1e-6] * 3, np.full(n - 3, np.inf)]))                # 3 valid
    mk.append(lambda: np.full(n, 1e-8))                                                    # nothing valid
    while len(mk) < 48:
        s = float(10.0 ** rng.uniform(-2, 3))
        m = float(10.0 ** rng.uniform(0, 4))
        odd = len(mk) % 2
        def f(s=s, m=m, odd=odd):
            a = rng.normal(m, s, n)
            if odd:
                a[0] = np.nan                                                               # flips the parity of the count
            return a
        mk.append(f)
    for i, f in enumerate(mk[:48]):
        ty, tx = divmod(i, 8)
        img[ty * T:(ty + 1) * T, tx * T:(tx + 1) * T] = np.asarray(f(), np.float64).astype(np.float32).reshape(T, T)
    return img


@pytest.mark.parametrize("tile,seed", [(16, 1), (32, 2), (37, 3), (64, 4), (100, 5), (200, 6), (256, 7)])
def test_every_tile_matches_the_oracle(ctx, oracle, tile, seed):
    with np.errstate(all="ignore"):
        img = adversarial_image(tile, seed)
    check(ctx, oracle, img, tile)


def test_ragged_edge_tiles(ctx, oracle):
    rng = np.random.default_rng(11)
    img = rng.normal(300, 12, (700, 901)).astype(np.float32)       # 3 x 4 tiles of 256 with thin right / bottom remainders
    img[rng.random(img.shape) < 0.01] *= 20
    check(ctx, oracle, img, 256)
    check(ctx, oracle, img[:258, :259], 256)                        # remainders of 2 and 3 pixels: fewer than 8 valid in the corner


def test_normalised_frame_tiles(ctx, oracle):
    """what the registration path feeds the tile kernel: a percentile-normalised star field (values clamped to [0, 1])"""
    from astroburst_amd import synth
    y, x, flux = synth.star_catalog(1024, 1024, 400, seed=3)
    frame = synth.make_frame(1024, 1024, 2, cat=(y, x, flux * 25.0)).numpy()
    norm = oracle.normalize_for_detection(frame)
    check(ctx, oracle, norm, 128)
    check(ctx, oracle, frame, 128)


def test_round1_kernel_behind_ab_tile_legacy(dev_build):
    """AB_TILE_LEGACY=1 (read once per process) selects round 1's radix-select tile kernel: same per-tile answers.
    Run in a child process so that both kernels are exercised by one `pytest -m gpu`."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "import astroburst_amd as ab\n"
        "from oracle import pyoracle as oracle\n"
        "from tests.test_gpu_tile_stats import adversarial_image, check\n"
        "ctx = ab.Context(0)\n"
        "with np.errstate(all='ignore'):\n"
        "    for tile, seed in ((64, 4), (256, 7)):\n"
        "        check(ctx, oracle, adversarial_image(tile, seed), tile)\n"
        "print('legacy-ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AB_TILE_LEGACY="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "legacy-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
