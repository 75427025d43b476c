"""An independent restatement of extract_background in numpy / plain Python floats.

Written from core/imaging/background.rs:55-459 and math/median.rs:46-63 alone: the kappa-sigma gated cell medians (f32), the ridge
least squares on the monomial basis (f64, partial-pivot elimination in the source's order), the model in f64 -> f32, the two
correction modes.  Nothing is shared with oracle/orc_background.c or csrc/background.hip.  The reference's own tests for this file
(background.rs:465-592) are behavioural; this is the bit-level second opinion (VERDICT r4 missing 5).  TEST INFRASTRUCTURE.
"""
import math

import numpy as np

F = np.float32
MAD_TO_SIGMA = 1.4826                                                  # types/constants.rs:7


class BackgroundError(ValueError):
    pass


def median_f32(values) -> np.float32:                                  # math/median.rs:46-63
    v = np.asarray(values, F)
    n = v.size
    if n == 0:
        return F(0.0)
    s = np.sort(v)
    if n % 2 == 0:
        return F((s[n // 2 - 1] + s[n // 2]) / F(2.0))                # (max of the left part + right) / 2, f32
    return s[n // 2]


def powi(a: float, b: int) -> float:
    """f64::powi = compiler-rt's __powidf2: square and multiply from the low bit"""
    r = 1.0
    while True:
        if b & 1:
            r *= a
        b //= 2
        if b == 0:
            break
        a *= a
    return r


def poly_basis(y, x, degree):                                          # :209-220
    out = []
    for total in range(degree + 1):
        for y_pow in range(total, -1, -1):
            out.append(powi(y, y_pow) * powi(x, total - y_pow))
    return out


def min_samples_for_degree(degree):                                    # :204-207
    return (degree + 1) * (degree + 2) // 2 + 2


def auto_sample_grid(image, grid, degree, sigma_clip, iterations):    # :118-202 -> [(y, x, value)] as f32
    rows, cols = image.shape
    cell_h, cell_w = rows // grid, cols // grid
    if cell_h < 4 or cell_w < 4:
        raise BackgroundError(f"Image too small for grid_size={grid}")
    margin_h, margin_w = cell_h // 4, cell_w // 4
    inner_h, inner_w = cell_h - 2 * margin_h, cell_w - 2 * margin_w
    kappa = F(sigma_clip)
    with np.errstate(invalid="ignore"):
        allpx = image[np.isfinite(image) & (image > F(0.0))]
    global_median = median_f32(allpx)
    global_mad = median_f32(np.abs(allpx - global_median))
    sigma = global_mad * F(MAD_TO_SIGMA)
    lo, hi = global_median - kappa * sigma, global_median + kappa * sigma
    samples = []
    for gy in range(grid):
        for gx in range(grid):
            y0, x0 = gy * cell_h + margin_h, gx * cell_w + margin_w
            cell = image[y0:min(y0 + inner_h, rows), x0:min(x0 + inner_w, cols)]
            with np.errstate(invalid="ignore"):
                ok = np.isfinite(cell) & (cell > F(1e-7))
            vals = cell[ok]
            zero_count = cell.size - vals.size
            if vals.size == 0 or zero_count / float(inner_h * inner_w) > 0.3:
                continue
            cell_median = median_f32(vals)
            if cell_median >= lo and cell_median <= hi:
                samples.append((F(y0 + inner_h // 2), F(x0 + inner_w // 2), cell_median))
    for _ in range(1, iterations):
        if len(samples) < min_samples_for_degree(degree):
            break
        values = np.array([s[2] for s in samples], F)
        med = median_f32(values)
        mad = median_f32(np.abs(values - med))
        sig = mad * F(MAD_TO_SIGMA)
        lo2, hi2 = med - kappa * sig, med + kappa * sig
        samples = [s for s in samples if s[2] >= lo2 and s[2] <= hi2]
    return samples


def solve_linear_system(a, b, n):                                      # :404-459, in place on Python lists
    for col in range(n):
        max_row, max_val = col, abs(a[col * n + col])
        for row in range(col + 1, n):
            v = abs(a[row * n + col])
            if v > max_val:
                max_val, max_row = v, row
        if max_val < 1e-14:
            raise BackgroundError("Failed to solve polynomial fit: Singular matrix in polynomial fit")
        if max_row != col:
            for k in range(n):
                a[col * n + k], a[max_row * n + k] = a[max_row * n + k], a[col * n + k]
            b[col], b[max_row] = b[max_row], b[col]
        pivot = a[col * n + col]
        for row in range(col + 1, n):
            factor = a[row * n + col] / pivot
            for k in range(col, n):
                a[row * n + k] -= factor * a[col * n + k]
            b[row] -= factor * b[col]
    for col in range(n - 1, -1, -1):
        s = b[col]
        for k in range(col + 1, n):
            s -= a[col * n + k] * b[k]
        b[col] = s / a[col * n + col]


def fit_polynomial_surface(samples, rows, cols, degree):               # :240-284
    n = (degree + 1) * (degree + 2) // 2
    ata, atb = [0.0] * (n * n), [0.0] * n
    for (sy, sx, sv) in samples:
        ny, nx, val = float(sy) / float(rows) - 0.5, float(sx) / float(cols) - 0.5, float(sv)
        basis = poly_basis(ny, nx, degree)
        for i in range(n):
            atb[i] += basis[i] * val
            for j in range(n):
                ata[i * n + j] += basis[i] * basis[j]
    for i in range(n):
        ata[i * n + i] += 1e-8
    solve_linear_system(ata, atb, n)
    return atb


def _pows(v, degree):
    p = [0.0] * 7
    p[0] = 1.0
    for i in range(1, min(degree, 6) + 1):
        p[i] = p[i - 1] * v
    return p


def eval_poly(coeffs, degree, y_pows, x_pows):                         # :222-238 (x_pows may be numpy rows)
    val, idx = 0.0, 0
    for total in range(degree + 1):
        for y_pow in range(total, -1, -1):
            val = val + coeffs[idx] * y_pows[y_pow] * x_pows[total - y_pow]
            idx += 1
    return val


def evaluate_polynomial_surface(coeffs, rows, cols, degree):           # :306-337
    nx = np.arange(cols, dtype=np.float64) / float(cols) - 0.5
    x_pows = [np.ones(cols)] + [None] * 6
    for i in range(1, min(degree, 6) + 1):
        x_pows[i] = x_pows[i - 1] * nx
    for i in range(min(degree, 6) + 1, 7):
        x_pows[i] = np.zeros(cols)
    out = np.empty((rows, cols), F)
    for y in range(rows):
        out[y] = eval_poly(coeffs, degree, _pows(y / float(rows) - 0.5, degree), x_pows).astype(F)
    return out


def apply_correction(image, model, mode):                              # :339-374
    with np.errstate(invalid="ignore"):
        fin = model[np.isfinite(model) & (model > F(0.0))]
    model_median = F(0.0) if fin.size == 0 else median_f32(fin)
    with np.errstate(all="ignore"):
        if mode == 0:
            return ((image - model) + model_median).astype(F)
        return np.where(np.abs(model) > F(1e-10), (image / model) * model_median, image).astype(F)


def rms_residual(samples, coeffs, rows, cols, degree):                 # :376-402
    tot = 0.0
    for (sy, sx, sv) in samples:
        ny, nx = float(sy) / float(rows) - 0.5, float(sx) / float(cols) - 0.5
        d = float(sv) - eval_poly(coeffs, degree, _pows(ny, degree), _pows(nx, degree))
        tot += d * d
    return math.sqrt(tot / len(samples))


def extract_background(image, grid_size=8, poly_degree=3, sigma_clip=2.5, iterations=3, mode=0):
    """-> (model, corrected, sample_count, rms_residual, coeffs); raises BackgroundError with the reference's messages"""
    image = np.asarray(image, F)
    rows, cols = image.shape
    samples = auto_sample_grid(image, grid_size, poly_degree, sigma_clip, iterations)
    if len(samples) < min_samples_for_degree(poly_degree):
        raise BackgroundError(f"Not enough background samples ({len(samples)}) for polynomial degree {poly_degree}")
    coeffs = fit_polynomial_surface(samples, rows, cols, poly_degree)
    model = evaluate_polynomial_surface(coeffs, rows, cols, poly_degree)
    return model, apply_correction(image, model, mode), len(samples), rms_residual(samples, coeffs, rows, cols, poly_degree), coeffs
