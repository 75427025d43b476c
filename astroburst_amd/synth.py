"""Deterministic synthetic frame stacks (the reference's sample FITS files are not shipped).

Modelled on the reference's own generator (core/synth/star_field.rs:44-50 Pareto fluxes,
core/synth/psf.rs:123-158 Gaussian PSF, core/synth/noise.rs:17-30 shot + read noise) plus what
exercises the stacking path: cosmic-ray hits so kappa-sigma really rejects, non-finite pixels
(clustered: bad columns / patches, as in real detector masks) and zero-padded borders on some
frames (what registration leaves behind, align.rs:50-52).  Pure torch so the same code makes the
small CPU fixtures and the 4 GB device-resident benchmark stacks.
"""
from __future__ import annotations

import math

import torch


def star_catalog(rows: int, cols: int, n_stars: int, seed: int = 42):
    g = torch.Generator().manual_seed(seed)
    y = torch.rand(n_stars, generator=g, dtype=torch.float64) * (rows - 1)
    x = torch.rand(n_stars, generator=g, dtype=torch.float64) * (cols - 1)
    u = torch.rand(n_stars, generator=g, dtype=torch.float64)
    flux = 100.0 * (1.0 - u).pow(-1.0 / 2.5)  # Pareto(alpha = 2.5), min 100
    return y, x, flux.clamp(max=50000.0)


def render_stars(rows: int, cols: int, cat, fwhm: float = 3.0, device="cpu", dy: float = 0.0, dx: float = 0.0):
    """Sum of Gaussian PSFs normalised to flux, rendered in 15x15 patches."""
    ys, xs, flux = cat
    sigma = fwhm / 2.3548200450309493
    img = torch.zeros(rows * cols, dtype=torch.float32, device=device)
    r = 7
    oy, ox = torch.meshgrid(torch.arange(-r, r + 1), torch.arange(-r, r + 1), indexing="ij")
    oy = oy.reshape(1, -1).to(device)
    ox = ox.reshape(1, -1).to(device)
    cy = (ys + dy).to(device)
    cx = (xs + dx).to(device)
    iy = cy.round().long().reshape(-1, 1) + oy
    ix = cx.round().long().reshape(-1, 1) + ox
    d2 = (iy.double() - cy.reshape(-1, 1)) ** 2 + (ix.double() - cx.reshape(-1, 1)) ** 2
    val = (flux.to(device).reshape(-1, 1) / (2.0 * math.pi * sigma * sigma)) * torch.exp(-d2 / (2.0 * sigma * sigma))
    ok = (iy >= 0) & (iy < rows) & (ix >= 0) & (ix < cols)
    img.index_add_(0, (iy * cols + ix)[ok], val[ok].float())
    return img.reshape(rows, cols)


def make_frame(rows: int, cols: int, k: int, cat=None, device="cpu", sky: float = 200.0, bias: float = 1000.0,
               gain: float = 1.5, read_noise: float = 8.0, cosmic_rate: float = 1e-4, bad_patch_rate: float = 2e-7,
               border: int = 0, shift=(0.0, 0.0), truth=None):
    """Frame k of a stack: truth(sky + stars, shifted) + shot/read noise + cosmic rays + NaN patches
    + optional zero border.  Seed = 123 + 7919 k (core/synth/pipeline.rs:101-103)."""
    g = torch.Generator(device=device).manual_seed(123 + 7919 * k)
    if truth is None:
        truth = torch.full((rows, cols), sky, dtype=torch.float32, device=device)
        if cat is not None:
            truth = truth + render_stars(rows, cols, cat, device=device, dy=shift[0], dx=shift[1])
    electrons = truth * gain
    noise = torch.randn((rows, cols), generator=g, device=device, dtype=torch.float32)
    frame = bias + (electrons + noise * torch.sqrt(electrons.clamp(min=0.0) + read_noise * read_noise)) / gain
    if cosmic_rate > 0:
        hit = torch.rand((rows, cols), generator=g, device=device) < cosmic_rate
        amp = 20.0 + 30.0 * torch.rand((rows, cols), generator=g, device=device)
        frame = torch.where(hit, frame * amp, frame)
    if bad_patch_rate > 0:
        n_patch = max(1, int(bad_patch_rate * rows * cols)) if rows * cols >= 4096 else 0
        for _ in range(n_patch):
            py = int(torch.randint(0, rows, (1,), generator=g, device=device))
            px = int(torch.randint(0, cols, (1,), generator=g, device=device))
            h = int(torch.randint(1, 9, (1,), generator=g, device=device))
            w = int(torch.randint(1, 65, (1,), generator=g, device=device))
            frame[py:py + h, px:px + w] = float("nan")
    if border > 0:
        frame[:border, :] = 0.0
        frame[-border:, :] = 0.0
        frame[:, :border] = 0.0
        frame[:, -border:] = 0.0
    return frame.contiguous()


def make_stack(n: int, rows: int, cols: int, device="cpu", stars_per_mpix: float = 120.0, border_every: int = 10,
               cosmic_rate: float = 1e-4, bad_patch_rate: float = 2e-7):
    """n frames of the same field (no shifts): the north-star stacking workload."""
    n_stars = max(8, int(stars_per_mpix * rows * cols / 1e6))
    cat = star_catalog(rows, cols, n_stars)
    truth = torch.full((rows, cols), 200.0, dtype=torch.float32, device=device) + render_stars(rows, cols, cat,
                                                                                               device=device)
    frames = []
    for k in range(n):
        border = 16 if (border_every and k % border_every == border_every - 1 and min(rows, cols) > 64) else 0
        frames.append(make_frame(rows, cols, k, device=device, truth=truth, border=border, cosmic_rate=cosmic_rate,
                                 bad_patch_rate=bad_patch_rate))
    return frames
