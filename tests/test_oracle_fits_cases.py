"""Pin the FITS pixel-codec oracle against the reference's unit tests (infra/fits/reader.rs:570-623) and against
numpy's own big-endian views (an independent restatement of the same byte layouts)."""
import struct

import numpy as np


def test_decode_pixels_i16(oracle):                                  # reader.rs:571-578
    px = oracle.fits_decode_pixels(bytes([0x01, 0x00, 0xFF, 0xFF]), 16, 1.0, 0.0)
    assert len(px) == 2 and abs(px[0] - 256.0) < 1e-6 and abs(px[1] + 1.0) < 1e-6


def test_decode_pixels_f32(oracle):                                  # :580-586
    px = oracle.fits_decode_pixels(bytes([0x3F, 0x80, 0x00, 0x00]), -32, 1.0, 0.0)
    assert len(px) == 1 and abs(px[0] - 1.0) < 1e-6


def test_decode_pixels_with_scaling(oracle):                         # :588-593
    assert abs(oracle.fits_decode_pixels(bytes([100]), 8, 2.0, 10.0)[0] - 210.0) < 1e-6


def test_decode_pixels_identity_f32_fast_path(oracle):               # :609-615
    val = np.float32(np.pi)
    assert oracle.fits_decode_pixels(struct.pack(">f", val), -32, 1.0, 0.0)[0] == val


def test_unknown_bitpix_is_empty(oracle):                            # :99
    assert len(oracle.fits_decode_pixels(bytes(16), 24)) == 0


def test_decode_matches_numpy_big_endian_views(oracle):
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, 4096, dtype=np.uint8)
    for bitpix, dt in ((8, "u1"), (16, ">i2"), (32, ">i4")):
        v = raw.view(dt).astype(np.float64)
        assert np.array_equal(oracle.fits_decode_pixels(raw, bitpix), raw.view(dt).astype(np.float32))
        want = (v * 0.25 + 32768.0).astype(np.float32)
        assert np.array_equal(oracle.fits_decode_pixels(raw, bitpix, 0.25, 32768.0), want)
    f = rng.normal(0, 100, 1000).astype(">f4")
    assert np.array_equal(oracle.fits_decode_pixels(f.view(np.uint8), -32), f.astype(np.float32))
    assert np.array_equal(oracle.fits_decode_pixels(f.view(np.uint8), -32, 2.0, -1.0), (f.astype(np.float64) * 2.0 - 1.0).astype(np.float32))
    d = rng.normal(0, 1e5, 1000).astype(">f8")
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.fits_decode_pixels(d.view(np.uint8), -64), d.astype(np.float32))
        assert np.array_equal(oracle.fits_decode_pixels(d.view(np.uint8), -64, 3.0, 7.0), (d.astype(np.float64) * 3.0 + 7.0).astype(np.float32))
    assert len(oracle.fits_decode_pixels(raw[:7], 32)) == 1           # chunks_exact drops the ragged tail


def test_encode_and_roundtrip(oracle):
    rng = np.random.default_rng(1)
    img = rng.normal(1000, 300, (40, 50)).astype(np.float32)
    img[3, 4] = np.nan
    img[5, 6] = np.inf
    assert np.array_equal(oracle.fits_encode_pixels(img, -32).view(">f4"), img.ravel().astype(">f4"), equal_nan=True)
    assert np.array_equal(oracle.fits_encode_pixels(img, -64).view(">f8"), img.ravel().astype(">f8"), equal_nan=True)
    back = oracle.fits_decode_pixels(oracle.fits_encode_pixels(img, -32), -32)
    assert np.array_equal(back, img.ravel(), equal_nan=True)         # f32 BE round trip is lossless
    bz, bs = oracle.fits_compute_bzero_bscale(img)
    fin = img[np.isfinite(img)].astype(np.float64)
    assert bs == (fin.max() - fin.min()) / 65535.0 and bz == fin.min() + bs * 32768.0
    enc = oracle.fits_encode_pixels(img, 16, bz, bs)
    phys = np.clip((img.ravel().astype(np.float64) - bz) / bs, -32768.0, 32767.0)
    want = np.where(np.isnan(phys), 0.0, np.sign(phys) * np.floor(np.abs(phys) + 0.5)).astype(np.int16)   # round half away
    assert np.array_equal(enc.view(">i2").astype(np.int16), want)
    dec = oracle.fits_decode_pixels(enc, 16, bs, bz)
    ok = np.isfinite(img.ravel())
    assert np.abs(dec[ok] - img.ravel()[ok]).max() <= bs * 0.5 + 1e-3   # quantisation step of the i16 encoding
    assert oracle.fits_compute_bzero_bscale(np.full((4, 4), 7.0, np.float32)) == (32768.0, 1.0)    # :153-155
