"""Seeded synthetic JPEG-like inputs (quantised DCT coefficient planes).

The reference ships no sample images or fixtures, so tests
and bench.py build their inputs here: a natural-image-like plane (smooth
gradients + a checker pattern + noise + a dark rectangle, the formula of
SURVEY.md section 8d) is forward-DCT'd per 8x8 block and quantised with the
IJG example tables scaled to a JPEG quality, which is what
jpeg_read_coefficients() would hand to do_quantsmooth() (reference
quantsmooth.c:549-550): int16 blocks in natural order, still quantised.

Uniform-random coefficients are deliberately NOT used: they push most blocks
into the saturated / a3 == 0 regimes and make CPU timings unrepresentative.
"""
from __future__ import annotations

import numpy as np

# ITU T.81 Annex K example tables (the ones libjpeg's jpeg_set_quality scales).
STD_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61,
    12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77,
    24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101,
    72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int32)

STD_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99,
    18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99], dtype=np.int32)


def quality_table(base: np.ndarray, quality: int) -> np.ndarray:
    """libjpeg's jpeg_quality_scaling + jpeg_add_quant_table (baseline clamp)."""
    quality = min(max(int(quality), 1), 100)
    scale = 5000 // quality if quality < 50 else 200 - quality * 2
    t = (base.astype(np.int64) * scale + 50) // 100
    return np.clip(t, 1, 255).astype(np.uint16)


def _dct_matrix() -> np.ndarray:
    k = np.arange(8)[:, None]
    n = np.arange(8)[None, :]
    d = np.cos((2 * n + 1) * k * np.pi / 16) * 0.5
    d[0, :] *= np.sqrt(0.5)
    return d  # orthonormal 8-point DCT-II


def synth_pixels(width: int, height: int, seed: int = 1234, variant: int = 0) -> np.ndarray:
    """uint8 plane [height, width]; `variant` shifts the periods (chroma planes)."""
    rng = np.random.default_rng(seed + 7919 * variant)
    x = np.arange(width, dtype=np.float32)[None, :]
    y = np.arange(height, dtype=np.float32)[:, None]
    px, py = 17.0 + 5 * variant, 23.0 + 3 * variant
    img = 128.0 + 60.0 * np.sin(x / px) + 50.0 * np.cos(y / py)
    checker = (((np.arange(width) // (37 + 4 * variant))[None, :] +
                (np.arange(height) // (29 + 2 * variant))[:, None]) & 1).astype(np.float32)
    img = img + 40.0 * (checker - 0.5)
    img = img + rng.normal(0.0, 6.0, size=(height, width)).astype(np.float32)
    y0, y1 = height // 5, height // 5 + max(height // 7, 1)
    x0, x1 = width // 3, width // 3 + max(width // 4, 1)
    img[y0:y1, x0:x1] *= 0.45
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def quantise_plane(pixels: np.ndarray, quant: np.ndarray) -> np.ndarray:
    """uint8 plane (dims multiples of 8) -> int16 [hblk, wblk, 64] quantised coefs."""
    h, w = pixels.shape
    assert h % 8 == 0 and w % 8 == 0
    d = _dct_matrix().astype(np.float32)
    q = quant.astype(np.float32).reshape(8, 8)
    out = np.empty((h // 8, w // 8, 64), dtype=np.int16)
    rows_per_chunk = max(1, (1 << 22) // max(w, 1))  # bound temporaries
    rows_per_chunk = max(8, rows_per_chunk // 8 * 8)
    for r0 in range(0, h, rows_per_chunk):
        r1 = min(h, r0 + rows_per_chunk)
        blk = pixels[r0:r1].astype(np.float32) - 128.0
        blk = blk.reshape((r1 - r0) // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)
        c = np.matmul(np.matmul(d, blk), d.T)
        c = np.rint(c / q)
        out[r0 // 8:r1 // 8] = c.reshape((r1 - r0) // 8, w // 8, 64).astype(np.int16)
    return out


def pad_to_blocks(pixels: np.ndarray, mult_x: int = 8, mult_y: int = 8) -> np.ndarray:
    """edge-replicate to a multiple of the block/MCU size (what an encoder does)."""
    h, w = pixels.shape
    ph = (h + mult_y - 1) // mult_y * mult_y
    pw = (w + mult_x - 1) // mult_x * mult_x
    return np.pad(pixels, ((0, ph - h), (0, pw - w)), mode="edge")


def synth_gray(width: int, height: int, quality: int = 50, seed: int = 1234):
    """-> (coef int16 [hblk, wblk, 64], quant uint16 [64])"""
    quant = quality_table(STD_LUMA, quality)
    pix = pad_to_blocks(synth_pixels(width, height, seed))
    return quantise_plane(pix, quant), quant


def synth_ycc(width: int, height: int, hs: int = 2, vs: int = 2, quality: int = 50, seed: int = 1234):
    """YCbCr job with luma sampling hs x vs and 1x1 chroma (4:2:0 by default).

    -> dict(coefs=[Y, Cb, Cr], quants=[qY, qC, qC], wblk, hblk, hsamp, vsamp)
    following libjpeg's component geometry (width_in_blocks =
    ceil(image_width * h_samp / (max_h * 8)))."""
    qy = quality_table(STD_LUMA, quality)
    qc = quality_table(STD_CHROMA, quality)
    coefs, wblk, hblk = [], [], []
    for ci in range(3):
        h_s, v_s = (hs, vs) if ci == 0 else (1, 1)
        cw = -(-width * h_s // hs)
        ch = -(-height * v_s // vs)
        pix = synth_pixels(cw, ch, seed, variant=ci)
        if ci:  # chroma is smoother and nearer mid-grey in natural images
            pix = np.clip(128 + (pix.astype(np.int32) - 128) // 3, 0, 255).astype(np.uint8)
        wb = -(-width * h_s // (hs * 8))
        hb = -(-height * v_s // (vs * 8))
        pix = np.pad(pix, ((0, hb * 8 - ch), (0, wb * 8 - cw)), mode="edge")
        coefs.append(quantise_plane(pix, qy if ci == 0 else qc))
        wblk.append(wb); hblk.append(hb)
    return dict(coefs=coefs, quants=[qy, qc, qc.copy()], wblk=wblk, hblk=hblk,
                hsamp=[hs, 1, 1], vsamp=[vs, 1, 1], width=width, height=height)
