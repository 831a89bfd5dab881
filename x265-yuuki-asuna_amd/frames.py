"""Synthetic clips and picture-plane layout for the frame pipeline.

Plane layout follows the reference's PicYuv (source/common/picyuv.cpp:87-114): the picture is
allocated in whole 64x64 CTUs with a luma margin of maxCU+32 = 96 pixels left/right and
maxCU+16 = 80 rows above/below, stride = ctusW*64 + 2*96; margins replicate the edge pixels
(extendPicBorder, source/common/pixel.cpp:1027-1041) so motion search may read +-merange (+ filter
taps) outside the picture.

The clip generator is the survey's seeded recipe: low-passed random
texture + gradient + sinusoid, global translation (3,2) px/frame, N(0,3) noise per frame.
"""
from __future__ import annotations

import numpy as np

CTU = 64
MARGIN_X = CTU + 32      # picyuv.cpp:93  m_lumaMarginX
MARGIN_Y = CTU + 16      # picyuv.cpp:94  m_lumaMarginY


def padded_dims(width: int, height: int):
    w64 = (width + CTU - 1) // CTU * CTU
    h64 = (height + CTU - 1) // CTU * CTU
    stride = w64 + 2 * MARGIN_X
    rows = h64 + 2 * MARGIN_Y
    org = MARGIN_Y * stride + MARGIN_X          # element offset of pixel (0,0)
    return w64, h64, stride, rows, org


def pad_plane(img: np.ndarray):
    """Picture -> padded plane (CTU-multiple size, replicated margins). Returns (buf, stride, org, w64, h64)."""
    h, w = img.shape
    w64, h64, stride, rows, org = padded_dims(w, h)
    buf = np.pad(img, ((MARGIN_Y, MARGIN_Y + h64 - h), (MARGIN_X, MARGIN_X + w64 - w)), mode="edge")
    assert buf.shape == (rows, stride)
    return np.ascontiguousarray(buf), stride, org, w64, h64


CHROMA_MARGIN_X = MARGIN_X          # picyuv.cpp:97-98: the chroma margin keeps the luma margin's width ...
CHROMA_MARGIN_Y = MARGIN_Y >> 1     # ... and halves its height (4:2:0)


def pad_chroma(img: np.ndarray, w64: int, h64: int):
    """4:2:0 chroma picture -> padded plane of the reference's PicYuv geometry (m_strideC = w64 / 2 + 2 * margin).
    Returns (buf, stride, org)."""
    cw, ch = w64 // 2, h64 // 2
    h, w = img.shape
    buf = np.pad(img, ((CHROMA_MARGIN_Y, CHROMA_MARGIN_Y + ch - h), (CHROMA_MARGIN_X, CHROMA_MARGIN_X + cw - w)), mode="edge")
    stride = cw + 2 * CHROMA_MARGIN_X
    return np.ascontiguousarray(buf), stride, CHROMA_MARGIN_Y * stride + CHROMA_MARGIN_X


def _box5(a):
    k = np.ones(5) / 5.0
    a = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, a)
    return np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, a)


def synth_clip(width: int, height: int, nframes: int, depth: int = 8, seed: int = 265, fade=None):
    """Return list of (Y, U, V) numpy planes (4:2:0), dtype uint8 (depth 8) or uint16.
    fade = (gain of the first picture, gain of the last one): a linear luma fade towards black level 16 - the content x265's default
    --weightp exists for (weightAnalyse picks a weight per reference, encoder/weightPrediction.cpp:222+); None = constant brightness
    (the same pictures as before the parameter existed)."""
    rng = np.random.default_rng(seed)
    tw, th = width + 3 * nframes + 8, height + 2 * nframes + 8
    yy, xx = np.mgrid[0:th, 0:tw].astype(np.float32)
    noise = rng.uniform(0, 255, size=(th, tw)).astype(np.float32)
    # separable 5x5 box filter without scipy
    c = np.cumsum(np.pad(noise, ((0, 0), (3, 2)), mode="edge"), axis=1)
    noise = (c[:, 5:] - c[:, :-5]) / 5.0
    c = np.cumsum(np.pad(noise, ((3, 2), (0, 0)), mode="edge"), axis=0)
    noise = (c[5:, :] - c[:-5, :]) / 5.0
    grad = (xx / tw + yy / th) * 127.5
    tex = (grad + 2.0 * noise) / 3.0 + 40.0 * np.sin(xx / 17.0) * np.cos(yy / 23.0)
    scale = 1 << (depth - 8)
    maxv = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    out = []
    for n in range(nframes):
        crop = tex[2 * n:2 * n + height, 3 * n:3 * n + width]
        y = crop * scale + rng.normal(0.0, 3.0 * scale, size=crop.shape).astype(np.float32)
        if fade is not None:
            gain = fade[0] + (fade[1] - fade[0]) * (n / max(1, nframes - 1))
            y = (y - 16.0 * scale) * np.float32(gain) + 16.0 * scale
        y = np.clip(np.rint(y), 0, maxv).astype(dt)
        sub = y[::2, ::2].astype(np.int32)
        u = np.clip(sub // 2 + 64 * scale, 0, maxv).astype(dt)
        v = np.clip(maxv - sub // 2 - 64 * scale, 0, maxv).astype(dt)
        out.append((y, u, v))
    return out


def write_y4m(path: str, frames, width: int, height: int, depth: int = 8, fps: int = 30):
    """YUV4MPEG2 writer (C420jpeg / C420p10; reference reader: source/input/y4m.cpp:252-285)."""
    tag = "C420jpeg" if depth == 8 else f"C420p{depth}"
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{width} H{height} F{fps}:1 Ip A1:1 {tag}\n".encode())
        for y, u, v in frames:
            f.write(b"FRAME\n")
            for pl in (y, u, v):
                f.write(np.ascontiguousarray(pl).astype("<u2" if depth > 8 else np.uint8).tobytes())


def qpel_cost_table(rng_r: int, lam: float = 4.0, qmax: int = 0):
    """uint16 bit-cost of a QUARTER-pel mv component q in [-4R-8, 4R+8] (index q + qoff), same formula as
    mv_cost_table; the integer-search table is its every-4th entry.  qmax widens the table (the search drivers index
    it with mv - predictor differences and, for one STAR candidate, with 8x the integer mv).  Returns (table, qoff)."""
    qoff = max(4 * rng_r + 8, qmax)
    q = np.arange(-qoff, qoff + 1)
    i = np.abs(q).astype(np.float32)
    bits = (np.log(i + np.float32(1.0)) * np.float32(2.0) / np.log(np.float32(2.0)) + np.float32(1.718)).astype(np.float32)
    bits[i == 0] = np.float32(0.718)
    cost = np.minimum(bits * np.float32(lam) + np.float32(0.5), np.float32(32767.0))
    return cost.astype(np.uint16), qoff


def mv_cost_table(rng_r: int, lam: float = 4.0):
    """uint16 bit-cost of an integer mv component in [-R, R], the reference's formula
    (source/encoder/bitcost.cpp:51-55,103-118: s_bitsizes[i] = log2(i+1)*2 + 1.718 in float,
    cost = min(bits*lambda + 0.5, 32767)), evaluated for quarter-pel distance 4*|d| on the host."""
    d = np.arange(-rng_r, rng_r + 1)
    i = (np.abs(d) * 4).astype(np.float32)
    bits = (np.log(i + np.float32(1.0)) * np.float32(2.0) / np.log(np.float32(2.0)) + np.float32(1.718)).astype(np.float32)
    bits[i == 0] = np.float32(0.718)
    cost = np.minimum(bits * np.float32(lam) + np.float32(0.5), np.float32(32767.0))
    return cost.astype(np.uint16)
