"""Fixture-side block quantisers (oracle; test infrastructure only).

The reference has no float->GGUF quantiser (its converter relies on the
un-vendored ``gguf`` package), so these are this repo's own, used only to
*manufacture* legal GGUF blocks for fixtures and synthetic benchmarks.  They are
simple min/max quantisers, not ggml's error-minimising search; any legal block
is a valid input for the dequantisers under test.
"""
from __future__ import annotations

import numpy as np

QK_K = 256


def _f16_bytes(x: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(x.astype("<f2")).view(np.uint8).reshape(x.shape + (2,))


def pack_scale_min_k4(sc: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Inverse of get_scale_min_k4: 6-bit sc[nb,8], m[nb,8] -> scales[nb,12]."""
    sc = sc.astype(np.uint8)
    m = m.astype(np.uint8)
    s = np.zeros((sc.shape[0], 12), dtype=np.uint8)
    for j in range(4):
        s[:, j] = (sc[:, j] & 63) | ((sc[:, j + 4] >> 4) << 6)
        s[:, j + 4] = (m[:, j] & 63) | ((m[:, j + 4] >> 4) << 6)
        s[:, j + 8] = (sc[:, j + 4] & 0xF) | ((m[:, j + 4] & 0xF) << 4)
    return s


def _k4_scales(x: np.ndarray, qmax: int):
    """x [nb, 8, 32] -> d, dmin (f16-rounded f32), sc, m (6 bit) for w = d*sc*q - dmin*m."""
    mn = np.minimum(x.min(axis=2), 0.0)
    mx = x.max(axis=2)
    scale = np.maximum(mx - mn, 1e-8) / qmax           # per sub-block step
    neg = -mn                                          # per sub-block offset (>= 0)
    d = (scale.max(axis=1) / 63.0).astype(np.float16).astype(np.float32)
    dmin = (neg.max(axis=1) / 63.0).astype(np.float16).astype(np.float32)
    d = np.where(d > 0, d, np.float32(1e-7).astype(np.float16).astype(np.float32))
    dmin_safe = np.where(dmin > 0, dmin, 1.0)
    sc = np.clip(np.rint(scale / d[:, None]), 1, 63).astype(np.uint8)
    m = np.clip(np.rint(neg / dmin_safe[:, None]), 0, 63).astype(np.uint8)
    return d, dmin, sc, m


def quantize_q4_k(w: np.ndarray) -> np.ndarray:
    x = w.astype(np.float32).reshape(-1, 8, 32)
    nb = x.shape[0]
    d, dmin, sc, m = _k4_scales(x, 15)
    step = d[:, None] * sc.astype(np.float32)
    off = dmin[:, None] * m.astype(np.float32)
    q = np.clip(np.rint((x + off[:, :, None]) / step[:, :, None]), 0, 15).astype(np.uint8)
    q = q.reshape(nb, 4, 2, 32)
    qs = (q[:, :, 0, :] | (q[:, :, 1, :] << 4)).reshape(nb, 128)
    out = np.empty((nb, 144), dtype=np.uint8)
    out[:, 0:2] = _f16_bytes(d)
    out[:, 2:4] = _f16_bytes(dmin)
    out[:, 4:16] = pack_scale_min_k4(sc, m)
    out[:, 16:144] = qs
    return out.reshape(-1)


def quantize_q5_k(w: np.ndarray) -> np.ndarray:
    x = w.astype(np.float32).reshape(-1, 8, 32)
    nb = x.shape[0]
    d, dmin, sc, m = _k4_scales(x, 31)
    step = d[:, None] * sc.astype(np.float32)
    off = dmin[:, None] * m.astype(np.float32)
    q = np.clip(np.rint((x + off[:, :, None]) / step[:, :, None]), 0, 31).astype(np.uint8)
    q = q.reshape(nb, 4, 2, 32)
    ql = ((q[:, :, 0, :] & 0xF) | ((q[:, :, 1, :] & 0xF) << 4)).reshape(nb, 128)
    qh = np.zeros((nb, 32), dtype=np.uint8)
    for j in range(4):
        qh |= ((q[:, j, 0, :] >> 4) & 1) << (2 * j)
        qh |= ((q[:, j, 1, :] >> 4) & 1) << (2 * j + 1)
    out = np.empty((nb, 176), dtype=np.uint8)
    out[:, 0:2] = _f16_bytes(d)
    out[:, 2:4] = _f16_bytes(dmin)
    out[:, 4:16] = pack_scale_min_k4(sc, m)
    out[:, 16:48] = qh
    out[:, 48:176] = ql
    return out.reshape(-1)


def quantize_q6_k(w: np.ndarray) -> np.ndarray:
    x = w.astype(np.float32).reshape(-1, 16, 16)       # 16 sub-blocks of 16
    nb = x.shape[0]
    amax = np.abs(x).max(axis=2)                       # [nb,16]
    scale = np.maximum(amax, 1e-8) / 31.0
    d = (scale.max(axis=1) / 127.0).astype(np.float16).astype(np.float32)
    d = np.where(d > 0, d, np.float32(1e-7).astype(np.float16).astype(np.float32))
    sc = np.clip(np.rint(scale / d[:, None]), 1, 127).astype(np.int8)
    step = d[:, None] * sc.astype(np.float32)
    q = (np.clip(np.rint(x / step[:, :, None]), -32, 31).astype(np.int16) + 32).astype(np.uint8)
    q = q.reshape(nb, 256)
    ql = np.zeros((nb, 128), dtype=np.uint8)
    qh = np.zeros((nb, 64), dtype=np.uint8)
    for n in range(2):
        e = q[:, 128 * n:128 * n + 128]
        q1, q2, q3, q4 = e[:, 0:32], e[:, 32:64], e[:, 64:96], e[:, 96:128]
        ql[:, 64 * n:64 * n + 32] = (q1 & 0xF) | ((q3 & 0xF) << 4)
        ql[:, 64 * n + 32:64 * n + 64] = (q2 & 0xF) | ((q4 & 0xF) << 4)
        qh[:, 32 * n:32 * n + 32] = (q1 >> 4) | ((q2 >> 4) << 2) | ((q3 >> 4) << 4) | ((q4 >> 4) << 6)
    out = np.empty((nb, 210), dtype=np.uint8)
    out[:, 0:128] = ql
    out[:, 128:192] = qh
    out[:, 192:208] = sc.view(np.uint8)
    out[:, 208:210] = _f16_bytes(d)
    return out.reshape(-1)


def quantize_q8_0(w: np.ndarray) -> np.ndarray:
    x = w.astype(np.float32).reshape(-1, 32)
    nb = x.shape[0]
    d = (np.abs(x).max(axis=1) / 127.0).astype(np.float16).astype(np.float32)
    d = np.where(d > 0, d, np.float32(1e-7).astype(np.float16).astype(np.float32))
    q = np.clip(np.rint(x / d[:, None]), -127, 127).astype(np.int8)
    out = np.empty((nb, 34), dtype=np.uint8)
    out[:, 0:2] = _f16_bytes(d)
    out[:, 2:34] = q.view(np.uint8)
    return out.reshape(-1)


def quantize_q4_0(w: np.ndarray) -> np.ndarray:
    """Element order follows the reference's decoder (interleaved lo/hi per byte)."""
    x = w.astype(np.float32).reshape(-1, 32)
    nb = x.shape[0]
    d = (np.abs(x).max(axis=1) / 7.0).astype(np.float16).astype(np.float32)
    d = np.where(d > 0, d, np.float32(1e-7).astype(np.float16).astype(np.float32))
    q = (np.clip(np.rint(x / d[:, None]), -8, 7).astype(np.int8) + 8).astype(np.uint8).reshape(nb, 16, 2)
    out = np.empty((nb, 18), dtype=np.uint8)
    out[:, 0:2] = _f16_bytes(d)
    out[:, 2:18] = q[:, :, 0] | (q[:, :, 1] << 4)
    return out.reshape(-1)


def quantize_q2_k(w: np.ndarray) -> np.ndarray:
    """Fixture quantiser (not ggml's search): per 16-element group min/scale on 4-bit grids of the block's d / dmin."""
    x = w.astype(np.float32).reshape(-1, 16, 16)                      # [nb, group, elem]; group order = decoder's `is`
    nb = x.shape[0]
    mn = np.minimum(x.min(axis=2), 0.0)
    rng = x.max(axis=2) - mn
    dmin = (np.abs(mn).max(axis=1) / 15.0).astype(np.float16).astype(np.float32)
    d = (rng.max(axis=1) / 3.0 / 15.0).astype(np.float16).astype(np.float32)
    dmin = np.where(dmin > 0, dmin, np.float32(1e-6)); d = np.where(d > 0, d, np.float32(1e-6))
    m4 = np.clip(np.rint(-mn / dmin[:, None]), 0, 15).astype(np.uint8)
    s4 = np.clip(np.rint(rng / 3.0 / d[:, None]), 1, 15).astype(np.uint8)
    q = np.clip(np.rint((x + (dmin[:, None] * m4)[:, :, None]) / (d[:, None] * s4)[:, :, None]), 0, 3).astype(np.uint8)
    out = np.zeros((nb, 84), np.uint8)
    out[:, 0:16] = s4 | (m4 << 4)
    # element (n, j, h, l) = group n*8 + j*2 + h, lives in qs[n*32 + h*16 + l] bits 2j..2j+1
    qs = np.zeros((nb, 64), np.uint8)
    for n in range(2):
        for j in range(4):
            for h in range(2):
                qs[:, n * 32 + h * 16: n * 32 + h * 16 + 16] |= q[:, n * 8 + j * 2 + h, :] << (2 * j)
    out[:, 16:80] = qs
    out[:, 80:82] = _f16_bytes(d.astype(np.float16).astype(np.float32))
    out[:, 82:84] = _f16_bytes(dmin.astype(np.float16).astype(np.float32))
    return out.reshape(-1)


def quantize_q3_k(w: np.ndarray) -> np.ndarray:
    """Fixture quantiser: 3-bit signed codes in [-4, 3] with a 6-bit (sc - 32) scale per 16 elements."""
    x = w.astype(np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    amax = np.abs(x).max(axis=2)
    d = (amax.max(axis=1) / 4.0 / 31.0).astype(np.float16).astype(np.float32)
    d = np.where(d > 0, d, np.float32(1e-6))
    sc = np.clip(np.rint(amax / 4.0 / d[:, None]), 1, 31).astype(np.int32)        # (stored - 32) in [1, 31]
    q = np.clip(np.rint(x / (d[:, None] * sc)[:, :, None]), -4, 3).astype(np.int32)
    stored = (sc + 32).astype(np.uint8)                                            # 6-bit value
    out = np.zeros((nb, 110), np.uint8)
    hmask = np.zeros((nb, 32), np.uint8); qs = np.zeros((nb, 64), np.uint8)
    for n in range(2):
        for j in range(4):
            for h in range(2):
                g = q[:, n * 8 + j * 2 + h, :]
                hi = (g >= 0)                                   # hbit set <=> no -4 offset
                lo = np.where(hi, g, g + 4).astype(np.uint8)
                qs[:, n * 32 + h * 16: n * 32 + h * 16 + 16] |= lo << (2 * j)
                hmask[:, h * 16: h * 16 + 16] |= hi.astype(np.uint8) << (n * 4 + j)
    # pack the sixteen 6-bit scales: inverse of the decoder's kmask shuffle (low 4 bits in aux0/aux1 nibbles, high 2 in tmp)
    lo4, hi2 = (stored & 0xF).astype(np.uint32), (stored >> 4).astype(np.uint32)
    a0 = np.zeros(nb, np.uint32); a1 = np.zeros(nb, np.uint32); tmp = np.zeros(nb, np.uint32)
    for k in range(4):
        a0 |= (lo4[:, k] | (lo4[:, 8 + k] << 4)) << (8 * k)
        a1 |= (lo4[:, 4 + k] | (lo4[:, 12 + k] << 4)) << (8 * k)
        tmp |= (hi2[:, k] | (hi2[:, 4 + k] << 2) | (hi2[:, 8 + k] << 4) | (hi2[:, 12 + k] << 6)) << (8 * k)
    out[:, 0:32] = hmask; out[:, 32:96] = qs
    out[:, 96:108] = np.stack([a0, a1, tmp], axis=1).astype("<u4").view(np.uint8).reshape(nb, 12)
    out[:, 108:110] = _f16_bytes(d)
    return out.reshape(-1)


QUANTIZE = {
    "Q4_K": quantize_q4_k,
    "Q5_K": quantize_q5_k,
    "Q6_K": quantize_q6_k,
    "Q8_0": quantize_q8_0,
    "Q4_0": quantize_q4_0,
    "Q2_K": quantize_q2_k,
    "Q3_K": quantize_q3_k,
    "F16": lambda w: np.ascontiguousarray(w.astype("<f2")).view(np.uint8).reshape(-1),
    "F32": lambda w: np.ascontiguousarray(w.astype("<f4")).view(np.uint8).reshape(-1),
}
