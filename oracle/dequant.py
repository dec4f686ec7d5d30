"""GGUF block dequantisers (oracle; test infrastructure only).

Restates ``/root/reference/src/runtime/gguf.rs:11-274`` (which follow ggml).
Every function returns float32 values *before* the reference's final
``f16::from_f32`` and, with ``round_f16=True``, the f16-rounded values the
reference actually stores (at HEAD every K-quant tensor is
dequantised to f16 on the CPU at load).

f32 arithmetic order matters for bit-exactness of the f16 rounding; each
expression keeps the reference's association (Rust does not contract a*b-c into
an FMA).
"""
from __future__ import annotations

import numpy as np

QK_K = 256
BLOCK_BYTES = {"F32": 4, "F16": 2, "Q4_0": 18, "Q8_0": 34, "Q2_K": 84, "Q3_K": 110, "Q4_K": 144, "Q5_K": 176, "Q6_K": 210}
BLOCK_ELEMS = {"F32": 1, "F16": 1, "Q4_0": 32, "Q8_0": 32, "Q2_K": 256, "Q3_K": 256, "Q4_K": 256, "Q5_K": 256, "Q6_K": 256}
# gguf.rs:888-923 (GgmlType ids)
GGML_TYPE_ID = {"F32": 0, "F16": 1, "Q4_0": 2, "Q8_0": 8, "Q2_K": 10, "Q3_K": 11, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14, "BF16": 30}
GGML_TYPE_NAME = {v: k for k, v in GGML_TYPE_ID.items()}


def _as_u8(data) -> np.ndarray:
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return a.view(np.uint8).reshape(-1)


def _f16_field(blocks: np.ndarray, off: int) -> np.ndarray:
    """Little-endian f16 at byte offset ``off`` of every block -> float32."""
    return np.ascontiguousarray(blocks[:, off:off + 2]).view("<f2").reshape(-1).astype(np.float32)


def _finish(val: np.ndarray, round_f16: bool) -> np.ndarray:
    val = val.astype(np.float32, copy=False).reshape(-1)
    if round_f16:
        # f16::from_f32 is round-to-nearest-even, as is numpy's cast.
        with np.errstate(over="ignore"):
            return val.astype(np.float16).astype(np.float32)
    return val


def get_scale_min_k4(scales: np.ndarray):
    """gguf.rs:81-89.  scales: uint8 [nb, 12] -> (sc [nb, 8], m [nb, 8]) uint8."""
    s = scales.astype(np.uint8)
    sc = np.empty((s.shape[0], 8), dtype=np.uint8)
    m = np.empty((s.shape[0], 8), dtype=np.uint8)
    for j in range(8):
        if j < 4:
            sc[:, j] = s[:, j] & 63
            m[:, j] = s[:, j + 4] & 63
        else:
            sc[:, j] = (s[:, j + 4] & 0xF) | ((s[:, j - 4] >> 6) << 4)
            m[:, j] = (s[:, j + 4] >> 4) | ((s[:, j] >> 6) << 4)
    return sc, m


def dequantize_q8_0(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:11-37: block = [d f16][32 x i8]; val = q * d."""
    blocks = _as_u8(data)[: (num_elements // 32) * 34].reshape(-1, 34)
    d = _f16_field(blocks, 0)
    q = blocks[:, 2:34].view(np.int8).astype(np.float32)
    return _finish(q * d[:, None], round_f16)


def dequantize_q4_0(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:42-75: block = [d f16][16 bytes]; per byte emits (lo-8)*d then (hi-8)*d.

    NOTE: this interleaved element order is the *reference's* (it differs from
    ggml's lo-half/hi-half order); the oracle restates the reference.
    """
    blocks = _as_u8(data)[: (num_elements // 32) * 18].reshape(-1, 18)
    d = _f16_field(blocks, 0)
    qs = blocks[:, 2:18]
    lo = (qs & 0x0F).astype(np.int8) - 8
    hi = ((qs >> 4) & 0x0F).astype(np.int8) - 8
    q = np.stack([lo, hi], axis=2).reshape(-1, 32).astype(np.float32)
    return _finish(q * d[:, None], round_f16)


def dequantize_q4_k(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:95-143: [d f16][dmin f16][scales 12][qs 128]; val = d*sc*q - dmin*m."""
    blocks = _as_u8(data)[: (num_elements // QK_K) * 144].reshape(-1, 144)
    nb = blocks.shape[0]
    d = _f16_field(blocks, 0)
    dmin = _f16_field(blocks, 2)
    sc, m = get_scale_min_k4(blocks[:, 4:16])
    qs = blocks[:, 16:144].reshape(nb, 4, 32)
    q = np.stack([qs & 0xF, qs >> 4], axis=2).reshape(nb, 8, 32).astype(np.float32)
    d1 = d[:, None] * sc.astype(np.float32)          # d * (sc as f32)
    m1 = dmin[:, None] * m.astype(np.float32)        # dmin * (m as f32)
    val = d1[:, :, None] * q - m1[:, :, None]        # d1 * q - m1 (two roundings)
    return _finish(val, round_f16)


def dequantize_q5_k(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:149-204: [d][dmin][scales 12][qh 32][ql 128]."""
    blocks = _as_u8(data)[: (num_elements // QK_K) * 176].reshape(-1, 176)
    nb = blocks.shape[0]
    d = _f16_field(blocks, 0)
    dmin = _f16_field(blocks, 2)
    sc, m = get_scale_min_k4(blocks[:, 4:16])
    qh = blocks[:, 16:48]                             # [nb, 32]
    ql = blocks[:, 48:176].reshape(nb, 4, 32)
    q = np.empty((nb, 8, 32), dtype=np.float32)
    for j in range(4):
        hb_lo = ((qh >> (2 * j)) & 1).astype(np.uint8) * 16       # u1 = 1 << 2j
        hb_hi = ((qh >> (2 * j + 1)) & 1).astype(np.uint8) * 16   # u2 = 2 << 2j
        q[:, 2 * j, :] = ((ql[:, j, :] & 0xF) + hb_lo).astype(np.float32)
        q[:, 2 * j + 1, :] = ((ql[:, j, :] >> 4) + hb_hi).astype(np.float32)
    d1 = d[:, None] * sc.astype(np.float32)
    m1 = dmin[:, None] * m.astype(np.float32)
    val = d1[:, :, None] * q - m1[:, :, None]
    return _finish(val, round_f16)


def dequantize_q6_k(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:210-274: [ql 128][qh 64][scales 16 x i8][d f16]; val = d*sc*q."""
    blocks = _as_u8(data)[: (num_elements // QK_K) * 210].reshape(-1, 210)
    nb = blocks.shape[0]
    ql = blocks[:, 0:128]
    qh = blocks[:, 128:192]
    scales = blocks[:, 192:208].view(np.int8).astype(np.float32)   # [nb, 16]
    d = _f16_field(blocks, 208)
    out = np.empty((nb, 256), dtype=np.float32)
    ls = np.arange(32) // 16                                       # is = l / 16
    for n in range(2):
        qln = ql[:, 64 * n:64 * n + 64]
        qhn = qh[:, 32 * n:32 * n + 32]
        scn = scales[:, 8 * n:8 * n + 8]
        q1 = ((qln[:, 0:32] & 0xF) | (((qhn >> 0) & 3) << 4)).astype(np.int8) - 32
        q2 = ((qln[:, 32:64] & 0xF) | (((qhn >> 2) & 3) << 4)).astype(np.int8) - 32
        q3 = ((qln[:, 0:32] >> 4) | (((qhn >> 4) & 3) << 4)).astype(np.int8) - 32
        q4 = ((qln[:, 32:64] >> 4) | (((qhn >> 6) & 3) << 4)).astype(np.int8) - 32
        for k, q in enumerate((q1, q2, q3, q4)):
            s = scn[:, ls + 2 * k]                                  # scales[sc_idx + is + 2k]
            ds = d[:, None] * s                                     # (d * sc) ...
            out[:, 128 * n + 32 * k:128 * n + 32 * k + 32] = ds * q.astype(np.float32)  # ... * q
    return _finish(out, round_f16)


def dequantize_q2_k(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:372-423: block = [scales 16][qs 64][d f16][dmin f16] = 84 B; per 128 elements four 2-bit planes of
    qs[32], each plane two 16-element groups with scale byte (low nibble scale, high nibble min):
    val = d*(sc & 15)*q - dmin*(sc >> 4)."""
    nb = num_elements // QK_K
    blocks = _as_u8(data)[: nb * 84].reshape(nb, 84)
    scales, qs = blocks[:, 0:16], blocks[:, 16:80]
    d, dmin = _f16_field(blocks, 80), _f16_field(blocks, 82)
    out = np.empty((nb, QK_K), np.float32)
    for n in range(2):
        for j in range(4):
            for h in range(2):
                is_ = n * 8 + j * 2 + h
                sc = scales[:, is_]
                dl = d * (sc & 0xF).astype(np.float32)
                ml = dmin * (sc >> 4).astype(np.float32)
                q = ((qs[:, n * 32 + h * 16: n * 32 + h * 16 + 16] >> (2 * j)) & 3).astype(np.float32)
                out[:, n * 128 + j * 32 + h * 16: n * 128 + j * 32 + h * 16 + 16] = dl[:, None] * q - ml[:, None]
    return _finish(out, round_f16)


def dequantize_q3_k(data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """gguf.rs:280-366: block = [hmask 32][qs 64][scales 12][d f16] = 110 B; 6-bit scales unpacked with the
    llama.cpp kmask shuffle, val = d*(sc - 32)*(q2 + (hbit ? 0 : -4))."""
    nb = num_elements // QK_K
    blocks = _as_u8(data)[: nb * 110].reshape(nb, 110)
    hmask, qs = blocks[:, 0:32], blocks[:, 32:96]
    raw = np.ascontiguousarray(blocks[:, 96:108]).view("<u4").reshape(nb, 3)
    d_all = _f16_field(blocks, 108)
    K1, K2 = np.uint32(0x03030303), np.uint32(0x0F0F0F0F)
    a0, a1, tmp = raw[:, 0], raw[:, 1], raw[:, 2]
    aux = np.empty((nb, 4), np.uint32)
    aux[:, 2] = ((a0 >> 4) & K2) | (((tmp >> 4) & K1) << 4)
    aux[:, 3] = ((a1 >> 4) & K2) | (((tmp >> 6) & K1) << 4)
    aux[:, 0] = (a0 & K2) | (((tmp >> 0) & K1) << 4)
    aux[:, 1] = (a1 & K2) | (((tmp >> 2) & K1) << 4)
    scales = np.ascontiguousarray(aux).view(np.int8).reshape(nb, 16)
    out = np.empty((nb, QK_K), np.float32)
    for n in range(2):
        for j in range(4):
            m = np.uint8(1 << (n * 4 + j))
            for h in range(2):
                is_ = n * 8 + j * 2 + h
                dl = d_all * (scales[:, is_].astype(np.int32) - 32).astype(np.float32)
                q = ((qs[:, n * 32 + h * 16: n * 32 + h * 16 + 16] >> (2 * j)) & 3).astype(np.int32)
                hv = np.where(hmask[:, h * 16: h * 16 + 16] & m != 0, 0, -4).astype(np.int32)
                out[:, n * 128 + j * 32 + h * 16: n * 128 + j * 32 + h * 16 + 16] = dl[:, None] * (q + hv).astype(np.float32)
    return _finish(out, round_f16)


DEQUANT = {
    "Q8_0": dequantize_q8_0,
    "Q4_0": dequantize_q4_0,
    "Q2_K": dequantize_q2_k,
    "Q3_K": dequantize_q3_k,
    "Q4_K": dequantize_q4_k,
    "Q5_K": dequantize_q5_k,
    "Q6_K": dequantize_q6_k,
}


def data_size(type_name: str, num_elements: int) -> int:
    """gguf.rs:1137-1147 (TensorInfo::data_size)."""
    be, bb = BLOCK_ELEMS[type_name], BLOCK_BYTES[type_name]
    return num_elements * bb if be == 1 else (num_elements // be) * bb


def dequantize(type_name: str, data, num_elements: int, round_f16: bool = True) -> np.ndarray:
    """Reader::tensor for quantised types (gguf.rs:1692-1734)."""
    if type_name == "F32":
        v = np.frombuffer(_as_u8(data)[: num_elements * 4].tobytes(), dtype="<f4").astype(np.float32)
        return _finish(v, round_f16)   # loader.rs:117-121 (F32 -> f16::from_f32)
    if type_name == "F16":
        return np.frombuffer(_as_u8(data)[: num_elements * 2].tobytes(), dtype="<f2").astype(np.float32)
    return DEQUANT[type_name](data, num_elements, round_f16)
