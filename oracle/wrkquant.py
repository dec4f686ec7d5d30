"""web-rwkv's own matrix formats Int8 and NF4 (oracle; test infrastructure only).

Restates
  quantize_mat_int8 : src/shaders/quant_mat_int8.wgsl:24-59   (per 128 elements: min/max f16, code = pack4x8unorm)
  matmul int8 decode: src/shaders/matmul_vec_int8.wgsl:89-92  (w = fma(code/255, max - min, min))
  quantize_mat_nf4  : src/shaders/quant_mat_nf4.wgsl:24-81    (per 64 elements: absmax; nearest of 16 levels, ties -> last)
  NF4 levels        : src/tensor/matrix.rs:50-67
  matmul nf4 decode : src/shaders/matmul_vec_nf4.wgsl:47-80   (w = level[q] * absmax_f16)
  repack_q8_0_to_int8 : src/runtime/gguf.rs:429-520           (direct Q8_0 -> Int8 at load, loader.rs:808-820)
Blocks run over the FLATTENED row-major matrix, as in the shaders.
"""
from __future__ import annotations

import numpy as np

INT8_BLOCK = 128
NF4_BLOCK = 64
NF4_LEVELS = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                       -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                       0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                       0.7229568362236023, 1.0], dtype=np.float32)


def quantize_int8(w16: np.ndarray):
    """w16: f16 values (any shape, size % 128 == 0) -> (codes u8 flat, minmax f16 [nblk, 2])."""
    v = w16.astype(np.float16).astype(np.float32).reshape(-1, INT8_BLOCK)
    mn, mx = v.min(axis=1), v.max(axis=1)
    minmax = np.stack([mn, mx], axis=1).astype(np.float16)
    m0, m1 = minmax[:, 0:1].astype(np.float32), minmax[:, 1:2].astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (v - m0) / (m1 - m0)
    x = np.clip(np.nan_to_num(x, nan=0.0, posinf=1.0, neginf=0.0), 0.0, 1.0)          # saturate
    codes = np.floor(np.float32(0.5) + np.float32(255.0) * x).astype(np.uint8)       # pack4x8unorm
    return codes.reshape(-1), minmax


def dequantize_int8(codes: np.ndarray, minmax: np.ndarray) -> np.ndarray:
    c = codes.reshape(-1, INT8_BLOCK).astype(np.float32) / np.float32(255.0)            # unpack4x8unorm
    m0, m1 = minmax[:, 0:1].astype(np.float32), minmax[:, 1:2].astype(np.float32)
    return (c * (m1 - m0) + m0).astype(np.float32).reshape(-1)


def quantize_nf4(w16: np.ndarray):
    """-> (packed nibbles u8 flat [n/2] (element 2i in the low nibble), absmax f16 [nblk])."""
    v = w16.astype(np.float16).astype(np.float32).reshape(-1, NF4_BLOCK)
    amax = np.abs(v).max(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        x = v * (np.float32(1.0) / amax)[:, None]
    err = np.abs(NF4_LEVELS[None, None, :] - x[:, :, None])                               # [nblk, 64, 16]
    err = np.where(np.isnan(err), np.inf, err)
    # "if abs(q - x) <= min_err" starting from 1.0 keeps the LAST minimal level
    idx = (15 - np.argmin(err[:, :, ::-1], axis=2)).astype(np.uint8)
    idx = np.where(err.min(axis=2) <= 1.0, idx, 0).astype(np.uint8)
    flat = idx.reshape(-1)
    packed = (flat[0::2] | (flat[1::2] << 4)).astype(np.uint8)
    return packed, amax.astype(np.float16)


def dequantize_nf4(packed: np.ndarray, absmax: np.ndarray) -> np.ndarray:
    idx = np.empty(packed.size * 2, np.uint8)
    idx[0::2], idx[1::2] = packed & 0xF, packed >> 4
    return (NF4_LEVELS[idx].reshape(-1, NF4_BLOCK) * absmax.astype(np.float32)[:, None]).astype(np.float32).reshape(-1)


def repack_q8_0_to_int8(data: np.ndarray, num_elements: int):
    """gguf.rs:429-520 (full 128-element blocks only; the tail branch is not needed for K % 128 == 0)."""
    blocks = np.asarray(data, np.uint8)[: (num_elements // 32) * 34].reshape(-1, 34)
    d = np.ascontiguousarray(blocks[:, 0:2]).view("<f2").reshape(-1).astype(np.float32)
    val = (blocks[:, 2:34].view(np.int8).astype(np.float32) * d[:, None]).reshape(-1, INT8_BLOCK)
    mn, mx = val.min(axis=1), val.max(axis=1)
    rng = mx - mn
    inv = np.where(rng > 0, np.float32(255.0) / np.where(rng > 0, rng, 1), np.float32(0.0)).astype(np.float32)
    t = (val - mn[:, None]) * inv[:, None]
    codes = np.where(t >= 0, np.floor(t + np.float32(0.5)), np.ceil(t - np.float32(0.5)))      # f32::round: half away from zero
    return np.clip(codes, 0, 255).astype(np.uint8).reshape(-1), np.stack([mn, mx], axis=1).astype(np.float16)


def repack_q4_0_to_nf4(data: np.ndarray, num_elements: int):
    """gguf.rs:528-627 (the live (Q4_0, Quant::NF4) arm, loader.rs:901-918): two Q4_0 blocks -> one NF4 block.
    Byte j of a Q4_0 block yields elements 2j (low nibble) and 2j+1 of the output, nearest level with the FIRST
    minimum (Iterator::min_by), absmax == 0 -> inv 0 -> level 7."""
    blocks = np.asarray(data, np.uint8)[: (num_elements // 32) * 18].reshape(-1, 18)
    d = np.ascontiguousarray(blocks[:, 0:2]).view("<f2").reshape(-1).astype(np.float32)
    qs = blocks[:, 2:18]
    vals = np.empty((blocks.shape[0], 32), np.float32)
    vals[:, 0::2] = ((qs & 0xF).astype(np.int8) - 8).astype(np.float32) * d[:, None]
    vals[:, 1::2] = ((qs >> 4).astype(np.int8) - 8).astype(np.float32) * d[:, None]
    v = vals.reshape(-1, NF4_BLOCK)
    amax = np.abs(v).max(axis=1)
    inv = np.where(amax > 0, np.float32(1.0) / np.where(amax > 0, amax, 1), np.float32(0.0)).astype(np.float32)
    x = v * inv[:, None]
    idx = np.argmin(np.abs(NF4_LEVELS[None, None, :] - x[:, :, None]), axis=2).astype(np.uint8).reshape(-1)
    return (idx[0::2] | (idx[1::2] << 4)).astype(np.uint8), amax.astype(np.float16)


def quantile_student(nu: float) -> np.ndarray:
    """quantile_student (matrix.rs:29-44): Student-t quantiles (scipy.stats.t.ppf stands in for statrs 0.18's inverse_cdf) at the
    reference's 16 probabilities, normalised by the largest; f32 like the reference's Vec<f32>."""
    from scipy.stats import t as student
    delta = (1.0 / 32.0 + 1.0 / 30.0) / 2.0
    p = [delta + (0.5 - delta) / 7.0 * i for i in range(7)] + [0.5 + (1.0 - delta - 0.5) / 8.0 * i for i in range(9)]
    q = student.ppf(np.asarray(p, np.float64), nu)
    return (q / q.max()).astype(np.float32)
