"""Oracle checks for web-rwkv's Int8 / NF4 formats (oracle/wrkquant.py): internal consistency only --
the reference holds no fixture for these, so they stay "parity unpinned"."""
import numpy as np

from oracle import dequant as dq
from oracle import quantize as qz
from oracle import wrkquant as wq


def test_int8_roundtrip_error_bound():
    w = (np.random.default_rng(0).standard_normal(128 * 40) * 0.1).astype(np.float16)
    codes, mm = wq.quantize_int8(w)
    d = wq.dequantize_int8(codes, mm).reshape(-1, 128)
    rng = (mm[:, 1].astype(np.float32) - mm[:, 0].astype(np.float32))[:, None]
    assert np.all(np.abs(d - w.astype(np.float32).reshape(-1, 128)) <= rng / 255 * 0.5001 + 1e-7)
    assert codes.reshape(-1, 128).min(axis=1).max() == 0 and codes.reshape(-1, 128).max(axis=1).min() == 255


def test_int8_constant_block_is_min():
    w = np.full(128, 0.5, np.float16)
    codes, mm = wq.quantize_int8(w)
    assert np.all(codes == 0) and np.all(mm == np.float16(0.5))
    assert np.all(wq.dequantize_int8(codes, mm) == 0.5)


def test_nf4_levels_are_fixed_points_and_ties_take_last():
    amax = np.float16(2.0)
    w = (wq.NF4_LEVELS.astype(np.float32) * 2.0)
    w = np.tile(w, 4).astype(np.float32)
    # only values exactly representable in f16 are fixed points; use the representable subset
    w16 = w.astype(np.float16)
    packed, am = wq.quantize_nf4(w16)
    assert am[0] == amax
    idx = np.empty(64, np.uint8); idx[0::2], idx[1::2] = packed & 15, packed >> 4
    assert np.array_equal(idx[:16][[0, 7, 15]], [0, 7, 15])
    # a value midway between level 7 (0) and level 8: "<=" keeps the later level
    mid = np.float32(wq.NF4_LEVELS[8] / 2)
    blk = np.zeros(64, np.float16); blk[0] = 1.0; blk[1] = np.float16(mid)
    p, _ = wq.quantize_nf4(blk)
    x = np.float32(blk[1])
    e7, e8 = abs(wq.NF4_LEVELS[7] - x), abs(wq.NF4_LEVELS[8] - x)
    assert (p[0] >> 4) == (8 if e8 <= e7 else 7)


def test_nf4_zero_block():
    p, am = wq.quantize_nf4(np.zeros(64, np.float16))
    assert np.all(p == 0) and am[0] == 0
    assert np.all(wq.dequantize_nf4(p, am) == 0)


def test_repack_q8_0_to_int8_tracks_q8_0():
    k, m = 256, 6
    w = (np.random.default_rng(1).standard_normal((m, k)) * 0.05).astype(np.float32)
    raw = qz.QUANTIZE["Q8_0"](w)
    ref = dq.dequantize("Q8_0", raw, k * m, round_f16=False)
    codes, mm = wq.repack_q8_0_to_int8(raw, k * m)
    d = wq.dequantize_int8(codes, mm)
    rng = np.repeat(mm[:, 1].astype(np.float32) - mm[:, 0].astype(np.float32), 128)
    edge = np.repeat(np.abs(mm.astype(np.float32)).max(axis=1), 128)      # min/max are stored as f16
    assert np.all(np.abs(d - ref) <= rng / 255 * 0.51 + edge * 2.0 ** -10 + 1e-6)
