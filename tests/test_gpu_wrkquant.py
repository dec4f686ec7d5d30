"""GPU parity of web-rwkv's own matrix formats: Matrix::quant_u8 / quant_nf4 / quant_sf4 on the
device, Matrix::Int8 / Matrix::Fp4 uploads (the direct-load arms), and their matmul, against oracle/wrkquant.py.

Quantised planes are integer work: the side tables must match bit for bit, and the codes bit for bit except
where the reference's own arithmetic is not pinned -- quant_mat_int8.wgsl divides in f32 on the GPU and WGSL
lets a division be off by 2.5 ULP, so a value that lands within that of a rounding boundary may legitimately
take either neighbouring code.  The test therefore allows code differences of exactly 1 on at most 1e-4 of
the elements and requires every one of them to sit on such a boundary.
"""
import numpy as np
import pytest

import wrk
from oracle import quantize as qz
from oracle import wrkquant as wq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def weights(k, m, seed, zero_block=False):
    r = np.random.default_rng(seed)
    w = (r.standard_normal((m, k)) / np.sqrt(k)).astype(np.float16)
    if zero_block:
        w[0, :128] = 0            # max == min and absmax == 0 blocks
        w[1, :128] = 0.25
    return w


SHAPES = [(128, 5), (256, 64), (320, 64), (768, 33), (2048, 40), (8192, 8)]     # 320: Int8 blocks straddle rows


@pytest.mark.parametrize("k,m", SHAPES)
def test_quant_u8_codes_and_minmax(ctx, k, m):
    w = weights(k, m, k + m, zero_block=True)
    mat = wrk.Matrix.quant_u8(wrk.Buffer(ctx, w.nbytes, w), k, m)
    blob = mat.export()
    codes, minmax = blob[: k * m], blob[k * m:].view(np.float16).reshape(-1, 2)
    want_codes, want_mm = wq.quantize_int8(w)
    assert np.array_equal(minmax.view(np.uint16), want_mm.view(np.uint16))
    diff = codes.astype(np.int32) - want_codes.astype(np.int32)
    bad = np.flatnonzero(diff)
    assert bad.size <= 1e-4 * codes.size and np.all(np.abs(diff[bad]) == 1), (bad.size, np.abs(diff).max())
    if bad.size:       # each must be a rounding-boundary case
        v = w.astype(np.float32).reshape(-1)[bad]
        mm = want_mm.astype(np.float32)[bad // 128]
        t = 255.0 * (v - mm[:, 0]) / (mm[:, 1] - mm[:, 0]) + 0.5
        assert np.all(np.abs(t - np.round(t)) < 1e-3)
    assert mat.stream_bytes == k * m + (k * m // 128) * 4
    # and the quantised matrix multiplies like the oracle's reconstruction of the DEVICE codes
    wd = wq.dequantize_int8(codes, minmax).reshape(m, k)
    x = np.random.default_rng(0).standard_normal(k).astype(np.float16)
    out = ctx.zeros([m, 1, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, 1, 1]), out)
    want = wd.astype(np.float64) @ x.astype(np.float64)
    bound = 4e-6 * (np.abs(wd).astype(np.float64) @ np.abs(x).astype(np.float64)) + 1e-6
    assert np.all(np.abs(out.back().reshape(m) - want) <= bound)


@pytest.mark.parametrize("k,m", [(64, 3)] + SHAPES)
def test_quant_nf4_codes_and_absmax(ctx, k, m):
    w = weights(k, m, 3 * k + m, zero_block=k >= 128)
    mat = wrk.Matrix.quant_nf4(wrk.Buffer(ctx, w.nbytes, w), k, m)
    blob = mat.export()
    n = k * m
    packed, absmax, levels = blob[: n // 2], blob[n // 2: n // 2 + n // 64 * 2].view(np.float16), blob[-64:].view(np.float32)
    want_packed, want_amax = wq.quantize_nf4(w)
    assert np.array_equal(levels, wq.NF4_LEVELS)
    assert np.array_equal(absmax.view(np.uint16), want_amax.view(np.uint16))
    assert np.array_equal(packed, want_packed)
    assert mat.stream_bytes == n // 2 + n // 64 * 2


@pytest.mark.parametrize("student", [False, True])
def test_quant_sf4_levels(ctx, student):
    """quant_sf4 uses the same kernel with the caller's level table: Float4Quant::new_student(5.0) from the host's
    quantile_student (matrix.rs:29-44, 251-271), or any other 16 levels."""
    k, m = 256, 16
    if student:
        levels = wrk.quantile_student(5.0)
        assert np.allclose(levels, wq.quantile_student(5.0), rtol=1e-6, atol=1e-7)
    else:
        levels = np.sort(np.tanh(np.linspace(-2.0, 2.0, 16))).astype(np.float32)
        levels /= levels.max()
    w = weights(k, m, 9)
    mat = wrk.Matrix.quant_sf4(wrk.Buffer(ctx, w.nbytes, w), k, m, levels)
    blob = mat.export()
    n = k * m
    saved = wq.NF4_LEVELS.copy()
    try:
        wq.NF4_LEVELS[:] = levels
        want_packed, want_amax = wq.quantize_nf4(w)
        assert np.array_equal(blob[: n // 2], want_packed)
        x = np.random.default_rng(2).standard_normal(k).astype(np.float16)
        out = ctx.zeros([m, 1, 1], np.float32)
        mat.matmul_op(ctx.tensor(x, [k, 1, 1]), out)
        wd = wq.dequantize_nf4(want_packed, want_amax).reshape(m, k)
    finally:
        wq.NF4_LEVELS[:] = saved
    want = wd.astype(np.float64) @ x.astype(np.float64)
    bound = 4e-6 * (np.abs(wd).astype(np.float64) @ np.abs(x).astype(np.float64)) + 1e-6
    assert np.all(np.abs(out.back().reshape(m) - want) <= bound)


@pytest.mark.parametrize("kind", ["INT8", "NF4"])
@pytest.mark.parametrize("k,m", SHAPES)
@pytest.mark.parametrize("T,B", [(1, 1), (3, 2), (9, 1), (33, 1)])
def test_matmul_uploaded_planes(ctx, kind, k, m, T, B):
    """Matrix::Int8 {w, m} / Matrix::Fp4 {w, q, m} built on the host (the direct-load arms) and multiplied."""
    w = weights(k, m, k * 3 + m)
    if kind == "INT8":
        codes, mm = wq.quantize_int8(w)
        blob = np.concatenate([codes, mm.reshape(-1).view(np.uint8)])
        wd = wq.dequantize_int8(codes, mm).reshape(m, k)
    else:
        packed, amax = wq.quantize_nf4(w)
        blob = np.concatenate([packed, amax.view(np.uint8)])
        wd = wq.dequantize_nf4(packed, amax).reshape(m, k)
    mat = wrk.Matrix(ctx, kind, k, m, blob)
    assert np.array_equal(mat.export()[: blob.size], blob)
    x = np.random.default_rng(T + B).standard_normal((B, T, k)).astype(np.float16)
    out = ctx.zeros([m, T, B], np.float32)
    mat.matmul_op(ctx.tensor(x), out, turbo=T >= 16)
    got = out.back().reshape(B, T, m)
    want = x.astype(np.float64) @ wd.astype(np.float64).T
    bound = 4e-6 * (np.abs(x).astype(np.float64) @ np.abs(wd).astype(np.float64).T) + 1e-6
    assert np.all(np.abs(got - want) <= bound), np.abs(got - want).max()


@pytest.mark.parametrize("c,r,t", [(2560, 2048, 64), (320, 64, 320)])
@pytest.mark.parametrize("kind", ["INT8", "NF4"])
def test_reference_test_shapes(ctx, kind, c, r, t):
    """The shapes and value ranges of the reference's own test_matmul_int8 / test_matmul_nf4 (ops.rs:3642-3983):
    matrix and input uniform in [-5, 5) (NF4: normal), quantise on the device, matmul_vec and matmul_mat against
    the CPU definition; the reference accepts |a - b| <= max(0.01, 0.01 * max|a|,|b|) and codes off by < 2."""
    rng = np.random.default_rng(42)
    if kind == "INT8":
        w = (10.0 * (rng.random((r, c), np.float32) - 0.5)).astype(np.float16)
    else:
        w = rng.standard_normal((r, c)).astype(np.float16)
    x = (10.0 * (rng.random((t, c), np.float32) - 0.5)).astype(np.float16)
    src = wrk.Buffer(ctx, w.nbytes, w)
    mat = wrk.Matrix.quant_u8(src, c, r) if kind == "INT8" else wrk.Matrix.quant_nf4(src, c, r)
    blob = mat.export()
    n = c * r
    if kind == "INT8":
        codes, mm = blob[:n], blob[n:].view(np.float16).reshape(-1, 2)
        wc, wmm = wq.quantize_int8(w)
        assert np.array_equal(mm.view(np.uint16), wmm.view(np.uint16))
        assert np.abs(codes.astype(np.int32) - wc.astype(np.int32)).max() < 2
        wd = wq.dequantize_int8(codes, mm).reshape(r, c)
    else:
        packed, am = blob[: n // 2], blob[n // 2: n // 2 + n // 64 * 2].view(np.float16)
        wp, wam = wq.quantize_nf4(w)
        assert np.array_equal(am.view(np.uint16), wam.view(np.uint16)) and np.array_equal(packed, wp)
        wd = wq.dequantize_nf4(packed, am).reshape(r, c)
    ans = x.astype(np.float64) @ wd.astype(np.float64).T
    for turbo in (False, True):
        out = ctx.zeros([r, t, 1], np.float32)
        mat.matmul_op(ctx.tensor(x, [c, t, 1]), out, turbo=turbo)
        got = out.back().reshape(t, r)
        assert np.all(np.abs(got - ans) <= np.maximum(0.01, 0.01 * np.maximum(np.abs(got), np.abs(ans))))
        bound = 4e-6 * (np.abs(x).astype(np.float64) @ np.abs(wd).astype(np.float64).T) + 1e-5     # our own, tighter bar
        assert np.all(np.abs(got - ans) <= bound), np.abs(got - ans).max()


def test_repack_q8_0_to_int8_matmul(ctx):
    """loader.rs:808-820: a Q8_0 GGUF tensor loaded under Quant::Int8 is repacked on the host (gguf.rs:429-520)."""
    k, m = 512, 24
    w = (np.random.default_rng(4).standard_normal((m, k)) / np.sqrt(k)).astype(np.float32)
    raw = qz.QUANTIZE["Q8_0"](w)
    codes, mm = wq.repack_q8_0_to_int8(raw, k * m)
    mat = wrk.Matrix(ctx, "INT8", k, m, np.concatenate([codes, mm.reshape(-1).view(np.uint8)]))
    wd = wq.dequantize_int8(codes, mm).reshape(m, k)
    x = np.random.default_rng(5).standard_normal(k).astype(np.float16)
    out = ctx.zeros([m, 1, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, 1, 1]), out)
    want = wd.astype(np.float64) @ x.astype(np.float64)
    bound = 4e-6 * (np.abs(wd).astype(np.float64) @ np.abs(x).astype(np.float64)) + 1e-6
    assert np.all(np.abs(out.back().reshape(m) - want) <= bound)
    assert np.abs(wd - w).max() < 0.02       # still a faithful copy of the Q8_0 weights


def test_bad_shapes_rejected(ctx):
    with pytest.raises(wrk.WrkError):
        wrk.Matrix(ctx, "INT8", 200, 16, np.zeros(200 * 16 + 100, np.uint8))    # K % 16
    with pytest.raises(wrk.WrkError):
        wrk.Matrix(ctx, "NF4", 128, 4, np.zeros(10, np.uint8))                  # wrong byte count
    w = weights(96, 4, 1)
    with pytest.raises(wrk.WrkError):
        wrk.Matrix.quant_nf4(wrk.Buffer(ctx, w.nbytes, w), 96, 4)
