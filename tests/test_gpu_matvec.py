"""GPU parity of the inline-dequant matvec (C ABI wrk_op_matmul) against the oracle dequantisers.

The oracle dequantises with the reference's CPU routines (gguf.rs:11-274) and contracts in f64;
the kernel accumulates exact f16 x f16 products in f32, so for f32 outputs
  |got - want| <= 4e-6 * sum|w||x| + 1e-6        (f32 accumulation of K terms)
and f16 outputs are within one f16 ulp.  WRK_MATRIX_ROUND_F16 is checked against weights rounded
to f16 (the reference's effective path); the default mode against f32-exact weights.
"""
import numpy as np
import pytest

import wrk
from oracle import dequant as dq
from oracle import quantize as qz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def make(kind, k, m, seed):
    r = np.random.default_rng(seed)
    w = (r.standard_normal((m, k)) / np.sqrt(k)).astype(np.float32)
    raw = qz.QUANTIZE[kind](w)
    return raw


CASES = [("Q4_K", 256, 7), ("Q4_K", 2048, 64), ("Q4_K", 2560, 33), ("Q4_K", 8192, 16),
         ("Q5_K", 512, 9), ("Q5_K", 4096, 32), ("Q6_K", 256, 5), ("Q6_K", 2048, 130), ("Q6_K", 2560, 12),
         ("Q8_0", 96, 10), ("Q8_0", 4096, 40), ("Q8_0", 64, 2048),
         ("F16", 96, 2048), ("F16", 2048, 96), ("F16", 768, 100), ("F32", 256, 64),
         # rows longer than 8192 (block kinds) / 2048 (F16): K split over the waves with two chunk iterations (RWKV-6 7B shapes)
         ("Q5_K", 14336, 20), ("Q4_K", 16384, 8), ("Q6_K", 10240, 12), ("Q8_0", 12288, 9), ("F16", 4096, 40)]


@pytest.mark.parametrize("kind,k,m", CASES)
@pytest.mark.parametrize("flags", [wrk.MATRIX_EXACT, wrk.MATRIX_ROUND_F16])
def test_matvec_single_token(ctx, kind, k, m, flags):
    raw = make(kind, k, m, k + m)
    mat = wrk.Matrix(ctx, kind, k, m, raw, flags)
    assert mat.stream_bytes == (raw.size if kind != "F32" else raw.size // 2)
    w = dq.dequantize(kind, raw, k * m, round_f16=(flags == wrk.MATRIX_ROUND_F16 or kind == "F32")).reshape(m, k)
    x = (np.random.default_rng(1).standard_normal(k)).astype(np.float16)
    out = ctx.zeros([m, 1, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, 1, 1]), out)
    got = out.back().reshape(m)
    want = w.astype(np.float64) @ x.astype(np.float64)
    bound = 4e-6 * (np.abs(w).astype(np.float64) @ np.abs(x).astype(np.float64)) + 1e-6
    assert np.all(np.abs(got - want) <= bound), np.abs(got - want).max()


@pytest.mark.parametrize("kind,k,m", [("Q4_K", 2048, 48), ("Q6_K", 512, 20), ("Q5_K", 256, 8), ("Q8_0", 1024, 24), ("F16", 256, 40)])
@pytest.mark.parametrize("T,B", [(2, 1), (3, 1), (5, 2), (9, 1), (1, 4)])
def test_matvec_stacked_tokens_activation_f16_out(ctx, kind, k, m, T, B):
    raw = make(kind, k, m, 3)
    mat = wrk.Matrix(ctx, kind, k, m, raw)
    w = dq.dequantize(kind, raw, k * m, round_f16=False).reshape(m, k)
    x = np.random.default_rng(T * 7 + B).standard_normal((B, T, k)).astype(np.float16)
    for act, fn in (("none", lambda z: z), ("squared_relu", lambda z: np.maximum(z, 0) ** 2), ("tanh", np.tanh),
                    ("sigmoid", lambda z: 1 / (1 + np.exp(-z)))):
        out = ctx.zeros([m, T, B])
        mat.matmul_op(ctx.tensor(x), out, act)
        got = out.back().reshape(B, T, m).astype(np.float32)
        want = fn(x.astype(np.float64) @ w.astype(np.float64).T)
        tol = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10 + 2e-5
        assert np.all(np.abs(got - want) <= tol), (act, np.abs(got - want).max())


def test_matvec_view_offsets(ctx):
    """Matrix::matmul_op on views: head rows of x into a slice of a wider output (v7.rs:1026-1031)."""
    k, m, T = 256, 24, 4
    raw = make("Q4_K", k, m, 11)
    mat = wrk.Matrix(ctx, "Q4_K", k, m, raw)
    w = dq.dequantize("Q4_K", raw, k * m, round_f16=False).reshape(m, k)
    x = np.random.default_rng(5).standard_normal((T, k)).astype(np.float16)
    out = ctx.zeros([m, T, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, T, 1]).view(None, (1, 3)), out.view(None, (2, 4)))
    got = out.back().reshape(T, m)
    want = x[1:3].astype(np.float64) @ w.astype(np.float64).T
    np.testing.assert_allclose(got[2:4], want, rtol=1e-4, atol=1e-5)
    assert not got[:2].any()


def test_matrix_create_rejects_bad_input(ctx):
    with pytest.raises(wrk.WrkError):
        wrk.Matrix(ctx, "Q4_K", 300, 4, np.zeros(144 * 4, np.uint8))          # K % 256 != 0 (loader.rs:824-827)
    with pytest.raises(wrk.WrkError):
        wrk.Matrix(ctx, "Q4_K", 256, 4, np.zeros(100, np.uint8))              # wrong byte count
    mat = wrk.Matrix(ctx, "Q8_0", 64, 4, make("Q8_0", 64, 4, 1))
    with pytest.raises(wrk.WrkError):
        mat.matmul_op(ctx.zeros([32, 1, 1]), ctx.zeros([4, 1, 1]))            # K mismatch


# ------------------------------------------------------------------ MFMA dequant-GEMM (turbo path; tiles padded to 16 tokens)
GEMM_CASES = [("Q4_K", 2048, 100), ("Q4_K", 512, 64), ("Q5_K", 1024, 48), ("Q6_K", 512, 130), ("Q6_K", 2048, 33),
              ("Q8_0", 96 * 2, 40), ("Q8_0", 1024, 64), ("F16", 96, 256), ("F16", 2048, 70), ("F32", 256, 64)]


@pytest.mark.parametrize("kind,k,m", GEMM_CASES)
@pytest.mark.parametrize("T,B", [(2, 1), (3, 1), (9, 1), (16, 1), (32, 1), (37, 1), (128, 1), (8, 3)])
def test_gemm_matches_oracle(ctx, kind, k, m, T, B):
    """Matrix::matmul_op(turbo=true) (matrix.rs:185-196) on the matrix cores: exact-weight f32 contraction."""
    raw = make(kind, k, m, k * 3 + m)
    mat = wrk.Matrix(ctx, kind, k, m, raw)
    w = dq.dequantize(kind, raw, k * m, round_f16=(kind == "F32")).reshape(m, k)
    x = np.random.default_rng(T + B).standard_normal((B, T, k)).astype(np.float16)
    out = ctx.zeros([m, T, B], np.float32)
    mat.matmul_op(ctx.tensor(x), out, turbo=True)
    got = out.back().reshape(B, T, m)
    want = x.astype(np.float64) @ w.astype(np.float64).T
    bound = 4e-6 * (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    assert np.all(np.abs(got - want) <= bound), np.abs(got - want).max()


@pytest.mark.parametrize("kind,k,m", [("Q4_K", 512, 6416), ("Q5_K", 256, 6500), ("F16", 256, 6416), ("Q4_K", 2048, 8192)])
@pytest.mark.parametrize("T", [5, 16])
def test_gemm_two_row_tiles_per_workgroup(ctx, kind, k, m, T, monkeypatch):
    """Round 3: launches of >= 400 row tiles at <= 16 tokens multiply two 16-row tiles per workgroup with the same B fragments
    (gemm_pair_kernel): ragged last pair (6416 = 200 pairs + one tile, 6500 = rows in fours), against the oracle and bit-identical to the
    one-tile kernel (same per-wave block order, same combine order)."""
    raw = make(kind, k, m, k + m)
    mat = wrk.Matrix(ctx, kind, k, m, raw)
    w = dq.dequantize(kind, raw, k * m, round_f16=False).reshape(m, k)
    x = np.random.default_rng(T).standard_normal((1, T, k)).astype(np.float16)
    res = np.random.default_rng(T + 1).standard_normal((1, T, m)).astype(np.float16)
    got = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("WRK_GEMM_PAIR", pair)
        out = ctx.zeros([m, T, 1], np.float32)
        mat.matmul_op(ctx.tensor(x), out, turbo=True)
        got[pair] = out.back().reshape(T, m)
    want = x[0].astype(np.float64) @ w.astype(np.float64).T
    bound = 4e-6 * (np.abs(x[0]).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    assert np.all(np.abs(got["1"] - want) <= bound), np.abs(got["1"] - want).max()
    assert np.array_equal(got["1"], got["0"])
    del res


def test_gemm_activation_residual_f16_out_and_vec_agreement(ctx):
    k, m, T = 512, 72, 48
    raw = make("Q4_K", k, m, 5)
    mat = wrk.Matrix(ctx, "Q4_K", k, m, raw)
    x = np.random.default_rng(2).standard_normal((T, k)).astype(np.float16)
    a, b = ctx.zeros([m, T, 1]), ctx.zeros([m, T, 1])
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), a, "squared_relu", turbo=True)
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), b, "squared_relu", turbo=False)
    ga, gb = a.back().astype(np.float32), b.back().astype(np.float32)
    assert np.all(np.abs(ga - gb) <= np.maximum(np.abs(gb), 2.0 ** -14) * 2.0 ** -10 + 1e-6)     # <= 1 f16 ulp apart


# every case satisfies the dispatcher's tile rule (>= 96 tiles of 64 rows x 64 tokens, or >= 64 when K <= 2560), so these run on
# the LDS-tiled kernel; the first four (12..16 tiles) stay on the K-split kernel and pin the two against the same bound
@pytest.mark.parametrize("kind,k,m,T", [
    ("Q4_K", 2048, 130, 200), ("Q5_K", 1024, 200, 100), ("Q4_K", 256, 64, 48), ("Q6_K", 2560, 72, 64),
    ("Q4_K", 2048, 520, 500), ("Q5_K", 1024, 1030, 250), ("Q4_K", 256, 2050, 130), ("Q5_K", 2560, 520, 470),
    ("Q6_K", 2048, 520, 500), ("Q6_K", 768, 1030, 250), ("Q6_K", 4096, 1100, 400), ("Q6_K", 256, 2050, 130),
    ("F16", 2048, 520, 500), ("F16", 96, 2050, 130), ("Q4_K", 4096, 1100, 400),
    ("F16", 2048, 96, 128), ("F16", 2560, 320, 70), ("F16", 2048, 64, 48),       # LoRA down-projections: tiled whatever their tile count
    # >= 512 tokens, Q4_K / Q5_K, rows in fours: the second-generation tile kernel (half-block stages), ragged row and token tails
    ("Q4_K", 2048, 520, 600), ("Q5_K", 1024, 1028, 530), ("Q4_K", 256, 2052, 640), ("Q4_K", 4096, 1100, 520), ("Q5_K", 2560, 260, 1030),
    # Q4_K with >= 512 tokens: the third-generation tile (wrk_gemm3.hip: shared LDS A-tile, min term over sub-block input sums); 2560 = ten
    # blocks: a last group of two; 8192: the ffn value shape
    ("Q4_K", 2560, 260, 530), ("Q4_K", 8192, 132, 515), ("Q4_K", 1280, 128, 512),
    # the same tile from 128 tokens on, K split over workgroups (slices of 4 / 2 / 1 blocks, partial tiles added in slice order): slices that start
    # inside a group of four blocks, a last slice shorter than the others, one token tile and two, ragged rows
    ("Q5_K", 2048, 260, 200), ("Q5_K", 4096, 132, 128), ("Q5_K", 3584, 130, 520),        # Q5_K on the same tile (round 3): the fifth bit from the qh plane
    ("Q4_K", 8192, 132, 140), ("Q4_K", 2048, 260, 128), ("Q4_K", 2560, 384, 256), ("Q4_K", 1024, 2052, 130), ("Q4_K", 1536, 128, 250), ("Q4_K", 2048, 6144, 128),
    ("Q8_0", 2048, 520, 600), ("Q8_0", 1024, 1028, 530), ("Q8_0", 4096, 1100, 520), ("Q8_0", 128, 2052, 640),         # Q8_0: that kernel only
    ("Q8_0", 2048, 520, 200), ("Q8_0", 1024, 1028, 130), ("Q8_0", 128, 2052, 100), ("Q8_0", 2560, 260, 70),           # round 3: Q8_0 chunks of 48 .. 511 tokens on the same kernel
    ("F16", 2048, 96, 600), ("F16", 2560, 320, 530), ("F16", 256, 2052, 640), ("F16", 1024, 520, 1030)])      # F16 with K % 128 == 0
def test_gemm_prefill_tile_kernel(ctx, kind, k, m, T):
    """The LDS-tiled prefill kernel (>= 48 stacked tokens of one dense [K, T, 1] stack, >= 64 rows, Q4_K / Q5_K / Q6_K / F16):
    ragged row and token tails, fused activation with f16 output."""
    raw = make(kind, k, m, k + m + T)
    mat = wrk.Matrix(ctx, kind, k, m, raw)
    w = dq.dequantize(kind, raw, k * m, round_f16=False).reshape(m, k)
    x = np.random.default_rng(T).standard_normal((T, k)).astype(np.float16)
    out = ctx.zeros([m, T, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), out, turbo=True)
    got = out.back().reshape(T, m)
    want = x.astype(np.float64) @ w.astype(np.float64).T
    bound = 4e-6 * (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    assert np.all(np.abs(got - want) <= bound), np.abs(got - want).max()
    # x += W . relu(x)^2-style epilogue through the op-level residual of the model path is covered by the model tests;
    # here: activation + f16 store agree with the K-split kernel to one f16 ulp
    a, b = ctx.zeros([m, T, 1]), ctx.zeros([m, T, 1])
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), a, "tanh", turbo=True)
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), b, "tanh", turbo=False)
    ga, gb = a.back().astype(np.float32), b.back().astype(np.float32)
    # both f32 pre-activations are within `bound` of the exact value (tanh' <= 1), then each is rounded to f16
    d = np.abs(ga.reshape(T, m) - gb.reshape(T, m))
    ok = d <= np.maximum(np.abs(gb.reshape(T, m)), 2.0 ** -14) * 2.0 ** -10 + 2.0 * bound + 1e-6
    assert np.all(ok), (int((~ok).sum()), float(d.max()))


def test_gemm_prefill_tile_kernel_model_shape(ctx):
    """A shape the dispatcher really sends to the tile kernel at K = 2048 (>= 128 tiles): 2048 x 2048 x 256 tokens."""
    kind, k, m, T = "Q4_K", 2048, 2048, 256
    raw = make(kind, k, m, 77)
    mat = wrk.Matrix(ctx, kind, k, m, raw)
    w = dq.dequantize(kind, raw, k * m, round_f16=False).reshape(m, k)
    x = np.random.default_rng(9).standard_normal((T, k)).astype(np.float16)
    out = ctx.zeros([m, T, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), out, turbo=True)
    got = out.back().reshape(T, m)
    want = x.astype(np.float64) @ w.astype(np.float64).T
    bound = 4e-6 * (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    assert np.all(np.abs(got - want) <= bound), np.abs(got - want).max()


def _fuzz_cases(n=48, seed=2024):
    r = np.random.default_rng(seed)
    kinds = ["Q4_K", "Q5_K", "Q6_K", "Q8_0", "F16", "INT8", "NF4"]
    step = {"Q4_K": 256, "Q5_K": 256, "Q6_K": 256, "Q8_0": 32, "F16": 8, "INT8": 128, "NF4": 64}
    out = []
    for i in range(n):
        kind = kinds[i % len(kinds)]
        k = int(step[kind] * r.integers(1, 1 + (12 if step[kind] == 256 else 40)))
        if kind == "F16":
            k = max(k, 32) // 32 * 32 if r.random() < 0.5 else k
        m = int(r.integers(1, 300))
        if kind == "INT8":
            m = max(1, m // 8 * 8)                      # K*M % 128 == 0
        T = int(r.choice([1, 1, 2, 3, 5, 7, 12, 17, 31, 50, 70]))
        out.append((kind, k, m, T, bool(r.integers(0, 2))))
    return out


@pytest.mark.parametrize("kind,k,m,T,turbo", _fuzz_cases())
def test_fuzz_shapes_all_kinds(ctx, kind, k, m, T, turbo):
    """Seeded random shapes over every matrix kind and both dispatch flags: ragged row counts, odd token counts, K from
    one block to a dozen.  Same bound as the fixed cases (f32 accumulation of exact products)."""
    from oracle import wrkquant as wq
    rng = np.random.default_rng(k * 131 + m * 7 + T)
    w = (rng.standard_normal((m, k)) / np.sqrt(k)).astype(np.float32)
    if kind == "INT8":
        codes, mm = wq.quantize_int8(w.astype(np.float16))
        raw = np.concatenate([codes, mm.reshape(-1).view(np.uint8)])
        wd = wq.dequantize_int8(codes, mm).reshape(m, k)
    elif kind == "NF4":
        packed, am = wq.quantize_nf4(w.astype(np.float16))
        raw = np.concatenate([packed, am.view(np.uint8)])
        wd = wq.dequantize_nf4(packed, am).reshape(m, k)
    else:
        raw = qz.QUANTIZE[kind](w)
        wd = dq.dequantize(kind, raw, k * m, round_f16=False).reshape(m, k)
    mat = wrk.Matrix(ctx, kind, k, m, raw)
    x = rng.standard_normal((T, k)).astype(np.float16)
    out = ctx.zeros([m, T, 1], np.float32)
    mat.matmul_op(ctx.tensor(x, [k, T, 1]), out, turbo=turbo)
    got = out.back().reshape(T, m)
    want = x.astype(np.float64) @ wd.astype(np.float64).T
    bound = 4e-6 * (np.abs(x).astype(np.float64) @ np.abs(wd).astype(np.float64).T) + 1e-6
    assert np.all(np.abs(got - want) <= bound), (kind, k, m, T, turbo, float(np.abs(got - want).max()))
