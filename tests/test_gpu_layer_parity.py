"""Per-layer, teacher-forced parity: no propagation, so no "f16 flip" argument.

For every token the ORACLE runs the whole model with its trace on (examples/inspect.rs's buffer names).  The HIP path then
runs each layer ALONE (`wrk_v7_infer_layer`, the C-ABI analogue of a v7::HookMap that overwrites the Frame) on the
oracle's own layer input, layer-0 value and pre-token state, and every buffer the layer stores is compared with the
oracle's buffer of that stage.  Differences can only come from inside ONE layer: f32 summation order in the matmuls and
reductions, which may move an f16 store to the neighbouring f16 value.

Bars (fixed):
  * buffers produced directly from exact inputs (r, raw k / v, LoRA intermediates, LN outputs, token shifts):
    every element within 1 f16 ulp of the oracle's value (+ a 2^-13 * max|buffer| floor for elements that are a
    cancelling sum: the f32 accumulation error of a matmul scales with sum|w||x|, not with the result);
  * buffers downstream of those inside the layer (w, a, k, kk, v after the value residual, WKV output, gated att_x, att_o, x
    after attention, ffn buffers, layer output x): a 1-ulp flip of an upstream f16 value moves them by far less than 1 ulp,
    except next to a rounding boundary: 2 ulp, and >= 99.9 % of the elements within 1 ulp;
  * state (f32): max |delta| <= 2^-9 * max|state row|.
"""
import numpy as np
import pytest

import wrk
from oracle import gguf as ogguf
from oracle import rwkv7 as O
from oracle import synth
from oracle.rnn import stack_cursors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def ulps(got, want):
    """|got - want| in units of the f16 spacing at |want| (subnormal spacing below 2^-14), with the cancellation floor."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    mag = np.maximum(np.abs(want), 2.0 ** -14)
    ulp = 2.0 ** (np.floor(np.log2(mag)) - 10)
    floor = 2.0 ** -13 * float(np.abs(want).max() + 1e-30)
    return np.maximum(np.abs(got - want) - floor, 0.0) / ulp


def check(name, got, want, max_ulp, frac1=None):
    u = ulps(got.reshape(want.shape), want)
    assert u.max() <= max_ulp, f"{name}: {u.max():.2f} ulp (limit {max_ulp}); {int((u > 1).sum())} of {u.size} elements beyond 1 ulp"
    if frac1 is not None:
        assert (u <= 1.0).mean() >= frac1, f"{name}: only {(u <= 1.0).mean():.5f} of the elements within 1 ulp"
    return float(u.max()), float((u > 0).mean())


# (frame buffer name on the HIP side, oracle trace key, direct?)
MODE0 = [("att_rx", "att_rx", True), ("att_kx", "att_kx", True), ("att_gx", "att_gx", True), ("att_r", "r", True), ("aux_w", "aux_w", True),
         ("aux_a", "aux_a", True), ("aux_g", "aux_g", True), ("att_w", "w", False), ("att_a", "a", False), ("att_g", "g", False),
         ("att_k", "k", False), ("att_v", "v", False), ("att_kk", "kk", False), ("att_x", "att_x", False), ("att_o", "att_o", False),
         ("ffn_x", "ffn_x", False), ("ffn_kx", "ffn_kx", False), ("ffn_k", "ffn_k", False), ("ffn_v", "ffn_v", False), ("x", "x", False)]
# the fused decode layer keeps the element-wise chain in registers: these are the buffers it stores
MODE1 = [("att_x_ln", "att_x_ln", True), ("att_r", "r", True), ("att_k", "k_raw", True), ("att_v", "v_raw", True), ("aux_w", "aux_w", True),
         ("aux_a", "aux_a", True), ("aux_g", "aux_g", True), ("att_x", "att_x", False), ("ffn_x", "ffn_x", False), ("ffn_k", "ffn_k", False),
         ("x", "x", False)]
# merged multi-token launches (mode 1, T > 1) store the same buffers as the op list except the per-op temporaries
MERGED = [("att_r", "r", True), ("att_x", "att_x", False), ("ffn_k", "ffn_k", False), ("x", "x", False)]


def run_case(ctx, name, weights, kw, mode, chunk_lens, steps):
    data = synth.make_v7_gguf(synth.CONFIGS[name], 42, **kw)
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=len(chunk_lens), weights=weights)
    model = O.build_v7(ogguf.GgufReader(data), weights_f16=(weights != wrk.WEIGHTS_INLINE))
    oracle = O.V7Runtime(model, len(chunk_lens), act_f16=True)
    V, L = rt.info.num_vocab, rt.info.num_layer
    T = sum(chunk_lens)
    cursors = stack_cursors(chunk_lens)
    decode = all(n == 1 for n in chunk_lens)
    if mode == 0:
        table = MODE0
    elif decode:        # with several sequences LN + shifts run as their own launch and LN(x) is not stored separately
        table = MODE1 if T == 1 else [e for e in MODE1 if e[0] not in ("att_x_ln", "ffn_x")]
    else:
        table = MERGED
    worst = {}
    for step in range(steps):
        chunk = [synth.tokens(100 + step, f"b{b}", n, V) for b, n in enumerate(chunk_lens)]
        before = [oracle.state.back(b) for b in range(len(chunk_lens))]
        oracle.trace = {}
        oracle.infer_chunk(chunk, [T - 1])
        tr = oracle.trace
        for li in range(L):
            for b in range(len(chunk_lens)):
                rt.state_load(before[b], b)                       # teacher-forced state too
            x_in = tr["emb_x"] if li == 0 else tr[f"{li - 1}_x"]
            rt.infer_layer(li, x_in, tr["0_v"] if li else None, cursors, mode=mode)
            for buf, key, direct in table:
                if key == "aux_v" and li == 0:
                    continue
                got = rt.frame(buf, T).astype(np.float32)
                w = check(f"step {step} layer {li} {buf}", got, tr[f"{li}_{key}"], 1.0 if direct else 2.0, None if direct else 0.999)
                worst[buf] = max(worst.get(buf, (0, 0)), w)
            for b in range(len(chunk_lens)):
                got, want = rt.state_back(b)[li], oracle.state.back(b)[li]
                d = np.abs(got - want)
                assert d.max() <= 2.0 ** -9 * max(float(np.abs(want).max()), 1e-3), (step, li, b, float(d.max()))
    rt.close()
    return worst


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name,weights,kw", [
    ("tiny", wrk.WEIGHTS_INLINE, {}),
    ("small", wrk.WEIGHTS_INLINE_F16, {}),
    ("small", wrk.WEIGHTS_INLINE, {"mat_override": {"time_mix_value": "Q6_K", "channel_mix_value": "Q6_K"}}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q5_K", "head": "Q8_0"}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q8_0", "head": "F16", "lora": "F16"}),
])
def test_decode_layer_by_layer(ctx, name, weights, kw, mode):
    worst = run_case(ctx, name, weights, kw, mode, [1], 6)
    print(name, mode, {k: f"{v[0]:.2f} ulp" for k, v in worst.items()})


@pytest.mark.parametrize("mode", [0, 1])
def test_two_sequences_decode_layer_by_layer(ctx, mode):
    run_case(ctx, "tiny", wrk.WEIGHTS_INLINE, {}, mode, [1, 1], 4)        # multi-token matvec / MFMA launches of batched decode


@pytest.mark.parametrize("mode", [0, 1])
def test_prefill_chunk_layer_by_layer(ctx, mode):
    run_case(ctx, "small", wrk.WEIGHTS_INLINE, {}, mode, [70], 2)         # one 70-token chunk: tile GEMMs + chunk WKV
    run_case(ctx, "tiny", wrk.WEIGHTS_INLINE, {}, mode, [9, 0, 23], 2)    # ragged chunk of two sequences (one slot idle)
