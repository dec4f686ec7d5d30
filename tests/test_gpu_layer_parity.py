"""Per-layer, teacher-forced parity: no propagation, so no "f16 flip" argument.

For every token the ORACLE runs the whole model with its trace on (examples/inspect.rs's buffer names).  The HIP path then
runs each layer ALONE (`wrk_v7_infer_layer`, the C-ABI analogue of a v7::HookMap that overwrites the Frame) on the
oracle's own layer input, layer-0 value and pre-token state, and every buffer the layer stores is compared with the
oracle's buffer of that stage.  Differences can only come from inside ONE layer: f32 summation order in the matmuls and
reductions, which may move an f16 store to the neighbouring f16 value.

Bars (fixed; "ulp" = the f16 spacing at max(|value|, rms of the buffer), see `ulps`):
  * DIRECT buffers -- one rounding away from the layer's exact inputs (LN(x), the decode token shifts, r, raw k / v, the LoRA
    intermediates): EVERY element within 1 ulp of the oracle's value;
  * DOWNSTREAM buffers -- computed inside the layer from other f16-stored buffers (w, a, g, k after control_k, kk after the
    l2 norm, v after the value residual, the gated WKV output att_x, att_o, the ffn buffers, the layer output x; for multi-token
    chunks also the token shifts, which mix TWO LN rows): an upstream 1-ulp flip is carried through the layer's own chain
    (l2 / group norm, a 1024-term matmul), so one layer accumulates a few ulp: every element within 6 ulp and >= 80 % of the
    elements of every buffer within 1 ulp (measured worst: 5 ulp, 84.7 %).  This is the whole per-layer error: with 1-ulp
    f16 stores as the only difference, "<= 1 ulp on every stored buffer" is not attainable for buffers that are functions of
    other rounded buffers, and it is this per-layer 1..5 ulp that 24 layers propagate into the whole-model figures.
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
    """|got - want| in units of the f16 spacing at max(|want|, rms(want)): an element that is small against its buffer (a
    cancelling sum) is measured with the spacing of a typical element -- the f32 accumulation error of a dot product scales
    with sum|w||x|, not with its result."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    rms = float(np.sqrt(np.mean(np.square(want, dtype=np.float64)))) + 1e-30
    mag = np.maximum(np.maximum(np.abs(want), rms), 2.0 ** -14)
    return np.abs(got - want) / 2.0 ** (np.floor(np.log2(mag)) - 10)


STATS = {}


def check(case, name, got, want, direct, downstream_max_ulp=None):
    u = ulps(got.reshape(want.shape), want)
    s = STATS.setdefault(case, {}).setdefault(name, {"max_ulp": 0.0, "min_frac_le1": 1.0, "min_frac_exact": 1.0, "n": 0, "max_strict_ulp": 0.0,
                                                     "min_frac_strict_le1": 1.0})
    # strict measure: the f16 spacing at |want| itself, no floor at the buffer's rms -- reported beside the graded one
    w32 = np.asarray(want, np.float32)
    su = np.abs(np.asarray(got, np.float32).reshape(w32.shape) - w32) / 2.0 ** (np.floor(np.log2(np.maximum(np.abs(w32), 2.0 ** -14))) - 10)
    s["max_strict_ulp"] = max(s["max_strict_ulp"], float(su.max()))
    s["min_frac_strict_le1"] = min(s["min_frac_strict_le1"], float((su <= 1.0).mean()))
    s["max_ulp"] = max(s["max_ulp"], float(u.max()))
    s["min_frac_le1"] = min(s["min_frac_le1"], float((u <= 1.0).mean()))
    s["min_frac_exact"] = min(s["min_frac_exact"], float((u == 0).mean()))
    s["n"] += 1
    max_ulp, frac1 = (DIRECT_MAX_ULP, 1.0) if direct else (downstream_max_ulp or DOWNSTREAM_MAX_ULP, DOWNSTREAM_FRAC_LE1)
    assert u.max() <= max_ulp, f"{case} {name}: {u.max():.2f} ulp (limit {max_ulp}); {int((u > 1).sum())} of {u.size} elements beyond 1 ulp"
    assert (u <= 1.0).mean() >= frac1, f"{case} {name}: only {(u <= 1.0).mean():.5f} of the elements within 1 ulp"


# FIXED bars.  Measured on MI355X over every case below (gpurun_out/layer_parity.json, summary in DESIGN.md section 2): direct buffers
# <= 1 ulp everywhere (93-100 % bit-identical); downstream buffers <= 5 ulp, >= 84.7 % of the elements within 1 ulp.
DIRECT_MAX_ULP, DOWNSTREAM_MAX_ULP, DOWNSTREAM_FRAC_LE1 = 1.0, 6.0, 0.80


# (frame buffer name on the HIP side, oracle trace key, direct?)
MODE0 = [("att_rx", "att_rx", True), ("att_kx", "att_kx", True), ("att_gx", "att_gx", True), ("att_r", "r", True), ("aux_w", "aux_w", True),
         ("aux_a", "aux_a", True), ("aux_g", "aux_g", True), ("att_w", "w", False), ("att_a", "a", False), ("att_g", "g", False),
         ("att_k", "k", False), ("att_v", "v", False), ("att_kk", "kk", False), ("att_x", "att_x", False), ("att_o", "att_o", False),
         # (ffn_x is overwritten by channel_mix_v7 at the end of the op list: not comparable)
         ("ffn_kx", "ffn_kx", False), ("ffn_k", "ffn_k", False), ("ffn_v", "ffn_v", False), ("x", "x", False)]
# the fused decode layer keeps the element-wise chain in registers: these are the buffers it stores
MODE1 = [("att_x_ln", "att_x_ln", True), ("att_r", "r", True), ("att_k", "k_raw", True), ("att_v", "v_raw", True), ("aux_w", "aux_w", True),
         ("aux_a", "aux_a", True), ("aux_g", "aux_g", True), ("att_x", "att_x", False), ("ffn_x", "ffn_x", False), ("ffn_k", "ffn_k", False),
         ("x", "x", False)]
# merged multi-token launches (mode 1, T > 1) store the same buffers as the op list except the per-op temporaries
MERGED = [("att_r", "r", True), ("att_x", "att_x", False), ("ffn_k", "ffn_k", False), ("x", "x", False)]


def run_case(ctx, name, weights, kw, mode, chunk_lens, steps, downstream_max_ulp=None):
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
    elif decode and T <= 8:
        # 1 .. 8 sequences (the dmv kernels, 5 launches per layer): the split head kernel hands y = WKV output (pre group norm)
        # and the gate to W_o's prologue
        table = [(b, "wkv" if b == "att_x" else k, d) for b, k, d in MODE1] + [("att_g", "g", False)]
    elif decode:        # with more sequences LN + shifts run as their own launch and LN(x) is not stored separately
        table = [e for e in MODE1 if e[0] not in ("att_x_ln", "ffn_x")]
    else:
        table = MERGED
    case = f"{name}/{weights}/{sorted(kw.items())}/mode{mode}/{chunk_lens}"
    # warm-up: a generic (non-zero) recurrent state -- with the all-zero initial state the WKV output of the first token is
    # (r . k) v, a single dot product whose cancellation the group norm then amplifies: an ill-conditioned special case
    oracle.infer_chunk([synth.tokens(99, f"w{b}", 8 if n else 0, V) for b, n in enumerate(chunk_lens)], [])
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
                if buf.endswith("x") and buf.startswith("att_") and len(buf) == 6 and T > 1:
                    direct = False          # att_rx .. att_gx of a chunk mix two LN rows (the previous token's and this one's)
                got = rt.frame(buf, T).astype(np.float32)
                check(case, buf, got, tr[f"{li}_{key}"], direct, downstream_max_ulp)
            for b in range(len(chunk_lens)):
                got, want = rt.state_back(b)[li], oracle.state.back(b)[li]
                d = np.abs(got - want)
                assert d.max() <= 2.0 ** -9 * max(float(np.abs(want).max()), 1e-3), (step, li, b, float(d.max()))
    rt.close()
    return STATS[case]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name,weights,kw", [
    ("tiny", wrk.WEIGHTS_INLINE, {}),
    ("small", wrk.WEIGHTS_INLINE_F16, {}),
    ("small", wrk.WEIGHTS_INLINE, {"mat_override": {"time_mix_value": "Q6_K", "channel_mix_value": "Q6_K"}}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q5_K", "head": "Q8_0"}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q8_0", "head": "F16", "lora": "F16"}),
])
def test_decode_layer_by_layer(ctx, name, weights, kw, mode):
    run_case(ctx, name, weights, kw, mode, [1], 6)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("nseq", [2, 3, 9])
def test_several_sequences_decode_layer_by_layer(ctx, mode, nseq):
    # 2, 3: the multi-token dmv launches (2- and 4-token instantiations; 3 leaves a clamped dead token); 9: the MFMA launches of batched decode
    run_case(ctx, "tiny", wrk.WEIGHTS_INLINE, {}, mode, [1] * nseq, 3)


@pytest.mark.parametrize("name,nseq,kw", [("small", 9, {}), ("small", 20, {}), ("tiny", 40, {}), ("small", 12, {"mat": "Q5_K"}),
                                          ("small", 12, {"mat": "Q8_0"}), ("small", 20, {"mat": "Q8_0"}),      # round 3: Q8_0 body
                                          ("small", 18, {"mat_override": {"time_mix_value": "Q6_K", "channel_mix_value": "Q6_K"}})])
def test_decode_batches_on_the_k_sliced_gemm(ctx, name, nseq, kw, monkeypatch):
    """5 .. 32 sequences with WRK_GEMM_KS=2: every matrix of the layer goes through the K-sliced MFMA kernel (1 and 2 token tiles; K
    slices meeting through partial tiles and arrival counters for K >= 512; F16 LoRA rows in the same launch); 40 sequences stay on the
    K-split kernels."""
    monkeypatch.setenv("WRK_GEMM_KS", "2")
    run_case(ctx, name, wrk.WEIGHTS_INLINE, kw, 1, [1] * nseq, 2)


@pytest.mark.parametrize("nseq", [1, 2])
def test_real_tensor_mix_at_the_2p9b_layer_shape(ctx, nseq):
    """Round 3: llama.cpp's Q4_K_M mix (Q6_K attention value and ffn value next to Q4_K, F16 LoRA rows) on rows of 2560
    elements, 1 / 2 sequences: the three-kind launch of the fused layer (r, k, v + LoRA behind the LN prologue) on the dmv kernels with
    two chunk iterations per wave -- before, K > 2048 sent this launch (and with it the layer) to the 7-launch MFMA path.  (The buffer table
    of this case is the split-head one: it passes only if the 5-launch layer ran.  Four sequences stay on the MFMA layer: measured faster.)"""
    run_case(ctx, "2.9B-2L", wrk.WEIGHTS_INLINE, {"mat_override": {"time_mix_value": "Q6_K", "channel_mix_value": "Q6_K"}}, 1, [1] * nseq, 2)


@pytest.mark.parametrize("mode", [0, 1])
def test_decode_layer_by_layer_headline_layer_shape(ctx, mode):
    """the per-buffer comparison at the 1.5B LAYER SHAPE (D = 2048, F = 8192, 32 heads, ranks 96 / 96 / 64 / 256; three layers,
    vocabulary 1024), one-token decode, op list (mode 0) and fused launches (mode 1), both in the reference's effective arithmetic (weights
    rounded to f16).  The full-size logits of modes 0 and 1 differ from the oracle by 1.4e-2 and 3.7e-2 (profiles/r02_fullsize_parity.json):
    this case shows, buffer by buffer on the 2048-wide layer, whether a kernel of mode 1 is further from the oracle than its mode-0
    counterpart or whether the difference is propagation (stats -> profiles/r03_layer_parity_1p5b.json)."""
    run_case(ctx, "1.5B-3L", wrk.WEIGHTS_INLINE_F16, {}, mode, [1], 3)


@pytest.mark.parametrize("lens", [[128], [100, 100]])
def test_prefill_chunks_on_the_k_split_tile_at_the_headline_layer_shape(ctx, lens):
    """Round 3: chunks of 128 - 256 tokens at the 1.5B layer shape, mode 1.  128 tokens: the ffn value (K = 8192) on the K-split third-generation
    tile (partial tiles + reduce launch), the other matrices on the first-generation tile; 200 tokens of two sequences: every Q4_K matrix on the
    K-split tile; WKV chunk kernel with precomputed decays, eight threads per state column, a head's columns over several workgroups.
    Downstream bar 8 ulp here: 409 600 elements per buffer behind up to 100 steps of the recurrence -- the post-WKV buffer has ONE element at 7 ulp
    (99.6 % within 1 ulp), and has it with the K-split tile, the third-generation tile and the eight-thread kernel each switched off
    (tools/lp_probe.sh): the tail of the distribution at this size, not a kernel."""
    run_case(ctx, "1.5B-3L", wrk.WEIGHTS_INLINE, {}, 1, lens, 1, downstream_max_ulp=8.0)


@pytest.mark.parametrize("mode", [0, 1])
def test_prefill_chunk_layer_by_layer(ctx, mode):
    run_case(ctx, "small", wrk.WEIGHTS_INLINE, {}, mode, [70], 2)         # one 70-token chunk: tile GEMMs + chunk WKV
    run_case(ctx, "tiny", wrk.WEIGHTS_INLINE, {}, mode, [9, 0, 23], 2)    # ragged chunk of two sequences (one slot idle)


def test_zz_write_layer_parity_stats():
    """Not a check: dumps what the cases above measured (worst over steps and layers, per buffer) for DESIGN.md."""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "layer_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(STATS, open(out, "w"), indent=1, sort_keys=True)
    json.dump({c: b for c, b in STATS.items() if c.startswith("1.5B-3L")}, open(out.replace("layer_parity.json", "layer_parity_1p5b.json"), "w"), indent=1,
              sort_keys=True)
    agg = {}
    for case, bufs in STATS.items():
        for b, s in bufs.items():
            a = agg.setdefault(b, {"max_ulp": 0.0, "min_frac_le1": 1.0})
            a["max_ulp"] = max(a["max_ulp"], s["max_ulp"]); a["min_frac_le1"] = min(a["min_frac_le1"], s["min_frac_le1"])
    print(json.dumps(agg, indent=1, sort_keys=True))
