"""Whole-model parity of the HIP path (through libwrk_runtime + libwrk_hip) against the oracle.

Bars (north_star: greedy-token-identical, logits within 1e-3 of the reference path):
  * logits vs the oracle run with the SAME arithmetic variant (f16 activation rounding points,
    exact or f16-rounded weights): max |delta| <= LOGIT_TOL.  The two sides accumulate in f32 in a
    different order, which can move an f16 store by one ulp, and such flips propagate through the
    recurrent state.  The oracle itself moves by max 2.6e-3 / mean 6e-4 on the 21-token prompt
    when only its matmul accumulation is switched from f32 to f64 (DESIGN.md "tolerances"), so
    1e-3 as a max-norm is below the arithmetic's own noise floor; the bars used are
    max |delta| <= 1e-2 (LOGIT_TOL) AND mean |delta| <= 1.5e-3 (LOGIT_MEAN_TOL).
  * greedy tokens identical; recurrent state within 1e-3 relative.
PARITY UNPINNED against the real reference (it cannot be run here or on the box); the oracle is the
line-by-line restatement described in oracle/rwkv7.py.
"""
import json
import os

import numpy as np
import pytest

import wrk
from oracle import gguf as ogguf
from oracle import rwkv7 as O
from oracle import synth
from oracle.rnn import FULL, LAST, RnnInput, RnnInputBatch, stack_cursors

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-2
LOGIT_MEAN_TOL = 1.5e-3
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def assert_state_close(got, want):
    """f32 recurrent state: same noise source as the logits (f16 flips upstream of the WKV update)."""
    d = np.abs(got - want)
    assert d.max() <= 2e-2 * max(1.0, float(np.abs(want).max())), d.max()
    assert d.mean() <= 1e-3, d.mean()


def build(ctx, name, weights, num_batch, **kw):
    data = synth.make_v7_gguf(synth.CONFIGS[name], 42, **kw)
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=num_batch, weights=weights)
    model = O.build_v7(ogguf.GgufReader(data), weights_f16=(weights != wrk.WEIGHTS_INLINE))
    return rt, O.V7Runtime(model, num_batch, act_f16=True)


VARIANTS = [
    ("tiny", wrk.WEIGHTS_INLINE, {}),
    ("tiny", wrk.WEIGHTS_INLINE_F16, {}),
    ("tiny", wrk.WEIGHTS_REFERENCE, {}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q5_K", "head": "Q8_0"}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q8_0", "head": "F16", "lora": "F16"}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "F16", "head": "F16"}),
    ("tiny", wrk.WEIGHTS_INLINE, {"mat": "Q3_K", "head": "Q2_K"}),          # load-time-only kinds (gguf.rs:280-423)
    ("tiny", wrk.WEIGHTS_REFERENCE, {"mat": "Q4_0", "head": "Q4_0"}),
    ("small", wrk.WEIGHTS_INLINE, {"mat_override": {"time_mix_value": "Q6_K", "channel_mix_value": "Q6_K"}}),
]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name,weights,kw", VARIANTS)
def test_prefill_then_greedy_decode(ctx, name, weights, kw, mode):
    rt, oracle = build(ctx, name, weights, 1, **kw)
    V = rt.info.num_vocab
    prompt = synth.tokens(3, "prompt", 21, V)
    inp = wrk.RnnInput([prompt], 32)
    got = rt.infer(inp, mode=mode)[0]
    want = oracle.infer_chunk([prompt], [len(prompt) - 1])
    assert got.shape == (1, V)
    assert np.abs(got - want).max() <= LOGIT_TOL, np.abs(got - want).max()
    assert np.abs(got - want).mean() <= LOGIT_MEAN_TOL, np.abs(got - want).mean()
    tok = int(want[0].argmax())
    assert int(got[0].argmax()) == tok
    # 12 greedy steps on the device-resident loop vs the oracle stepping one token at a time
    toks, ms, last = rt.generate_greedy([tok], 12, mode=mode, want_logits=True)
    otoks = []
    for _ in range(12):
        ol = oracle.infer_chunk([[tok]], [0])
        tok = int(ol[0].argmax())
        otoks.append(tok)
    assert toks[:, 0].tolist() == otoks
    assert np.abs(last - ol).max() <= LOGIT_TOL
    st = rt.state_back(0)
    assert_state_close(st, oracle.state.back(0))
    rt.close()


def test_chunked_prefill_scheduler_and_ragged_batches(ctx):
    """4 sequences of different length, Last and Full options, chunk size 32: the host scheduler
    (C++) must cut the same chunks as oracle/rnn.py and every chunk's logits must match."""
    rt, oracle = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 4)
    V = rt.info.num_vocab
    lens, opts = [45, 1, 0, 19], [LAST, LAST, FULL, FULL]
    toks = [synth.tokens(5, f"b{b}", n, V) for b, n in enumerate(lens)]
    inp = wrk.RnnInput(toks, 32, [wrk.RNN_LAST if o == LAST else wrk.RNN_FULL for o in opts])
    oin = RnnInput([RnnInputBatch(list(t), o) for t, o in zip(toks, opts)], 32)
    steps = 0
    while oin.num_token() > 0:
        info = next(oin.iter())
        red = info.redirect()
        chunk = oin.chunk()
        want = oracle.infer_chunk(chunk, red.headers)
        got = rt.infer(inp, mode=steps % 2)               # alternate op-by-op / fused
        for b, (s, e) in enumerate(red.outputs):
            assert got[b].shape == (e - s, V)
            if e > s:
                assert np.abs(got[b] - want[s:e]).max() <= LOGIT_TOL
        oin.step()
        steps += 1
        assert [inp.remaining(b) for b in range(4)] == [len(b.tokens) for b in oin.batches]
    assert steps >= 3
    with pytest.raises(wrk.WrkError):                      # RuntimeError::InputExhausted
        rt.infer(inp)
    for b in range(4):
        assert_state_close(rt.state_back(b), oracle.state.back(b))
    rt.close()


def test_chunk_split_invariance_and_state_carry(ctx):
    """Prefill in one 64-token chunk == 2 x 32 == state saved/restored in between (State::back/load)."""
    rt, _ = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 2)
    V = rt.info.num_vocab
    p = synth.tokens(9, "carry", 64, V)
    a = rt.infer(wrk.RnnInput([p, []], 64), mode=1)[0]
    inp = wrk.RnnInput([[], p], 32)
    rt.infer(inp, mode=1)
    saved = rt.state_back(1)
    rt.state_load(np.zeros_like(saved), 1)
    rt.state_load(saved, 1)
    b = rt.infer(inp, mode=1)[1]
    assert np.abs(a - b).max() <= 2e-3
    assert_state_close(rt.state_back(0), rt.state_back(1))
    rt.close()


def test_device_resident_state_snapshots(ctx):
    """State::read / State::write (v7.rs:229-262): snapshot a sequence's state in HBM, move it to another batch slot, and
    continue there -- the continuation equals the original sequence's, bit for bit (same kernels, same data)."""
    rt, _ = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 3)
    V = rt.info.num_vocab
    p = synth.tokens(21, "snap", 40, V)
    rt.infer(wrk.RnnInput([p, [], []], 64))
    snap = rt.state_read(0)
    assert np.array_equal(snap.read(np.float32, rt.state_back(0).size).reshape(rt.state_back(0).shape), rt.state_back(0))
    rt.state_write(snap, 2)
    assert np.array_equal(rt.state_back(2), rt.state_back(0))
    nxt = synth.tokens(22, "cont", 8, V)
    a = rt.infer(wrk.RnnInput([nxt, [], []], 64))[0]
    b = rt.infer(wrk.RnnInput([[], [], nxt], 64))[2]
    assert np.array_equal(a, b)
    with pytest.raises(wrk.WrkError):
        rt.state_write(wrk.Buffer(ctx, 64), 1)                # snapshot too small
    with pytest.raises(wrk.WrkError):
        rt.state_write(snap, 7)                               # batch out of range
    rt.close()


def test_batched_decode_matches_single_streams(ctx):
    """Independent sequences stacked in one dispatch give the same tokens as running them alone."""
    rt4, _ = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 4)
    rt1, _ = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 1)
    first = [3, 77, 200, 411]
    t4, _ = rt4.generate_greedy(first, 10, mode=1)
    for b, f in enumerate(first):
        t1, _ = rt1.generate_greedy([f], 10, mode=1)
        assert t1[:, 0].tolist() == t4[:, b].tolist()
        rt1.state_load(np.zeros_like(rt1.state_back(0)), 0)
    rt4.close(); rt1.close()


def test_invalid_inputs_are_rejected_before_launch(ctx):
    rt, _ = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 1)
    V = rt.info.num_vocab
    cur = stack_cursors([2])
    with pytest.raises(wrk.WrkError):
        rt.infer_raw([1, V], cur, [1])                     # token id out of vocab
    with pytest.raises(wrk.WrkError):
        rt.infer_raw([1, 2], [cur[0] | 5, cur[1]], [1])    # cursor names batch 5 of 1
    with pytest.raises(wrk.WrkError):
        rt.infer_raw([1, 2], cur, [2])                     # header row out of range
    with pytest.raises(wrk.WrkError):
        rt.generate_greedy([V + 3], 2)
    rt.close()


def test_golden_logits(ctx):
    """Committed fixture (tests/golden/tiny_q4k_golden.npz, made by tests/golden/make_golden.py from
    the oracle): guards both the oracle and the HIP path against silent drift."""
    g = np.load(os.path.join(GOLD, "tiny_q4k_golden.npz"))
    rt, _ = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 1)
    got = rt.infer(wrk.RnnInput([g["prompt"].tolist()], 32), mode=1)[0]
    assert np.abs(got - g["logits_inline"]).max() <= LOGIT_TOL
    toks, _ = rt.generate_greedy([int(g["logits_inline"][0].argmax())], len(g["greedy_inline"]), mode=1)
    assert toks[:, 0].tolist() == g["greedy_inline"].tolist()
    rt.close()


def test_empty_and_maximum_chunks(ctx):
    """Edge sizes of one job: no tokens at all (v7.rs:626-635 returns an empty output), the longest chunk a cursor
    can describe (255 tokens, `Cursor::pack` keeps `len` in 8 bits, tensor/mod.rs:53-60) with EVERY row's logits
    (RnnOption::Full -> header rows = all tokens; exercises the tiled prefill GEMM and the chunk WKV kernel inside a
    whole model), and one token more (rejected, not truncated)."""
    rt, oracle = build(ctx, "tiny", wrk.WEIGHTS_INLINE, 1)
    V = rt.info.num_vocab
    assert rt.infer_raw([], [], []).shape == (0, V)                     # T = 0: no-op
    toks = synth.tokens(13, "long", 255, V)
    got = rt.infer_raw(toks, stack_cursors([255]), list(range(255)))
    want = oracle.infer_chunk([toks], list(range(255)))
    assert got.shape == want.shape == (255, V)
    d = np.abs(got - want)
    # 255 tokens of recurrence: the f16-flip noise floor grows with depth (DESIGN.md "Tolerances"); rows near the start
    # are tight, the bound below is for the whole chunk
    assert d[:32].max() <= LOGIT_TOL and d.max() <= 4 * LOGIT_TOL and d.mean() <= 2 * LOGIT_MEAN_TOL, (d[:32].max(), d.max(), d.mean())
    assert (got.argmax(axis=1) == want.argmax(axis=1)).mean() >= 0.99
    assert_state_close(rt.state_back(0), oracle.state.back(0))
    with pytest.raises(wrk.WrkError):
        rt.infer(wrk.RnnInput([synth.tokens(1, "x", 300, V)], 512))     # 300 tokens of one sequence in one chunk
    rt.close()


@pytest.mark.parametrize("name,kw", [("tiny", {}), ("small", {"mat_override": {"time_mix_value": "Q6_K", "channel_mix_value": "Q6_K"}})])
def test_merged_prefill_launches_are_bit_identical_to_the_op_list(ctx, name, kw):
    """Mode 1 runs a multi-token chunk with fewer launches (six shifts in one pass, projections grouped per stage, the
    element-wise chains before and after the WKV kernel in one launch each, W_o's add in the GEMM epilogue).  Every
    intermediate is rounded where the separate ops store it and the reductions are the same, so above 64 stacked
    tokens (where the GEMM kernel choice cannot differ) logits and state must EQUAL mode 0's, not just be close."""
    data = synth.make_v7_gguf(synth.CONFIGS[name], 7, **kw)
    V = synth.CONFIGS[name].num_vocab
    p0, p1 = synth.tokens(5, "merged-a", 70, V), synth.tokens(5, "merged-b", 26, V)
    out = []
    for mode in (0, 1):
        rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=2, weights=wrk.WEIGHTS_INLINE)
        inp = wrk.RnnInput([p0, p1], 96, [wrk.RNN_FULL, wrk.RNN_LAST])
        logits = rt.infer(inp, mode=mode)
        nxt = rt.infer(wrk.RnnInput([[3] * 40, [5] * 33], 128), mode=mode)      # second chunk on the carried state
        out.append((logits, nxt, rt.state_back(0), rt.state_back(1)))
        rt.close()
    (l0, n0, a0, b0), (l1, n1, a1, b1) = out
    assert l0[0].shape == (70, V) and l0[1].shape == (1, V)
    for x, y in zip(l0 + n0, l1 + n1):
        assert np.array_equal(x, y), float(np.abs(x - y).max())
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)


@pytest.mark.parametrize("B,groups", [(4, 2), (6, 3), (5, 5), (8, 2)])
def test_concurrent_pipelines_equal_separate_runs(ctx, B, groups):
    """generate_greedy with G concurrent pipelines (mode bits 8-15): the sequences are independent (separate state slices,
    v7.rs:519-521), so every block must produce exactly what it produces when decoded alone through the ordinary call --
    same kernels, same shapes, bit-identical tokens, logits and state -- and the oracle's greedy tokens."""
    data = synth.make_v7_gguf(synth.CONFIGS["small"], 42)
    V = synth.CONFIGS["small"].num_vocab
    first = [(13 + 97 * b) % (V - 1) for b in range(B)]
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=B)
    toks, ms, logits = rt.generate_greedy(first, 9, mode=1, want_logits=True, groups=groups)
    states = [rt.state_back(b) for b in range(B)]
    rt.close()
    assert toks.shape == (9, B) and ms > 0
    for g in range(groups):
        b0, b1 = B * g // groups, B * (g + 1) // groups
        alone = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=b1 - b0)
        t, _, l = alone.generate_greedy(first[b0:b1], 9, mode=1, want_logits=True)
        assert np.array_equal(toks[:, b0:b1], t)
        assert np.array_equal(logits[b0:b1], l)
        for b in range(b0, b1):
            assert np.array_equal(states[b], alone.state_back(b - b0))
        alone.close()
    oracle = O.V7Runtime(O.build_v7(ogguf.GgufReader(data), weights_f16=False), B, act_f16=True)
    cur = list(first)
    for step in range(9):
        ol = oracle.infer_chunk([[t] for t in cur], list(range(B)))
        cur = [int(ol[b].argmax()) for b in range(B)]
        assert toks[step].tolist() == cur, step


def test_frame_dtype_switch_reaches_the_pipeline_lanes(ctx):
    """lanes of a grouped generate_greedy used to keep the frame dtype they were created with.  After a switch to f32 frames (and
    back) every block of a grouped call must still equal the same block decoded alone at the SAME frame dtype, bit for bit."""
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 11)
    V = synth.CONFIGS["tiny"].num_vocab
    B, groups = 4, 2
    first = [(3 + 41 * b) % (V - 1) for b in range(B)]
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=B)
    rt.generate_greedy(first, 3, mode=1, groups=groups)                 # lanes now exist, with f16 frames
    for dtype in (wrk.F32, wrk.F16):
        rt.set_frame_dtype(dtype)
        for b in range(B):
            rt.state_load(np.zeros_like(rt.state_back(b)), b)
        toks, _, logits = rt.generate_greedy(first, 6, mode=1, want_logits=True, groups=groups)
        for g in range(groups):
            b0, b1 = B * g // groups, B * (g + 1) // groups
            alone = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=b1 - b0)
            alone.set_frame_dtype(dtype)
            t, _, l = alone.generate_greedy(first[b0:b1], 6, mode=1, want_logits=True)
            assert np.array_equal(toks[:, b0:b1], t), (dtype, g)
            assert np.array_equal(logits[b0:b1], l), (dtype, g)
            alone.close()
    rt.close()


@pytest.mark.parametrize("B", [9, 20])
def test_k_sliced_gemm_runs_are_bit_identical(ctx, B, monkeypatch):
    """Every GEMM of a 9 / 20-sequence decode on the K-sliced kernel (WRK_GEMM_KS=2): its K slices meet in-kernel through write-through
    partial tiles and arrival counters, added in slice order -- so two runs from the same state must agree bit for bit (a stale tile
    would not), and the greedy tokens must be the oracle's."""
    monkeypatch.setenv("WRK_GEMM_KS", "2")
    data = synth.make_v7_gguf(synth.CONFIGS["small"], 42)
    V = synth.CONFIGS["small"].num_vocab
    first = [(5 + 31 * b) % (V - 1) for b in range(B)]
    runs = []
    for _ in range(3):
        rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=B)
        toks, _, logits = rt.generate_greedy(first, 24, mode=1, want_logits=True)
        runs.append((toks.copy(), logits.copy(), [rt.state_back(b) for b in range(B)]))
        rt.close()
    for toks, logits, states in runs[1:]:
        assert np.array_equal(toks, runs[0][0]) and np.array_equal(logits, runs[0][1])
        assert all(np.array_equal(x, y) for x, y in zip(states, runs[0][2]))
    oracle = O.V7Runtime(O.build_v7(ogguf.GgufReader(data), weights_f16=False), B, act_f16=True)
    cur = list(first)
    for step in range(8):
        ol = oracle.infer_chunk([[t] for t in cur], list(range(B)))
        cur = [int(ol[b].argmax()) for b in range(B)]
        assert runs[0][0][step].tolist() == cur, step
