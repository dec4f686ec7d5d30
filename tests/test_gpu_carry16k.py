"""BASELINE cfg 5's mechanism: a 16k-token state carry.

One sequence of 16 384 + tokens goes through `runtime.infer` chunk by chunk (`RnnIter`, rnn.rs:280-335) with the recurrent
state resident on the device the whole time (v7.rs:152-217); the reference cannot take more than 255 tokens of a sequence per
dispatch (cursor `len` is u8, tensor/mod.rs:53-60), so the carry IS ~70-130 dispatches.  Half way the state makes a
`State::back -> load` round trip through the host.  At the end the state and the logits of the last token are compared with
the oracle fed the same stream (NumPy, chunks of 240 tokens: matmuls vectorised over the chunk, the recurrence stepped per
token).  RWKV-6 (`v6-tiny`, cfg 5's family) and RWKV-7 (`tiny`), chunk sizes 128 and 64.  (128 is the largest chunk ONE sequence
can use: `RnnInput::new` rounds the chunk size up to a power of two, rnn.rs:204-253, and 256 tokens of one sequence do not fit the
cursor -- the host refuses such a dispatch with WRK_E_UNSUPPORTED, which the test checks too.)

Bars: as the short-sequence whole-model tests (logits max 1e-2 / mean 1.5e-3, state 2e-2 relative max / 1e-3 mean): decay < 1
forgets old rounding differences, so 16k tokens must not be worse than 21.
"""
import numpy as np
import pytest

import wrk
from oracle import gguf as ogguf
from oracle import rwkv6 as O6
from oracle import rwkv7 as O7
from oracle import synth

pytestmark = pytest.mark.gpu
N_TOKENS = 16384 + 57          # not a multiple of any chunk size


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def carry(ctx, data, oracle, V, chunk, mode):
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=2, weights=wrk.WEIGHTS_INLINE)
    toks = synth.tokens(77, "carry", N_TOKENS, V)
    inp = wrk.RnnInput([[], toks], chunk)              # slot 1 of 2: the carry must leave slot 0 untouched
    calls, last, round_trip = 0, None, False
    while inp.remaining(1) > 0:
        out = rt.infer(inp, mode=mode)
        calls += 1
        if out[1].shape[0]:
            last = out[1]
        if not round_trip and inp.remaining(1) <= N_TOKENS // 2:
            saved = rt.state_back(1)                    # State::back(1)
            rt.state_load(np.full_like(saved, 7.0), 1)  # clobber, then State::load
            rt.state_load(saved, 1)
            round_trip = True
    assert round_trip and last is not None and last.shape == (1, V)
    want = None
    for i in range(0, N_TOKENS, 240):       # any split gives the same recurrence; a cursor holds at most 255 tokens
        part = toks[i:i + 240]
        want = oracle.infer_chunk([[], part], [len(part) - 1])
    d = np.abs(last - want)
    ost = oracle.state.back(1) if hasattr(oracle.state, "back") else oracle.state[:, 1]
    ds = np.abs(rt.state_back(1) - ost)
    untouched = not rt.state_back(0).any()
    rt.close()
    return calls, float(d.max()), float(d.mean()), int(last.argmax()) == int(want.argmax()), float(ds.max()), float(ds.mean()), float(np.abs(ost).max()), untouched


def test_more_than_255_tokens_of_one_sequence_per_dispatch_is_refused(ctx):
    cfg = synth.CONFIGS["tiny"]
    rt = wrk.Runtime(ctx, wrk.GgufReader(synth.make_v7_gguf(cfg, 42)), num_batch=1)
    with pytest.raises(wrk.WrkError):
        rt.infer(wrk.RnnInput([synth.tokens(1, "x", 600, cfg.num_vocab)], 255))       # 255 -> 256 (power of two) > u8
    rt.close()


@pytest.mark.parametrize("family,chunk,mode", [("v7", 128, 1), ("v7", 64, 0), ("v6", 64, 1), ("v6", 128, 0)])
def test_16k_token_state_carry(ctx, family, chunk, mode):
    if family == "v7":
        cfg = synth.CONFIGS["tiny"]
        data = synth.make_v7_gguf(cfg, 42)
        oracle = O7.V7Runtime(O7.build_v7(ogguf.GgufReader(data), weights_f16=False), 2, act_f16=True)
    else:
        cfg = synth.V6_CONFIGS["tiny"]
        data = synth.make_v6_gguf(cfg, 42)
        oracle = O6.V6Runtime(O6.build_v6(ogguf.GgufReader(data), weights_f16=False), 2, act_f16=True)
    calls, dmax, dmean, same_tok, smax, smean, sref, untouched = carry(ctx, data, oracle, cfg.num_vocab, chunk, mode)
    # the dispatch count of the reference scheduler (oracle/rnn.py restates RnnIter): full chunks, then the tail in multiples of 32
    from oracle.rnn import LAST, RnnInput, RnnInputBatch
    sched, want_calls = RnnInput([RnnInputBatch([], LAST), RnnInputBatch([0] * N_TOKENS, LAST)], chunk), 0
    while sched.num_token() > 0:
        sched.step()
        want_calls += 1
    assert calls == want_calls and calls >= N_TOKENS // chunk, (calls, want_calls)
    print(f"{family} chunk {chunk} mode {mode}: {calls} dispatches; logits max {dmax:.2e} mean {dmean:.2e}; state max {smax:.2e} mean {smean:.2e} (|state| <= {sref:.1f})")
    assert untouched
    assert same_tok and dmax <= 1e-2 and dmean <= 1.5e-3, (dmax, dmean)
    assert smax <= 2e-2 * max(1.0, sref) and smean <= 1e-3, (smax, smean)
