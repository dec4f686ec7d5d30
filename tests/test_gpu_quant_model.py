"""ModelBuilder::quant (v7.rs:1089, v6.rs:1045; SURVEY a9 / BASELINE cfg 5): layers loaded as web-rwkv Int8 / NF4
matrices.  Covers the three load routes of Loader::load_matrix / load_matrix_discount (loader.rs:756-951):
  * Q8_0 source + Int8  -> host repack_q8_0_to_int8, no f16 staging      (live direct arm)
  * Q4_0 source + NF4   -> host repack_q4_0_to_nf4                       (live direct arm)
  * anything else       -> f16 (+ discount) -> on-device quant_u8 / quant_nf4
against the oracle built with the same map (oracle/rwkv7.py::_mat).  Same bars as test_gpu_model.py; the
on-device int8 quantiser may differ from the oracle's by one code on a rounding boundary (see
test_gpu_wrkquant.py), which is far below the logit noise floor.  PARITY UNPINNED (no reference fixture)."""
import numpy as np
import pytest

import wrk
from oracle import gguf as ogguf
from oracle import rwkv6 as O6
from oracle import rwkv7 as O
from oracle import synth

LOGIT_TOL, LOGIT_MEAN_TOL = 1e-2, 1.5e-3      # tests/test_gpu_model.py


def assert_state_close(got, want):
    d = np.abs(got - want)
    assert d.max() <= 2e-2 * max(1.0, float(np.abs(want).max())), d.max()
    assert d.mean() <= 1e-3, d.mean()


pytestmark = pytest.mark.gpu
QID = {"int8": wrk.QUANT_INT8, "nf4": wrk.QUANT_NF4}


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


CASES = [
    ("Q8_0", {0: "int8", 1: "int8"}, 1024),      # direct repack
    ("Q4_0", {0: "nf4", 1: "nf4"}, 1024),        # direct repack
    ("Q4_K", {0: "int8"}, 1024),                 # K-quant -> f16 -> quant_u8 on layer 0 only; layer 1 stays inline Q4_K
    ("Q4_K", {1: "nf4"}, 1024),
    ("F16", {0: "nf4", 1: "int8"}, 1024),
    ("Q8_0", {0: "int8", 1: "int8"}, 1),         # rescale 1: layer 1 has discount 1/2 -> output/value take the f16 route
]


@pytest.mark.parametrize("mat,quant,rescale", CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_v7_quant_layers(ctx, mat, quant, rescale, mode):
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 7, mat=mat)
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=1, weights=wrk.WEIGHTS_INLINE, rescale=rescale,
                     quant={l: QID[q] for l, q in quant.items()})
    oracle = O.V7Runtime(O.build_v7(ogguf.GgufReader(data), weights_f16=False, rescale=rescale, quant=quant), 1, act_f16=True)
    V = rt.info.num_vocab
    prompt = synth.tokens(11, "prompt", 19, V)
    got = rt.infer(wrk.RnnInput([prompt], 32), mode=mode)[0]
    want = oracle.infer_chunk([prompt], [len(prompt) - 1])
    assert np.abs(got - want).max() <= LOGIT_TOL, np.abs(got - want).max()
    assert np.abs(got - want).mean() <= LOGIT_MEAN_TOL, np.abs(got - want).mean()
    tok = int(want[0].argmax())
    assert int(got[0].argmax()) == tok
    toks, _, last = rt.generate_greedy([tok], 8, mode=mode, want_logits=True)
    otoks = []
    for _ in range(8):
        ol = oracle.infer_chunk([[tok]], [0])
        tok = int(ol[0].argmax())
        otoks.append(tok)
    assert toks[:, 0].tolist() == otoks
    assert np.abs(last - ol).max() <= LOGIT_TOL
    assert_state_close(rt.state_back(0), oracle.state.back(0))
    rt.close()


def test_v7_quant_changes_stream_bytes(ctx):
    """Int8 layers stream 1 + 4/128 bytes per weight, NF4 0.5 + 2/64: the roofline denominator follows the map."""
    cfg = synth.CONFIGS["tiny"]
    data = synth.make_v7_gguf(cfg, 7, mat="F16")
    base = wrk.Runtime(ctx, wrk.GgufReader(data), weights=wrk.WEIGHTS_INLINE)
    q8 = wrk.Runtime(ctx, wrk.GgufReader(data), weights=wrk.WEIGHTS_INLINE, quant={0: wrk.QUANT_INT8})
    q4 = wrk.Runtime(ctx, wrk.GgufReader(data), weights=wrk.WEIGHTS_INLINE, quant={0: wrk.QUANT_NF4})
    n = 4 * cfg.num_emb ** 2 + 2 * cfg.num_emb * cfg.num_hidden          # quantised weights per layer
    assert base.token_bytes() - q8.token_bytes() == 2 * n - (n + n // 128 * 4)
    assert base.token_bytes() - q4.token_bytes() == 2 * n - (n // 2 + n // 64 * 2)
    for r in (base, q8, q4):
        r.close()


def test_bad_quant_value_rejected(ctx):
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 7)
    with pytest.raises(wrk.WrkError):
        wrk.Runtime(ctx, wrk.GgufReader(data), quant={0: 9})


@pytest.mark.parametrize("quant", [{0: "int8", 1: "nf4"}, {6: "int8"}])
def test_v6_quant_layers(ctx, quant):
    cfg = synth.V6_CONFIGS["small" if 6 in quant else "tiny"]
    data = synth.make_v6_gguf(cfg, 5, mat="Q8_0")
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=1, weights=wrk.WEIGHTS_INLINE, quant={l: QID[q] for l, q in quant.items()})
    model = O6.build_v6(ogguf.GgufReader(data), weights_f16=False, quant=quant)
    oracle = O6.V6Runtime(model, 1, act_f16=True)
    V = rt.info.num_vocab
    prompt = synth.tokens(2, "prompt", 17, V)
    got = rt.infer(wrk.RnnInput([prompt], 32))[0]
    want = oracle.infer_chunk([prompt], [len(prompt) - 1])
    # V6 noise floor is higher (7 layers, rescale halvings): same derivation as tests/test_gpu_v6.py
    assert np.abs(got - want).max() <= 3e-2, np.abs(got - want).max()
    assert np.abs(got - want).mean() <= 6e-3, np.abs(got - want).mean()
    assert int(got[0].argmax()) == int(want[0].argmax())
    rt.close()
