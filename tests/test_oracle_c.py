"""The C restatement (oracle/c) against the NumPy oracle: two independent restatements of the
reference's effective decode path must agree (logits to accumulation-order noise, greedy tokens
exactly, dequantised weights bit for bit)."""
import numpy as np
import pytest

from oracle import cport, dequant as dq, gguf, quantize as qz, rwkv7, synth


@pytest.mark.parametrize("tn", ["Q4_K", "Q5_K", "Q6_K", "Q8_0"])
def test_c_dequant_bit_exact(tn):
    w = synth.normal(5, tn, 256 * 24) * np.float32(0.05)
    raw = np.ascontiguousarray(qz.QUANTIZE[tn](w))
    out = np.empty(w.size, np.uint16)
    assert cport.lib.orc_dequant_f16(dq.GGML_TYPE_ID[tn], raw.ctypes.data, w.size, out.ctypes.data) == 0
    want = dq.DEQUANT[tn](raw, w.size, round_f16=True).astype(np.float16).view(np.uint16)
    assert np.array_equal(out, want)


def test_c_decode_matches_numpy_oracle():
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 42, mat_override={"time_mix_value": "Q5_K", "channel_mix_value": "Q8_0"})
    cm = cport.CModel(data)
    om = rwkv7.V7Runtime(rwkv7.build_v7(gguf.GgufReader(data), weights_f16=True), 1, act_f16=True)
    tok = 11
    for step in range(12):
        a = cm.decode(tok).copy()
        b = om.infer_chunk([[tok]], [0])[0]
        assert np.abs(a - b).max() < 1e-2 and np.abs(a - b).mean() < 1.5e-3, (step, np.abs(a - b).max())
        assert int(a.argmax()) == int(b.argmax())
        tok = int(b.argmax())
    d = np.abs(cm.state - om.state.back(0))
    assert d.max() < 2e-2 and d.mean() < 1e-3
