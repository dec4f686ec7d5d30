"""The oracle reproduces its committed golden vectors (CPU; guards against numpy / code drift)."""
import os

import numpy as np

from oracle import gguf, rwkv7, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tiny_q4k_golden.npz")


def test_oracle_matches_golden():
    g = np.load(GOLD)
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 42)
    for tag, wf in (("inline", False), ("ref", True)):
        rt = rwkv7.V7Runtime(rwkv7.build_v7(gguf.GgufReader(data), weights_f16=wf), 1, act_f16=True)
        logits = rt.infer_chunk([g["prompt"].tolist()], [len(g["prompt"]) - 1])
        np.testing.assert_allclose(logits, g[f"logits_{tag}"], rtol=0, atol=1e-5)
        tok = int(logits[0].argmax())
        for want in g[f"greedy_{tag}"][:4]:
            tok = int(rt.infer_chunk([[tok]], [0])[0].argmax())
            assert tok == int(want)


def test_f16_weight_rounding_changes_logits_slightly():
    """inline (f32) vs reference-effective (f16-rounded) weights differ, but only at
    the 1e-3 level on this model -- both deltas are reported in DESIGN.md."""
    g = np.load(GOLD)
    d = np.abs(g["logits_inline"] - g["logits_ref"]).max()
    assert 0 < d < 5e-2
