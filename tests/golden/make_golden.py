"""Generates tests/golden/tiny_q4k_golden.npz from the oracle (run from the repo root:
`python tests/golden/make_golden.py`).  Inputs: the deterministic 'tiny' Q4_K/Q6_K-head GGUF
(oracle/synth.py, sha256 pinned in tiny_q4k.sha256) and a 21-token prompt.  Outputs: last-token
logits, 16 greedy tokens and the recurrent state for both weight modes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gguf, rwkv7, synth  # noqa: E402

cfg = synth.CONFIGS["tiny"]
data = synth.make_v7_gguf(cfg, 42)
prompt = synth.tokens(3, "prompt", 21, cfg.num_vocab)
out = {"prompt": np.array(prompt, np.uint32)}
for tag, wf in (("inline", False), ("ref", True)):
    rt = rwkv7.V7Runtime(rwkv7.build_v7(gguf.GgufReader(data), weights_f16=wf), 1, act_f16=True)
    logits = rt.infer_chunk([prompt], [len(prompt) - 1])
    tok, toks = int(logits[0].argmax()), []
    for _ in range(16):
        tok = int(rt.infer_chunk([[tok]], [0])[0].argmax())
        toks.append(tok)
    out[f"logits_{tag}"] = logits
    out[f"greedy_{tag}"] = np.array(toks, np.uint32)
    out[f"state_{tag}"] = rt.state.back(0).astype(np.float16)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_q4k_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
