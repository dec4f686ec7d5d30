"""Which merged stage of the mode-1 prefill differs from the op list?  WRK_MERGE_MASK bit per stage, max |logit diff| vs mode 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))
import numpy as np
import wrk
from oracle import synth

ctx = wrk.Context(0)
cfg = synth.CONFIGS["tiny"]
data = synth.make_v7_gguf(cfg, 7)
V = cfg.num_vocab
p0, p1 = synth.tokens(5, "merged-a", 70, V), synth.tokens(5, "merged-b", 26, V)

def run(mode, mask):
    os.environ["WRK_MERGE_MASK"] = str(mask)
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=2, weights=wrk.WEIGHTS_INLINE)
    l = rt.infer(wrk.RnnInput([p0, p1], 96, [wrk.RNN_FULL, wrk.RNN_LAST]), mode=mode)
    st = rt.state_back(0)
    rt.close()
    return l[0], st

ref, rst = run(0, 0)
for mask in (0, 1, 2, 4, 8, 16, 32, 64, 127):
    l, st = run(1, mask)
    d = np.abs(l - ref)
    rows = np.nonzero(d.max(axis=1))[0]
    print(f"mask {mask:2d}: max logit diff {d.max():.3e}  first differing row {rows[0] if rows.size else None}  state diff {np.abs(st - rst).max():.3e}", flush=True)
