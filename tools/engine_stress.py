#!/usr/bin/env python3
"""Decode-engine soak: N greedy tokens of the headline model, three runs from the same state; tokens, logits and state must be bit-identical
(a hand-off that read a stale granule would not be) and no launch may give up.  usage: python tools/engine_stress.py [--steps 3000]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))
import bench  # noqa: E402

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=3000); ap.add_argument("--model", default="1.5B")
    a = ap.parse_args()
    import wrk
    ctx = wrk.Context(0)
    data = bench.make_model_gguf(a.model, seed=42)
    runs = []
    for i in range(3):
        rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=1, weights=wrk.WEIGHTS_INLINE)
        t0 = time.time()
        toks, ms, logits = rt.generate_greedy([17], a.steps, mode=1, want_logits=True)
        runs.append((toks.copy(), logits.copy(), rt.state_back(0)))
        print(f"run {i}: engine {rt.engine_status()}, {a.steps} tokens in {time.time() - t0:.2f} s, {ms / a.steps:.4f} ms/token", flush=True)
        rt.close()
    ok = all(np.array_equal(r[0], runs[0][0]) and np.array_equal(r[1], runs[0][1]) and np.array_equal(r[2], runs[0][2]) for r in runs[1:])
    print("bit-identical over three runs:", ok)
    sys.exit(0 if ok else 1)

if __name__ == "__main__":
    main()
