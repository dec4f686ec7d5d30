#!/usr/bin/env python3
"""Determinism stress for the K-sliced GEMM's in-kernel meeting (agent-scope sc1 stores / loads + arrival counters, wrk_gemm.hip):
decode B sequences for N steps twice on fresh runtimes with EVERY eligible launch on the K-sliced kernel (WRK_GEMM_KS=2) and compare
tokens, final logits and states bit for bit.  A stale partial tile (a read that beat a write-through store) would show as a difference;
the sum order is fixed (slice order), so identical inputs must give identical bits.  usage: python tools/ks_stress.py [--batch 16] [--steps 200]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))
os.environ["WRK_GEMM_KS"] = "2"
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="1.5B")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    import wrk
    gg = bench.make_model_gguf(a.model, seed=42)
    ctx = wrk.Context(0)
    first = [(17 + 101 * g) % 60000 for g in range(a.batch)]
    ref = None
    for rep in range(a.repeat):
        rt = wrk.Runtime(ctx, wrk.GgufReader(gg), num_batch=a.batch, weights=wrk.WEIGHTS_INLINE)
        toks, ms, logits = rt.generate_greedy(first, a.steps, mode=1, want_logits=True)
        states = [rt.state_back(b) for b in range(a.batch)]
        rt.close()
        cur = (toks.copy(), logits.copy(), states)
        if ref is None:
            ref = cur
            print(f"run 0: {a.batch} sequences x {a.steps} steps, {ms / a.steps:.4f} ms per step")
            continue
        same_t = np.array_equal(ref[0], cur[0])
        same_l = np.array_equal(ref[1], cur[1])
        same_s = all(np.array_equal(x, y) for x, y in zip(ref[2], cur[2]))
        print(f"run {rep}: tokens identical {same_t}, logits identical {same_l}, states identical {same_s}")
        if not (same_t and same_l and same_s):
            raise SystemExit("K-sliced GEMM: runs differ")
    print("ok")


if __name__ == "__main__":
    main()
