#!/usr/bin/env python3
"""Prefill throughput (BASELINE cfg 3 shape of work: chunked prefill through the MFMA dequant-GEMM).

Feeds `--batch` sequences of `--prompt` tokens through `Runtime.infer` in chunks of `--chunk` tokens
(RnnInput / RnnIter as in examples/bench.rs:176-222), option Last, and reports tokens/s and the MFMA rate of
the matrix work (2 * (12 D^2 L + LoRA) FLOP per token + head rows).  Logit readback of the single
header row per chunk is included (it is what `runtime.infer` returns)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))
import bench  # noqa: E402  (synthetic GGUF generator)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="1.5B")
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--mixed", action="store_true", help="llama.cpp Q4_K_M type mix (Q6_K for half the value matrices and the head)")
    a = ap.parse_args()
    import wrk
    ctx = wrk.Context(0)
    if a.model in bench.CONFIGS_V6:         # RWKV-6 (cfg 4 / 5 models): 5 D^2 att + 2 D F + D^2 ffn matrices, two LoRA stacks
        L, D, F, V, R, W = bench.CONFIGS_V6[a.model]
        rt = wrk.Runtime(ctx, wrk.GgufReader(bench.make_model_gguf_v6(a.model, seed=42)), num_batch=a.batch, weights=wrk.WEIGHTS_INLINE)
        flop_tok = 2.0 * L * (6.0 * D * D + 2.0 * D * F + 10.0 * D * R + 2.0 * D * W)
    else:
        L, D, F, V, lw, la, lv, lg = bench.CONFIGS[a.model]
        rt = wrk.Runtime(ctx, wrk.GgufReader(bench.make_model_gguf(a.model, seed=42, mixed=a.mixed)), num_batch=a.batch, weights=wrk.WEIGHTS_INLINE)
        flop_tok = 2.0 * (12.0 * D * D * L + L * 2.0 * D * (lw + la + lg) + (L - 1) * 2.0 * D * lv)
    best = None
    for rep in range(a.repeat + 1):
        toks = [[(7 + 13 * i + 101 * b + rep) % (V - 1) for i in range(a.prompt)] for b in range(a.batch)]
        inp = wrk.RnnInput(toks, a.chunk)
        ctx.sync()
        t0 = time.perf_counter()
        n = 0
        while sum(inp.remaining(b) for b in range(a.batch)) > 0:
            rt.infer(inp, mode=a.mode)
            n += 1
        ctx.sync()
        dt = time.perf_counter() - t0
        if rep and (best is None or dt < best):
            best = dt
    total = a.prompt * a.batch
    print(json.dumps({"workload": f"{'RWKV-6' if a.model in bench.CONFIGS_V6 else 'RWKV-7'} {a.model} {('Q8_0' if a.model.startswith('v6-14B') else 'Q5_K_M') if a.model in bench.CONFIGS_V6 else ('Q4_K_M mix' if a.mixed else 'Q4_K')} prefill, {a.batch} x {a.prompt} tokens, chunk {a.chunk}", "chunks": n,
                      "tokens_per_s": round(total / best, 1), "ms": round(best * 1e3, 3),
                      "matrix_TFLOPs": round(total * flop_tok / best / 1e12, 2), "mfma_peak_TFLOPs_f16_dense": 2500.0}))


if __name__ == "__main__":
    main()
