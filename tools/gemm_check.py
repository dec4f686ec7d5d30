#!/usr/bin/env python3
"""Numeric cross-check of the MFMA GEMM kernels against the single-token matvec on model-sized shapes (the oracle-based
tests use small shapes; this covers the 8-wave K split, long rows and the Q6_K head)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))
sys.path.insert(0, ROOT)
import wrk  # noqa: E402
from tools.microbench import BLOCK  # noqa: E402


def rand_matrix(ctx, kind, k, m, rng):
    be, bb = BLOCK[kind]
    if kind == "F16":
        raw = (rng.standard_normal(k * m).astype(np.float16) * np.float16(0.02)).view(np.uint8)
    else:
        raw = rng.integers(0, 256, k * m // be * bb, dtype=np.uint8).reshape(-1, bb)
        for o in {"Q4_K": (0, 2), "Q5_K": (0, 2), "Q6_K": (208,), "Q8_0": (0,)}[kind]:
            raw[:, o:o + 2] = np.frombuffer(np.float16(0.01).tobytes(), np.uint8)
        raw = raw.reshape(-1)
    return wrk.Matrix(ctx, kind, k, m, raw)


def main():
    ctx = wrk.Context(0)
    rng = np.random.default_rng(1)
    bad = 0
    for kind, k, m in [("Q4_K", 8192, 2048), ("Q4_K", 2048, 8192), ("Q4_K", 2048, 2048), ("Q6_K", 2048, 65536), ("Q5_K", 14336, 4096),
                       ("Q5_K", 4096, 14336), ("F16", 2048, 96), ("F16", 96, 2048), ("Q8_0", 4096, 4096), ("Q6_K", 8192, 2048)]:
        mat = rand_matrix(ctx, kind, k, m, rng)
        for n in (2, 3, 16, 40, 130):
            x = rng.standard_normal((n, k)).astype(np.float16)
            ref = np.empty((n, m), np.float32)
            for t in range(n):          # single-token matvec, one row at a time
                o = ctx.zeros([m, 1, 1], np.float32)
                mat.matmul_op(ctx.tensor(x[t:t + 1], [k, 1, 1]), o)
                ref[t] = o.back().reshape(m)
            out = ctx.zeros([m, n, 1], np.float32)
            os.environ["WRK_GEMM_MIN"] = "2"
            mat.matmul_op(ctx.tensor(x, [k, n, 1]), out, turbo=True)
            got = out.back().reshape(n, m)
            # turbo only takes >= 16 tokens through wrk_op_matmul; the runtime's own path (gemm_min_tokens) is exercised by the model tests
            err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-9)
            flag = "" if err < 1e-4 else "   <-- MISMATCH"
            bad += err >= 1e-4
            print(f"{kind:5s} K={k:5d} M={m:6d} N={n:4d}: rel err {err:.2e}{flag}", flush=True)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
