#!/usr/bin/env python3
"""Per-shape matvec microbenchmark through the C ABI (wrk_op_matmul), in the spirit of the
reference's examples/bench_q4k_shaders.rs (K=M=2560, 10 warm-up + 100 timed).  N launches are
captured into one program (hipGraph) and replayed; reports us/launch and GB/s of stored bytes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import wrk  # noqa: E402

BLOCK = {"Q4_K": (256, 144), "Q5_K": (256, 176), "Q6_K": (256, 210), "Q8_0": (32, 34), "F16": (1, 2), "INT8": (128, 132), "NF4": (64, 34)}


def run(ctx, kind, k, m, nin=1, reps=200, copies=4, turbo=False):
    be, bb = BLOCK[kind]
    rng = np.random.default_rng(0)
    mats = []
    for _ in range(copies):      # several copies so consecutive launches do not re-read the same bytes from cache
        if kind == "INT8":      # codes ++ (min, max) f16 per 128 flattened elements
            raw = np.concatenate([rng.integers(0, 256, k * m, dtype=np.uint8), np.tile(np.array([-0.1, 0.1], np.float16).view(np.uint8), k * m // 128)])
            mats.append(wrk.Matrix(ctx, kind, k, m, raw))
            continue
        if kind == "NF4":       # nibbles ++ absmax f16 per 64 flattened elements
            raw = np.concatenate([rng.integers(0, 256, k * m // 2, dtype=np.uint8), np.tile(np.array([0.1], np.float16).view(np.uint8), k * m // 64)])
            mats.append(wrk.Matrix(ctx, kind, k, m, raw))
            continue
        raw = rng.integers(0, 256, k * m // be * bb, dtype=np.uint8)
        if kind != "F16":
            raw = raw.reshape(-1, bb)
            off = {"Q4_K": (0, 2), "Q5_K": (0, 2), "Q6_K": (208,), "Q8_0": (0,)}[kind]
            for o in off:
                raw[:, o:o + 2] = np.frombuffer(np.float16(0.01).tobytes(), np.uint8)
            raw = raw.reshape(-1)
        else:
            raw = (rng.standard_normal(k * m).astype(np.float16)).view(np.uint8)
        mats.append(wrk.Matrix(ctx, kind, k, m, raw))
    x = ctx.tensor(rng.standard_normal((nin, k)).astype(np.float16), [k, nin, 1])
    out = ctx.zeros([m, nin, 1])
    for mt in mats:
        mt.matmul_op(x, out, turbo=turbo)
    ctx.sync()
    ctx.check(wrk.hip.wrk_capture_begin(ctx.h))
    for i in range(reps):
        mats[i % copies].matmul_op(x, out, turbo=turbo)
    prog = C.c_void_p()
    ctx.check(wrk.hip.wrk_capture_end(ctx.h, C.byref(prog)))
    best = 1e9
    for _ in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(wrk.hip.wrk_program_launch(ctx.h, prog))
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    wrk.hip.wrk_program_destroy(prog)
    gb = mats[0].stream_bytes / 1e9
    tf = 2.0 * k * m * nin / best / 1e12
    print(f"{kind:5s} K={k:5d} M={m:6d} T={nin:4d}{' mfma' if turbo else '     '}: {best * 1e6:8.2f} us/launch  {gb / best:8.1f} GB/s  ({gb * 1e3:.2f} MB)  {tf:7.2f} TFLOP/s", flush=True)
    return best


if __name__ == "__main__":
    ctx = wrk.Context(0)
    shapes = [("Q4_K", 2048, 2048), ("Q4_K", 2048, 8192), ("Q4_K", 8192, 2048), ("Q6_K", 2048, 65536), ("F16", 2048, 65536),
              ("Q4_K", 2560, 2560), ("Q8_0", 4096, 4096), ("Q5_K", 4096, 4096), ("F16", 2048, 96), ("F16", 96, 2048),
              ("INT8", 4096, 4096), ("NF4", 4096, 4096), ("Q5_K", 14336, 4096), ("Q5_K", 4096, 14336)]
    if len(sys.argv) > 1:
        shapes = [s for s in shapes if s[0] in sys.argv[1:]]
    for kind, k, m in shapes:
        run(ctx, kind, k, m)
    run(ctx, "Q4_K", 2048, 8192, nin=4)
    run(ctx, "Q4_K", 2048, 8192, nin=8)
    for n in (16, 32, 64, 128, 512):
        run(ctx, "Q4_K", 2048, 8192, nin=n, reps=50, turbo=True)
    for n in (16, 128):
        run(ctx, "Q4_K", 8192, 2048, nin=n, reps=50, turbo=True)
        run(ctx, "Q6_K", 2048, 8192, nin=n, reps=50, turbo=True)
        run(ctx, "F16", 2048, 2048, nin=n, reps=50, turbo=True)
    ctx.close()
