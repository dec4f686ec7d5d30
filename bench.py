#!/usr/bin/env python3
"""bench.py -- RWKV-7 decode throughput of the HIP path on N GPUs of one node.

Metric (BASELINE.json): tokens/sec RWKV-7 1.5B Q4_K_M batch-1 decode on one MI355X, with the
achieved HBM GB/s against the roofline.  A "step" is one decoded token per stream: the device-
resident greedy loop (gather embedding -> 24 layers -> head -> argmax -> next token), i.e. the
reference's bench loop (examples/bench.rs:224-236) with sampling kept on the device.

Weights are random-initialised blocks of the 1.5B architecture (no network for checkpoints):
Q4_K for the twelve big matrices per layer, Q6_K head, F16 embedding table, F32 LoRA/vectors
(converted to f16 at load, as the reference does) -- the BASELINE.md section 2 byte tally.

N > 1: one process per GPU, every rank runs its own independent stream(s) (global stream ids
`replicas.partition_streams`) on a full weight replica; no collective on the data path,
barrier-bracketed timing, max over ranks, value = streams * steps / time  ("scaling": "weak").
`python bench.py --gpus N` with no launcher environment spawns the N ranks ITSELF (fresh child
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started before the parent touches any GPU; the
parent only forwards rank 0's JSON line); under `torch.distributed.run` (RANK already set) it is one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0   # float4 copy rate measured on MI355X (same guide, chip-level parameters)


PMC_PROFILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")      # newest first


def measured_traffic(model, batch, mode):
    """(HBM bytes per decode step, source) from the committed PMC run of this same command (profiles/rNN_pmc_traffic.json,
    made by tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: hardware counters cannot be
    read from inside the process, so this figure is STATIC -- the JSON line says so in `traffic_source`).
    (None, None) when no profile of this model / batch / mode exists."""
    if not (model == "1.5B" and batch == 1 and mode == 1):
        return None, None
    for name in PMC_PROFILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            return int(json.load(open(path))["hbm_bytes_per_step"]), f"profiles/{name} (static: rocprofv3 --pmc run of this command, not measured in this run)"
        except Exception:
            continue
    return None, None

CONFIGS = {     # L, D, F, V, lora w/a/v/g
    "tiny": (2, 256, 1024, 512, 32, 32, 32, 64),
    "0.1B": (12, 768, 3072, 65536, 64, 64, 32, 128),
    "1.5B": (24, 2048, 8192, 65536, 96, 96, 64, 256),
    "2.9B": (32, 2560, 10240, 65536, 96, 96, 64, 320),
}


# --------------------------------------------------------------------------- synthetic GGUF (harness side)
def _f16b(x):
    return np.asarray(x, dtype="<f2").view(np.uint8)


def _q4k_blocks(rng, n_elem, std):
    """Legal random Q4_K blocks with w = d*sc*q - dmin*m ~ zero-mean, given std."""
    nb = n_elem // 256
    out = np.empty((nb, 144), np.uint8)
    d = np.float16(std / (47.5 * 4.61))
    out[:, 0:2] = _f16b([d])
    out[:, 2:4] = _f16b([np.float16(8.0) * d])
    sc = rng.integers(32, 64, (nb, 8), dtype=np.uint8)
    m = np.rint(sc.astype(np.float32) * (7.5 / 8.0)).astype(np.uint8)
    s = np.zeros((nb, 12), np.uint8)
    for j in range(4):
        s[:, j] = (sc[:, j] & 63) | ((sc[:, j + 4] >> 4) << 6)
        s[:, j + 4] = (m[:, j] & 63) | ((m[:, j + 4] >> 4) << 6)
        s[:, j + 8] = (sc[:, j + 4] & 0xF) | ((m[:, j + 4] & 0xF) << 4)
    out[:, 4:16] = s
    out[:, 16:144] = rng.integers(0, 256, (nb, 128), dtype=np.uint8)
    return out.reshape(-1)


def _q6k_blocks(rng, n_elem, std):
    nb = n_elem // 256
    out = np.empty((nb, 210), np.uint8)
    out[:, 0:192] = rng.integers(0, 256, (nb, 192), dtype=np.uint8)
    out[:, 192:208] = rng.integers(40, 101, (nb, 16), dtype=np.uint8)      # int8 scales 40..100
    out[:, 208:210] = _f16b([np.float16(std / (70.0 * 18.5))])
    return out.reshape(-1)


def _q5k_blocks(rng, n_elem, std):
    """Legal random Q5_K blocks (176 B: d, dmin, scales 12, qh 32, qs 128), w = d*sc*q - dmin*m ~ zero-mean."""
    nb = n_elem // 256
    out = np.empty((nb, 176), np.uint8)
    d = np.float16(std / (47.5 * 9.23))
    out[:, 0:2] = _f16b([d])
    out[:, 2:4] = _f16b([np.float16(16.0) * d])
    sc = rng.integers(32, 64, (nb, 8), dtype=np.uint8)
    m = np.rint(sc.astype(np.float32) * (15.5 / 16.0)).astype(np.uint8)
    s = np.zeros((nb, 12), np.uint8)
    for j in range(4):
        s[:, j] = (sc[:, j] & 63) | ((sc[:, j + 4] >> 4) << 6)
        s[:, j + 4] = (m[:, j] & 63) | ((m[:, j + 4] >> 4) << 6)
        s[:, j + 8] = (sc[:, j + 4] & 0xF) | ((m[:, j + 4] & 0xF) << 4)
    out[:, 4:16] = s
    out[:, 16:176] = rng.integers(0, 256, (nb, 160), dtype=np.uint8)
    return out.reshape(-1)


CONFIGS_V6 = {      # SURVEY section 8 cfg 4 / 5: L, D, F, V, time_mix rank, time_decay rank
    "v6-tiny": (2, 256, 896, 512, 32, 64),
    "v6-7B": (32, 4096, 14336, 65536, 64, 128),
    "v6-14B": (61, 4096, 14336, 65536, 64, 128),        # cfg 5: Q8_0 file + per-layer Int8 / NF4 map (--quant)
    # three layers of the cfg 4 / cfg 5 layer shape with a small vocabulary: the real matrix widths (K = 4096, 14336) at a size the
    # NumPy oracle can follow (tests/test_gpu_v6_fullshape.py)
    "v6-7B-3L": (3, 4096, 14336, 8192, 64, 128),
    "v6-14B-3L": (3, 4096, 14336, 8192, 64, 128),
}


def _q8_0_blocks(rng, n_elem, std):
    nb = n_elem // 32
    out = np.empty((nb, 34), np.uint8)
    out[:, 0:2] = _f16b([np.float16(std / 73.9)])            # int8 uniform in [-128, 127]: std 73.9
    out[:, 2:34] = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    return out.reshape(-1)


def make_model_gguf_v6(name, seed=42):
    """RWKV-6 "World" architecture, Q5_K_M-style: Q5_K matrices, Q6_K head, F16 embedding, F32 LoRA / vectors
    (names: gguf.rs:1198-1251); the 14B config is a Q8_0 file (cfg 5)."""
    L, D, F, V, R, W = CONFIGS_V6[name]
    q8 = name.startswith("v6-14B")
    mat_id = 8 if q8 else 13
    mat_blocks = _q8_0_blocks if q8 else _q5k_blocks
    rng = np.random.default_rng(seed)
    tensors = []

    def f32(nm, dims, vals):
        tensors.append((nm, dims, 0, np.ascontiguousarray(vals, dtype="<f4").view(np.uint8).reshape(-1)))

    def nrm(n, std):
        return rng.standard_normal(n, dtype=np.float32) * np.float32(std)

    tensors.append(("token_embd.weight", [D, V], 1, _f16b(nrm(V * D, 1.0))))
    f32("token_embd_norm.weight", [D], 1 + nrm(D, 0.1)); f32("token_embd_norm.bias", [D], nrm(D, 0.05))
    f32("output_norm.weight", [D], 1 + nrm(D, 0.1)); f32("output_norm.bias", [D], nrm(D, 0.05))
    tensors.append(("output.weight", [D, V], 14, _q6k_blocks(rng, V * D, 1.0 / np.sqrt(D))))
    for l in range(L):
        p = f"blk.{l}."
        for nm in ("attn_norm", "attn_norm_2", "attn_ln_x"):
            f32(p + nm + ".weight", [D], 1 + nrm(D, 0.1)); f32(p + nm + ".bias", [D], nrm(D, 0.05))
        f32(p + "attn_time_decay", [D], rng.random(D, dtype=np.float32) * 3.5 - 3.0)
        f32(p + "attn_time_first", [64, D // 64], nrm(D, 0.3))
        for nm in ("x", "w", "k", "v", "r", "g"):
            f32(p + f"attn_time_mix_{nm}", [D], rng.random(D, dtype=np.float32))
        f32(p + "attn_time_mix_w1", [D, 5 * R], nrm(5 * R * D, 1.0 / np.sqrt(D)))
        f32(p + "attn_time_mix_w2", [R, D, 5], nrm(5 * D * R, 0.3 / np.sqrt(R)))
        f32(p + "attn_time_decay_w1", [D, W], nrm(W * D, 1.0 / np.sqrt(D)))
        f32(p + "attn_time_decay_w2", [W, D], nrm(D * W, 0.5 / np.sqrt(W)))
        for nm in ("k", "v", "r", "g", "output"):
            tensors.append((p + f"attn_{nm}.weight", [D, D], mat_id, mat_blocks(rng, D * D, (0.5 if nm == "k" else (0.1 if nm == "output" else 1.0)) / np.sqrt(D))))
        f32(p + "ffn_time_mix_k", [D], rng.random(D, dtype=np.float32)); f32(p + "ffn_time_mix_r", [D], rng.random(D, dtype=np.float32))
        tensors.append((p + "ffn_k.weight", [D, F], mat_id, mat_blocks(rng, F * D, 1.0 / np.sqrt(D))))
        tensors.append((p + "ffn_v.weight", [F, D], mat_id, mat_blocks(rng, D * F, 0.05 / np.sqrt(F))))
        tensors.append((p + "ffn_r.weight", [D, D], mat_id, mat_blocks(rng, D * D, 1.0 / np.sqrt(D))))
    meta = [("general.architecture", 8, "rwkv6"), ("general.alignment", 4, 32), ("rwkv6.wkv.head_size", 4, 64),
            ("rwkv6.block_count", 4, L), ("rwkv6.embedding_length", 4, D), ("rwkv6.feed_forward_length", 4, F)]
    return _write_gguf(tensors, meta)


def use_more_bits(i_layer, n_layer):
    """llama.cpp's Q4_K_M recipe: attn_v and ffn_down get Q6_K in the first and last eighth and every third layer between."""
    return i_layer < n_layer // 8 or i_layer >= 7 * n_layer // 8 or (i_layer - n_layer // 8) % 3 == 2


def make_model_gguf(name, seed=42, mixed=False):
    if name in CONFIGS_V6:
        return make_model_gguf_v6(name, seed)
    L, D, F, V, rw, ra, rv, rg = CONFIGS[name]
    rng = np.random.default_rng(seed)
    tensors = []        # (name, dims, type_id, raw)

    def f32(nm, dims, vals):
        tensors.append((nm, dims, 0, np.ascontiguousarray(vals, dtype="<f4").view(np.uint8).reshape(-1)))

    def nrm(n, std):
        return rng.standard_normal(n, dtype=np.float32) * np.float32(std)

    tensors.append(("token_embd.weight", [D, V], 1, _f16b(nrm(V * D, 1.0))))
    f32("token_embd_norm.weight", [D], 1 + nrm(D, 0.1)); f32("token_embd_norm.bias", [D], nrm(D, 0.05))
    f32("output_norm.weight", [D], 1 + nrm(D, 0.1)); f32("output_norm.bias", [D], nrm(D, 0.05))
    tensors.append(("output.weight", [D, V], 14, _q6k_blocks(rng, V * D, 1.0 / np.sqrt(D))))
    for l in range(L):
        p = f"blk.{l}."
        for nm in ("attn_norm", "attn_norm_2"):
            f32(p + nm + ".weight", [D], 1 + nrm(D, 0.1)); f32(p + nm + ".bias", [D], nrm(D, 0.05))
        f32(p + "time_mix_lerp_fused.weight", [D, 1, 1, 6], rng.random(6 * D, dtype=np.float32))
        f32(p + "time_mix_w0.weight", [D], rng.random(D, dtype=np.float32) * 3 - 1.5)
        f32(p + "time_mix_a0.weight", [D], nrm(D, 0.5)); f32(p + "time_mix_v0.weight", [D], nrm(D, 0.5))
        for nm, r, s2 in (("w", rw, 1.0), ("a", ra, 1.0), ("v", rv if l else ra, 1.0), ("g", rg, 2.0)):
            f32(p + f"time_mix_{nm}1.weight", [D, r], nrm(r * D, 1.0 / np.sqrt(D)))
            f32(p + f"time_mix_{nm}2.weight", [r, D], nrm(D * r, s2 / np.sqrt(r)))
        f32(p + "time_mix_r_k.weight", [D], nrm(D, 0.3))
        f32(p + "time_mix_k_k.weight", [D], 1 + nrm(D, 0.2)); f32(p + "time_mix_k_a.weight", [D], 1 + nrm(D, 0.2))
        f32(p + "time_mix_ln.weight", [D], 1 + nrm(D, 0.1)); f32(p + "time_mix_ln.bias", [D], nrm(D, 0.05))
        # branch outputs are scaled down (as trained models' are relative to the residual stream): with unit-gain random
        # branches a 24-layer stack amplifies one f16 rounding flip into O(1) logit changes and nothing could be compared
        q6 = mixed and use_more_bits(l, L)
        for nm in ("key", "value", "receptance", "output"):
            std = (0.1 if nm == "output" else 1.0) / np.sqrt(D)
            if q6 and nm == "value":
                tensors.append((p + "time_mix_value.weight", [D, D], 14, _q6k_blocks(rng, D * D, std)))
            else:
                tensors.append((p + f"time_mix_{nm}.weight", [D, D], 12, _q4k_blocks(rng, D * D, std)))
        f32(p + "channel_mix_lerp_k.weight", [D], rng.random(D, dtype=np.float32))
        tensors.append((p + "channel_mix_key.weight", [D, F], 12, _q4k_blocks(rng, F * D, 1.0 / np.sqrt(D))))
        if q6:
            tensors.append((p + "channel_mix_value.weight", [F, D], 14, _q6k_blocks(rng, D * F, 0.05 / np.sqrt(F))))
        else:
            tensors.append((p + "channel_mix_value.weight", [F, D], 12, _q4k_blocks(rng, D * F, 0.05 / np.sqrt(F))))

    meta = [("general.architecture", 8, "rwkv7"), ("general.alignment", 4, 32), ("rwkv7.wkv.head_size", 4, 64),
            ("rwkv7.block_count", 4, L), ("rwkv7.embedding_length", 4, D), ("rwkv7.feed_forward_length", 4, F)]
    return _write_gguf(tensors, meta)


def _write_gguf(tensors, meta):
    import struct

    def wstr(s):
        b = s.encode()
        return struct.pack("<Q", len(b)) + b

    head = bytearray(struct.pack("<IIQQ", 0x46554747, 3, len(tensors), len(meta)))
    for k, t, v in meta:
        head += wstr(k) + struct.pack("<I", t) + (wstr(v) if t == 8 else struct.pack("<I", v))
    off, offs = 0, []
    for nm, dims, tid, raw in tensors:
        offs.append(off)
        head += wstr(nm) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", tid, off)
        off = (off + raw.size + 31) & ~31
    base = (len(head) + 31) & ~31
    buf = np.zeros(base + off, np.uint8)
    buf[: len(head)] = np.frombuffer(bytes(head), np.uint8)
    for (nm, dims, tid, raw), o in zip(tensors, offs):
        buf[base + o: base + o + raw.size] = raw
    return buf


# --------------------------------------------------------------------------- cpu baseline (oracle port)
def cpu_baseline(gguf_bytes, first_token, seconds=15.0):
    """The oracle's C restatement of the reference's effective path (f16 weights, f16 activations,
    f32 accumulate) timed on the host cores on a bounded sample of the same decode workload."""
    try:
        from oracle import cport
    except Exception as e:      # checker not built: report, never substitute
        return {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    return cport.time_decode(gguf_bytes, first_token, seconds)


MFMA_PEAK_TFLOPS = 2500.0   # dense f16 / bf16 MFMA peak of the chip (MI355X_MICROARCH.md; the 5 PF headline includes 2:1 sparsity)


def prefill_leg(wrk, ctx, rt, model, batch, prompt, chunk, repeats=5, warmups=2, mode=1, robust=False):
    """The reference's prefill protocol (examples/bench.rs:176-222, bench_format.rs:34-35): `batch` prompts of `prompt` random tokens fed
    through runtime.infer in chunks of `chunk` tokens (token_chunk_size), option Last; `warmups` untimed runs, then `repeats` timed ones
    (wall clock around the whole prompt, logits read-back of the last row included, as the reference times it).  Reports the mean."""
    L, D, F, V, lw, la, lv, lg = CONFIGS[model]
    flop_tok = 2.0 * (12.0 * D * D * L + L * 2.0 * D * (lw + la + lg) + (L - 1) * 2.0 * D * lv)      # dense contraction per token
    times = []
    for rep in range(warmups + repeats):
        toks = [[(7 + 13 * i + 101 * b + rep) % (V - 1) for i in range(prompt)] for b in range(batch)]
        inp = wrk.RnnInput(toks, chunk)
        ctx.sync()
        t0 = time.perf_counter()
        chunks = 0
        while sum(inp.remaining(b) for b in range(batch)) > 0:
            rt.infer(inp, mode=mode)
            chunks += 1
        ctx.sync()
        if rep >= warmups:
            times.append(time.perf_counter() - t0)
    mean = sum(times) / len(times)
    med = sorted(times)[len(times) // 2]
    if robust:      # the stacked leg: the median (one repetition in a few stalls for ~60 ms on some boxes -- host side, the kernels are the same); mean kept beside it
        mean_ms, mean = mean * 1e3, med
    total = prompt * batch
    tflops = total * flop_tok / mean / 1e12
    extra = {"statistic": "median", "ms_mean": round(mean_ms, 3)} if robust else {"statistic": "mean"}
    return {"tokens": prompt, "streams": batch, "chunk": chunk, "chunks": chunks, "repeats": repeats, "warmups": warmups, **extra,
            "tokens_per_s": round(total / mean, 1), "ms": round(mean * 1e3, 3), "ms_best": round(min(times) * 1e3, 3),
            "matrix_tflops": round(tflops, 2), "peak_tflops": MFMA_PEAK_TFLOPS, "mfma_frac": round(tflops / MFMA_PEAK_TFLOPS, 4),
            "protocol": "examples/bench.rs: random prompt, token_chunk_size chunks, wall clock incl. logits read-back of the last row"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_replicas(n, argv):
    """`bench.py --gpus N` without a launcher environment: spawn the N ranks as fresh child processes (one per GPU) BEFORE
    this process touches any GPU -- the parent never imports wrk or torch.cuda -- and forward rank 0's JSON line.
    Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT exactly as torch.distributed.run sets them."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # watch every rank: the first one that dies takes the others down with it (a rank that never reaches the rendezvous would otherwise
    # leave the rest in init_process_group / the barrier until the backend's own timeout -- minutes -- and the caller's window with it)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.05)
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed} exited with code {procs[failed].returncode}; stopping the other ranks\n")
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    rcs = [p.returncode for p in procs]
    out = b"".join(c for c in chunks if c)
    for line in out.decode(errors="replace").splitlines():      # the JSON line to stdout; library chatter (e.g. gloo's connection notes) to stderr
        (sys.stdout if (failed is None and line.lstrip().startswith("{")) else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    return 0


def metric_name(model, batch, world):
    """BASELINE.json's metric string for the headline workload; the same wording with the model / quantisation / stream
    count actually run for every other configuration."""
    fam, size, quant = ("RWKV-6", model[3:], "Q8_0" if model.startswith("v6-14B") else "Q5_K_M") if model in CONFIGS_V6 else ("RWKV-7", model, "Q4_K_M")
    streams = "" if batch == 1 else f" batch={batch}/GPU"
    return f"tokens/sec {fam} {size} {quant} decode{streams} @{world} GPU; achieved HBM GB/s vs roofline"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--model", default="1.5B", choices=sorted(CONFIGS) + sorted(CONFIGS_V6))
    ap.add_argument("--batch", type=int, default=1, help="independent streams per GPU")
    ap.add_argument("--mode", type=int, default=1, help="1 = fused decode kernels, 0 = one kernel per reference op")
    ap.add_argument("--groups", type=int, default=1, help="deal the --batch streams of a GPU over this many concurrent decode pipelines (RWKV-7; 1 = one batched step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the prefill legs (pp512 of the reference's protocol, and 32 x 128 stacked tokens)")
    ap.add_argument("--quant", default="", help="ModelBuilder::quant map, e.g. int8:0-29,nf4:30-60 (layers inclusive)")
    ap.add_argument("--mixed", action="store_true", help="llama.cpp Q4_K_M tensor mix: Q6_K for attn value / ffn value in about half of the layers")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="backend of the timing barrier / max-reduce (the only communication); nccl == RCCL.  gloo allows a rehearsal of N ranks on a box with fewer GPUs")
    ap.add_argument("--device-map", default="", help="rehearsal only: comma list, local rank -> HIP device (e.g. 0,0 runs two ranks on one GPU; needs --dist-backend gloo)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for one rank (exercises the RCCL path on a one-GPU box)")
    ap.add_argument("--stub-device", action="store_true",
                    help="TEST ONLY (tests/test_bench_launcher.py): exercise the launcher, the rank rendezvous (gloo) and the aggregation "
                         "without a GPU; the decode step is replaced by a sleep and the JSON line is marked \"data\": \"stub\"")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0, help="seconds a rank waits for the others in init_process_group / the barriers")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="TEST ONLY: with --stub-device this rank exits with code 3 before the rendezvous")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_replicas(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.stub_device and rank == args.stub_fail_rank:
        sys.stderr.write(f"bench.py: rank {rank} failing on request (--stub-fail-rank)\n")
        sys.exit(3)
    dist = None
    device = local_rank
    if args.device_map:
        dm = [int(v) for v in args.device_map.split(",")]
        device = dm[local_rank]
        if len(set(dm)) < len(dm):
            # ranks SHARE a GPU (rehearsal): the persistent decode engine needs every CU of its GPU for itself -- two of them would each
            # hold part of the chip and wait for the rest until their bounded spins give up -- so the shared ranks keep the launches
            os.environ["WRK_ENGINE"] = "0"
    use_gloo = args.stub_device or args.dist_backend == "gloo"
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        import datetime
        rdv = datetime.timedelta(seconds=args.rendezvous_timeout)      # a missing rank fails the rendezvous in seconds, not in the backend's half hour
        if use_gloo:
            dist.init_process_group("gloo", timeout=rdv)
        else:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device), timeout=rdv)

    sys.path.insert(0, os.path.join(ROOT, "web-rwkv-gguf_amd", "wrk"))
    import replicas         # no GPU dependency (the stub ranks import nothing else of the package)
    group = replicas.ReplicaGroup(dist, device="cuda" if (dist is not None and not use_gloo) else None)
    B = args.batch
    streams = group.my_streams(world * B)           # global ids of the independent sequences this rank owns
    assert len(streams) == B

    if args.stub_device:
        group.barrier()
        t0 = time.perf_counter()
        time.sleep(0.001 * args.steps * (1 + 0.1 * rank))
        dev_ms = (time.perf_counter() - t0) * 1e3
        group.barrier()
        ms = group.max_over_ranks(dev_ms)
        total = group.sum_over_ranks(len(streams))
        if rank == 0:
            print(json.dumps({"metric": "STUB " + metric_name(args.model, B, world), "value": round(total * args.steps / (ms / 1e3), 2),
                              "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(ms / args.steps, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "none", "data": "stub", "config": {"workload": "launcher self-test, no device work", "streams": int(total)}}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import wrk
    t0 = time.time()
    # every rank builds the SAME weights (a replica): seed 42
    gg = make_model_gguf(args.model, seed=42, mixed=args.mixed) if args.model not in CONFIGS_V6 else make_model_gguf(args.model, seed=42)
    ctx = wrk.Context(device)
    reader = wrk.GgufReader(gg)
    quant = {}
    for part in filter(None, args.quant.split(",")):
        kind, rng_ = part.split(":")
        lo, hi = (rng_.split("-") + [rng_])[:2]
        for l in range(int(lo), int(hi) + 1):
            quant[l] = {"int8": wrk.QUANT_INT8, "nf4": wrk.QUANT_NF4}[kind]
    runtime = wrk.Runtime(ctx, reader, num_batch=B, weights=wrk.WEIGHTS_INLINE, quant=quant or None)
    load_s = time.time() - t0
    engine = None
    if args.model not in CONFIGS_V6 and B == 1 and args.mode == 1:
        ok, why = runtime.engine_status()       # the persistent batch-1 decode engine (one launch per token) or the five-launch layer
        engine = "on" if ok and os.environ.get("WRK_ENGINE", "1") != "0" else f"off ({why or 'WRK_ENGINE=0'})"
    first = [(17 + 101 * g) % (runtime.info.num_vocab - 1) for g in streams]
    token_bytes = runtime.token_bytes(B)

    def barrier():
        ctx.sync()
        if dist is not None:
            group.barrier()
            if not use_gloo:
                import torch
                torch.cuda.synchronize()

    if args.warmup > 0:
        runtime.generate_greedy(first, args.warmup, mode=args.mode, groups=args.groups)
    barrier()
    w0 = time.perf_counter()
    toks, dev_ms = runtime.generate_greedy(first, args.steps, mode=args.mode, groups=args.groups)     # HIP events on the ctx stream
    ctx.sync()
    wall_ms = (time.perf_counter() - w0) * 1e3
    barrier()
    ms = group.max_over_ranks(max(dev_ms, 0.0))
    wall_ms = group.max_over_ranks(wall_ms)

    if rank == 0:
        ms_per_step = ms / args.steps
        value = world * B * args.steps / (ms / 1e3)
        achieved = token_bytes / (ms_per_step / 1e3) / 1e9
        traffic, traffic_source = measured_traffic(args.model, B, args.mode)
        out = {
            "metric": metric_name(args.model, B, world),
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": (f"RWKV-6 World {args.model[3:]} {'Q8_0' if args.model.startswith('v6-14B') else 'Q5_K_M (Q5_K matrices)'}, Q6_K head, F16 LoRA"
                                    f"{', quant map ' + args.quant if args.quant else ''}, batch={B} greedy decode, "
                                    if args.model in CONFIGS_V6 else
                                    f"RWKV-7 World {args.model} Q4_K_M ({'llama.cpp mix: Q4_K + Q6_K attn/ffn value' if args.mixed else 'Q4_K matrices'}, Q6_K head, F16 LoRA) batch={B} greedy decode, ") +
                                   f"{'fused kernels' if args.mode == 1 else 'one kernel per reference op'} under hipGraph",
                       **({"decode_engine": engine} if engine else {}), "streams_per_gpu": B, "pipelines_per_gpu": args.groups, "streams": world * B, "parallelism": f"replicas x{world}" if world > 1 else "single",
                       **({"rehearsal": f"device map {args.device_map}, {args.dist_backend} barrier: ranks SHARE GPUs, not a scaling number"} if args.device_map else {})},
            # per GPU: every replica streams its own copy of the weights
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "peak_measured": HBM_MEASURED_GBS, "frac_of_measured": round(achieved / HBM_MEASURED_GBS, 4),
                         "kernel": "one decode step = one hipGraph launch", "algorithmic_bytes_per_launch": token_bytes},
            "wall_ms_per_step_incl_host": round(wall_ms / args.steps, 5), "load_seconds": round(load_s, 1),
        }
        # the reference's bench is pp512 + tg128 (examples/bench.rs:33,94-97): the decode above is the tg leg (headline metric); the prefill
        # legs ride in the same line.  RWKV-7 models, one GPU, after the timed region.
        if world == 1 and not args.no_prefill and args.model in CONFIGS:
            try:
                out["prefill"] = prefill_leg(wrk, ctx, runtime, args.model, 1, 512, 128, mode=args.mode)
                runtime.close()
                rt32 = wrk.Runtime(ctx, reader, num_batch=32, weights=wrk.WEIGHTS_INLINE, quant=quant or None)      # cfg-3 regime: 32 prompts stacked
                out["prefill_batched"] = prefill_leg(wrk, ctx, rt32, args.model, 32, 128, 32 * 128, repeats=5, warmups=2, mode=args.mode, robust=True)     # one chunk of 4096 stacked tokens
                rt32.close()
                runtime = None
            except Exception as e:      # a failing extra leg must not take the headline line with it
                out["prefill"] = out.get("prefill") or {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gg, first[0]) if args.model not in CONFIGS_V6 else None     # the C port is RWKV-7 only
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if runtime is not None:
        runtime.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
