"""Deterministic synthetic RWKV-7 models as GGUF bytes (oracle; test infrastructure only).

No reference counterpart: the reference ships no weights (`.MISSING_LARGE_BLOBS`) and there is no
network, so fixtures and the benchmark model are random-initialised with the *architecture* the
reference loads (tensor names and ggml dims as its converter writes them:
assets/scripts/convert_hf_to_gguf.py:534-602; name map src/runtime/gguf.rs:1173-1329).

Randomness is SplitMix64 implemented with integer ops only, so the bytes are identical on every
machine and numpy version; tests pin the sha256 of the generated file.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .gguf import write_gguf
from .quantize import QUANTIZE

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _key(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little")


def uniform(seed: int, name: str, n: int) -> np.ndarray:
    """float32 in [0, 1) with 24 random bits."""
    z = splitmix64(_key(seed, name), n)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def normal(seed: int, name: str, n: int) -> np.ndarray:
    """Irwin-Hall(4) approximation of N(0,1): integer->float ops only (bit-stable)."""
    z = splitmix64(_key(seed, name), n)
    acc = np.zeros(n, dtype=np.float32)
    for s in range(4):
        acc += ((z >> np.uint64(16 * s)) & np.uint64(0xFFFF)).astype(np.float32)
    return ((acc * np.float32(1.0 / 65536.0) - np.float32(2.0)) * np.float32(1.7320508)).astype(np.float32)


def tokens(seed: int, name: str, n: int, vocab: int) -> List[int]:
    """Uniform ids in [0, vocab-1) like examples/bench.rs:184-186 (PRNG differs: fastrand is not available)."""
    z = splitmix64(_key(seed, name), n)
    return [int(v) for v in (z % np.uint64(vocab - 1))]


@dataclass
class V7Config:
    num_layer: int = 2
    num_emb: int = 256
    num_hidden: int = 1024
    num_vocab: int = 512
    head_size: int = 64
    lora_w: int = 32
    lora_a: int = 32
    lora_v: int = 32
    lora_g: int = 64

    @property
    def num_head(self) -> int:
        return self.num_emb // self.head_size


# named configurations
CONFIGS: Dict[str, V7Config] = {
    "tiny": V7Config(2, 256, 1024, 512, 64, 32, 32, 32, 64),
    "small": V7Config(3, 512, 2048, 1000, 64, 32, 32, 32, 64),
    "0.1B": V7Config(12, 768, 3072, 65536, 64, 64, 64, 32, 128),
    "1.5B": V7Config(24, 2048, 8192, 65536, 64, 96, 96, 64, 256),
    # three layers of the 1.5B LAYER SHAPE (D = 2048, F = 8192, 32 heads, the real LoRA ranks) with a small vocabulary: the widths of the
    # headline model at a size the NumPy oracle follows buffer by buffer (tests/test_gpu_layer_parity.py)
    "1.5B-3L": V7Config(3, 2048, 8192, 1024, 64, 96, 96, 64, 256),
    "2.9B-2L": V7Config(2, 2560, 10240, 1024, 64, 96, 96, 64, 320),      # the 2.9B layer shape (rows of 2560 elements: two chunk iterations per wave)
    "2.9B": V7Config(32, 2560, 10240, 65536, 64, 96, 96, 64, 320),
}


def v7_tensor_plan(cfg: V7Config, seed: int):
    """Yield (gguf_name, ggml_dims, kind, float32 values).  kind: 'mat' (big matrix), 'lora',
    'vec', 'emb', 'head'.  Values are scaled so activations stay O(1) with f16 storage."""
    D, F, V, H, S = cfg.num_emb, cfg.num_hidden, cfg.num_vocab, cfg.num_head, cfg.head_size

    def N(name, n, std):
        return normal(seed, name, n) * np.float32(std)

    def U(name, n, lo, hi):
        return uniform(seed, name, n) * np.float32(hi - lo) + np.float32(lo)

    yield "token_embd.weight", [D, V], "emb", N("emb", V * D, 1.0)
    yield "token_embd_norm.weight", [D], "vec", 1.0 + N("ln0.w", D, 0.1)
    yield "token_embd_norm.bias", [D], "vec", N("ln0.b", D, 0.05)
    yield "output_norm.weight", [D], "vec", 1.0 + N("lnout.w", D, 0.1)
    yield "output_norm.bias", [D], "vec", N("lnout.b", D, 0.05)
    yield "output.weight", [D, V], "head", N("head", V * D, 1.0 / np.sqrt(D))
    for l in range(cfg.num_layer):
        p = f"blk.{l}."
        k = f"L{l}."
        yield p + "attn_norm.weight", [D], "vec", 1.0 + N(k + "ln1.w", D, 0.1)
        yield p + "attn_norm.bias", [D], "vec", N(k + "ln1.b", D, 0.05)
        yield p + "attn_norm_2.weight", [D], "vec", 1.0 + N(k + "ln2.w", D, 0.1)
        yield p + "attn_norm_2.bias", [D], "vec", N(k + "ln2.b", D, 0.05)
        yield p + "time_mix_lerp_fused.weight", [D, 1, 1, 6], "vec", U(k + "maa", 6 * D, 0.0, 1.0)
        yield p + "time_mix_w0.weight", [D], "vec", U(k + "w0", D, -1.5, 1.5)
        yield p + "time_mix_w1.weight", [D, cfg.lora_w], "lora", N(k + "w1", cfg.lora_w * D, 1.0 / np.sqrt(D))
        yield p + "time_mix_w2.weight", [cfg.lora_w, D], "lora", N(k + "w2", D * cfg.lora_w, 1.0 / np.sqrt(cfg.lora_w))
        yield p + "time_mix_a0.weight", [D], "vec", N(k + "a0", D, 0.5)
        yield p + "time_mix_a1.weight", [D, cfg.lora_a], "lora", N(k + "a1", cfg.lora_a * D, 1.0 / np.sqrt(D))
        yield p + "time_mix_a2.weight", [cfg.lora_a, D], "lora", N(k + "a2", D * cfg.lora_a, 1.0 / np.sqrt(cfg.lora_a))
        # layer 0 carries dummy v0/v1/v2 copies of a0/a1/a2 (convert_hf_to_gguf.py:596-599)
        if l == 0:
            yield p + "time_mix_v0.weight", [D], "vec", N(k + "a0", D, 0.5)
            yield p + "time_mix_v1.weight", [D, cfg.lora_a], "lora", N(k + "a1", cfg.lora_a * D, 1.0 / np.sqrt(D))
            yield p + "time_mix_v2.weight", [cfg.lora_a, D], "lora", N(k + "a2", D * cfg.lora_a, 1.0 / np.sqrt(cfg.lora_a))
        else:
            yield p + "time_mix_v0.weight", [D], "vec", N(k + "v0", D, 0.5)
            yield p + "time_mix_v1.weight", [D, cfg.lora_v], "lora", N(k + "v1", cfg.lora_v * D, 1.0 / np.sqrt(D))
            yield p + "time_mix_v2.weight", [cfg.lora_v, D], "lora", N(k + "v2", D * cfg.lora_v, 1.0 / np.sqrt(cfg.lora_v))
        yield p + "time_mix_g1.weight", [D, cfg.lora_g], "lora", N(k + "g1", cfg.lora_g * D, 1.0 / np.sqrt(D))
        yield p + "time_mix_g2.weight", [cfg.lora_g, D], "lora", N(k + "g2", D * cfg.lora_g, 2.0 / np.sqrt(cfg.lora_g))
        yield p + "time_mix_r_k.weight", [D], "vec", N(k + "r_k", D, 0.3)
        yield p + "time_mix_k_k.weight", [D], "vec", 1.0 + N(k + "k_k", D, 0.2)
        yield p + "time_mix_k_a.weight", [D], "vec", 1.0 + N(k + "k_a", D, 0.2)
        yield p + "time_mix_ln.weight", [D], "vec", 1.0 + N(k + "lnx.w", D, 0.1)
        yield p + "time_mix_ln.bias", [D], "vec", N(k + "lnx.b", D, 0.05)
        for nm in ("key", "value", "receptance", "output"):
            yield p + f"time_mix_{nm}.weight", [D, D], "mat", N(k + "att." + nm, D * D, 1.0 / np.sqrt(D))
        yield p + "channel_mix_lerp_k.weight", [D], "vec", U(k + "ffn.x_k", D, 0.0, 1.0)
        yield p + "channel_mix_key.weight", [D, F], "mat", N(k + "ffn.key", F * D, 1.0 / np.sqrt(D))
        yield p + "channel_mix_value.weight", [F, D], "mat", N(k + "ffn.value", D * F, 0.5 / np.sqrt(F))


def make_v7_gguf(cfg: V7Config, seed: int = 42, mat: str = "Q4_K", head: str = "Q6_K", emb: str = "F16",
                 lora: str = "F32", vec: str = "F32", mat_override: Dict[str, str] | None = None) -> bytes:
    """Build a complete GGUF file.  ``mat_override`` maps a substring of a tensor name to a type
    (to make Q4_K_M-style mixtures)."""
    kinds = {"mat": mat, "head": head, "emb": emb, "lora": lora, "vec": vec}
    tensors: List[Tuple[str, List[int], str, np.ndarray]] = []
    for name, dims, kind, vals in v7_tensor_plan(cfg, seed):
        tn = kinds[kind]
        if mat_override and kind == "mat":
            for sub, t in mat_override.items():
                if sub in name:
                    tn = t
        tensors.append((name, dims, tn, QUANTIZE[tn](vals.astype(np.float32))))
    meta = [
        ("general.architecture", "str", "rwkv7"),
        ("general.alignment", "u32", 32),
        ("rwkv7.block_count", "u32", cfg.num_layer),
        ("rwkv7.embedding_length", "u32", cfg.num_emb),
        ("rwkv7.feed_forward_length", "u32", cfg.num_hidden),
        ("rwkv7.wkv.head_size", "u32", cfg.head_size),
        ("rwkv7.attention.layer_norm_epsilon", "f32", 1e-5),
    ]
    return write_gguf(meta, tensors)


# --------------------------------------------------------------------------- RWKV-6
@dataclass
class V6Config:
    num_layer: int = 2
    num_emb: int = 256
    num_hidden: int = 896
    num_vocab: int = 512
    head_size: int = 64
    time_mix: int = 32
    time_decay: int = 64

    @property
    def num_head(self) -> int:
        return self.num_emb // self.head_size


V6_CONFIGS: Dict[str, V6Config] = {
    "tiny": V6Config(2, 256, 896 + 128, 512, 64, 32, 64),
    "small": V6Config(7, 512, 1792, 1000, 64, 32, 64),           # 7 layers: crosses the rescale-every-6 boundary
    "7B": V6Config(32, 4096, 14336, 65536, 64, 64, 128),
}


def v6_tensor_plan(cfg: V6Config, seed: int):
    """GGUF names the reference's map understands for V6 (gguf.rs:1198-1251: attn_* / ffn_* forms)."""
    D, F, V, H, S, R, W = cfg.num_emb, cfg.num_hidden, cfg.num_vocab, cfg.num_head, cfg.head_size, cfg.time_mix, cfg.time_decay

    def N(name, n, std):
        return normal(seed, name, n) * np.float32(std)

    def U(name, n, lo, hi):
        return uniform(seed, name, n) * np.float32(hi - lo) + np.float32(lo)

    yield "token_embd.weight", [D, V], "emb", N("emb", V * D, 1.0)
    yield "token_embd_norm.weight", [D], "vec", 1.0 + N("ln0.w", D, 0.1)
    yield "token_embd_norm.bias", [D], "vec", N("ln0.b", D, 0.05)
    yield "output_norm.weight", [D], "vec", 1.0 + N("lnout.w", D, 0.1)
    yield "output_norm.bias", [D], "vec", N("lnout.b", D, 0.05)
    yield "output.weight", [D, V], "head", N("head", V * D, 1.0 / np.sqrt(D))
    for l in range(cfg.num_layer):
        p, k = f"blk.{l}.", f"V6L{l}."
        yield p + "attn_norm.weight", [D], "vec", 1.0 + N(k + "ln1.w", D, 0.1)
        yield p + "attn_norm.bias", [D], "vec", N(k + "ln1.b", D, 0.05)
        yield p + "attn_norm_2.weight", [D], "vec", 1.0 + N(k + "ln2.w", D, 0.1)
        yield p + "attn_norm_2.bias", [D], "vec", N(k + "ln2.b", D, 0.05)
        yield p + "attn_time_decay", [D], "vec", U(k + "td", D, -3.0, 0.5)
        yield p + "attn_time_first", [S, H], "vec", N(k + "tf", D, 0.3)
        for nm in ("x", "w", "k", "v", "r", "g"):
            yield p + f"attn_time_mix_{nm}", [D], "vec", U(k + "tm" + nm, D, 0.0, 1.0)
        yield p + "attn_time_mix_w1", [D, 5 * R], "lora", N(k + "tmw1", 5 * R * D, 1.0 / np.sqrt(D))
        yield p + "attn_time_mix_w2", [R, D, 5], "lora", N(k + "tmw2", 5 * D * R, 0.3 / np.sqrt(R))
        yield p + "attn_time_decay_w1", [D, W], "lora", N(k + "tdw1", W * D, 1.0 / np.sqrt(D))
        yield p + "attn_time_decay_w2", [W, D], "lora", N(k + "tdw2", D * W, 0.5 / np.sqrt(W))
        yield p + "attn_ln_x.weight", [D], "vec", 1.0 + N(k + "lnx.w", D, 0.1)
        yield p + "attn_ln_x.bias", [D], "vec", N(k + "lnx.b", D, 0.05)
        for nm in ("k", "v", "r", "g", "output"):
            yield p + f"attn_{nm}.weight", [D, D], "mat", N(k + "att." + nm, D * D, (0.5 if nm == "k" else 1.0) / np.sqrt(D))
        yield p + "ffn_time_mix_k", [D], "vec", U(k + "ffn.tmk", D, 0.0, 1.0)
        yield p + "ffn_time_mix_r", [D], "vec", U(k + "ffn.tmr", D, 0.0, 1.0)
        yield p + "ffn_k.weight", [D, F], "mat", N(k + "ffn.key", F * D, 1.0 / np.sqrt(D))
        yield p + "ffn_v.weight", [F, D], "mat", N(k + "ffn.value", D * F, 0.5 / np.sqrt(F))
        yield p + "ffn_r.weight", [D, D], "mat", N(k + "ffn.rec", D * D, 1.0 / np.sqrt(D))


# attn_* / ffn_* spelling (what the reference's map knows) -> llama.cpp's RWKV-6 spelling (what its own converter emits,
# assets/scripts/convert_hf_to_gguf.py:455-525); same dims and data
_V6_LLAMA = {"attn_time_decay": "time_mix_decay.weight", "attn_time_first": "time_mix_first.weight",
             "attn_time_mix_x": "time_mix_lerp_x.weight", "attn_time_mix_w": "time_mix_lerp_w.weight", "attn_time_mix_k": "time_mix_lerp_k.weight",
             "attn_time_mix_v": "time_mix_lerp_v.weight", "attn_time_mix_r": "time_mix_lerp_r.weight", "attn_time_mix_g": "time_mix_lerp_g.weight",
             "attn_time_mix_w1": "time_mix_w1.weight", "attn_time_mix_w2": "time_mix_w2.weight",
             "attn_time_decay_w1": "time_mix_decay_w1.weight", "attn_time_decay_w2": "time_mix_decay_w2.weight",
             "attn_ln_x.weight": "time_mix_ln.weight", "attn_ln_x.bias": "time_mix_ln.bias",
             "attn_k.weight": "time_mix_key.weight", "attn_v.weight": "time_mix_value.weight", "attn_r.weight": "time_mix_receptance.weight",
             "attn_g.weight": "time_mix_gate.weight", "attn_output.weight": "time_mix_output.weight",
             "ffn_time_mix_k": "channel_mix_lerp_k.weight", "ffn_time_mix_r": "channel_mix_lerp_r.weight",
             "ffn_k.weight": "channel_mix_key.weight", "ffn_v.weight": "channel_mix_value.weight", "ffn_r.weight": "channel_mix_receptance.weight"}


def make_v6_gguf(cfg: V6Config, seed: int = 42, mat: str = "Q5_K", head: str = "Q6_K", emb: str = "F16", lora: str = "F32",
                 vec: str = "F32", mat_override: Dict[str, str] | None = None, names: str = "attn") -> bytes:
    """names: "attn" = the spellings the reference's map knows; "llama" = llama.cpp's RWKV-6 tensor names."""
    kinds = {"mat": mat, "head": head, "emb": emb, "lora": lora, "vec": vec}
    tensors = []
    for name, dims, kind, vals in v6_tensor_plan(cfg, seed):
        if names == "llama" and name.startswith("blk."):
            pre, rem = name.split(".", 2)[:2], name.split(".", 2)[2]
            name = ".".join(pre) + "." + _V6_LLAMA.get(rem, rem)
        tn = kinds[kind]
        if mat_override and kind == "mat":
            for sub, t in mat_override.items():
                if sub in name:
                    tn = t
        tensors.append((name, dims, tn, QUANTIZE[tn](vals.astype(np.float32))))
    meta = [("general.architecture", "str", "rwkv6"), ("general.alignment", "u32", 32), ("rwkv6.block_count", "u32", cfg.num_layer),
            ("rwkv6.embedding_length", "u32", cfg.num_emb), ("rwkv6.wkv.head_size", "u32", cfg.head_size)]
    return write_gguf(meta, tensors)
