"""RWKV-6 forward pass (oracle; test infrastructure only).

NumPy restatement of the reference's V6 model path:
  build .............. src/runtime/v6.rs:995-1170 (time_mix stack [C,1,5] = w,k,v,r,g :1053-1071)
  dispatch_layer ..... src/runtime/v6.rs:701-958
  time_mix_v6 ........ src/shaders/time_mix_v6.wgsl:83-155
  channel_mix (V6) ... src/shaders/channel_mix.wgsl:83-107 (x <- sigmoid(r) * v)
  token_shift with a per-token factor tensor ... src/shaders/token_shift.wgsl:70-117
  activations ........ src/tensor/ops.rs:205-235 (StableExp = exp(-exp(x)), Silu)
  rescale / discount . src/runtime/v6.rs:49,1046,953-955
Buffer dtypes follow Runtime<f16> (v6.rs:265-300): att_k/att_v/att_r and time_decay are f32, the
rest f16.  PARITY UNPINNED (no reference test covers any of this).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from .gguf import GgufReader
from .rnn import stack_cursors
from .rwkv7 import (ACT, GN_EPS, LN_EPS, ModelInfo, _id, _mat, _mat16, _vec16, custom_tanh, layer_norm, mix, r16, sigmoid)


def loader_info_v6(reader: GgufReader) -> ModelInfo:
    """loader.rs:238-371, V6 branch."""
    num_layer = 0
    for name in reader.names():
        if name.startswith("blocks."):
            rest = name[len("blocks."):]
            num_layer = max(num_layer, int(rest[: max(rest.find("."), 0)]))
    num_layer += 1
    need = ["time_mix_x", "time_mix_w", "time_mix_k", "time_mix_v", "time_mix_r", "time_mix_g", "time_mix_w1", "time_mix_w2",
            "time_decay_w1", "time_decay_w2"]
    if not (all(reader.contains(f"blocks.0.att.{n}") for n in need) and reader.contains("blocks.0.ffn.time_mix_k")
            and reader.contains("blocks.0.ffn.time_mix_r")):
        raise ValueError("invalid model version")
    embed = reader.shape("emb.weight")
    ffn = reader.shape("blocks.0.ffn.key.weight")
    custom = {"time_mix": reader.shape("blocks.0.att.time_mix_w1")[0] // 5, "time_decay": reader.shape("blocks.0.att.time_decay_w1")[0]}
    return ModelInfo(num_layer, embed[1], ffn[0], embed[0], reader.shape("blocks.0.att.time_first")[0], custom)


@dataclass
class V6Model:
    info: ModelInfo
    emb: np.ndarray
    ln0: tuple
    ln_out: tuple
    head: np.ndarray
    layers: List[Dict[str, np.ndarray]]
    rescale: int = 6


def build_v6(reader: GgufReader, weights_f16: bool = True, rescale: int = 6, quant=None) -> V6Model:
    """quant: layer -> "int8" | "nf4" (ModelBuilder::quant, v6.rs:1045)."""
    quant = quant or {}
    info = loader_info_v6(reader)
    D, R = info.num_emb, info.custom["time_mix"]
    layers = []
    for l in range(info.num_layer):
        discount = np.float32(2.0 ** (-(l // rescale)))
        att, ffn = f"blocks.{l}.att", f"blocks.{l}.ffn"
        p: Dict[str, np.ndarray] = {}
        p["ln1_w"], p["ln1_b"] = _vec16(reader, f"blocks.{l}.ln1.weight"), _vec16(reader, f"blocks.{l}.ln1.bias")
        p["ln2_w"], p["ln2_b"] = _vec16(reader, f"blocks.{l}.ln2.weight"), _vec16(reader, f"blocks.{l}.ln2.bias")
        p["time_decay"] = _vec16(reader, f"{att}.time_decay")
        p["time_first"] = _vec16(reader, f"{att}.time_first")                         # load_vector_f32: f16-rounded values as f32
        p["time_mix_x"] = _vec16(reader, f"{att}.time_mix_x")
        p["time_mix"] = np.stack([_vec16(reader, f"{att}.time_mix_{n}") for n in "wkvrg"])          # [5, D]
        p["time_decay_w1"], p["time_decay_w2"] = _mat16(reader, f"{att}.time_decay_w1"), _mat16(reader, f"{att}.time_decay_w2")
        p["time_mix_w1"] = _mat16(reader, f"{att}.time_mix_w1")                                       # [5R, D]
        shp = reader.shape(f"{att}.time_mix_w2")                                                     # [5, D, R]
        p["time_mix_w2"] = r16(reader.tensor(f"{att}.time_mix_w2")[2]).reshape(shp)
        p["gn_w"], p["gn_b"] = _vec16(reader, f"{att}.ln_x.weight"), _vec16(reader, f"{att}.ln_x.bias")
        q = quant.get(l, "none")
        for n, k in (("w_k", "key"), ("w_v", "value"), ("w_r", "receptance"), ("w_g", "gate"), ("w_o", "output")):
            p[n] = _mat(reader, f"{att}.{k}.weight", weights_f16, q, discount if n == "w_o" else np.float32(1.0))
        p["ffn_mix_k"], p["ffn_mix_r"] = _vec16(reader, f"{ffn}.time_mix_k"), _vec16(reader, f"{ffn}.time_mix_r")
        for n, k in (("ffn_w_k", "key"), ("ffn_w_v", "value"), ("ffn_w_r", "receptance")):
            p[n] = _mat(reader, f"{ffn}.{k}.weight", weights_f16, q, discount if n == "ffn_w_v" else np.float32(1.0))
        if discount != 1.0 and q == "none":
            for n in ("w_o", "ffn_w_v"):
                p[n] = r16(discount * p[n]) if weights_f16 else (discount * p[n]).astype(np.float32)
        layers.append(p)
    return V6Model(info, _mat16(reader, "emb.weight"), (_vec16(reader, "blocks.0.ln0.weight"), _vec16(reader, "blocks.0.ln0.bias")),
                   (_vec16(reader, "ln_out.weight"), _vec16(reader, "ln_out.bias")), _mat(reader, "head.weight", weights_f16), layers, rescale)


def silu(x):
    return (x / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


class V6Runtime:
    def __init__(self, model: V6Model, num_batch: int, act_f16: bool = True):
        self.model = model
        info = model.info
        self.state = np.zeros((info.num_layer, num_batch, info.head_size + 2, info.num_emb), np.float32)
        self.rnd = r16 if act_f16 else _id

    def _mm(self, w, x, act="none", f32_out=False):
        y = ACT[act](np.matmul(x.astype(np.float32), w.T.astype(np.float32)).astype(np.float32))
        return y if f32_out else self.rnd(y)

    def infer_chunk(self, chunk_tokens: List[List[int]], headers: List[int]) -> np.ndarray:
        m, info, rnd = self.model, self.model.info, self.rnd
        D, S, H, R = info.num_emb, info.head_size, info.num_head, info.custom["time_mix"]
        lens = [len(c) for c in chunk_tokens]
        T = sum(lens)
        if T == 0:
            return np.zeros((0, info.num_vocab), np.float32)
        cur = stack_cursors(lens)
        batches = np.array([c & 0xFF for c in cur]); starts = np.array([(c >> 8) & 0xFFFF for c in cur]); clens = np.array([c >> 24 for c in cur])
        idx = np.arange(T)
        firsts, lasts = idx == starts, (idx - starts + 1) == clens
        tokens = np.concatenate([np.asarray(c, dtype=np.int64) for c in chunk_tokens if len(c)])

        def shift(xv, state_row, fac):                     # fac [D] or [T, D]
            prev = np.empty_like(xv)
            prev[1:] = xv[:-1]
            for t in np.nonzero(firsts)[0]:
                prev[t] = state_row[batches[t]]
            return rnd(mix(xv, prev, fac if fac.ndim == 2 else fac[None, :]))

        x = rnd(layer_norm(m.emb[tokens], m.ln0[0], m.ln0[1], LN_EPS))
        for li, p in enumerate(m.layers):
            st = self.state[li]
            att_x = rnd(layer_norm(x, p["ln1_w"], p["ln1_b"], LN_EPS))
            row0 = st[:, 0, :]
            att_xx = shift(att_x, row0, p["time_mix_x"])
            tmx = self._mm(p["time_mix_w1"], att_xx, "tanh")                                   # [T, 5R] == [R, 5, T]
            tmx = tmx.reshape(T, 5, R)
            tm = np.stack([self._mm(p["time_mix_w2"][i], tmx[:, i, :]) for i in range(5)])     # [5, T, D]
            tm = rnd(p["time_mix"][:, None, :] + tm)                                           # add(time_mix, buffer.time_mix)
            sx = [shift(att_x, row0, tm[i]) for i in range(5)]                                 # w, k, v, r, g
            k = self._mm(p["w_k"], sx[1], f32_out=True)
            v = self._mm(p["w_v"], sx[2], f32_out=True)
            r = self._mm(p["w_r"], sx[3], f32_out=True)
            g = self._mm(p["w_g"], sx[4])
            aw = self._mm(p["time_decay_w1"], sx[0], "tanh")
            td = self._mm(p["time_decay_w2"], aw, f32_out=True)
            td = (p["time_decay"][None, :] + td).astype(np.float32)
            td = np.exp(-np.exp(td, dtype=np.float32), dtype=np.float32)                       # StableExp, f32 buffer
            u = p["time_first"].reshape(H, S)
            y = np.empty((T, D), np.float32)
            for t in range(T):
                b = batches[t]
                if lasts[t]:
                    st[b, 0, :] = att_x[starts[t] + clens[t] - 1]
                Sm = st[b, 1:S + 1, :].reshape(S, H, S).transpose(1, 0, 2)                     # [H, j, i]
                kt, vt, rt, wt = (z[t].reshape(H, S) for z in (k, v, r, td))
                kv = kt[:, :, None] * vt[:, None, :]
                y[t] = np.einsum("hj,hji->hi", rt, u[:, :, None] * kv + Sm).reshape(D)
                st[b, 1:S + 1, :] = (wt[:, :, None] * Sm + kv).astype(np.float32).transpose(1, 0, 2).reshape(S, D)
            aux = rnd(y)
            aux = rnd(layer_norm(aux.reshape(T, H, S), p["gn_w"].reshape(H, S)[None], p["gn_b"].reshape(H, S)[None], GN_EPS).reshape(T, D))
            att_x = rnd(silu(g) * aux)                                                         # mul_activate(att_g Silu, att_x)
            o = self._mm(p["w_o"], att_x)
            x = rnd(o + x)
            ffn_x = rnd(layer_norm(x, p["ln2_w"], p["ln2_b"], LN_EPS))
            rowf = st[:, S + 1, :]
            kx, rx = shift(ffn_x, rowf, p["ffn_mix_k"]), shift(ffn_x, rowf, p["ffn_mix_r"])
            fk = self._mm(p["ffn_w_k"], kx, "squared_relu")
            fv = self._mm(p["ffn_w_v"], fk)
            fr = self._mm(p["ffn_w_r"], rx)
            for t in np.nonzero(lasts)[0]:
                st[batches[t], S + 1, :] = ffn_x[t]
            ffn_x = rnd(sigmoid(fr) * fv)                                                      # channel_mix
            x = rnd(ffn_x + x)
            if (li + 1) % m.rescale == 0:
                x = rnd(np.float32(0.5) * x)
        if not headers:
            return np.zeros((0, info.num_vocab), np.float32)
        hx = rnd(layer_norm(x[np.asarray(headers)], m.ln_out[0], m.ln_out[1], LN_EPS))
        return np.matmul(hx, m.head.T.astype(np.float32)).astype(np.float32)
