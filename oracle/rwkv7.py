"""RWKV-7 forward pass (oracle; test infrastructure only).

NumPy restatement of the reference's V7 model path:
  build (load)  .... src/runtime/v7.rs:1038-1227, src/runtime/loader.rs:104-132,563-951
  dispatch ......... src/runtime/v7.rs:598-713 (embed LN :649-659)
  dispatch_layer ... src/runtime/v7.rs:716-1007  (op numbers below = SURVEY 3.2)
  dispatch_header .. src/runtime/v7.rs:1009-1036
  state layout ..... src/runtime/v7.rs:146-208  ([D, S+2, B] f32 per layer)
and of the WGSL arithmetic each op dispatches:
  layer_norm / group_norm .. shaders/layer_norm.wgsl:63-121
  token_shift (REVERSED) ... shaders/token_shift.wgsl:85-117
  activations .............. tensor/ops.rs:205-235
  add / mul ................ shaders/binary.wgsl:38-78
  l2_norm .................. shaders/normalize.wgsl:117-152
  control_k_v7 ............. shaders/control_k_v7.wgsl:60-75
  lerp (REVERSED) .......... shaders/lerp.wgsl:74-92
  time_mix / time_first .... shaders/time_mix_v7.wgsl:68-70,143-262
  channel_mix (V7) ......... shaders/channel_mix.wgsl:83-107
  matmul (f32 accumulate) .. shaders/matmul_vec_fp16.wgsl:48-110

Two orthogonal switches reproduce the arithmetic variants discussed in SURVEY F1/F4:
  act_f16      every TensorOp output that the reference stores in a ``Runtime<f16>`` buffer is
               rounded to f16 (examples instantiate ``Bundle::<f16>``); logits and state stay f32.
  weights_f16  big matrices are the reference-at-HEAD effective weights (K-quants dequantised to
               f16 at load); False = ggml-canonical inline dequantisation in f32 (north_star).

PARITY UNPINNED for everything in this file (no reference test/fixture covers it).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import dequant as dq
from .gguf import GgufReader
from .rnn import stack_cursors

LN_EPS = np.float32(1.0e-5)     # v7.rs:47
GN_EPS = np.float32(64.0e-5)    # v7.rs:48
L2_EPS = np.float32(1.0e-12)    # v7.rs:46
W_SCALE = np.float32(-0.606531)  # time_mix_v7.wgsl:69


def r16(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def _id(x):
    return np.asarray(x, dtype=np.float32)


# ----------------------------------------------------------------------------- activations
def sigmoid(x):
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def custom_tanh(x):
    return np.where(x > 42.0, np.float32(1.0), np.tanh(x, dtype=np.float32)).astype(np.float32)


def squared_relu(x):
    p = np.maximum(x, np.float32(0.0))
    return (p * p).astype(np.float32)


ACT = {"none": _id, "tanh": custom_tanh, "sigmoid": sigmoid, "squared_relu": squared_relu}


# ----------------------------------------------------------------------------- elementwise ops
def layer_norm(x, w, b, eps):
    """x [..., C]; statistics in f64 then f32 (the reference uses an f32 Welford merge; both are
    roundings of the same mean / biased variance)."""
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=-1, keepdims=True)
    var = ((x64 - mean) ** 2).mean(axis=-1, keepdims=True)
    mean = mean.astype(np.float32)
    dev = (np.float32(1.0) / np.sqrt(var.astype(np.float32) + eps)).astype(np.float32)
    value = (x - mean) * dev
    return (value * w + b).astype(np.float32)       # fma(value, w, b)


def l2_norm(x, eps):
    s = (x.astype(np.float64) ** 2).sum(axis=-1, keepdims=True).astype(np.float32)
    norm = (np.float32(1.0) / np.sqrt(s + eps)).astype(np.float32)
    return (x * norm).astype(np.float32)


def mix(x, y, a):
    """WGSL mix(x, y, a) = x*(1-a) + y*a."""
    return (x * (np.float32(1.0) - a) + y * a).astype(np.float32)


# ----------------------------------------------------------------------------- model
@dataclass
class ModelInfo:
    num_layer: int
    num_emb: int
    num_hidden: int
    num_vocab: int
    num_head: int
    custom: Dict[str, int]

    @property
    def head_size(self) -> int:
        return self.num_emb // self.num_head

    @property
    def num_vocab_padded(self) -> int:          # model.rs:60-62, PAD_MAT[1] = 8
        return -(-self.num_vocab // 8) * 8


def loader_info(reader: GgufReader) -> ModelInfo:
    """loader.rs:238-371 (V7 branch only; other versions are out of scope and rejected)."""
    num_layer = 0
    for name in reader.names():
        if name.startswith("blocks."):
            rest = name[len("blocks."):]
            idx = rest.find(".")
            num_layer = max(num_layer, int(rest[: max(idx, 0)]))
    num_layer += 1
    embed = reader.shape("emb.weight")
    ffn = reader.shape("blocks.0.ffn.key.weight")
    sep = ["x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "w1", "w2", "a0", "a1", "a2", "g1", "g2", "r_k", "k_k", "k_a"]
    fused = ["time_maa", "w0", "w1", "w2", "a0", "a1", "a2", "g1", "g2", "r_k", "k_k", "k_a"]
    v7 = all(reader.contains(f"blocks.0.att.{n}") for n in sep) or all(reader.contains(f"blocks.0.att.{n}") for n in fused)
    if not v7:
        raise ValueError("invalid model version")
    custom = {k: reader.shape(f"blocks.{1 if k == 'v' else 0}.att.{k}1")[0] for k in ("w", "a", "g", "v")}
    return ModelInfo(num_layer, embed[1], ffn[0], embed[0], reader.shape("blocks.0.att.r_k")[0], custom)


def _vec16(reader, name):
    """load_vector_f16 (loader.rs:563-615): any float dtype -> f16."""
    _, _, v = reader.tensor(name)
    return r16(v).reshape(-1)


def _mat16(reader, name):
    """load_matrix_f16 (loader.rs:617-641): [M, K] f16 (K-quants dequantised by Reader::tensor)."""
    _, shape, v = reader.tensor(name)
    return r16(v).reshape(shape[0], shape[1])


def _mat(reader: GgufReader, name: str, weights_f16: bool, quant: str = "none", discount=np.float32(1.0)) -> np.ndarray:
    """load_matrix / load_matrix_discount (loader.rs:756-789, 923-951).  weights_f16=True is the reference at
    HEAD (F1); False lifts the gate and keeps ggml-canonical f32 dequantised values (north_star path).
    quant "int8" / "nf4" (ModelBuilder::quant): the live direct arms for Q8_0 / Q4_0 sources loaded without a
    discount (loader.rs:808-820, 901-918), otherwise f16 -> discount -> Matrix::quant_u8 / quant_nf4; the
    returned array holds the values the matmul shaders reconstruct (matmul_vec_int8/nf4.wgsl)."""
    shape = reader.shape(name)
    if quant != "none":
        from . import wrkquant as wq
        n = shape[0] * shape[1]
        qt = reader.quantized_tensor(name) if discount == 1.0 else None
        if qt is not None and qt[0] == dq.GGML_TYPE_ID["Q8_0"] and quant == "int8":
            return wq.dequantize_int8(*wq.repack_q8_0_to_int8(np.frombuffer(qt[1], np.uint8), n)).reshape(shape[0], shape[1])
        if qt is not None and qt[0] == dq.GGML_TYPE_ID["Q4_0"] and quant == "nf4":
            return wq.dequantize_nf4(*wq.repack_q4_0_to_nf4(np.frombuffer(qt[1], np.uint8), n)).reshape(shape[0], shape[1])
        w = _mat16(reader, name)
        if discount != 1.0:
            w = r16(discount * w)
        w16 = w.astype(np.float16)
        if quant == "int8":
            return wq.dequantize_int8(*wq.quantize_int8(w16)).reshape(shape[0], shape[1])
        return wq.dequantize_nf4(*wq.quantize_nf4(w16)).reshape(shape[0], shape[1])
    if weights_f16:
        return _mat16(reader, name)
    tn, raw = reader.raw_tensor(name)
    # only the kinds with an inline-dequant kernel keep f32-exact weights; every other type is dequantised to f16 at
    # load exactly as the reference does (gguf.rs:1690-1720)
    inline = tn in ("Q4_K", "Q5_K", "Q6_K", "Q8_0", "F16")
    return dq.dequantize(tn, raw, shape[0] * shape[1], round_f16=not inline).reshape(shape[0], shape[1])


@dataclass
class V7Layer:
    p: Dict[str, np.ndarray] = field(default_factory=dict)


@dataclass
class V7Model:
    info: ModelInfo
    emb: np.ndarray
    ln0: tuple
    ln_out: tuple
    head: np.ndarray
    layers: List[V7Layer]
    rescale: int = 1024


def build_v7(reader: GgufReader, weights_f16: bool = True, rescale: int = 1024, quant: Optional[Dict[int, str]] = None) -> V7Model:
    """ModelBuilder::build_v7 (v7.rs:1038-1227).  LoRA blending is not restated (the GGUF path ignores LoRA).
    quant: layer -> "int8" | "nf4" (ModelBuilder::quant, v7.rs:1089)."""
    quant = quant or {}
    info = loader_info(reader)
    emb = _mat16(reader, "emb.weight")                                   # CPU f16, v7.rs:1065
    ln0 = (_vec16(reader, "blocks.0.ln0.weight"), _vec16(reader, "blocks.0.ln0.bias"))
    ln_out = (_vec16(reader, "ln_out.weight"), _vec16(reader, "ln_out.bias"))
    # head is Matrix::Fp16 in the reference (v7.rs:1073, F2); the inline path keeps its blocks.
    head = _mat(reader, "head.weight", weights_f16)
    layers = []
    for layer in range(info.num_layer):
        discount = np.float32(2.0 ** (-(layer // rescale)))              # v7.rs:1090
        att, ffn = f"blocks.{layer}.att", f"blocks.{layer}.ffn"
        p: Dict[str, np.ndarray] = {}
        p["ln1_w"], p["ln1_b"] = _vec16(reader, f"blocks.{layer}.ln1.weight"), _vec16(reader, f"blocks.{layer}.ln1.bias")
        p["ln2_w"], p["ln2_b"] = _vec16(reader, f"blocks.{layer}.ln2.weight"), _vec16(reader, f"blocks.{layer}.ln2.bias")
        for n in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "a0", "k_k", "k_a"):
            p[n] = _vec16(reader, f"{att}.{n}")
        for n in ("w1", "w2", "a1", "a2", "g1", "g2"):
            p[n] = _mat16(reader, f"{att}.{n}")
        if layer == 0:                                                   # v7.rs:1115-1116 placeholder
            p["v0"], p["v1"], p["v2"] = p["a0"], p["a1"], p["a2"]
        else:
            p["v0"] = _vec16(reader, f"{att}.v0")
            p["v1"], p["v2"] = _mat16(reader, f"{att}.v1"), _mat16(reader, f"{att}.v2")
        p["r_k"] = _mat16(reader, f"{att}.r_k").reshape(-1)              # [H, S] flattened
        p["gn_w"], p["gn_b"] = _vec16(reader, f"{att}.ln_x.weight"), _vec16(reader, f"{att}.ln_x.bias")
        q = quant.get(layer, "none")
        p["w_k"] = _mat(reader, f"{att}.key.weight", weights_f16, q)
        p["w_v"] = _mat(reader, f"{att}.value.weight", weights_f16, q)
        p["w_r"] = _mat(reader, f"{att}.receptance.weight", weights_f16, q)
        p["w_o"] = _mat(reader, f"{att}.output.weight", weights_f16, q, discount)
        p["ffn_x_k"] = _vec16(reader, f"{ffn}.x_k")
        p["ffn_w_k"] = _mat(reader, f"{ffn}.key.weight", weights_f16, q)
        p["ffn_w_v"] = _mat(reader, f"{ffn}.value.weight", weights_f16, q, discount)
        if discount != 1.0 and q == "none":                                              # load_matrix_f16_discount, loader.rs:643-652
            p["w_o"] = r16(discount * p["w_o"]) if weights_f16 else (discount * p["w_o"]).astype(np.float32)
            p["ffn_w_v"] = r16(discount * p["ffn_w_v"]) if weights_f16 else (discount * p["ffn_w_v"]).astype(np.float32)
        layers.append(V7Layer(p))
    return V7Model(info, emb, ln0, ln_out, head, layers, rescale)


class V7State:
    """v7.rs:146-208: per layer f32 [B][S+2 rows][D]; row 0 att shift, 1..S wkv, S+1 ffn shift."""

    def __init__(self, info: ModelInfo, num_batch: int):
        self.info = info
        self.data = np.zeros((info.num_layer, num_batch, info.head_size + 2, info.num_emb), dtype=np.float32)

    def back(self, batch: int) -> np.ndarray:
        """State::back -> [L, S+2, D] (reference shape [D, S+2, L, 1], x fastest)."""
        return self.data[:, batch].copy()

    def load(self, tensor: np.ndarray, batch: int) -> None:
        assert tensor.shape == self.data[:, batch].shape
        self.data[:, batch] = tensor


class V7Runtime:
    """v7::Bundle + dispatch: runs one chunk (``infer_chunk``) exactly as one ``RnnJob``."""

    def __init__(self, model: V7Model, num_batch: int, act_f16: bool = True):
        self.model = model
        self.state = V7State(model.info, num_batch)
        self.rnd = r16 if act_f16 else _id
        self.trace: Optional[Dict[str, np.ndarray]] = None     # filled like examples/inspect.rs:190-248

    # matmul_op: y[T, M] = act(x[T, K] @ W[M, K]^T), f32 accumulate, output rounded to the buffer dtype
    def _mm(self, w, x, act="none"):
        return self.rnd(ACT[act](np.matmul(x.astype(np.float32), w.T.astype(np.float32)).astype(np.float32)))

    def _tr(self, layer, name, val):
        if self.trace is not None:
            self.trace[f"{layer}_{name}"] = np.array(val, copy=True)

    def _token_shift(self, x, prev_state_row, mu, firsts, batches):
        """token_shift.wgsl REVERSED: out = mix(x_t, prev, mu); prev = state row (f32) on a
        sequence's first token of the chunk, else x_{t-1}."""
        prev = np.empty_like(x)
        prev[1:] = x[:-1]
        for t in np.nonzero(firsts)[0]:
            prev[t] = prev_state_row[batches[t]]
        return self.rnd(mix(x, prev, mu[None, :]))

    def infer_chunk(self, chunk_tokens: List[List[int]], headers: List[int]) -> np.ndarray:
        """chunk_tokens[b] = token ids of batch b in this chunk; headers = stacked row indices fed to
        the head (RnnRedirect.headers).  Returns f32 logits [len(headers), num_vocab]."""
        m, info, rnd = self.model, self.model.info, self.rnd
        D, S, H = info.num_emb, info.head_size, info.num_head
        lens = [len(c) for c in chunk_tokens]
        T = sum(lens)
        if T == 0:
            return np.zeros((0, info.num_vocab), np.float32)
        cursors = stack_cursors(lens)
        batches = np.array([c & 0xFF for c in cursors])
        starts = np.array([(c >> 8) & 0xFFFF for c in cursors])
        clens = np.array([(c >> 24) & 0xFF for c in cursors])
        idx = np.arange(T)
        firsts = idx == starts
        lasts = (idx - starts + 1) == clens
        tokens = np.concatenate([np.asarray(c, dtype=np.int64) for c in chunk_tokens if len(c)])

        # embed (v7.rs:438-474, 649-659): CPU gather of f16 rows, LN(ln0) in place, blit to x
        inp = m.emb[tokens]
        # Runtime::input is TensorGpu<f16> whatever F is (v7.rs:283): LN(ln0) runs in place on it, then blit -> x (F)
        x = rnd(r16(layer_norm(inp, m.ln0[0], m.ln0[1], LN_EPS)))
        self._tr("emb", "x", x)
        v0 = None
        for li, layer in enumerate(m.layers):
            p = layer.p
            st = self.state.data[li]                                     # [B, S+2, D]
            # 1-2
            att_x = rnd(layer_norm(x, p["ln1_w"], p["ln1_b"], LN_EPS))
            self._tr(li, "att_x_ln", att_x)
            # 3: six token shifts against state row 0
            row0 = st[:, 0, :]
            sx = {n: self._token_shift(att_x, row0, p[f"x_{n}"], firsts, batches) for n in "rwkvag"}
            # 4
            r = self._mm(p["w_r"], sx["r"])
            k = self._mm(p["w_k"], sx["k"])
            v = self._mm(p["w_v"], sx["v"])
            self._tr(li, "k_raw", k); self._tr(li, "v_raw", v)
            for n in "rwkvag":
                self._tr(li, f"att_{n}x", sx[n])
            # 5
            aux_w = self._mm(p["w1"], sx["w"], "tanh")
            self._tr(li, "aux_w", aux_w)
            w = self._mm(p["w2"], aux_w)
            w = rnd(p["w0"][None, :] + w)
            # 6
            aux_a = self._mm(p["a1"], sx["a"])
            self._tr(li, "aux_a", aux_a)
            a = self._mm(p["a2"], aux_a)
            a = rnd(sigmoid(p["a0"][None, :] + a))
            # 7
            aux_g = self._mm(p["g1"], sx["g"], "sigmoid")
            self._tr(li, "aux_g", aux_g)
            g = self._mm(p["g2"], aux_g)
            # 8
            kk = rnd(p["k_k"][None, :] * k)
            kk = rnd(l2_norm(kk.reshape(T, H, S), L2_EPS).reshape(T, D))
            # 9
            k = rnd(k * (np.float32(1.0) + (a - np.float32(1.0)) * p["k_a"][None, :]))
            # 10
            if li == 0:
                v0 = v.copy()
            else:
                aux_v = self._mm(p["v1"], sx["v"])
                self._tr(li, "aux_v", aux_v)
                vv = self._mm(p["v2"], aux_v)
                vv = rnd(sigmoid(p["v0"][None, :] + vv))
                self._tr(li, "vv", vv)
                v = rnd(mix(v, v0, vv))                                   # lerp REVERSED: y <- mix(y, x, f)
            self._tr(li, "r", r); self._tr(li, "w", w); self._tr(li, "k", k); self._tr(li, "v", v)
            self._tr(li, "a", a); self._tr(li, "g", g); self._tr(li, "kk", kk)
            # 12: WKV7, sequential over tokens (time_mix_v7.wgsl:143-221)
            y = np.empty((T, D), np.float32)
            ww = np.exp(W_SCALE * sigmoid(w), dtype=np.float32)
            aa = (-kk).astype(np.float32)
            bb = (kk * a).astype(np.float32)
            for t in range(T):
                b = batches[t]
                if lasts[t]:
                    st[b, 0, :] = att_x[starts[t] + clens[t] - 1]
                Sm = st[b, 1:S + 1, :].reshape(S, H, S).transpose(1, 0, 2)      # [H, j, i] view
                rt, wt, kt, vt = (z[t].reshape(H, S) for z in (r, ww, k, v))
                at, bt = aa[t].reshape(H, S), bb[t].reshape(H, S)
                sa = np.einsum("hj,hji->hi", at, Sm).astype(np.float32)
                Sn = (Sm * wt[:, :, None] + kt[:, :, None] * vt[:, None, :] + sa[:, None, :] * bt[:, :, None]).astype(np.float32)
                y[t] = np.einsum("hj,hji->hi", rt, Sn).astype(np.float32).reshape(D)
                st[b, 1:S + 1, :] = Sn.transpose(1, 0, 2).reshape(S, D)
            att_x = rnd(y)
            self._tr(li, "wkv", att_x)
            # 13: group norm per head
            gn = layer_norm(att_x.reshape(T, H, S), p["gn_w"].reshape(H, S)[None], p["gn_b"].reshape(H, S)[None], GN_EPS)
            att_x = rnd(gn.reshape(T, D))
            # 14: time_first
            xx = (p["r_k"][None, :] * k * r).reshape(T, H, S).astype(np.float64).sum(axis=-1).astype(np.float32)
            att_x = rnd(att_x + (xx[:, :, None] * v.reshape(T, H, S)).reshape(T, D))
            # 15
            att_x = rnd(g * att_x)
            self._tr(li, "att_x", att_x)
            # 16
            o = self._mm(p["w_o"], att_x)
            self._tr(li, "att_o", o)
            x = rnd(o + x)
            self._tr(li, "x_att", x)
            # 17
            ffn_x = rnd(layer_norm(x, p["ln2_w"], p["ln2_b"], LN_EPS))
            # 18
            kx = self._token_shift(ffn_x, st[:, S + 1, :], p["ffn_x_k"], firsts, batches)
            # 19-20
            fk = self._mm(p["ffn_w_k"], kx, "squared_relu")
            fv = self._mm(p["ffn_w_v"], fk)
            self._tr(li, "ffn_x", ffn_x); self._tr(li, "ffn_kx", kx); self._tr(li, "ffn_k", fk); self._tr(li, "ffn_v", fv)
            # 21: channel_mix_v7 saves the ffn shift state of each sequence's last token
            for t in np.nonzero(lasts)[0]:
                st[batches[t], S + 1, :] = ffn_x[t]
            # 22
            x = rnd(fv + x)
            # 23
            if (li + 1) % m.rescale == 0:
                x = rnd(np.float32(0.5) * x)
            self._tr(li, "x", x)
        # head
        if not headers:
            return np.zeros((0, info.num_vocab), np.float32)
        hx = x[np.asarray(headers)]
        hx = rnd(layer_norm(hx, m.ln_out[0], m.ln_out[1], LN_EPS))
        logits = np.matmul(hx, m.head.T.astype(np.float32)).astype(np.float32)   # head_o is f32
        return logits


def read_state(reader: GgufReader, info=None) -> np.ndarray:
    """read_state (v7.rs:1229-1262, same in v6.rs:1176-1208): the pre-trained initial state of a state-tuned model.
    `blocks.N.att.time_state` is stored [H, S, S]; after the reference's transpose + blit, state row 1 + j of channel
    h*S + c holds time_state[h][j][c]; rows 0 and S+1 (the token-shift rows) stay zero.  -> [L, S+2, D] f32."""
    info = info or loader_info(reader)
    S, H, D = info.head_size, info.num_head, info.num_emb
    out = np.zeros((info.num_layer, S + 2, D), np.float32)
    for layer in range(info.num_layer):
        _, shape, vals = reader.tensor(f"blocks.{layer}.att.time_state")
        ts = r16(vals).reshape(H, S, S)                      # load_matrix_f16: f16 values
        out[layer, 1:S + 1, :] = ts.transpose(1, 0, 2).reshape(S, D)      # [j][h][c] -> row j, channel h*S + c
    return out

