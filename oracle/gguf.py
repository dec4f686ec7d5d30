"""GGUF v2/v3 container: reader (restating the reference) and a writer for fixtures.

Oracle / test infrastructure only.

Reader restates ``/root/reference/src/runtime/gguf.rs``:
  * header + metadata + tensor-info parse ............ :1331-1402, cursor :1419-1538
  * data offset alignment (``general.alignment``/32) .. :1358-1361, :1388-1389, :1415-1417
  * RWKV name map GGUF -> safetensors names ........... :1160-1329
  * fused ``time_mix_lerp_fused`` slicing ............. :1545-1571, :1651-1679
  * ``r_k`` 1-D -> [num_head, head_size] .............. :1623-1640, :1741-1760
  * shape reversal to safetensors convention ......... :1642-1647
  * ``tensor()`` (CPU dequant to f16) ................. :1650-1773
  * ``quantized_tensor()`` gate ....................... :1775-1794
The writer has no reference counterpart (the reference only reads GGUF).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import dequant as dq

GGUF_MAGIC = 0x46554747
GGUF_DEFAULT_ALIGNMENT = 32


class GgufError(Exception):
    pass


def align_offset(offset: int, alignment: int) -> int:
    """gguf.rs:1415-1417."""
    return offset + (alignment - (offset % alignment)) % alignment


# --------------------------------------------------------------------------- name map
_TOP = {
    "token_embd.weight": "emb.weight",
    "output_norm.weight": "ln_out.weight",
    "output_norm.bias": "ln_out.bias",
    "output.weight": "head.weight",
    "token_embd_norm.weight": "blocks.0.ln0.weight",
    "token_embd_norm.bias": "blocks.0.ln0.bias",
}

_BLK = {
    "attn_norm.weight": "ln1.weight", "attn_norm.bias": "ln1.bias",
    "attn_norm_2.weight": "ln2.weight", "attn_norm_2.bias": "ln2.bias",
    "ffn_norm.weight": "ln2.weight", "ffn_norm.bias": "ln2.bias",
    "attn_k.weight": "att.key.weight", "attn_v.weight": "att.value.weight",
    "attn_r.weight": "att.receptance.weight", "attn_g.weight": "att.gate.weight",
    "attn_output.weight": "att.output.weight",
    "attn_time_decay": "att.time_decay", "attn_time_first": "att.time_first",
    "attn_time_mix_k": "att.time_mix_k", "attn_time_mix_v": "att.time_mix_v",
    "attn_time_mix_r": "att.time_mix_r", "attn_time_mix_g": "att.time_mix_g",
    "attn_time_mix_x": "att.time_mix_x", "attn_time_mix_w": "att.time_mix_w",
    "attn_time_mix_w1": "att.time_mix_w1", "attn_time_mix_w2": "att.time_mix_w2",
    "attn_time_decay_w1": "att.time_decay_w1", "attn_time_decay_w2": "att.time_decay_w2",
    "time_maa_w1": "att.time_mix_w1", "time_maa_w2": "att.time_mix_w2",
    "time_decay_w1": "att.time_decay_w1", "time_decay_w2": "att.time_decay_w2",
    "attn_ln_x.weight": "att.ln_x.weight", "attn_ln_x.bias": "att.ln_x.bias",
    "attn_time_state": "att.time_state",
    "ffn_k.weight": "ffn.key.weight", "ffn_v.weight": "ffn.value.weight",
    "ffn_r.weight": "ffn.receptance.weight",
    "ffn_time_mix_k": "ffn.time_mix_k", "ffn_time_mix_r": "ffn.time_mix_r",
    "ffn.key.weight": "ffn.key.weight", "ffn.value.weight": "ffn.value.weight",
    "ffn.receptance.weight": "ffn.receptance.weight",
    "channel_mix_key.weight": "ffn.key.weight", "channel_mix_value.weight": "ffn.value.weight",
    "channel_mix_receptance.weight": "ffn.receptance.weight",
    "channel_mix_lerp_k.weight": "ffn.x_k",
    "time_mix_key.weight": "att.key.weight", "time_mix_value.weight": "att.value.weight",
    "time_mix_receptance.weight": "att.receptance.weight", "time_mix_gate.weight": "att.gate.weight",
    "time_mix_output.weight": "att.output.weight",
    "time_mix_lerp_fused.weight": "att.time_maa",
    "time_mix_ln.weight": "att.ln_x.weight", "time_mix_ln.bias": "att.ln_x.bias",
    "ffn_x_k": "ffn.x_k",
}
for _v in ("w0", "w1", "w2", "a0", "a1", "a2", "g1", "g2", "v0", "v1", "v2", "r_k", "k_k", "k_a"):
    _BLK[f"time_mix_{_v}.weight"] = f"att.{_v}"
    _BLK[f"attn_{_v}"] = f"att.{_v}"
    _BLK[f"att_{_v}"] = f"att.{_v}"
for _v in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g"):
    _BLK[f"attn_{_v}"] = f"att.{_v}"
    _BLK[f"att_{_v}"] = f"att.{_v}"

_FUSED_SUFFIX = [(".att.x_r", 0), (".att.x_w", 1), (".att.x_k", 2), (".att.x_v", 3), (".att.x_a", 4), (".att.x_g", 5)]


# llama.cpp's RWKV-6 names, as the reference's own converter emits them (assets/scripts/convert_hf_to_gguf.py:455-525); the
# reference's map lacks them (gguf.rs:1212-1229) and maps time_mix_w1.weight to the V7 tensor (gguf.rs:1261): SURVEY H6.
# Applied when general.architecture == "rwkv6", before the shared rules.  Parity unpinned (gguf-py absent).
_BLK_V6 = {
    "time_mix_lerp_x.weight": "att.time_mix_x", "time_mix_lerp_w.weight": "att.time_mix_w", "time_mix_lerp_k.weight": "att.time_mix_k",
    "time_mix_lerp_v.weight": "att.time_mix_v", "time_mix_lerp_r.weight": "att.time_mix_r", "time_mix_lerp_g.weight": "att.time_mix_g",
    "time_mix_w1.weight": "att.time_mix_w1", "time_mix_w2.weight": "att.time_mix_w2",
    "time_mix_decay.weight": "att.time_decay", "time_mix_decay_w1.weight": "att.time_decay_w1", "time_mix_decay_w2.weight": "att.time_decay_w2",
    "time_mix_first.weight": "att.time_first",
    "channel_mix_lerp_k.weight": "ffn.time_mix_k", "channel_mix_lerp_r.weight": "ffn.time_mix_r",
}


def gguf_to_safetensors_name(name: str, arch_v6: bool = False) -> Optional[str]:
    """gguf.rs:1173-1329 (+ the llama.cpp RWKV-6 names when the file says rwkv6)."""
    if name in _TOP:
        return _TOP[name]
    if name.startswith("blk."):
        rest = name[4:]
        dot = rest.find(".")
        if dot >= 0:
            blk, rem = rest[:dot], rest[dot + 1:]
            if arch_v6 and rem in _BLK_V6:
                return f"blocks.{blk}.{_BLK_V6[rem]}"
            if rem in _BLK:
                return f"blocks.{blk}.{_BLK[rem]}"
    return None


# --------------------------------------------------------------------------- reader
class TensorInfo:
    def __init__(self, name: str, dimensions: List[int], type_id: int, offset: int):
        self.name, self.dimensions, self.type_id, self.offset = name, dimensions, type_id, offset

    @property
    def type_name(self) -> str:
        return dq.GGML_TYPE_NAME.get(self.type_id, f"Unknown({self.type_id})")

    def num_elements(self) -> int:
        n = 1
        for d in self.dimensions:
            n *= d
        return n

    def data_size(self) -> int:
        return dq.data_size(self.type_name, self.num_elements())


class _Cursor:
    def __init__(self, data: memoryview):
        self.data, self.pos = data, 0

    def take(self, n: int) -> bytes:
        if len(self.data) - self.pos < n:
            raise GgufError("unexpected end of file")
        b = bytes(self.data[self.pos:self.pos + n])
        self.pos += n
        return b

    def u8(self): return self.take(1)[0]
    def u32(self): return struct.unpack("<I", self.take(4))[0]
    def u64(self): return struct.unpack("<Q", self.take(8))[0]

    def string(self) -> str:
        n = self.u64()
        try:
            return self.take(n).decode("utf-8")
        except UnicodeDecodeError as e:
            raise GgufError("invalid utf-8 string") from e

    _SCALAR = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 10: "<Q", 11: "<q", 12: "<d"}

    def value_of_type(self, t: int):
        if t in self._SCALAR:
            f = self._SCALAR[t]
            return struct.unpack(f, self.take(struct.calcsize(f)))[0]
        if t == 7:
            return self.u8() != 0
        if t == 8:
            return self.string()
        if t == 9:
            at = self.u32()
            n = self.u64()
            return [self.value_of_type(at) for _ in range(n)]
        raise GgufError(f"invalid metadata value type: {t}")


class GgufReader:
    """Mirror of ``GgufReader`` + its ``Reader`` impl (names/contains/shape/tensor/quantized_tensor)."""

    def __init__(self, data):
        self.data = memoryview(data).cast("B") if not isinstance(data, memoryview) else data
        c = _Cursor(self.data)
        magic = c.u32()
        if magic != GGUF_MAGIC:
            raise GgufError(f"invalid magic number: expected 0x{GGUF_MAGIC:08X}, got 0x{magic:08X}")
        self.version = c.u32()
        if self.version < 2 or self.version > 3:
            raise GgufError(f"unsupported version: {self.version} (supported: 3)")
        self.tensor_count = c.u64()
        n_kv = c.u64()
        self.metadata: Dict[str, object] = {}
        for _ in range(n_kv):
            k = c.string()
            self.metadata[k] = c.value_of_type(c.u32())
        al = self.metadata.get("general.alignment")
        alignment = int(al) if isinstance(al, int) and not isinstance(al, bool) else GGUF_DEFAULT_ALIGNMENT
        self.tensors: Dict[str, TensorInfo] = {}
        for _ in range(self.tensor_count):
            name = c.string()
            nd = c.u32()
            dims = [c.u64() for _ in range(nd)]
            t = c.u32()
            off = c.u64()
            self.tensors[name] = TensorInfo(name, dims, t, off)
        self.tensor_data_offset = align_offset(c.pos, alignment)
        self.name_map: Dict[str, str] = {}
        arch_v6 = self.metadata.get("general.architecture") == "rwkv6"
        for g in self.tensors:                    # build_rwkv_name_map, gguf.rs:1160-1171
            s = gguf_to_safetensors_name(g, arch_v6)
            if s is not None:
                self.name_map[s] = g
            self.name_map[g] = g

    # -- helpers
    def get_tensor_data(self, info: TensorInfo) -> np.ndarray:
        start = self.tensor_data_offset + info.offset
        return np.frombuffer(self.data, dtype=np.uint8, count=info.data_size(), offset=start)

    def _head_size(self) -> Optional[int]:
        for k in ("rwkv7.wkv.head_size", "rwkv6.wkv.head_size"):
            v = self.metadata.get(k)
            if isinstance(v, int) and not isinstance(v, bool):
                return int(v)
        return None

    def _fused_slice(self, name: str) -> Optional[Tuple[str, int]]:
        if not name.startswith("blocks.") or ".att.x_" not in name:
            return None
        for suffix, idx in _FUSED_SUFFIX:
            if name.endswith(suffix):
                fused = name[: -len(suffix)] + ".att.time_maa"
                if fused in self.name_map:
                    return fused, idx
        return None

    def _info(self, name: str) -> TensorInfo:
        g = self.name_map.get(name)
        if g is None or g not in self.tensors:
            raise GgufError(f"tensor not found: {name}")
        return self.tensors[g]

    # -- Reader trait
    def names(self) -> List[str]:
        names = list(self.name_map.keys())
        for key in list(self.name_map.keys()):
            if key.endswith(".att.time_maa"):
                prefix = key[: -len(".att.time_maa")]
                for sfx in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g"):
                    v = f"{prefix}.att.{sfx}"
                    if v not in self.name_map:
                        names.append(v)
        return names

    def contains(self, name: str) -> bool:
        return name in self.name_map or self._fused_slice(name) is not None

    def shape(self, name: str) -> List[int]:
        fs = self._fused_slice(name)
        if fs is not None:
            return [int(self._info(fs[0]).dimensions[0])]
        info = self._info(name)
        shape = [int(d) for d in info.dimensions]
        if len(shape) == 1 and name.endswith(".att.r_k"):
            hs = self._head_size()
            if hs:
                return [shape[0] // hs, hs]
        if len(shape) > 1:
            shape.reverse()
        return shape

    def tensor(self, name: str):
        """-> (dtype_name, shape(safetensors order), values float32 with the reference's rounding).

        dtype_name is 'F16' for dequantised tensors, else the stored float type.
        """
        fs = self._fused_slice(name)
        if fs is not None:
            info = self._info(fs[0])
            if info.type_name not in ("F32", "F16"):
                raise GgufError(f"unsupported tensor type: {info.type_name}")
            emb = int(info.dimensions[0])
            es = dq.BLOCK_BYTES[info.type_name]
            raw = self.get_tensor_data(info)[fs[1] * emb * es:(fs[1] + 1) * emb * es]
            dt = "<f4" if info.type_name == "F32" else "<f2"
            return info.type_name, [emb, 1], np.frombuffer(raw.tobytes(), dtype=dt).astype(np.float32)
        info = self._info(name)
        n = info.num_elements()
        shape = [int(d) for d in info.dimensions]
        tn = info.type_name
        if tn in dq.DEQUANT:
            raw = self.get_tensor_data(info)
            vals = dq.DEQUANT[tn](raw, n, round_f16=True)
            if vals.size < n:                      # gguf.rs:1717-1724
                vals = np.concatenate([vals, np.zeros(n - vals.size, np.float32)])
            vals = vals[:n]
            if len(shape) > 1:
                shape.reverse()
            else:
                shape.append(1)
            return "F16", shape, vals
        if tn not in ("F32", "F16"):
            raise GgufError(f"unsupported tensor type: {tn}")
        dt = "<f4" if tn == "F32" else "<f2"
        vals = np.frombuffer(self.get_tensor_data(info).tobytes(), dtype=dt).astype(np.float32)
        if len(shape) == 1 and name.endswith(".att.r_k"):
            hs = self._head_size()
            if hs:
                return tn, [shape[0] // hs, hs], vals
        if len(shape) == 1:
            shape.append(1)
        else:
            shape.reverse()
        return tn, shape, vals

    def quantized_tensor(self, name: str):
        """gguf.rs:1775-1794: raw blocks only for Q8_0 / Q4_0 (K-quants gated off at HEAD)."""
        if self._fused_slice(name) is not None:
            return None
        g = self.name_map.get(name)
        if g is None:
            return None
        info = self.tensors[g]
        if info.type_name in ("Q8_0", "Q4_0"):
            return info.type_id, self.get_tensor_data(info)
        return None

    def raw_tensor(self, name: str):
        """NOT in the reference at HEAD: raw blocks for any type (the build's inline-dequant path,
        i.e. what ``quantized_tensor`` would return with the K-quant gate lifted)."""
        if self._fused_slice(name) is not None:
            return None
        g = self.name_map.get(name)
        if g is None:
            return None
        info = self.tensors[g]
        return info.type_name, self.get_tensor_data(info)


# --------------------------------------------------------------------------- writer (fixtures)
_KV_TYPE = {"u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "u64": 10}


def _w_str(s: str) -> bytes:
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def write_gguf(metadata: List[Tuple[str, str, object]], tensors: List[Tuple[str, List[int], str, np.ndarray]],
               alignment: int = GGUF_DEFAULT_ALIGNMENT, version: int = 3) -> bytes:
    """metadata: (key, kind, value); tensors: (name, ggml dims [fastest first], type_name, raw uint8)."""
    out = bytearray()
    out += struct.pack("<IIQQ", GGUF_MAGIC, version, len(tensors), len(metadata))
    for key, kind, val in metadata:
        out += _w_str(key) + struct.pack("<I", _KV_TYPE[kind])
        if kind == "str":
            out += _w_str(val)
        elif kind == "bool":
            out += struct.pack("<B", 1 if val else 0)
        else:
            out += struct.pack({"u32": "<I", "i32": "<i", "f32": "<f", "u64": "<Q"}[kind], val)
    offset = 0
    offs = []
    for name, dims, tn, raw in tensors:
        n = 1
        for d in dims:
            n *= d
        assert raw.dtype == np.uint8 and raw.size == dq.data_size(tn, n), (name, raw.size, dq.data_size(tn, n))
        offs.append(offset)
        out += _w_str(name) + struct.pack("<I", len(dims))
        for d in dims:
            out += struct.pack("<Q", d)
        out += struct.pack("<IQ", dq.GGML_TYPE_ID[tn], offset)
        offset = align_offset(offset + raw.size, alignment)
    pad = align_offset(len(out), alignment) - len(out)
    out += b"\0" * pad
    base = len(out)
    for (name, dims, tn, raw), off in zip(tensors, offs):
        cur = len(out) - base
        out += b"\0" * (off - cur)
        out += raw.tobytes()
    return bytes(out)
