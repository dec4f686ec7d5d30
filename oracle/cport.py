"""ctypes glue for oracle/c/wrk_oracle.c (test infrastructure only): builds the C model from a
GGUF byte string using the oracle's reader, runs decode steps, and times them for bench.py's
`cpu_baseline` leg."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import dequant as dq
from .gguf import GgufReader
from .rwkv7 import loader_info

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "libwrk_oracle.so")
if not os.path.exists(_LIB):
    raise ImportError(f"{_LIB} not built (make -C oracle/c)")
lib = C.CDLL(_LIB)

_H = C.c_void_p
_LAYER_FIELDS = ["ln1_w", "ln1_b", "ln2_w", "ln2_b", "x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "a0", "v0",
                 "w1", "w2", "a1", "a2", "g1", "g2", "v1", "v2", "r_k", "k_k", "k_a", "gn_w", "gn_b",
                 "w_k", "w_v", "w_r", "w_o", "ffn_x_k", "ffn_w_k", "ffn_w_v"]


class OrcLayer(C.Structure):
    _fields_ = [(n, _H) for n in _LAYER_FIELDS]


class OrcModel(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_layer", "num_emb", "num_hidden", "num_vocab", "num_head", "lora_w", "lora_a", "lora_g", "lora_v")] + \
               [(n, _H) for n in ("emb", "ln0_w", "ln0_b", "ln_out_w", "ln_out_b", "head")] + [("layers", C.POINTER(OrcLayer))]


lib.orc_dequant_f16.restype = C.c_int
lib.orc_dequant_f16.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
lib.orc_v7_decode.restype = None
lib.orc_v7_decode.argtypes = [C.POINTER(OrcModel), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
lib.orc_num_threads.restype = C.c_int
lib.orc_set_threads.restype = None
lib.orc_set_threads.argtypes = [C.c_int]


def usable_cpus() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box gives a
    16-CPU share of a much larger host; OpenMP's default of one thread per host core oversubscribes it)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("WRK_ORACLE_THREADS", "16"))))


class CModel:
    """The reference-effective (all-f16 weights) model in C."""

    def __init__(self, gguf_bytes):
        self.reader = GgufReader(gguf_bytes)
        self.info = loader_info(self.reader)
        self._keep = []
        info = self.info
        m = OrcModel(info.num_layer, info.num_emb, info.num_hidden, info.num_vocab, info.num_head,
                     info.custom["w"], info.custom["a"], info.custom["g"], info.custom["v"])
        m.emb, m.head = self._t("emb.weight"), self._t("head.weight")
        m.ln0_w, m.ln0_b = self._t("blocks.0.ln0.weight"), self._t("blocks.0.ln0.bias")
        m.ln_out_w, m.ln_out_b = self._t("ln_out.weight"), self._t("ln_out.bias")
        self.layers = (OrcLayer * info.num_layer)()
        names = {"ln1_w": "ln1.weight", "ln1_b": "ln1.bias", "ln2_w": "ln2.weight", "ln2_b": "ln2.bias", "gn_w": "att.ln_x.weight",
                 "gn_b": "att.ln_x.bias", "w_k": "att.key.weight", "w_v": "att.value.weight", "w_r": "att.receptance.weight",
                 "w_o": "att.output.weight", "ffn_x_k": "ffn.x_k", "ffn_w_k": "ffn.key.weight", "ffn_w_v": "ffn.value.weight"}
        for l in range(info.num_layer):
            for f in _LAYER_FIELDS:
                if l == 0 and f in ("v0", "v1", "v2"):
                    continue
                setattr(self.layers[l], f, self._t(f"blocks.{l}." + names.get(f, "att." + f)))
        m.layers = self.layers
        self.m = m
        S = info.head_size
        self.state = np.zeros((info.num_layer, S + 2, info.num_emb), np.float32)
        self.scratch = np.zeros(24 * info.num_emb + info.num_hidden + 1024, np.float32)
        self.logits = np.zeros(info.num_vocab, np.float32)

    def _t(self, name: str):
        raw = None if self.reader._fused_slice(name) is not None else self.reader.raw_tensor(name)
        if raw is not None and raw[0] in dq.GGML_TYPE_ID and raw[0] != "BF16":
            tn, data = raw
            n = self.reader._info(name).num_elements()
            out = np.empty(n, np.uint16)
            src = np.ascontiguousarray(data)
            rc = lib.orc_dequant_f16(dq.GGML_TYPE_ID[tn], src.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p))
            assert rc == 0, (name, tn)
        else:   # virtual slices of time_maa
            out = self.reader.tensor(name)[2].astype(np.float16).view(np.uint16).copy()
        self._keep.append(out)
        return out.ctypes.data_as(C.c_void_p).value

    def decode(self, token: int) -> np.ndarray:
        lib.orc_v7_decode(C.byref(self.m), self.state.ctypes.data_as(C.c_void_p), token, self.logits.ctypes.data_as(C.c_void_p),
                          self.scratch.ctypes.data_as(C.c_void_p))
        return self.logits


def time_decode(gguf_bytes, first_token: int, seconds: float = 15.0):
    """bench.py cpu_baseline: greedy decode for about `seconds` of CPU time on the host cores."""
    lib.orc_set_threads(usable_cpus())
    model = CModel(gguf_bytes)
    tok = int(first_token)
    tok = int(model.decode(tok).argmax())          # warm-up (page in the weights)
    n, t0 = 0, time.perf_counter()
    while True:
        tok = int(model.decode(tok).argmax())
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 4096:
            break
    return {"value": round(n / dt, 3), "unit": "tokens/s", "cores": int(lib.orc_num_threads()), "kind": "port",
            "sample": f"{n} greedy decode tokens of the same model in {dt:.1f} s; oracle/c restatement of the reference's effective "
                      f"path (all matrices dequantised to f16 at load, f16 activations, f32 accumulate), OpenMP over rows"}
