"""ctypes binding of libwrk_hip.so (include/wrk_hip.h) and libwrk_runtime.so (include/wrk_runtime.h).

This is the harness-side mirror of the reference's public API for the hot path
(`Context`, `TensorOp::*`, `Matrix`, `GgufReader`, `Loader::info`, `ModelBuilder::build_v7`,
`v7::Bundle`, `RnnInput`, `runtime.infer`).  All compute happens in the HIP library; there is NO
CPU fallback: importing this module raises if the libraries are not built, and creating a `Context`
raises if no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.environ.get("WRK_LIB_DIR") or os.path.join(os.path.dirname(_HERE), "lib")   # WRK_LIB_DIR: e.g. the `make TIMING=1` build
LIB_HIP = os.path.join(_LIBDIR, "libwrk_hip.so")
LIB_RT = os.path.join(_LIBDIR, "libwrk_runtime.so")


class WrkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{STATUS.get(code, code)}] {msg}")
        self.code = code


OK, E_ARG, E_OOM, E_HIP, E_UNSUPPORTED = 0, 1, 2, 3, 4
STATUS = {0: "WRK_OK", 1: "WRK_E_ARG", 2: "WRK_E_OOM", 3: "WRK_E_HIP", 4: "WRK_E_UNSUPPORTED"}
F16, F32, U8, U32 = 0, 1, 2, 3
ACT = {"none": 0, "squared_relu": 1, "tanh": 2, "stable_exp": 3, "opposite_exp": 4, "softplus": 5, "sigmoid": 6, "silu": 7}
MAT = {"F32": 0, "F16": 1, "Q8_0": 8, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14, "INT8": 100, "NF4": 101}
MATRIX_EXACT, MATRIX_ROUND_F16 = 0, 1
WEIGHTS_INLINE, WEIGHTS_INLINE_F16, WEIGHTS_REFERENCE = 0, 1, 2
RNN_NONE, RNN_LAST, RNN_FULL = -1, 0, 1
QUANT_NONE, QUANT_INT8, QUANT_NF4 = 0, 1, 2

if not (os.path.exists(LIB_HIP) and os.path.exists(LIB_RT)):
    raise ImportError(
        f"{LIB_HIP} / {LIB_RT} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the HIP path)")

hip = C.CDLL(LIB_HIP, mode=C.RTLD_GLOBAL)
rt = C.CDLL(LIB_RT, mode=C.RTLD_GLOBAL)


class View(C.Structure):
    _fields_ = [("shape", C.c_uint32 * 4), ("stride", C.c_uint32 * 4), ("offset", C.c_uint32 * 4)]


class TensorDesc(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("dtype", C.c_uint32), ("view", View)]


class ModelInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("version", "num_layer", "num_emb", "num_hidden", "num_vocab", "num_head",
                                           "lora_w", "lora_a", "lora_g", "lora_v")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class BuildOptions(C.Structure):
    _fields_ = [("rescale", C.c_uint32), ("weights", C.c_uint32), ("quant", C.POINTER(C.c_uint8)), ("num_quant", C.c_uint32)]


_P = C.c_void_p
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_TP = C.POINTER(TensorDesc)

# name -> (restype, argtypes); every symbol declared in include/*.h
HIP_SYMBOLS = {
    "wrk_abi_version": (C.c_int32, []),
    "wrk_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(_P)]),
    "wrk_ctx_destroy": (C.c_int32, [_P]),
    "wrk_last_error": (C.c_char_p, [_P]),
    "wrk_ctx_sync": (C.c_int32, [_P]),
    "wrk_ctx_stream": (_P, [_P]),
    "wrk_buf_create": (C.c_int32, [_P, C.c_size_t, _P, C.POINTER(_P)]),
    "wrk_buf_retain": (C.c_int32, [_P]),
    "wrk_buf_release": (C.c_int32, [_P]),
    "wrk_buf_size": (C.c_size_t, [_P]),
    "wrk_buf_device_ptr": (_P, [_P]),
    "wrk_buf_write": (C.c_int32, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "wrk_buf_read": (C.c_int32, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "wrk_buf_copy": (C.c_int32, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_size_t]),
    "wrk_capture_begin": (C.c_int32, [_P]),
    "wrk_capture_end": (C.c_int32, [_P, C.POINTER(_P)]),
    "wrk_program_launch": (C.c_int32, [_P, _P]),
    "wrk_program_destroy": (C.c_int32, [_P]),
    "wrk_matrix_create": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "wrk_matrix_quantize": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_float), C.POINTER(_P)]),
    "wrk_matrix_export": (C.c_int32, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "wrk_matrix_set_scale": (C.c_int32, [_P, C.c_float]),
    "wrk_matrix_release": (C.c_int32, [_P]),
    "wrk_matrix_stream_bytes": (C.c_size_t, [_P]),
    "wrk_op_matmul": (C.c_int32, [_P, _P, _TP, _TP, C.c_uint32, C.c_int32, C.c_int32]),
    "wrk_op_layer_norm": (C.c_int32, [_P, _P, _P, _TP, C.c_float]),
    "wrk_op_group_norm": (C.c_int32, [_P, _P, _P, _TP, C.c_float]),
    "wrk_op_l2_norm": (C.c_int32, [_P, _TP, C.c_float]),
    "wrk_op_token_shift": (C.c_int32, [_P, _P, _TP, _TP, _TP, _TP, C.c_int32]),
    "wrk_op_add": (C.c_int32, [_P, _TP, _TP, C.c_uint32, C.c_uint32, C.c_uint32]),
    "wrk_op_mul": (C.c_int32, [_P, _TP, _TP, C.c_uint32, C.c_uint32, C.c_uint32]),
    "wrk_op_lerp": (C.c_int32, [_P, _TP, _TP, _TP, C.c_int32]),
    "wrk_op_blit": (C.c_int32, [_P, _TP, _TP]),
    "wrk_op_affine": (C.c_int32, [_P, _TP, C.c_float, C.c_float]),
    "wrk_op_activate": (C.c_int32, [_P, _TP, C.c_uint32]),
    "wrk_op_control_k_v7": (C.c_int32, [_P, _P, _TP, _TP]),
    "wrk_op_time_mix_v7": (C.c_int32, [_P, _P, _TP, _TP, _TP, _TP, _TP]),
    "wrk_op_time_first_v7": (C.c_int32, [_P, _P, _TP, _TP, _TP]),
    "wrk_op_channel_mix_v7": (C.c_int32, [_P, _P, _TP, _TP, _TP]),
    "wrk_op_softmax": (C.c_int32, [_P, _TP]),
    "wrk_v7_model_create": (C.c_int32, [_P, _P, C.POINTER(_P)]),
    "wrk_v7_model_destroy": (C.c_int32, [_P]),
    "wrk_v7_model_token_bytes": (C.c_size_t, [_P, C.c_uint32]),
    "wrk_v7_state_create": (C.c_int32, [_P, _P, C.c_uint32, C.POINTER(_P)]),
    "wrk_v7_state_destroy": (C.c_int32, [_P]),
    "wrk_v7_state_load": (C.c_int32, [_P, _P, C.c_uint32, _f32p]),
    "wrk_v7_state_read": (C.c_int32, [_P, _P, C.c_uint32, _P]),
    "wrk_v7_state_write": (C.c_int32, [_P, _P, C.c_uint32, _P]),
    "wrk_v7_state_back": (C.c_int32, [_P, _P, C.c_uint32, _f32p]),
    "wrk_v7_infer": (C.c_int32, [_P, _P, _P, _u32p, C.POINTER(C.c_uint16), _u32p, C.c_uint32, _u32p, C.c_uint32, _f32p, _u32p, C.c_uint32]),
    "wrk_v7_model_set_frame_dtype": (C.c_int32, [_P, _P, C.c_uint32]),
    "wrk_v7_model_engine_status": (C.c_int32, [_P, _P, C.c_char_p, C.c_size_t]),
    "wrk_v7_infer_layer": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, _P, _u32p, C.c_uint32, C.c_uint32]),
    "wrk_v7_frame_read": (C.c_int32, [_P, _P, C.c_char_p, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "wrk_v7_generate_greedy": (C.c_int32, [_P, _P, _P, _u32p, C.c_uint32, C.c_uint32, _u32p, _f32p, _f32p, C.c_uint32]),
    "wrk_op_transpose": (C.c_int32, [_P, _TP, _TP]),
    "wrk_op_time_mix_v6": (C.c_int32, [_P, _P, _TP, _P, _TP, _TP, _TP, _TP, _TP]),
    "wrk_op_channel_mix": (C.c_int32, [_P, _P, _TP, _TP, _TP, _TP]),
    "wrk_v6_model_create": (C.c_int32, [_P, _P, C.POINTER(_P)]),
    "wrk_v6_model_destroy": (C.c_int32, [_P]),
    "wrk_v6_model_token_bytes": (C.c_size_t, [_P, C.c_uint32]),
    "wrk_v6_state_create": (C.c_int32, [_P, _P, C.c_uint32, C.POINTER(_P)]),
    "wrk_v6_infer": (C.c_int32, [_P, _P, _P, _u32p, C.POINTER(C.c_uint16), _u32p, C.c_uint32, _u32p, C.c_uint32, _f32p, _u32p, C.c_uint32]),
    "wrk_v6_generate_greedy": (C.c_int32, [_P, _P, _P, _u32p, C.c_uint32, C.c_uint32, _u32p, _f32p, _f32p, C.c_uint32]),
}
RT_SYMBOLS = {
    "wrk_host_last_error": (C.c_char_p, []),
    "wrk_gguf_open": (C.c_int32, [C.c_char_p, C.POINTER(_P)]),
    "wrk_gguf_from_memory": (C.c_int32, [_P, C.c_size_t, C.POINTER(_P)]),
    "wrk_gguf_close": (C.c_int32, [_P]),
    "wrk_gguf_version": (C.c_uint32, [_P]),
    "wrk_gguf_tensor_data_offset": (C.c_uint64, [_P]),
    "wrk_gguf_contains": (C.c_int32, [_P, C.c_char_p]),
    "wrk_gguf_shape": (C.c_int32, [_P, C.c_char_p, C.c_uint32 * 4, _u32p]),
    "wrk_gguf_tensor_f16": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_uint16), C.c_size_t, C.POINTER(C.c_size_t)]),
    "wrk_gguf_raw": (C.c_int32, [_P, C.c_char_p, _u32p, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "wrk_gguf_meta_u64": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_uint64)]),
    "wrk_gguf_read_state": (C.c_int32, [_P, _f32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "wrk_gguf_info": (C.c_int32, [_P, C.POINTER(ModelInfo)]),
    "wrk_quantile_student": (C.c_int32, [C.c_double, _f32p]),
    "wrk_rnn_input_create": (C.c_int32, [C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "wrk_rnn_input_destroy": (C.c_int32, [_P]),
    "wrk_rnn_input_token_chunk_size": (C.c_uint32, [_P]),
    "wrk_rnn_input_append": (C.c_int32, [_P, C.c_uint32, _u32p, C.c_uint32]),
    "wrk_rnn_input_set_option": (C.c_int32, [_P, C.c_uint32, C.c_int32]),
    "wrk_rnn_input_remaining": (C.c_uint32, [_P, C.c_uint32]),
    "wrk_rnn_input_step": (C.c_int32, [_P]),
    "wrk_rnn_iter_create": (C.c_int32, [_P, C.POINTER(_P)]),
    "wrk_rnn_iter_destroy": (C.c_int32, [_P]),
    "wrk_rnn_iter_next": (C.c_int32, [_P, _u32p, _i32p]),
    "wrk_rnn_redirect": (C.c_int32, [_u32p, _i32p, C.c_uint32, _u32p, _u32p, _u32p, _u32p]),
    "wrk_runtime_create": (C.c_int32, [_P, _P, C.POINTER(BuildOptions), C.c_uint32, C.POINTER(_P)]),
    "wrk_runtime_destroy": (C.c_int32, [_P]),
    "wrk_runtime_info": (C.c_int32, [_P, C.POINTER(ModelInfo)]),
    "wrk_runtime_model": (_P, [_P]),
    "wrk_runtime_state": (_P, [_P]),
    "wrk_runtime_model_v6": (_P, [_P]),
    "wrk_runtime_infer": (C.c_int32, [_P, _P, _f32p, C.c_size_t, _u32p, C.c_uint32]),
}
for _lib, _tab in ((hip, HIP_SYMBOLS), (rt, RT_SYMBOLS)):
    for _name, (_res, _args) in _tab.items():
        _f = getattr(_lib, _name)      # AttributeError here == header/library mismatch
        _f.restype = _res
        _f.argtypes = _args


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


# ----------------------------------------------------------------------------- backend objects
class Context:
    """`Context` (src/context.rs:51-64): one HIP device + submission stream."""

    def __init__(self, device: int = 0):
        h = _P()
        rc = hip.wrk_ctx_create(device, C.byref(h))
        if rc != OK:
            raise WrkError(rc, f"wrk_ctx_create(device={device}) failed: no usable HIP device (the HIP path has no CPU fallback)")
        self.h = h

    def check(self, rc: int):
        if rc != OK:
            raise WrkError(rc, (hip.wrk_last_error(self.h) or b"").decode())

    def sync(self):
        self.check(hip.wrk_ctx_sync(self.h))

    def close(self):
        if self.h:
            hip.wrk_ctx_destroy(self.h)
            self.h = None

    # -- programs
    def encode(self, build) -> "Program":
        """`Context::encode(&TensorOp)` (ops.rs:79-143): run `build()` (which calls `TensorOp.*` / `matmul_op`) with the ops
        recorded into a `Program` instead of executed.  Safe from several threads at once: a capture belongs to the calling
        thread and records on a private stream (runtime/mod.rs:139-167 encodes on spawn_blocking workers)."""
        self.check(hip.wrk_capture_begin(self.h))
        try:
            build()
        finally:
            h = _P()
            rc = hip.wrk_capture_end(self.h, C.byref(h))
        self.check(rc)
        return Program(self, h)

    # -- tensors
    def tensor(self, array: np.ndarray, shape: Optional[Sequence[int]] = None) -> "Tensor":
        """context.tensor_from_data: numpy float16/float32 array; `shape` is [x fastest, y, z, w]."""
        a = np.ascontiguousarray(array)
        dt = {np.dtype(np.float16): F16, np.dtype(np.float32): F32}[a.dtype]
        if shape is None:
            shape = list(reversed(a.shape))
        shape = list(shape) + [1] * (4 - len(shape))
        assert int(np.prod(shape)) == a.size
        return Tensor(self, Buffer(self, a.nbytes, a), dt, shape)

    def zeros(self, shape: Sequence[int], dtype=np.float16) -> "Tensor":
        shape = list(shape) + [1] * (4 - len(shape))
        return self.tensor(np.zeros(int(np.prod(shape)), dtype=dtype), shape)

    def buffer(self, array: np.ndarray) -> "Buffer":
        a = np.ascontiguousarray(array)
        return Buffer(self, a.nbytes, a)


class Program:
    """The `Vec<CommandBuffer>` an `RnnJob` keeps (v7.rs:423-432): a captured hipGraph; `launch` == `queue.submit`."""

    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def launch(self):
        self.ctx.check(hip.wrk_program_launch(self.ctx.h, self.h))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.sync()
                hip.wrk_program_destroy(self.h)
        except Exception:
            pass
        self.h = None


class Buffer:
    def __init__(self, ctx: Context, nbytes: int, init: Optional[np.ndarray] = None):
        self.ctx = ctx
        h = _P()
        p = init.ctypes.data_as(_P) if init is not None else None
        ctx.check(hip.wrk_buf_create(ctx.h, nbytes, p, C.byref(h)))
        self.h, self.nbytes = h, nbytes

    def write(self, array: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(array)
        self.ctx.check(hip.wrk_buf_write(self.ctx.h, self.h, offset, a.ctypes.data_as(_P), a.nbytes))

    def read(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        self.ctx.check(hip.wrk_buf_read(self.ctx.h, self.h, offset, out.ctypes.data_as(_P), out.nbytes))
        return out

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                hip.wrk_buf_release(self.h)
        except Exception:
            pass
        self.h = None


class Tensor:
    """TensorGpu / TensorGpuView: buffer + dtype + View{shape, stride, offset}."""

    def __init__(self, ctx, buf: Buffer, dtype: int, shape, stride=None, offset=None):
        self.ctx, self.buf, self.dtype = ctx, buf, dtype
        self.shape = list(shape)
        self.stride = list(stride) if stride is not None else list(shape)
        self.offset = list(offset) if offset is not None else [0, 0, 0, 0]

    def view(self, *slices) -> "Tensor":
        """tensor.view(.., a..b, .., ..): each arg is None (full), int, or (start, end)."""
        shape, off = list(self.shape), list(self.offset)
        for i, s in enumerate(slices):
            if s is None:
                continue
            a, b = (s, s + 1) if isinstance(s, int) else s
            assert 0 <= a <= b <= self.shape[i]
            shape[i], off[i] = b - a, self.offset[i] + a
        return Tensor(self.ctx, self.buf, self.dtype, shape, self.stride, off)

    def reshape(self, shape) -> "Tensor":
        shape = list(shape) + [1] * (4 - len(shape))
        assert int(np.prod(shape)) == int(np.prod(self.shape)) and self.offset == [0, 0, 0, 0] and self.stride == self.shape
        return Tensor(self.ctx, self.buf, self.dtype, shape)

    def desc(self) -> TensorDesc:
        d = TensorDesc()
        d.buf = self.buf.h
        d.dtype = self.dtype
        for i in range(4):
            d.view.shape[i], d.view.stride[i], d.view.offset[i] = self.shape[i], self.stride[i], self.offset[i]
        return d

    def back(self) -> np.ndarray:
        """TensorGpu::back -> numpy array of the PARENT tensor, numpy shape reversed ([w, z, y, x])."""
        n = int(np.prod(self.stride))
        dt = np.float16 if self.dtype == F16 else np.float32
        return self.buf.read(dt, n).reshape(list(reversed(self.stride)))


class Matrix:
    """`enum Matrix` (src/tensor/matrix.rs:82-131) created from raw GGUF blocks or f16/f32 values."""

    def __init__(self, ctx: Context, kind: str, k: int, m: int, data: np.ndarray, flags: int = MATRIX_EXACT):
        a = np.ascontiguousarray(data)
        h = _P()
        ctx.check(hip.wrk_matrix_create(ctx.h, MAT[kind], k, m, a.ctypes.data_as(_P), a.nbytes, flags, C.byref(h)))
        self.ctx, self.h, self.k, self.m, self.kind = ctx, h, k, m, kind

    @classmethod
    def _quant(cls, kind: str, matrix: "Buffer", k: int, m: int, levels=None) -> "Matrix":
        ctx = matrix.ctx
        h = _P()
        lv = None
        if levels is not None:
            lv = np.ascontiguousarray(levels, np.float32)
            assert lv.size == 16
            lv = lv.ctypes.data_as(C.POINTER(C.c_float))
        ctx.check(hip.wrk_matrix_quantize(ctx.h, MAT[kind], k, m, matrix.h, lv, C.byref(h)))
        self = cls.__new__(cls)
        self.ctx, self.h, self.k, self.m, self.kind = ctx, h, k, m, kind
        return self

    @classmethod
    def quant_u8(cls, matrix: "Buffer", k: int, m: int) -> "Matrix":
        """`Matrix::quant_u8` (matrix.rs:211-227): f16 [K, M] buffer -> Int8 matrix, quantised on the device."""
        return cls._quant("INT8", matrix, k, m)

    @classmethod
    def quant_nf4(cls, matrix: "Buffer", k: int, m: int) -> "Matrix":
        """`Matrix::quant_nf4` (matrix.rs:229-249)."""
        return cls._quant("NF4", matrix, k, m)

    @classmethod
    def quant_sf4(cls, matrix: "Buffer", k: int, m: int, levels) -> "Matrix":
        """`Matrix::quant_sf4` (matrix.rs:251-271) with the caller's 16 levels (`Float4Quant::new_student`)."""
        return cls._quant("NF4", matrix, k, m, levels)

    def export(self) -> np.ndarray:
        """Int8 / NF4 planes in `wrk_matrix_create`'s layout (codes ++ side table [++ levels])."""
        n = C.c_size_t()
        self.ctx.check(hip.wrk_matrix_export(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint8)
        self.ctx.check(hip.wrk_matrix_export(self.h, out.ctypes.data_as(_P), out.nbytes, C.byref(n)))
        return out

    def set_scale(self, scale: float):
        """y = act(scale * (W . x)): `load_matrix_discount`'s 2^-k factor without leaving the quantised form."""
        self.ctx.check(hip.wrk_matrix_set_scale(self.h, scale))

    @property
    def stream_bytes(self) -> int:
        return hip.wrk_matrix_stream_bytes(self.h)

    def matmul_op(self, inp: Tensor, out: Tensor, act: str = "none", turbo: bool = False, sparse: bool = False):
        di, do = inp.desc(), out.desc()
        self.ctx.check(hip.wrk_op_matmul(self.ctx.h, self.h, C.byref(di), C.byref(do), ACT[act], int(turbo), int(sparse)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                hip.wrk_matrix_release(self.h)
        except Exception:
            pass
        self.h = None


class TensorOp:
    """Constructors named like `TensorOp::*` (src/tensor/ops.rs); each enqueues on the context."""

    @staticmethod
    def _c(t: Tensor):
        return t.ctx

    @staticmethod
    def layer_norm(w: Buffer, b: Buffer, x: Tensor, eps: float):
        d = x.desc(); x.ctx.check(hip.wrk_op_layer_norm(x.ctx.h, w.h, b.h, C.byref(d), eps))

    @staticmethod
    def group_norm(w: Buffer, b: Buffer, x: Tensor, eps: float):
        d = x.desc(); x.ctx.check(hip.wrk_op_group_norm(x.ctx.h, w.h, b.h, C.byref(d), eps))

    @staticmethod
    def l2_norm(x: Tensor, eps: float):
        d = x.desc(); x.ctx.check(hip.wrk_op_l2_norm(x.ctx.h, C.byref(d), eps))

    @staticmethod
    def token_shift(cursors: Buffer, time_mix: Tensor, state: Tensor, inp: Tensor, out: Tensor, reversed_: bool):
        m, s, i, o = time_mix.desc(), state.desc(), inp.desc(), out.desc()
        inp.ctx.check(hip.wrk_op_token_shift(inp.ctx.h, cursors.h, C.byref(m), C.byref(s), C.byref(i), C.byref(o), int(reversed_)))

    @staticmethod
    def transpose(inp: Tensor, out: Tensor):
        i, o = inp.desc(), out.desc()
        out.ctx.check(hip.wrk_op_transpose(out.ctx.h, C.byref(i), C.byref(o)))

    @staticmethod
    def time_mix_v6(cursors: Buffer, time_decay: Tensor, time_first: Buffer, state: Tensor, k: Tensor, v: Tensor, r: Tensor, x: Tensor):
        dd, ds, dk, dv, dr, dx = time_decay.desc(), state.desc(), k.desc(), v.desc(), r.desc(), x.desc()
        x.ctx.check(hip.wrk_op_time_mix_v6(x.ctx.h, cursors.h, C.byref(dd), time_first.h, C.byref(ds), C.byref(dk), C.byref(dv), C.byref(dr), C.byref(dx)))

    @staticmethod
    def channel_mix(cursors: Buffer, state: Tensor, r: Tensor, v: Tensor, x: Tensor):
        ds, dr, dv, dx = state.desc(), r.desc(), v.desc(), x.desc()
        x.ctx.check(hip.wrk_op_channel_mix(x.ctx.h, cursors.h, C.byref(ds), C.byref(dr), C.byref(dv), C.byref(dx)))

    @staticmethod
    def add_activate(inp: Tensor, out: Tensor, act_x="none", act_y="none", act_out="none"):
        i, o = inp.desc(), out.desc()
        out.ctx.check(hip.wrk_op_add(out.ctx.h, C.byref(i), C.byref(o), ACT[act_x], ACT[act_y], ACT[act_out]))

    add = add_activate

    @staticmethod
    def mul_activate(inp: Tensor, out: Tensor, act_x="none", act_y="none", act_out="none"):
        i, o = inp.desc(), out.desc()
        out.ctx.check(hip.wrk_op_mul(out.ctx.h, C.byref(i), C.byref(o), ACT[act_x], ACT[act_y], ACT[act_out]))

    mul = mul_activate

    @staticmethod
    def lerp(x: Tensor, y: Tensor, f: Tensor, reversed_: bool):
        a, b, c = x.desc(), y.desc(), f.desc()
        y.ctx.check(hip.wrk_op_lerp(y.ctx.h, C.byref(a), C.byref(b), C.byref(c), int(reversed_)))

    @staticmethod
    def blit(inp: Tensor, out: Tensor):
        i, o = inp.desc(), out.desc()
        out.ctx.check(hip.wrk_op_blit(out.ctx.h, C.byref(i), C.byref(o)))

    @staticmethod
    def affine(x: Tensor, scale: float, bias: float):
        d = x.desc(); x.ctx.check(hip.wrk_op_affine(x.ctx.h, C.byref(d), scale, bias))

    @staticmethod
    def activate(x: Tensor, act: str):
        d = x.desc(); x.ctx.check(hip.wrk_op_activate(x.ctx.h, C.byref(d), ACT[act]))

    @staticmethod
    def control_k_v7(p: Buffer, a: Tensor, k: Tensor):
        da, dk = a.desc(), k.desc()
        k.ctx.check(hip.wrk_op_control_k_v7(k.ctx.h, p.h, C.byref(da), C.byref(dk)))

    @staticmethod
    def time_mix_v7(cursors: Buffer, state: Tensor, r: Tensor, w: Tensor, n: Tensor, x: Tensor):
        ds, dr, dw, dn, dx = state.desc(), r.desc(), w.desc(), n.desc(), x.desc()
        x.ctx.check(hip.wrk_op_time_mix_v7(x.ctx.h, cursors.h, C.byref(ds), C.byref(dr), C.byref(dw), C.byref(dn), C.byref(dx)))

    @staticmethod
    def time_first_v7(u: Buffer, r: Tensor, n: Tensor, x: Tensor):
        dr, dn, dx = r.desc(), n.desc(), x.desc()
        x.ctx.check(hip.wrk_op_time_first_v7(x.ctx.h, u.h, C.byref(dr), C.byref(dn), C.byref(dx)))

    @staticmethod
    def channel_mix_v7(cursors: Buffer, state: Tensor, v: Tensor, x: Tensor):
        ds, dv, dx = state.desc(), v.desc(), x.desc()
        x.ctx.check(hip.wrk_op_channel_mix_v7(x.ctx.h, cursors.h, C.byref(ds), C.byref(dv), C.byref(dx)))

    @staticmethod
    def softmax(x: Tensor):
        d = x.desc(); x.ctx.check(hip.wrk_op_softmax(x.ctx.h, C.byref(d)))


# ----------------------------------------------------------------------------- host layer
def _host_check(rc: int):
    if rc != OK:
        raise WrkError(rc, (rt.wrk_host_last_error() or b"").decode())


def quantile_student(nu: float = 5.0) -> np.ndarray:
    """`quantile_student(nu)` (src/tensor/matrix.rs:29-44): the 16 f32 levels of `Float4Quant::new_student(nu)` (SF4)."""
    out = np.zeros(16, np.float32)
    _host_check(rt.wrk_quantile_student(float(nu), _ptr(out, _f32p)))
    return out


class GgufReader:
    """`GgufReader` + `Reader` trait (src/runtime/gguf.rs)."""

    def __init__(self, data=None, path: Optional[str] = None):
        h = _P()
        if path is not None:
            _host_check(rt.wrk_gguf_open(path.encode(), C.byref(h)))
            self._keep = None
        else:
            self._keep = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
            _host_check(rt.wrk_gguf_from_memory(self._keep.ctypes.data_as(_P), self._keep.nbytes, C.byref(h)))
        self.h = h

    @property
    def version(self) -> int:
        return rt.wrk_gguf_version(self.h)

    @property
    def tensor_data_offset(self) -> int:
        return rt.wrk_gguf_tensor_data_offset(self.h)

    def contains(self, name: str) -> bool:
        return bool(rt.wrk_gguf_contains(self.h, name.encode()))

    def shape(self, name: str) -> List[int]:
        dims = (C.c_uint32 * 4)()
        nd = C.c_uint32()
        _host_check(rt.wrk_gguf_shape(self.h, name.encode(), dims, C.byref(nd)))
        return [int(dims[i]) for i in range(nd.value)]

    def tensor_f16(self, name: str) -> np.ndarray:
        n = C.c_size_t()
        _host_check(rt.wrk_gguf_tensor_f16(self.h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint16)
        _host_check(rt.wrk_gguf_tensor_f16(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_uint16)), out.size, C.byref(n)))
        return out.view(np.float16)

    def raw(self, name: str):
        t, p, n = C.c_uint32(), _P(), C.c_size_t()
        _host_check(rt.wrk_gguf_raw(self.h, name.encode(), C.byref(t), C.byref(p), C.byref(n)))
        return t.value, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))

    def read_state(self) -> np.ndarray:
        """`read_state(context, info, reader)` (v7.rs:1229-1262): pre-trained initial state -> [L, S+2, D] f32."""
        n = C.c_size_t()
        _host_check(rt.wrk_gguf_read_state(self.h, None, 0, C.byref(n)))
        mi = self.info()
        out = np.empty(n.value, np.float32)
        _host_check(rt.wrk_gguf_read_state(self.h, _ptr(out, _f32p), out.size, C.byref(n)))
        S = mi.num_emb // mi.num_head
        return out.reshape(mi.num_layer, S + 2, mi.num_emb)

    def info(self) -> ModelInfo:
        """`Loader::info(&reader)`."""
        mi = ModelInfo()
        _host_check(rt.wrk_gguf_info(self.h, C.byref(mi)))
        return mi

    def close(self):
        if self.h:
            rt.wrk_gguf_close(self.h)
            self.h = None


class RnnInput:
    """`RnnInput::new(batches, token_chunk_size)` (src/runtime/infer/rnn.rs:204-253)."""

    def __init__(self, batches: Sequence[Sequence[int]], token_chunk_size: int = 128, options: Optional[Sequence[int]] = None):
        h = _P()
        _host_check(rt.wrk_rnn_input_create(len(batches), token_chunk_size, C.byref(h)))
        self.h, self.num_batch = h, len(batches)
        for b, toks in enumerate(batches):
            self.append(b, toks)
            if options is not None:
                _host_check(rt.wrk_rnn_input_set_option(self.h, b, options[b]))

    @property
    def token_chunk_size(self) -> int:
        return rt.wrk_rnn_input_token_chunk_size(self.h)

    def append(self, batch: int, tokens: Sequence[int]):
        a = _u32(tokens)
        _host_check(rt.wrk_rnn_input_append(self.h, batch, _ptr(a, _u32p), a.size))

    def remaining(self, batch: int) -> int:
        return rt.wrk_rnn_input_remaining(self.h, batch)

    def step(self):
        _host_check(rt.wrk_rnn_input_step(self.h))

    def iter(self):
        it = _P()
        _host_check(rt.wrk_rnn_iter_create(self.h, C.byref(it)))
        nb = self.num_batch
        try:
            while True:
                lens = np.zeros(nb, np.uint32)
                opts = np.zeros(nb, np.int32)
                _host_check(rt.wrk_rnn_iter_next(it, _ptr(lens, _u32p), _ptr(opts, _i32p)))
                yield [(int(l), int(o)) for l, o in zip(lens, opts)]
        finally:
            rt.wrk_rnn_iter_destroy(it)

    def __del__(self):
        try:
            if self.h:
                rt.wrk_rnn_input_destroy(self.h)
        except Exception:
            pass
        self.h = None


def redirect(info):
    """`RnnInfo::redirect` -> (headers, inputs, outputs)."""
    nb = len(info)
    lens = _u32([l for l, _ in info])
    opts = np.ascontiguousarray(np.asarray([o for _, o in info], dtype=np.int32))
    headers = np.zeros(max(int(lens.sum()), 1), np.uint32)
    nh = C.c_uint32()
    inputs = np.zeros(2 * nb, np.uint32)
    outputs = np.zeros(2 * nb, np.uint32)
    _host_check(rt.wrk_rnn_redirect(_ptr(lens, _u32p), _ptr(opts, _i32p), nb, _ptr(headers, _u32p), C.byref(nh),
                                    _ptr(inputs, _u32p), _ptr(outputs, _u32p)))
    return ([int(x) for x in headers[: nh.value]], [tuple(int(v) for v in inputs[2 * b:2 * b + 2]) for b in range(nb)],
            [tuple(int(v) for v in outputs[2 * b:2 * b + 2]) for b in range(nb)])


class Runtime:
    """`ModelBuilder::new(&context, reader).build_v7()` -> `v7::Bundle::<f16>::new(model, num_batch)`
    -> `SimpleRuntime::new(bundle)`; `infer(input)` as src/runtime/mod.rs:238-263."""

    def __init__(self, ctx: Context, reader: GgufReader, num_batch: int = 1, weights: int = WEIGHTS_INLINE, rescale: int = 0,
                 quant: Optional[Dict[int, int]] = None):
        """`quant`: `ModelBuilder::quant`'s layer -> QUANT_* map."""
        opt = BuildOptions(rescale, weights, None, 0)
        if quant:
            q = np.zeros(max(quant) + 1, np.uint8)
            for l, v in quant.items():
                q[l] = v
            opt.quant, opt.num_quant = q.ctypes.data_as(C.POINTER(C.c_uint8)), q.size
        h = _P()
        _host_check(rt.wrk_runtime_create(ctx.h, reader.h, C.byref(opt), num_batch, C.byref(h)))
        self.ctx, self.h, self.num_batch = ctx, h, num_batch
        self.info = ModelInfo()
        rt.wrk_runtime_info(h, C.byref(self.info))
        self.model = rt.wrk_runtime_model(h)
        self.model6 = rt.wrk_runtime_model_v6(h)
        self.state = rt.wrk_runtime_state(h)

    def token_bytes(self, num_batch: int = 1) -> int:
        if self.model6:
            return hip.wrk_v6_model_token_bytes(self.model6, num_batch)
        return hip.wrk_v7_model_token_bytes(self.model, num_batch)

    def infer(self, inp: RnnInput, mode: int = 1) -> List[np.ndarray]:
        """(input, output) = runtime.infer(input): runs one chunk; returns per-batch logits [rows, V]."""
        V = self.info.num_vocab
        cap = inp.token_chunk_size + inp.num_batch
        logits = np.empty((cap, V), dtype=np.float32)
        rows = np.zeros(inp.num_batch, np.uint32)
        _host_check(rt.wrk_runtime_infer(self.h, inp.h, _ptr(logits, _f32p), cap, _ptr(rows, _u32p), mode))
        out, p = [], 0
        for b in range(inp.num_batch):
            out.append(logits[p:p + rows[b]].copy())
            p += int(rows[b])
        return out

    def infer_raw(self, tokens, cursors, headers, mode: int = 1, want_argmax: bool = False):
        """One RnnJob on explicit stacked tokens / packed cursors / header rows."""
        t, c, h = _u32(tokens), _u32(cursors), _u32(headers)
        V = self.info.num_vocab
        logits = np.empty((max(h.size, 1), V), np.float32)
        am = np.zeros(max(h.size, 1), np.uint32)
        fn, mdl = (hip.wrk_v6_infer, self.model6) if self.model6 else (hip.wrk_v7_infer, self.model)
        self.ctx.check(fn(self.ctx.h, mdl, self.state, _ptr(t, _u32p), None, _ptr(c, _u32p), t.size,
                          _ptr(h, _u32p), h.size, _ptr(logits, _f32p), _ptr(am, _u32p) if want_argmax else None, mode))
        return (logits[: h.size], am[: h.size]) if want_argmax else logits[: h.size]

    def set_frame_dtype(self, dtype: int):
        """`Bundle::<f16>` (F16, default) or `Bundle::<f32>` (F32) -- v7.rs:281-364 is generic over the activation type."""
        self.ctx.check(hip.wrk_v7_model_set_frame_dtype(self.ctx.h, self.model, dtype))
        self.frame_dtype = dtype

    def engine_status(self):
        """(exists, reason) of the persistent batch-1 decode engine of this model (RWKV-7)."""
        buf = C.create_string_buffer(512)
        rc = hip.wrk_v7_model_engine_status(self.ctx.h, self.model, buf, 512)
        if rc < 0:
            self.ctx.check(rc)
        return rc == 1, buf.value.decode()

    def infer_layer(self, layer: int, x: np.ndarray, v_first: Optional[np.ndarray], cursors, mode: int = 0):
        """Teacher-forced run of one layer on the layer input `x` [T, D] (and the layer-0 value `v_first`)."""
        dt = np.float32 if getattr(self, "frame_dtype", F16) == F32 else np.float16
        xa = np.ascontiguousarray(x, dtype=dt)
        va = np.ascontiguousarray(v_first, dtype=dt) if v_first is not None else None
        c = _u32(cursors)
        self.ctx.check(hip.wrk_v7_infer_layer(self.ctx.h, self.model, self.state, layer, xa.ctypes.data_as(_P),
                                              va.ctypes.data_as(_P) if va is not None else None, _ptr(c, _u32p), c.size, mode))

    def frame(self, name: str, num_token: int) -> np.ndarray:
        """One `Runtime<F>` buffer of the last job, [T, C] (names: examples/inspect.rs:208-248)."""
        dt = np.float32 if getattr(self, "frame_dtype", F16) == F32 else np.float16
        n = C.c_size_t()
        self.ctx.check(hip.wrk_v7_frame_read(self.ctx.h, self.model, name.encode(), num_token, None, 0, C.byref(n)))
        out = np.empty(n.value // np.dtype(dt).itemsize, dt)
        self.ctx.check(hip.wrk_v7_frame_read(self.ctx.h, self.model, name.encode(), num_token, out.ctypes.data_as(_P), out.nbytes, C.byref(n)))
        return out.reshape(num_token, -1)

    def generate_greedy(self, first_tokens, steps: int, mode: int = 1, want_logits: bool = False, groups: int = 1):
        """Device-resident greedy loop; returns (tokens [steps, B], elapsed_ms[, last logits [B, V]]).
        groups > 1 (RWKV-7): the B independent sequences are dealt over that many concurrent pipelines (each a contiguous block of
        sequences with its own frame, decode program and HIP stream) instead of one batched step."""
        mode = (mode & 0xff) | ((groups & 0xff) << 8 if groups > 1 and not self.model6 else 0)
        ft = _u32(first_tokens)
        B = ft.size
        out = np.zeros((steps, B), np.uint32)
        ms = C.c_float()
        logits = np.empty((B, self.info.num_vocab), np.float32) if want_logits else None
        fn, mdl = (hip.wrk_v6_generate_greedy, self.model6) if self.model6 else (hip.wrk_v7_generate_greedy, self.model)
        self.ctx.check(fn(self.ctx.h, mdl, self.state, _ptr(ft, _u32p), B, steps, _ptr(out, _u32p),
                          _ptr(logits, _f32p) if want_logits else None, C.byref(ms), mode))
        return (out, ms.value, logits) if want_logits else (out, ms.value)

    def state_back(self, batch: int) -> np.ndarray:
        """`State::back(batch)` -> [L, S+2, D] f32."""
        L, D, S = self.info.num_layer, self.info.num_emb, self.info.num_emb // self.info.num_head
        out = np.empty((L, S + 2, D), np.float32)
        self.ctx.check(hip.wrk_v7_state_back(self.ctx.h, self.state, batch, _ptr(out, _f32p)))
        return out

    def state_read(self, batch: int) -> "Buffer":
        """`State::read(batch)` (v7.rs:246-262): a device-resident snapshot [L, S+2, D] f32 (no host round trip)."""
        L, D, S = self.info.num_layer, self.info.num_emb, self.info.num_emb // self.info.num_head
        buf = Buffer(self.ctx, L * (S + 2) * D * 4)
        self.ctx.check(hip.wrk_v7_state_read(self.ctx.h, self.state, batch, buf.h))
        return buf

    def state_write(self, snapshot: "Buffer", batch: int):
        """`State::write(tensor, batch)` (v7.rs:229-244)."""
        self.ctx.check(hip.wrk_v7_state_write(self.ctx.h, self.state, batch, snapshot.h))

    def state_load(self, tensor: np.ndarray, batch: int):
        a = np.ascontiguousarray(tensor, dtype=np.float32)
        L, D, S = self.info.num_layer, self.info.num_emb, self.info.num_emb // self.info.num_head
        assert a.shape == (L, S + 2, D)
        self.ctx.check(hip.wrk_v7_state_load(self.ctx.h, self.state, batch, _ptr(a, _f32p)))

    def close(self):
        if self.h:
            rt.wrk_runtime_destroy(self.h)
            self.h = None
