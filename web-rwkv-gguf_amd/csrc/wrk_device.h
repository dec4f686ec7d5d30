// Device-side helpers shared by all gfx950 kernels: f16 storage, View addressing, activations,
// wave64 / workgroup reductions.  Compiled with -ffp-contract=off: every fused multiply-add is
// written explicitly so rounding follows the reference's expressions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wrk_internal.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ------------------------------------------------------------------ rounding points
// Round an f32 to the f16 the reference would have stored in a Runtime<f16> buffer.
__device__ __forceinline__ float r16(float x) { return (float)(f16)x; }

// ------------------------------------------------------------------ View addressing
__device__ __forceinline__ size_t dt_index(const DTensor& d, uint32_t c, uint32_t t, uint32_t b) {
    return ((size_t)(b + d.offset[2]) * d.stride[1] + (t + d.offset[1])) * d.stride[0] + c + d.offset[0];
}
__device__ __forceinline__ size_t dt_index4(const DTensor& d, uint32_t c, uint32_t t, uint32_t b, uint32_t w) {
    return (((size_t)(w + d.offset[3]) * d.stride[2] + (b + d.offset[2])) * d.stride[1] + (t + d.offset[1])) * d.stride[0] + c + d.offset[0];
}
__device__ __forceinline__ float dt_load(const DTensor& d, size_t i) {
    return d.dtype == WRK_F16 ? (float)((const f16*)d.p)[i] : ((const float*)d.p)[i];
}
__device__ __forceinline__ void dt_store(const DTensor& d, size_t i, float v) {
    if (d.dtype == WRK_F16) ((f16*)d.p)[i] = (f16)v; else ((float*)d.p)[i] = v;
}
// value as it reads back from the buffer after a store (f16 buffers round)
__device__ __forceinline__ float dt_round(const DTensor& d, float v) { return d.dtype == WRK_F16 ? r16(v) : v; }

// ------------------------------------------------------------------ cursors (tensor/mod.rs:53-60)
struct Cursor { uint32_t batch, token, len; };
__device__ __forceinline__ Cursor unpack_cursor(uint32_t x) {
    Cursor c; c.batch = x & 0xffu; c.token = (x >> 8) & 0xffffu; c.len = (x >> 24) & 0xffu; return c;
}

// ------------------------------------------------------------------ activations (ops.rs:205-235)
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_apply(uint32_t act, float x) {
    switch (act) {
        case WRK_ACT_SQUARED_RELU: { float p = fmaxf(x, 0.0f); return p * p; }
        case WRK_ACT_TANH: return x > 42.0f ? 1.0f : tanhf(x);
        case WRK_ACT_STABLE_EXP: return __expf(-__expf(x));
        case WRK_ACT_OPPOSITE_EXP: return -__expf(x);
        case WRK_ACT_SOFTPLUS: return __logf(1.0f + __expf(x));
        case WRK_ACT_SIGMOID: return act_sigmoid(x);
        case WRK_ACT_SILU: return x / (1.0f + __expf(-x));
        default: return x;
    }
}
// WGSL mix(x, y, a) = x * (1 - a) + y * a
__device__ __forceinline__ float wgsl_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// ------------------------------------------------------------------ reductions
// wave64 sum on the VALU only: 4 DPP steps reduce every 16-lane row (quad xor 1, quad xor 2, half mirror,
// row mirror), then the 4 row sums are read through SGPRs.  No ds_bpermute, no LDS latency.  All 64 lanes
// must be active (EXEC full); the result is uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
// sum over a workgroup of NW waves; `red` is NW floats of LDS; result broadcast to all threads
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

// debug timing: thread 0 of the first / last workgroup stamps the 100 MHz wall clock.  Compiled in only by
// `make TIMING=1` (the sched barriers that pin a stamp also constrain the production schedule); run with WRK_TIMING=1.
#ifdef WRK_TIMING_BUILD
#define WRK_STAMP(p, k)                                                                                       \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);      /* keep the stamp where it is written */                     \
        if ((p) && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                      \
            (p)[(blockIdx.x ? 8 : 0) + (k)] = wall_clock64();                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#else
#define WRK_STAMP(p, k) do { } while (0)
#endif
