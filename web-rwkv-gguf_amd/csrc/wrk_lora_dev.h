// LoRA up-projection helpers of the RWKV-7 head kernels (wrk_v7_fused.hip) and of the persistent engine (wrk_v7_engine.hip).
#pragma once
#include "wrk_device.h"

namespace wrk {

// dot of a row slice of an f16 [D][rank] matrix with the token's f16 LoRA intermediate; 4 lanes share a
// row (lane `part` takes columns part*8 + 32*n ..+8).  Loads and arithmetic are separate calls so the kernel
// can put EVERY load of the launch in flight before the first wait (loads return in issue order).
template <int MAXCH>
struct LoraRegs { f16x8 w[MAXCH], x[MAXCH]; };

template <int MAXCH>
__device__ __forceinline__ void lora_load(LoraRegs<MAXCH>& r, const f16* __restrict__ wrow, const f16* __restrict__ aux, uint32_t rank, uint32_t part) {
#pragma unroll
    for (int n = 0; n < MAXCH; ++n) {
        // unconditional (column clamped; lora_dot masks the chunks beyond the rank): predicated loads make the compiler lose count of
        // what is in flight and wait vmcnt(0) between dependent groups
        const uint32_t c = min(part * 8 + 32 * n, rank - 8);
        r.w[n] = *(const f16x8*)(wrow + c);
        r.x[n] = *(const f16x8*)(aux + c);
    }
}

template <int MAXCH>
__device__ __forceinline__ float lora_dot(const LoraRegs<MAXCH>& r, uint32_t rank, uint32_t part) {
    float acc = 0.0f;
#pragma unroll
    for (int n = 0; n < MAXCH; ++n) {
        const uint32_t c = part * 8 + 32 * n;
        if (c < rank) {
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(r.w[n], r.w[n], 0, 1), __builtin_shufflevector(r.x[n], r.x[n], 0, 1), acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(r.w[n], r.w[n], 2, 3), __builtin_shufflevector(r.x[n], r.x[n], 2, 3), acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(r.w[n], r.w[n], 4, 5), __builtin_shufflevector(r.x[n], r.x[n], 4, 5), acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(r.w[n], r.w[n], 6, 7), __builtin_shufflevector(r.x[n], r.x[n], 6, 7), acc, false);
        }
    }
    acc += dpp_f32<0xB1>(acc);      // lane ^ 1, lane ^ 2 inside the quad by DPP (ds_bpermute = an LDS round trip each on the head kernels' critical path)
    acc += dpp_f32<0x4E>(acc);
    return acc;
}

}  // namespace wrk
