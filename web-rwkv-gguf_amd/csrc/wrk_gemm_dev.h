// Shared by the MFMA GEMM translation units (wrk_gemm.hip: K-split / K-sliced / tile kernels; wrk_gemm3.hip: the third-generation
// prefill tile): launch parameters, code -> f16 helpers, the MFMA wrapper.
#pragma once
#include <mutex>
#include <set>
#include <utility>

#include "wrk_device.h"

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: set it once per (device, kernel), under a lock (contexts
// on several GPUs, encoders on several threads) -- ADVICE r02: a process-wide `static bool done` left the second GPU's launch rejected.
inline bool lds_attr_once(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done.insert({dev, fn});
    return true;
}

namespace wrk {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x2 pk_bits(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ f16x2 splat(float a) { f16x2 r = {(f16)a, (f16)a}; return r; }

// 8 bytes (two dwords, each byte a code < 1024) -> f16x8 of subnormals code * 2^-24
__device__ __forceinline__ f16x8 codes8(uint32_t w0, uint32_t w1) {
    const f16x2 a = pk_bits(__builtin_amdgcn_perm(0u, w0, 0x0c010c00u)), b = pk_bits(__builtin_amdgcn_perm(0u, w0, 0x0c030c02u));
    const f16x2 c = pk_bits(__builtin_amdgcn_perm(0u, w1, 0x0c010c00u)), d = pk_bits(__builtin_amdgcn_perm(0u, w1, 0x0c030c02u));
    f16x8 r = {a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
    return r;
}

__device__ __forceinline__ f16x8 mul8(f16x8 v, float s) {
    const f16 h = (f16)s;
    f16x8 m = {h, h, h, h, h, h, h, h};
    return v * m;
}
__device__ __forceinline__ f16x8 add8(f16x8 v, float s) {
    const f16 h = (f16)s;
    f16x8 m = {h, h, h, h, h, h, h, h};
    return v + m;
}

__device__ __forceinline__ f32x4v mfma16(f16x8 a, f16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

struct GemmParams {
    const uint8_t* w;
    uint32_t kind, k, m, row_bytes, act;
    uint32_t n;                 // tokens
    uint32_t has_res;
    uint32_t wg_begin;          // first workgroup (in x) of this job
    float scale;                // wrk_matrix::out_scale
    unsigned long long* dbg;    // WRK_TIMING build: stamps of this launch
    DTensor in, out, res;       // [K, T, B], [M, T, B]   (the LDS-tile kernels address through these)
    // dense token stacks for the K-split kernel: token tok at base + tok * stride (elements)
    const f16* x; const void* res_p; void* out_p;
    uint32_t xs, rs, os, out32, res32;
    const float* levels;        // NF4 / SF4: the 16 f32 levels (device)
};

constexpr int GEMM_MAX_JOBS = 8;
struct GemmBatch {
    GemmParams jobs[GEMM_MAX_JOBS];
    int njobs;
};

// token index -> (t, b) of the [C, T, B] views
__device__ __forceinline__ void tok_tb(const DTensor& d, uint32_t tok, uint32_t& t, uint32_t& b) { t = tok % d.shape[1]; b = tok / d.shape[1]; }


// wrk_gemm3.hip: third-generation prefill tile (Q4_K, >= 512 stacked tokens).  `xsum`: scratch of xsum_cap bytes for the per-sub-block
// input sums.  0 = launched, -1 = not applicable (scratch too small, ...)
int gemm_tile3_launch(hipStream_t s, const GemmBatch& T3, uint32_t row_tiles, uint32_t n, void* xsum, size_t xsum_cap);

}  // namespace wrk
