// On-device quantisation to web-rwkv's own matrix formats.
//   quantize_mat_int8  src/shaders/quant_mat_int8.wgsl:24-59  (+ Matrix::quant_u8, matrix.rs:211-227)
//   quantize_mat_nf4   src/shaders/quant_mat_nf4.wgsl:24-81   (+ Matrix::quant_nf4 / quant_sf4, matrix.rs:229-271)
// One wave per block (128 resp. 64 consecutive elements of the flattened matrix; K % block == 0 keeps a block
// inside one row).  Output goes straight into the row-plane layout the matvec kernels read:
//   INT8 row = [codes K][(min, max) f16 x K/128]      NF4 row = [nibbles K/2][absmax f16 x K/64]
#include "wrk_device.h"

namespace wrk {

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
    return v;
}

__global__ void __launch_bounds__(256) quant_int8_kernel(const f16* __restrict__ src, uint8_t* __restrict__ dst, uint32_t k, uint32_t m,
                                                         uint32_t row_bytes) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= (size_t)k * m / 128) return;
    // blocks run over the flattened matrix; k % 16 == 0, so this lane's element pair stays inside one row
    const size_t e0 = blk * 128 + 2 * lane;
    const uint32_t row = (uint32_t)(e0 / k), col = (uint32_t)(e0 % k);
    const f16x2 v = *(const f16x2*)(src + e0);
    const float v0 = (float)v[0], v1 = (float)v[1];
    const float mn = wave_min_f(fminf(v0, v1)), mx = wave_max_f(fmaxf(v0, v1));   // exact f16 values
    const float range = mx - mn;
    // saturate((v - min) / (max - min)); pack4x8unorm = floor(0.5 + 255 * x)
    const float x0 = fminf(fmaxf((v0 - mn) / range, 0.0f), 1.0f), x1 = fminf(fmaxf((v1 - mn) / range, 0.0f), 1.0f);
    uint8_t* d = dst + (size_t)row * row_bytes;
    uint8_t c[2] = {(uint8_t)floorf(0.5f + 255.0f * x0), (uint8_t)floorf(0.5f + 255.0f * x1)};
    *(uint16_t*)(d + col) = (uint16_t)c[0] | ((uint16_t)c[1] << 8);
    // every row the block touches keeps its own copy of (min, max): entry = block - first block of that row
    const uint32_t row_last = (uint32_t)((blk * 128 + 127) / k);
    if (lane == 0 || (lane == 63 && row_last != (uint32_t)(blk * 128 / k))) {
        const uint32_t rr = lane == 0 ? (uint32_t)(blk * 128 / k) : row_last;
        const f16x2 mm = {(f16)mn, (f16)mx};
        *(f16x2*)(dst + (size_t)rr * row_bytes + k + (blk - (size_t)rr * k / 128) * 4) = mm;
    }
}

__global__ void __launch_bounds__(256) quant_nf4_kernel(const f16* __restrict__ src, const float* __restrict__ levels, uint8_t* __restrict__ dst,
                                                        uint32_t k, uint32_t m, uint32_t row_bytes) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t bpr = k / 64;
    if (blk >= (size_t)bpr * m) return;
    const uint32_t row = (uint32_t)(blk / bpr), bi = (uint32_t)(blk % bpr);
    const float v = (float)src[(size_t)row * k + (size_t)bi * 64 + lane];
    const float amax = wave_max_f(fabsf(v));
    const float x = v * (1.0f / amax);
    // nearest level; "<=" from min_err = 1.0 keeps the LAST of equally near levels (quant_mat_nf4.wgsl:63-72)
    float min_err = 1.0f;
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) {
        const float e = fabsf(levels[i] - x);
        if (e <= min_err) { min_err = e; idx = i; }
    }
    uint32_t word = idx << (4 * (lane & 7u));
    word |= __shfl_xor(word, 1, WAVE);
    word |= __shfl_xor(word, 2, WAVE);
    word |= __shfl_xor(word, 4, WAVE);
    uint8_t* d = dst + (size_t)row * row_bytes;
    if ((lane & 7u) == 0) *(uint32_t*)(d + (size_t)bi * 32 + (lane >> 3) * 4) = word;
    if (lane == 0) *(f16*)(d + (k >> 1) + (size_t)bi * 2) = (f16)amax;
}

void quantize_int8(hipStream_t s, const void* src_f16, uint8_t* dst, uint32_t k, uint32_t m, uint32_t row_bytes) {
    const size_t blocks = (size_t)k * m / 128;
    hipLaunchKernelGGL(quant_int8_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, s, (const f16*)src_f16, dst, k, m, row_bytes);
}
void quantize_nf4(hipStream_t s, const void* src_f16, const float* levels, uint8_t* dst, uint32_t k, uint32_t m, uint32_t row_bytes) {
    const size_t blocks = (size_t)(k / 64) * m;
    hipLaunchKernelGGL(quant_nf4_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, s, (const f16*)src_f16, levels, dst, k, m, row_bytes);
}

}  // namespace wrk
