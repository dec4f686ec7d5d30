// Inline-dequant matvec for gfx950: y[M] = act(W[M,K] . x[K]) for 1..8 input vectors per pass.
//
// Replaces (reference file:line)
//   matmul_vec_fp16   ops.rs:697-784   + shaders/matmul_vec_fp16.wgsl
//   matmul_vec_q4k    ops.rs:1403-1467 + shaders/matmul_vec_q4k_v2.wgsl
//   matmul_vec_q5k/q6k/q8_0 ops.rs:1543-1948 -- semantics from gguf.rs:11-37,149-274 (SURVEY F3:
//   the Q5_K/Q6_K/Q8_0 shaders are defective; the CPU dequantisers are canonical)
//
// Design (DESIGN.md "matvec"): the kernel is HBM-bound, so everything is arranged around 16-byte
// coalesced weight loads straight into VGPRs:
//   * blocks are re-laid-out per row at upload (repack_rows) into 16-byte aligned planes
//     (quants | high bits | scales/header), so lane L of a wave reads the L-th 16-byte chunk of the
//     quant plane: one global_load_dwordx4 per lane covers 1 KiB contiguous per wave-instruction;
//   * one wave64 owns a row; 4 waves (256 threads) per workgroup;
//   * the input vector(s) are staged once per workgroup in LDS as f16, together with the per-16
//     partial sums that carry the K-quant "min" term:  sum_l (d*sc*q_l - dmin*m) x_l
//       = d*sc * sum_l q_l x_l  -  dmin*m * sum_l x_l ;
//   * integer codes become f16 by byte-permute into 0x6400|q (= 1024+q, exact) and one packed
//     subtract, then v_dot2_f32_f16 accumulates exact f16*f16 products in f32;
//   * wave64 butterfly reduction, fused activation, store in the output buffer's dtype.
// WRK_MATRIX_ROUND_F16 instead rounds every dequantised weight to f16 first, reproducing the
// reference at HEAD (weights dequantised to f16 on the CPU at load, gguf.rs:95-274).
#include "wrk_device.h"

namespace wrk {

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

size_t repack_row_bytes(uint32_t kind, uint32_t k) {
    const size_t nb = k / 256;
    switch (kind) {
        case WRK_MAT_F16: return align16((size_t)k * 2);
        case WRK_MAT_Q8_0: return align16((size_t)k + (size_t)(k / 32) * 2);
        case WRK_MAT_Q4_K: return nb * 144;
        case WRK_MAT_Q5_K: return nb * 176;
        case WRK_MAT_Q6_K: return align16(nb * 208 + nb * 2);
        default: return 0;
    }
}

size_t stored_bytes(uint32_t kind, uint32_t k, uint32_t m) {
    const size_t n = (size_t)k * m;
    switch (kind) {
        case WRK_MAT_F32:
        case WRK_MAT_F16: return n * 2;                // F32 sources are held as f16 (loader.rs:117-121)
        case WRK_MAT_Q8_0: return n / 32 * 34;
        case WRK_MAT_Q4_K: return n / 256 * 144;
        case WRK_MAT_Q5_K: return n / 256 * 176;
        case WRK_MAT_Q6_K: return n / 256 * 210;
        case WRK_MAT_INT8: return n + n / 128 * 4;
        case WRK_MAT_NF4: return n / 2 + n / 64 * 2;
        default: return 0;
    }
}

// Host-side re-layout of raw GGUF blocks (row-major rows of k elements) into the device planes.
int repack_rows(uint32_t kind, uint32_t k, uint32_t m, const uint8_t* src, uint8_t* dst) {
    const size_t rb = repack_row_bytes(kind, k);
    const size_t nb = k / 256;
    if (rb == 0) return -1;
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < (long long)m; ++r) {
        uint8_t* d = dst + (size_t)r * rb;
        switch (kind) {
            case WRK_MAT_F16: {
                memcpy(d, src + (size_t)r * k * 2, (size_t)k * 2);
                break;
            }
            case WRK_MAT_Q8_0: {
                const uint8_t* s = src + (size_t)r * (k / 32) * 34;
                for (size_t b = 0; b < k / 32; ++b) {
                    memcpy(d + b * 32, s + b * 34 + 2, 32);
                    memcpy(d + k + b * 2, s + b * 34, 2);
                }
                break;
            }
            case WRK_MAT_Q4_K: {
                const uint8_t* s = src + (size_t)r * nb * 144;
                for (size_t b = 0; b < nb; ++b) {
                    memcpy(d + b * 128, s + b * 144 + 16, 128);
                    memcpy(d + nb * 128 + b * 16, s + b * 144, 16);
                }
                break;
            }
            case WRK_MAT_Q5_K: {
                const uint8_t* s = src + (size_t)r * nb * 176;
                for (size_t b = 0; b < nb; ++b) {
                    memcpy(d + b * 128, s + b * 176 + 48, 128);            // ql
                    memcpy(d + nb * 128 + b * 32, s + b * 176 + 16, 32);   // qh
                    memcpy(d + nb * 160 + b * 16, s + b * 176, 16);        // d, dmin, scales
                }
                break;
            }
            case WRK_MAT_Q6_K: {
                const uint8_t* s = src + (size_t)r * nb * 210;
                for (size_t b = 0; b < nb; ++b) {
                    memcpy(d + b * 128, s + b * 210, 128);                 // ql
                    memcpy(d + nb * 128 + b * 64, s + b * 210 + 128, 64);  // qh
                    memcpy(d + nb * 192 + b * 16, s + b * 210 + 192, 16);  // scales
                    memcpy(d + nb * 208 + b * 2, s + b * 210 + 208, 2);    // d
                }
                break;
            }
            default: break;
        }
    }
    return 0;
}

// ------------------------------------------------------------------ device: code -> f16 helpers
// bytes b0..b3 of `v` (each < 1024) -> two f16x2 holding (b0, b1) and (b2, b3) minus `bias`
__device__ __forceinline__ void bytes_to_h2(uint32_t v, f16x2 biasv, f16x2& lo, f16x2& hi) {
    const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, v, 0x04010400u);   // 0x6400|b0 , 0x6400|b1
    const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, v, 0x04030402u);   // 0x6400|b2 , 0x6400|b3
    lo = __builtin_bit_cast(f16x2, p0) - biasv;
    hi = __builtin_bit_cast(f16x2, p1) - biasv;
}

__device__ __forceinline__ f16x2 h2(float a) { f16x2 r = {(f16)a, (f16)a}; return r; }

// dot of 16 f16 codes (8 f16x2) with 16 f16 inputs read from LDS
__device__ __forceinline__ float dot16(const f16x2 (&q)[8], const f16* __restrict__ x) {
    const f16x8 xa = *(const f16x8*)x;
    const f16x8 xb = *(const f16x8*)(x + 8);
    float acc = 0.0f;
    acc = __builtin_amdgcn_fdot2(q[0], __builtin_shufflevector(xa, xa, 0, 1), acc, false);
    acc = __builtin_amdgcn_fdot2(q[1], __builtin_shufflevector(xa, xa, 2, 3), acc, false);
    acc = __builtin_amdgcn_fdot2(q[2], __builtin_shufflevector(xa, xa, 4, 5), acc, false);
    acc = __builtin_amdgcn_fdot2(q[3], __builtin_shufflevector(xa, xa, 6, 7), acc, false);
    acc = __builtin_amdgcn_fdot2(q[4], __builtin_shufflevector(xb, xb, 0, 1), acc, false);
    acc = __builtin_amdgcn_fdot2(q[5], __builtin_shufflevector(xb, xb, 2, 3), acc, false);
    acc = __builtin_amdgcn_fdot2(q[6], __builtin_shufflevector(xb, xb, 4, 5), acc, false);
    acc = __builtin_amdgcn_fdot2(q[7], __builtin_shufflevector(xb, xb, 6, 7), acc, false);
    return acc;
}

// get_scale_min_k4 (gguf.rs:81-89) on the 12 scale bytes held in three dwords
__device__ __forceinline__ void scale_min_k4(uint32_t is, uint32_t s0, uint32_t s1, uint32_t s2, float& sc, float& mn) {
    const uint32_t sh = (is & 3u) * 8u;
    const uint32_t a = (s0 >> sh) & 0xffu, b = (s1 >> sh) & 0xffu, c = (s2 >> sh) & 0xffu;
    uint32_t scv, mv;
    if (is < 4u) { scv = a & 63u; mv = b & 63u; }
    else { scv = (c & 0xfu) | ((a >> 6) << 4); mv = (c >> 4) | ((b >> 6) << 4); }
    sc = (float)scv;
    mn = (float)mv;
}

__device__ __forceinline__ float f16bits_to_f32(uint32_t bits) { return (float)__builtin_bit_cast(f16, (uint16_t)bits); }

// One decoded 16-element group: codes as f16 pairs plus the affine (scale, minv) with
//   contribution = scale * dot(q, x[xoff..xoff+16)) - minv * xsum[xoff/16]
struct Group {
    f16x2 q[8];
    float scale, minv;
    uint32_t xoff;
};

template <bool R16>
__device__ __forceinline__ float group_dot(Group& g, const f16* __restrict__ xs, const float* __restrict__ xsum) {
    if (R16) {
        // weight_e = f16(scale * q_e - minv): the exact value the reference stores after CPU dequant
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = (float)g.q[i][0] * g.scale - g.minv;
            const float b = (float)g.q[i][1] * g.scale - g.minv;
            g.q[i][0] = (f16)a;
            g.q[i][1] = (f16)b;
        }
        return dot16(g.q, xs + g.xoff);
    }
    return g.scale * dot16(g.q, xs + g.xoff) - g.minv * xsum[g.xoff >> 4];
}

template <bool R16, int NB>
__device__ __forceinline__ void groups_accumulate(const Group& lo, const Group& hi, bool two, const f16* xs, const float* xsum,
                                                  uint32_t kpad, float (&acc)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        Group l2 = lo;
        acc[n] += group_dot<R16>(l2, xs + (size_t)n * kpad, xsum + (size_t)n * (kpad >> 4));
        if (two) {
            Group h2g = hi;
            acc[n] += group_dot<R16>(h2g, xs + (size_t)n * kpad, xsum + (size_t)n * (kpad >> 4));
        }
    }
}

// ------------------------------------------------------------------ per-kind chunk load / decode
// A chunk is 16 bytes of a row's quant plane.  `load_raw` only ISSUES the global loads of chunk c
// (quants non-temporal: each weight byte is read once per token; side data through the caches),
// `dot_raw` decodes and accumulates -- split so the next chunk's loads are in flight while the
// current one is decoded, and so the first chunk is requested before the inputs are staged.
struct Raw {
    u32x4 w;    // quant chunk
    u32x4 a;    // Q4_K: header | Q5_K: high bits | Q6_K: high bits | Q8_0: a.x = d bits
    u32x4 b;    // Q5_K: header | Q6_K: b.x = sc_lo | sc_hi << 8 | d bits << 16
};

template <int KIND>
__device__ __forceinline__ uint32_t num_chunks(uint32_t k, uint32_t kpad) {
    return KIND == WRK_MAT_F16 ? (kpad >> 3) : (KIND == WRK_MAT_Q8_0 ? (k >> 4) : (k >> 8) * 8);
}

template <int KIND>
__device__ __forceinline__ Raw load_raw(const uint8_t* __restrict__ row, uint32_t k, uint32_t c) {
    Raw r;
    r.w = __builtin_nontemporal_load((const u32x4*)(row + (size_t)c * 16));
    const uint32_t nb = k >> 8, b = c >> 3;
    if (KIND == WRK_MAT_Q4_K) {
        r.a = *(const u32x4*)(row + (size_t)nb * 128 + (size_t)b * 16);
    } else if (KIND == WRK_MAT_Q5_K) {
        r.a = *(const u32x4*)(row + (size_t)nb * 128 + (size_t)b * 32 + (c & 1u) * 16);
        r.b = *(const u32x4*)(row + (size_t)nb * 160 + (size_t)b * 16);
    } else if (KIND == WRK_MAT_Q6_K) {
        const uint32_t sub = c & 7u, n128 = sub >> 2, part = (sub >> 1) & 1u, l0 = (sub & 1u) * 16u;
        r.a = *(const u32x4*)(row + (size_t)nb * 128 + (size_t)b * 64 + n128 * 32 + l0);
        const uint8_t* scp = row + (size_t)nb * 192 + (size_t)b * 16 + n128 * 8 + (l0 >> 4) + part * 2;
        const uint32_t dbits = *(const uint16_t*)(row + (size_t)nb * 208 + (size_t)b * 2);
        r.b.x = (uint32_t)scp[0] | ((uint32_t)scp[4] << 8) | (dbits << 16);
    } else if (KIND == WRK_MAT_Q8_0) {
        r.a.x = *(const uint16_t*)(row + (size_t)k + (size_t)(c >> 1) * 2);
    }
    return r;
}

template <int KIND, bool R16, int NB>
__device__ __forceinline__ void dot_raw(const Raw& r, uint32_t c, const f16* xs, const float* xsum, uint32_t kpad, float (&acc)[NB]) {
    const u32x4 w = r.w;
    if (KIND == WRK_MAT_F16) {
        const f16x8 wv = __builtin_bit_cast(f16x8, w);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const f16x8 x = *(const f16x8*)(xs + (size_t)n * kpad + c * 8);
            float a = acc[n];
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 0, 1), __builtin_shufflevector(x, x, 0, 1), a, false);
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 2, 3), __builtin_shufflevector(x, x, 2, 3), a, false);
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 4, 5), __builtin_shufflevector(x, x, 4, 5), a, false);
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 6, 7), __builtin_shufflevector(x, x, 6, 7), a, false);
            acc[n] = a;
        }
        return;
    }
    Group lo, hi;
    if (KIND == WRK_MAT_Q8_0) {
        const f16x2 bias = h2(1152.0f);    // 1024 + 128: int8 = (byte ^ 0x80) - 128
        bytes_to_h2(w.x ^ 0x80808080u, bias, lo.q[0], lo.q[1]);
        bytes_to_h2(w.y ^ 0x80808080u, bias, lo.q[2], lo.q[3]);
        bytes_to_h2(w.z ^ 0x80808080u, bias, lo.q[4], lo.q[5]);
        bytes_to_h2(w.w ^ 0x80808080u, bias, lo.q[6], lo.q[7]);
        lo.scale = f16bits_to_f32(r.a.x); lo.minv = 0.0f; lo.xoff = c * 16;
        hi = lo;
        groups_accumulate<R16, NB>(lo, hi, false, xs, xsum, kpad, acc);
        return;
    }
    const uint32_t b = c >> 3, sub = c & 7u;
    if (KIND == WRK_MAT_Q6_K) {
        const uint32_t n128 = sub >> 2, part = (sub >> 1) & 1u, l0 = (sub & 1u) * 16u;
        const u32x4 qh = r.a;
        const float sc_lo = (float)(int8_t)(r.b.x & 0xffu), sc_hi = (float)(int8_t)((r.b.x >> 8) & 0xffu);
        const float d = f16bits_to_f32(r.b.x >> 16);
        const uint32_t s_lo = part * 2, s_hi = s_lo + 4;
        const f16x2 bias = h2(1056.0f);    // 1024 + 32: q6 = code - 32, exact in f16
#define Q6LO(W, H) (((W) & 0x0f0f0f0fu) | ((((H) >> s_lo) & 0x03030303u) << 4))
#define Q6HI(W, H) ((((W) >> 4) & 0x0f0f0f0fu) | ((((H) >> s_hi) & 0x03030303u) << 4))
        bytes_to_h2(Q6LO(w.x, qh.x), bias, lo.q[0], lo.q[1]);
        bytes_to_h2(Q6LO(w.y, qh.y), bias, lo.q[2], lo.q[3]);
        bytes_to_h2(Q6LO(w.z, qh.z), bias, lo.q[4], lo.q[5]);
        bytes_to_h2(Q6LO(w.w, qh.w), bias, lo.q[6], lo.q[7]);
        bytes_to_h2(Q6HI(w.x, qh.x), bias, hi.q[0], hi.q[1]);
        bytes_to_h2(Q6HI(w.y, qh.y), bias, hi.q[2], hi.q[3]);
        bytes_to_h2(Q6HI(w.z, qh.z), bias, hi.q[4], hi.q[5]);
        bytes_to_h2(Q6HI(w.w, qh.w), bias, hi.q[6], hi.q[7]);
#undef Q6LO
#undef Q6HI
        lo.scale = d * sc_lo; lo.minv = 0.0f; lo.xoff = b * 256 + n128 * 128 + part * 32 + l0;
        hi.scale = d * sc_hi; hi.minv = 0.0f; hi.xoff = lo.xoff + 64;
        groups_accumulate<R16, NB>(lo, hi, true, xs, xsum, kpad, acc);
        return;
    }
    // Q4_K / Q5_K share the d/dmin/6-bit scale header
    const uint32_t j = sub >> 1, h = sub & 1u;
    const u32x4 hd = KIND == WRK_MAT_Q4_K ? r.a : r.b;
    const float d = f16bits_to_f32(hd.x & 0xffffu), dmin = f16bits_to_f32(hd.x >> 16);
    float sc0, m0, sc1, m1;
    scale_min_k4(2 * j, hd.y, hd.z, hd.w, sc0, m0);
    scale_min_k4(2 * j + 1, hd.y, hd.z, hd.w, sc1, m1);
    const f16x2 bias = h2(1024.0f);
    if (KIND == WRK_MAT_Q4_K) {
        bytes_to_h2(w.x & 0x0f0f0f0fu, bias, lo.q[0], lo.q[1]);
        bytes_to_h2(w.y & 0x0f0f0f0fu, bias, lo.q[2], lo.q[3]);
        bytes_to_h2(w.z & 0x0f0f0f0fu, bias, lo.q[4], lo.q[5]);
        bytes_to_h2(w.w & 0x0f0f0f0fu, bias, lo.q[6], lo.q[7]);
        bytes_to_h2((w.x >> 4) & 0x0f0f0f0fu, bias, hi.q[0], hi.q[1]);
        bytes_to_h2((w.y >> 4) & 0x0f0f0f0fu, bias, hi.q[2], hi.q[3]);
        bytes_to_h2((w.z >> 4) & 0x0f0f0f0fu, bias, hi.q[4], hi.q[5]);
        bytes_to_h2((w.w >> 4) & 0x0f0f0f0fu, bias, hi.q[6], hi.q[7]);
    } else {
        const u32x4 qh = r.a;
        const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
#define Q5LO(W, H) (((W) & 0x0f0f0f0fu) | ((((H) >> s0) & 0x01010101u) << 4))
#define Q5HI(W, H) ((((W) >> 4) & 0x0f0f0f0fu) | ((((H) >> s1) & 0x01010101u) << 4))
        bytes_to_h2(Q5LO(w.x, qh.x), bias, lo.q[0], lo.q[1]);
        bytes_to_h2(Q5LO(w.y, qh.y), bias, lo.q[2], lo.q[3]);
        bytes_to_h2(Q5LO(w.z, qh.z), bias, lo.q[4], lo.q[5]);
        bytes_to_h2(Q5LO(w.w, qh.w), bias, lo.q[6], lo.q[7]);
        bytes_to_h2(Q5HI(w.x, qh.x), bias, hi.q[0], hi.q[1]);
        bytes_to_h2(Q5HI(w.y, qh.y), bias, hi.q[2], hi.q[3]);
        bytes_to_h2(Q5HI(w.z, qh.z), bias, hi.q[4], hi.q[5]);
        bytes_to_h2(Q5HI(w.w, qh.w), bias, hi.q[6], hi.q[7]);
#undef Q5LO
#undef Q5HI
    }
    lo.scale = d * sc0; lo.minv = dmin * m0; lo.xoff = b * 256 + j * 64 + h * 16;
    hi.scale = d * sc1; hi.minv = dmin * m1; hi.xoff = lo.xoff + 32;
    groups_accumulate<R16, NB>(lo, hi, true, xs, xsum, kpad, acc);
}

// ------------------------------------------------------------------ the kernel
constexpr int MAX_JOBS = 8;

struct JobDev {
    const uint8_t* w;
    uint32_t kind, flags, k, m, row_bytes, act;
    uint32_t rows_per_wg, wg_begin;    // first workgroup (in x) of this job
    uint32_t has_res;                  // fused residual: out = round_out(act(acc)) + res   (matmul + TensorOp::add)
    DTensor in, out, res;
    float* amax_val;                   // optional fused arg-max partials: [num_wg][ntok] (value, row)
    uint32_t* amax_idx;
};

struct MatvecParams {
    JobDev jobs[MAX_JOBS];
    int njobs;
};

template <int KIND, bool R16, int NB>
__device__ __forceinline__ void matvec_body(const JobDev& J, unsigned char* smem) {
    const uint32_t K = J.k;
    const uint32_t kpad = (K + 15u) & ~15u;
    f16* xs = (f16*)smem;                                   // [NB][kpad]
    float* xsum = (float*)(smem + (size_t)NB * kpad * 2);   // [NB][kpad/16]
    const uint32_t ntok = J.in.shape[1] * J.in.shape[2];
    const uint32_t tok0 = blockIdx.y * NB;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r0 = (blockIdx.x - J.wg_begin) * J.rows_per_wg;
    const uint32_t r1 = min(r0 + J.rows_per_wg, J.m);

    // work items of this wave: (row, chunk iteration); item it -> row r0 + wave + 4 * (it / iters)
    const uint32_t nch = num_chunks<KIND>(K, kpad);
    const uint32_t iters = (nch + 63) >> 6;
    const uint32_t nrows = r0 + wave < r1 ? (r1 - r0 - wave + 3) >> 2 : 0;
    const uint32_t nitems = nrows * iters;

    // request the first chunk before touching the inputs: weights do not depend on activations
    Raw cur{};
    if (nitems > 0 && lane < nch) cur = load_raw<KIND>(J.w + (size_t)(r0 + wave) * J.row_bytes, K, lane);

    // stage inputs (f16) and their per-16 sums
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const uint32_t tk = tok0 + n;
        const bool live = tk < ntok;
        const uint32_t t = live ? tk % J.in.shape[1] : 0, b = live ? tk / J.in.shape[1] : 0;
        const size_t base = dt_index(J.in, 0, t, b);
        f16* xd = xs + (size_t)n * kpad;
        float* sd = xsum + (size_t)n * (kpad >> 4);
        if (J.in.dtype == WRK_F16 && ((base | K) & 7u) == 0) {
            const f16* src = (const f16*)J.in.p + base;
            for (uint32_t v = tid; v < (kpad >> 3); v += 256) {
                f16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
                if (live && v * 8 < K) x = *(const f16x8*)(src + v * 8);
                *(f16x8*)(xd + v * 8) = x;
                float s8 = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s8 += (float)x[e];
                s8 += __shfl_xor(s8, 1, WAVE);
                if ((v & 1u) == 0) sd[v >> 1] = s8;
            }
        } else {
            for (uint32_t i = tid; i < kpad; i += 256) xd[i] = (live && i < K) ? (f16)dt_load(J.in, base + i) : (f16)0.0f;
            __syncthreads();
            for (uint32_t i = tid; i < (kpad >> 4); i += 256) {
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) s += (float)xd[i * 16 + e];
                sd[i] = s;
            }
        }
    }
    __syncthreads();

    float acc[NB];
    float best_v[NB];
    uint32_t best_i[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) { acc[n] = 0.0f; best_v[n] = -3.0e38f; best_i[n] = 0xffffffffu; }
    for (uint32_t it = 0; it < nitems; ++it) {
        const uint32_t ri = it / iters, ci = it - ri * iters;
        const uint32_t r = r0 + wave + 4 * ri, c = lane + 64 * ci;
        // prefetch the next item
        Raw nxt{};
        if (it + 1 < nitems) {
            const uint32_t it2 = it + 1, ri2 = it2 / iters, ci2 = it2 - ri2 * iters, c2 = lane + 64 * ci2;
            if (c2 < nch) nxt = load_raw<KIND>(J.w + (size_t)(r0 + wave + 4 * ri2) * J.row_bytes, K, c2);
        }
        if (c < nch) dot_raw<KIND, R16, NB>(cur, c, xs, xsum, kpad, acc);
        if (ci + 1 == iters) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const float v = wave_sum(acc[n]);
                acc[n] = 0.0f;
                const uint32_t tk = tok0 + n;
                if (lane == 0 && tk < ntok) {
                    const uint32_t t = tk % J.in.shape[1], b = tk / J.in.shape[1];
                    float o = act_apply(J.act, v);
                    if (J.has_res) o = dt_round(J.out, o) + dt_load(J.res, dt_index(J.res, r, t, b));
                    dt_store(J.out, dt_index(J.out, r, t, b), o);
                    if (o > best_v[n]) { best_v[n] = o; best_i[n] = r; }
                }
            }
        }
        cur = nxt;
    }
    if (J.amax_val) {       // fused greedy sampling, stage 1: per-workgroup (max, first index) of the rows it produced
        __syncthreads();
        float* sv = (float*)smem;
        uint32_t* si = (uint32_t*)(smem + 4 * NB * 4);
        if (lane == 0) {
#pragma unroll
            for (int n = 0; n < NB; ++n) { sv[wave * NB + n] = best_v[n]; si[wave * NB + n] = best_i[n]; }
        }
        __syncthreads();
        if (tid < NB && tok0 + tid < ntok) {
            float bv = sv[tid];
            uint32_t bi = si[tid];
            for (int w = 1; w < 4; ++w) {
                const float v = sv[w * NB + tid];
                const uint32_t i2 = si[w * NB + tid];
                if (v > bv || (v == bv && i2 < bi)) { bv = v; bi = i2; }
            }
            const size_t o = (size_t)(blockIdx.x - J.wg_begin) * ntok + tok0 + tid;
            J.amax_val[o] = bv;
            J.amax_idx[o] = bi;
        }
    }
}

template <int NB>
__global__ void __launch_bounds__(256) matvec_kernel(const MatvecParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_JOBS; ++q)
        if (q < P.njobs && blockIdx.x >= P.jobs[q].wg_begin) ji = q;
    const JobDev& J = P.jobs[ji];
    const bool r16w = (J.flags & WRK_MATRIX_ROUND_F16) != 0;
    switch (J.kind) {
        case WRK_MAT_Q4_K: if (r16w) matvec_body<WRK_MAT_Q4_K, true, NB>(J, smem); else matvec_body<WRK_MAT_Q4_K, false, NB>(J, smem); break;
        case WRK_MAT_Q5_K: if (r16w) matvec_body<WRK_MAT_Q5_K, true, NB>(J, smem); else matvec_body<WRK_MAT_Q5_K, false, NB>(J, smem); break;
        case WRK_MAT_Q6_K: if (r16w) matvec_body<WRK_MAT_Q6_K, true, NB>(J, smem); else matvec_body<WRK_MAT_Q6_K, false, NB>(J, smem); break;
        case WRK_MAT_Q8_0: if (r16w) matvec_body<WRK_MAT_Q8_0, true, NB>(J, smem); else matvec_body<WRK_MAT_Q8_0, false, NB>(J, smem); break;
        default: matvec_body<WRK_MAT_F16, false, NB>(J, smem); break;
    }
}

template <int NB>
static int launch_matvec(hipStream_t s, const MatvecParams& P, uint32_t total_wg, uint32_t tok_groups, size_t smem) {
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)matvec_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    }
    matvec_kernel<NB><<<dim3(total_wg, tok_groups), 256, smem, s>>>(P);
    return 0;
}

uint32_t matvec_num_wg(const MatJob* jobs, int njobs, int num_cu, uint32_t* rows_per_wg) {
    uint32_t total_rows = 0;
    for (int j = 0; j < njobs; ++j) total_rows += jobs[j].m;
    // rows per workgroup: aim at >= 4 workgroups per CU, 4..32 rows (1..8 per wave)
    uint32_t rpw = (total_rows + (uint32_t)num_cu * 4 - 1) / ((uint32_t)num_cu * 4);
    rpw = (rpw + 3) & ~3u;
    rpw = rpw < 4 ? 4 : (rpw > 32 ? 32 : rpw);
    uint32_t wg = 0;
    for (int j = 0; j < njobs; ++j) wg += (jobs[j].m + rpw - 1) / rpw;
    if (rows_per_wg) *rows_per_wg = rpw;
    return wg;
}

// All jobs of one call must have the same number of input vectors (T*B); they run in ONE launch.
int matvec(hipStream_t s, const MatJob* jobs, int njobs, int num_cu) {
    if (njobs <= 0 || njobs > MAX_JOBS) return -1;
    MatvecParams P;
    P.njobs = njobs;
    const uint32_t ntok = jobs[0].in.shape[1] * jobs[0].in.shape[2];
    if (ntok == 0) return 0;
    uint32_t kmax = 0;
    for (int j = 0; j < njobs; ++j) {
        if (jobs[j].in.shape[1] * jobs[j].in.shape[2] != ntok) return -1;
        kmax = jobs[j].k > kmax ? jobs[j].k : kmax;
    }
    uint32_t rpw = 4;
    matvec_num_wg(jobs, njobs, num_cu, &rpw);
    uint32_t wg = 0;
    for (int j = 0; j < njobs; ++j) {
        JobDev& d = P.jobs[j];
        d.w = jobs[j].w; d.kind = jobs[j].kind; d.flags = jobs[j].flags; d.k = jobs[j].k; d.m = jobs[j].m;
        d.row_bytes = jobs[j].row_bytes; d.act = jobs[j].act; d.rows_per_wg = rpw; d.wg_begin = wg;
        d.in = jobs[j].in; d.out = jobs[j].out; d.res = jobs[j].res; d.has_res = jobs[j].has_res;
        d.amax_val = jobs[j].amax_val; d.amax_idx = jobs[j].amax_idx;
        wg += (jobs[j].m + rpw - 1) / rpw;
    }
    const uint32_t kpad = (kmax + 15u) & ~15u;
    // pick inputs-per-pass: LDS budget 144 KiB
    int nb = ntok >= 8 ? 8 : (ntok >= 4 ? 4 : (ntok >= 2 ? 2 : 1));
    while (nb > 1 && (size_t)nb * kpad * 2 + (size_t)nb * (kpad >> 4) * 4 > 144 * 1024) nb >>= 1;
    size_t smem = (size_t)nb * kpad * 2 + (size_t)nb * (kpad >> 4) * 4;
    if (smem < 256) smem = 256;
    const uint32_t groups = (ntok + nb - 1) / nb;
    switch (nb) {
        case 8: return launch_matvec<8>(s, P, wg, groups, smem);
        case 4: return launch_matvec<4>(s, P, wg, groups, smem);
        case 2: return launch_matvec<2>(s, P, wg, groups, smem);
        default: return launch_matvec<1>(s, P, wg, groups, smem);
    }
}

int matmul_mfma(hipStream_t, const MatJob&, int) { return -2; }   // provided by wrk_gemm.hip when built

}  // namespace wrk
