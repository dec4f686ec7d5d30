// Inline-dequant matvec for gfx950: y[M] = act(W[M,K] . x[K]) for 1..8 input vectors per pass.
//
// Replaces (reference file:line)
//   matmul_vec_fp16   ops.rs:697-784   + shaders/matmul_vec_fp16.wgsl
//   matmul_vec_q4k    ops.rs:1403-1467 + shaders/matmul_vec_q4k_v2.wgsl
//   matmul_vec_q5k/q6k/q8_0 ops.rs:1543-1948 -- semantics from gguf.rs:11-37,149-274 (SURVEY F3:
//   the Q5_K/Q6_K/Q8_0 shaders are defective; the CPU dequantisers are canonical)
//
//   matmul_vec_int8 / nf4   ops.rs:791-991   + shaders/matmul_vec_int8.wgsl, matmul_vec_nf4.wgsl (web-rwkv's own formats)
//
// Design: everything is arranged around 16-byte coalesced weight loads straight into VGPRs:
//   * blocks are re-laid-out per row at upload (repack_rows) into 16-byte aligned planes
//     (quants | high bits | scales/header), so lane L of a wave reads the L-th 16-byte chunk of the
//     quant plane: one global_load_dwordx4 per lane covers 1 KiB contiguous per wave-instruction;
//   * one wave64 owns a row; 4 waves (256 threads) per workgroup;
//   * ONE input vector (decode): matvec_reg_kernel -- the inputs a lane multiplies are the same for every row and
//     live in registers; optional fused LN + token-shift prologue (computed once per workgroup, shared through LDS),
//     fused residual / gate / state-carry / arg-max epilogues; long rows split K over the 4 waves;
//   * 2..8 input vectors: matvec_kernel<NB> stages them in LDS as f16 with per-16 partial sums;
//   * the K-quant "min" term is factored:  sum_l (d*sc*q_l - dmin*m) x_l = d*sc * sum_l q_l x_l - dmin*m * sum_l x_l ;
//   * integer codes become f16 with one byte-permute: a code in the low mantissa bits of an f16 is the subnormal
//     code * 2^-24 (exact; the 2^24 is folded into the group scale), then v_dot2_f32_f16 accumulates exact
//     f16*f16 products in f32;
//   * wave64 DPP reduction, fused activation, store in the output buffer's dtype.
// WRK_MATRIX_ROUND_F16 instead rounds every dequantised weight to f16 first, reproducing the
// reference at HEAD (weights dequantised to f16 on the CPU at load, gguf.rs:95-274).
#include <algorithm>
#include <cstdlib>

#include "wrk_device.h"

namespace wrk {

static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

uint32_t int8_row_blocks(uint32_t k) { return (k % 128) ? k / 128 + 2 : k / 128; }

size_t repack_row_bytes(uint32_t kind, uint32_t k) {
    const size_t nb = k / 256;
    switch (kind) {
        case WRK_MAT_F16: return align16((size_t)k * 2);
        case WRK_MAT_Q8_0: return align16((size_t)k + (size_t)(k / 32) * 2);
        case WRK_MAT_Q4_K: return align16(nb * 148);   // quants 128 | d,dmin 4 | unpacked 6-bit scales/mins 16
        case WRK_MAT_Q5_K: return align16(nb * 180);   // quants 128 | high bits 32 | d,dmin 4 | scales/mins 16
        case WRK_MAT_Q6_K: return align16(nb * 208 + nb * 2);
        // codes | (min, max) f16 of every 128-block of the FLATTENED matrix this row touches (k/128 when rows are
        // block aligned, else up to k/128 + 2: the reference's own test multiplies a K = 320 Int8 matrix, ops.rs:3786)
        case WRK_MAT_INT8: return (k % 16) ? 0 : align16((size_t)k + (size_t)int8_row_blocks(k) * 4);
        case WRK_MAT_NF4: return (k % 64) ? 0 : align16((size_t)k / 2 + (size_t)(k / 64) * 2);   // nibbles | absmax f16 per 64
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

// get_scale_min_k4 (gguf.rs:81-89) applied once at upload: 12 packed bytes -> 16 bytes laid out per 64-element
// group j as [sc(2j), sc(2j+1), m(2j), m(2j+1)], so a lane fetches its two (scale, min) pairs with one dword load.
static void unpack_scales_k4(const uint8_t* s, uint8_t* out) {
    for (int j = 0; j < 8; ++j) {
        uint8_t sc, m;
        if (j < 4) { sc = s[j] & 63; m = s[j + 4] & 63; }
        else { sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4); m = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
        out[(j >> 1) * 4 + (j & 1)] = sc;
        out[(j >> 1) * 4 + 2 + (j & 1)] = m;
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
                    memcpy(d + nb * 128 + b * 4, s + b * 144, 4);              // d, dmin
                    unpack_scales_k4(s + b * 144 + 4, d + nb * 132 + b * 16);
                }
                break;
            }
            case WRK_MAT_Q5_K: {
                const uint8_t* s = src + (size_t)r * nb * 176;
                for (size_t b = 0; b < nb; ++b) {
                    memcpy(d + b * 128, s + b * 176 + 48, 128);            // ql
                    memcpy(d + nb * 128 + b * 32, s + b * 176 + 16, 32);   // qh
                    memcpy(d + nb * 160 + b * 4, s + b * 176, 4);          // d, dmin
                    unpack_scales_k4(s + b * 176 + 4, d + nb * 164 + b * 16);
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
            // web-rwkv's formats: `src` = codes of the flattened matrix followed by the side table of the flattened
            // matrix (Matrix::Int8 { w, m } / Matrix::Fp4 { w, q, m }); k % block == 0 makes the side table per-row
            case WRK_MAT_INT8: {
                memcpy(d, src + (size_t)r * k, k);
                const size_t total = (size_t)k * m / 128, first = (size_t)r * k / 128;
                const size_t cnt = std::min<size_t>(int8_row_blocks(k), total - first);
                memcpy(d + k, src + (size_t)m * k + first * 4, cnt * 4);
                break;
            }
            case WRK_MAT_NF4: {
                memcpy(d, src + (size_t)r * (k / 2), k / 2);
                memcpy(d + k / 2, src + (size_t)m * (k / 2) + (size_t)r * (k / 64) * 2, (size_t)(k / 64) * 2);
                break;
            }
            default: break;
        }
    }
    return 0;
}


// ------------------------------------------------------------------ device: code -> f16 helpers
// An integer code c < 1024 placed in the low bits of an f16 lane IS the subnormal c * 2^-24 (subnormals are
// linear in the mantissa), so a byte becomes an exact f16 with one byte-permute and no arithmetic; the 2^24
// (or 2^20 when the code sits in the high nibble, i.e. is 16*q) is folded into the group scale.
// bytes b0..b3 of `v` -> two f16x2 holding (b0, b1) and (b2, b3) as subnormals
__device__ __forceinline__ void bytes_to_h2(uint32_t v, f16x2& lo, f16x2& hi) {
    lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, v, 0x0c010c00u));   // 0x00 b1 0x00 b0
    hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_perm(0u, v, 0x0c030c02u));   // 0x00 b3 0x00 b2
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

__device__ __forceinline__ float f16bits_to_f32(uint32_t bits) { return (float)__builtin_bit_cast(f16, (uint16_t)bits); }

// One decoded 16-element group.  Codes are f16 subnormals (code * qinv); the weight of element e is
//   w_e = scale * (code_e - off) - minv         (ggml: d*sc*q - dmin*m, or d*sc*(q6 - 32), or d*(i8))
// so  sum_e w_e x_e = scale * (qmul * dot(q, x) - off * sum x) - minv * sum x     with qmul = 1 / qinv.
struct Group {
    f16x2 q[8];
    float scale, minv, off, qmul;
    uint32_t xoff;
};

// ROUND_F16: w_e <- f16(scale * (code_e - off) - minv), the exact value the reference stores after CPU dequant
__device__ __forceinline__ void round_group(Group& g) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float a = ((float)g.q[i][0] * g.qmul - g.off) * g.scale - g.minv;
        const float b = ((float)g.q[i][1] * g.qmul - g.off) * g.scale - g.minv;
        g.q[i][0] = (f16)a;
        g.q[i][1] = (f16)b;
    }
}

template <bool R16>
__device__ __forceinline__ float group_dot(Group& g, const f16* __restrict__ xs, const float* __restrict__ xsum) {
    if (R16) {
        round_group(g);
        return dot16(g.q, xs + g.xoff);
    }
    const float sx = xsum[g.xoff >> 4];
    return g.scale * (g.qmul * dot16(g.q, xs + g.xoff) - g.off * sx) - g.minv * sx;
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

// ------------------------------------------------------------------ launch parameters
constexpr int MAX_JOBS = 8;

struct JobDev {
    const uint8_t* w;
    const float* aux;                  // NF4/SF4: the 16 f32 levels of Matrix::Fp4 { q }
    uint32_t kind, flags, k, m, row_bytes, act;
    uint32_t rows_per_wg, wg_begin;    // first workgroup (in x) of this job
    uint32_t has_res;                  // fused residual: out = round_out(act(acc)) + res   (matmul + TensorOp::add)
    DTensor in, out, res;
    float* amax_val;                   // optional fused arg-max partials: [num_wg][ntok] (value, row)
    uint32_t* amax_idx;
    // optional fused prologue (single-token decode only): the input is x_in = mix(LN(in), prev, mixw)  (layer_norm +
    // token_shift REVERSED); the first workgroup of the job also stores LN(in) to ln_out for the later state carry
    uint32_t pro;
    float pro_eps;
    const f16 *ln_w, *ln_b, *mixw;
    const float* prev;
    f16* ln_out;
    // optional fused shift-state carry in the epilogue: carry_dst[row] = carry_src[row] (channel_mix_v7's state write)
    const f16* carry_src;
    float* carry_dst;
    const f16* gate;
    float scale;
    unsigned long long* dbg;
};

struct MatvecParams {
    JobDev jobs[MAX_JOBS];
    int njobs;
};

// ------------------------------------------------------------------ per-kind chunk load / decode
// A chunk is 16 bytes of a row's quant plane.  `load_raw` only ISSUES the global loads of chunk c
// (quants non-temporal: each weight byte is read once per token; side data through the caches),
// `dot_raw` decodes and accumulates -- split so the next chunk's loads are in flight while the
// current one is decoded, and so the first chunk is requested before the inputs are staged.
struct Raw {
    u32x4 w;    // quant chunk
    u32x4 a;    // Q5_K / Q6_K: high bits | Q4_K: a.x = d,dmin  a.y = sc,sc,m,m of the chunk's group | Q8_0: a.x = d bits
    u32x2 b;    // Q5_K: b.x = d,dmin  b.y = sc,sc,m,m | Q6_K: b.x = sc_lo | sc_hi << 8 | d bits << 16
};

template <int KIND>
__device__ __forceinline__ uint32_t num_chunks(uint32_t k, uint32_t kpad) {
    if (KIND == WRK_MAT_INT8) return k >> 4;
    if (KIND == WRK_MAT_NF4) return k >> 5;
    return KIND == WRK_MAT_F16 ? (kpad >> 3) : (KIND == WRK_MAT_Q8_0 ? (k >> 4) : (k >> 8) * 8);
}

template <int KIND>
__device__ __forceinline__ Raw load_raw(const uint8_t* __restrict__ row, uint32_t k, uint32_t c, uint32_t phase = 0) {
    // every offset is 32-bit and relative to the (wave-uniform) row pointer: with a scalar base the loads take the
    // `saddr + voffset` form and the per-lane 64-bit address arithmetic disappears (it was ~1/4 of the kernel's VALU work)
    Raw r;
    r.w = __builtin_nontemporal_load((const u32x4*)(row + c * 16u));
    const uint32_t nb = k >> 8, b = c >> 3;
    if (KIND == WRK_MAT_Q4_K) {
        r.a.x = *(const uint32_t*)(row + (nb * 128u + b * 4u));
        r.a.y = *(const uint32_t*)(row + (nb * 132u + b * 16u + ((c & 7u) >> 1) * 4u));
    } else if (KIND == WRK_MAT_Q5_K) {
        r.a = *(const u32x4*)(row + (nb * 128u + b * 32u + (c & 1u) * 16u));
        r.b.x = *(const uint32_t*)(row + (nb * 160u + b * 4u));
        r.b.y = *(const uint32_t*)(row + (nb * 164u + b * 16u + ((c & 7u) >> 1) * 4u));
    } else if (KIND == WRK_MAT_Q6_K) {
        const uint32_t sub = c & 7u, n128 = sub >> 2, part = (sub >> 1) & 1u, l0 = (sub & 1u) * 16u;
        r.a = *(const u32x4*)(row + (nb * 128u + b * 64u + n128 * 32u + l0));
        const uint8_t* scp = row + (nb * 192u + b * 16u + n128 * 8u + (l0 >> 4) + part * 2u);
        const uint32_t dbits = *(const uint16_t*)(row + (nb * 208u + b * 2u));
        r.b.x = (uint32_t)scp[0] | ((uint32_t)scp[4] << 8) | (dbits << 16);
    } else if (KIND == WRK_MAT_Q8_0) {
        r.a.x = *(const uint16_t*)(row + (k + (c >> 1) * 2u));
    } else if (KIND == WRK_MAT_INT8) {
        r.a.x = *(const uint32_t*)(row + (k + ((c + phase) >> 3) * 4u));   // (min, max) f16 of the 128-block
    } else if (KIND == WRK_MAT_NF4) {
        r.a.x = *(const uint16_t*)(row + ((k >> 1) + (c >> 1) * 2u));     // absmax f16 of the 64-block
    }
    return r;
}

// decode one chunk into one or two 16-element groups (not used for F16)
template <int KIND>
__device__ __forceinline__ void decode_raw(const Raw& r, uint32_t c, Group& lo, Group& hi) {
    const u32x4 w = r.w;
    constexpr float Q24 = 16777216.0f, Q20 = 1048576.0f;       // 2^24, 2^20
    if (KIND == WRK_MAT_Q8_0) {
        bytes_to_h2(w.x ^ 0x80808080u, lo.q[0], lo.q[1]);        // u = int8 + 128
        bytes_to_h2(w.y ^ 0x80808080u, lo.q[2], lo.q[3]);
        bytes_to_h2(w.z ^ 0x80808080u, lo.q[4], lo.q[5]);
        bytes_to_h2(w.w ^ 0x80808080u, lo.q[6], lo.q[7]);
        lo.scale = f16bits_to_f32(r.a.x); lo.minv = 0.0f; lo.off = 128.0f; lo.qmul = Q24; lo.xoff = c * 16;
        return;
    }
    if (KIND == WRK_MAT_INT8) {
        // matmul_vec_int8.wgsl:89-92: w = fma(code / 255, max - min, min)
        bytes_to_h2(w.x, lo.q[0], lo.q[1]);
        bytes_to_h2(w.y, lo.q[2], lo.q[3]);
        bytes_to_h2(w.z, lo.q[4], lo.q[5]);
        bytes_to_h2(w.w, lo.q[6], lo.q[7]);
        const float mn = f16bits_to_f32(r.a.x & 0xffffu), mx = f16bits_to_f32(r.a.x >> 16);
        lo.scale = mx - mn; lo.minv = -mn; lo.off = 0.0f; lo.qmul = Q24 / 255.0f; lo.xoff = c * 16;
        return;
    }
    const uint32_t b = c >> 3, sub = c & 7u;
    if (KIND == WRK_MAT_Q6_K) {
        const uint32_t n128 = sub >> 2, part = (sub >> 1) & 1u, l0 = (sub & 1u) * 16u;
        const u32x4 qh = r.a;
        const float sc_lo = (float)(int8_t)(r.b.x & 0xffu), sc_hi = (float)(int8_t)((r.b.x >> 8) & 0xffu);
        const float d = f16bits_to_f32(r.b.x >> 16);
        const uint32_t s_lo = part * 2, s_hi = s_lo + 4;
        // bits s, s+1 of every byte of H -> bits 4, 5 of the same byte with one rotate by (s - 4) mod 32 and one mask
        const uint32_t r_lo = (s_lo + 28u) & 31u, r_hi = (s_hi + 28u) & 31u;
#define Q6LO(W, H) (((W) & 0x0f0f0f0fu) | (__builtin_amdgcn_alignbit((H), (H), r_lo) & 0x30303030u))
#define Q6HI(W, H) ((((W) >> 4) & 0x0f0f0f0fu) | (__builtin_amdgcn_alignbit((H), (H), r_hi) & 0x30303030u))
        bytes_to_h2(Q6LO(w.x, qh.x), lo.q[0], lo.q[1]);
        bytes_to_h2(Q6LO(w.y, qh.y), lo.q[2], lo.q[3]);
        bytes_to_h2(Q6LO(w.z, qh.z), lo.q[4], lo.q[5]);
        bytes_to_h2(Q6LO(w.w, qh.w), lo.q[6], lo.q[7]);
        bytes_to_h2(Q6HI(w.x, qh.x), hi.q[0], hi.q[1]);
        bytes_to_h2(Q6HI(w.y, qh.y), hi.q[2], hi.q[3]);
        bytes_to_h2(Q6HI(w.z, qh.z), hi.q[4], hi.q[5]);
        bytes_to_h2(Q6HI(w.w, qh.w), hi.q[6], hi.q[7]);
#undef Q6LO
#undef Q6HI
        lo.scale = d * sc_lo; lo.minv = 0.0f; lo.off = 32.0f; lo.qmul = Q24; lo.xoff = b * 256 + n128 * 128 + part * 32 + l0;
        hi.scale = d * sc_hi; hi.minv = 0.0f; hi.off = 32.0f; hi.qmul = Q24; hi.xoff = lo.xoff + 64;
        return;
    }
    // Q4_K / Q5_K: d, dmin and the pre-unpacked (sc, sc', m, m') bytes of this chunk's 64-element group
    const uint32_t j = sub >> 1, h = sub & 1u;
    const uint32_t dd = KIND == WRK_MAT_Q4_K ? r.a.x : r.b.x, sm = KIND == WRK_MAT_Q4_K ? r.a.y : r.b.y;
    const float d = f16bits_to_f32(dd & 0xffffu), dmin = f16bits_to_f32(dd >> 16);
    const float sc0 = (float)(sm & 0xffu), sc1 = (float)((sm >> 8) & 0xffu), m0 = (float)((sm >> 16) & 0xffu), m1 = (float)(sm >> 24);
    if (KIND == WRK_MAT_Q4_K) {
        bytes_to_h2(w.x & 0x0f0f0f0fu, lo.q[0], lo.q[1]);
        bytes_to_h2(w.y & 0x0f0f0f0fu, lo.q[2], lo.q[3]);
        bytes_to_h2(w.z & 0x0f0f0f0fu, lo.q[4], lo.q[5]);
        bytes_to_h2(w.w & 0x0f0f0f0fu, lo.q[6], lo.q[7]);
        bytes_to_h2(w.x & 0xf0f0f0f0u, hi.q[0], hi.q[1]);        // 16 * q, the 1/16 lives in qmul
        bytes_to_h2(w.y & 0xf0f0f0f0u, hi.q[2], hi.q[3]);
        bytes_to_h2(w.z & 0xf0f0f0f0u, hi.q[4], hi.q[5]);
        bytes_to_h2(w.w & 0xf0f0f0f0u, hi.q[6], hi.q[7]);
        hi.qmul = Q20;
    } else {
        const u32x4 qh = r.a;
        const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
        // bit s of every byte of H -> bit 4 of the same byte with ONE rotate (by s - 4 mod 32: the bit that lands on
        // position 8b + 4 is always 8b + s) and one mask; v_and_or_b32 then merges it with the nibble
        const uint32_t r0 = (s0 + 28u) & 31u, r1 = (s1 + 28u) & 31u;
#define Q5LO(W, H) (((W) & 0x0f0f0f0fu) | (__builtin_amdgcn_alignbit((H), (H), r0) & 0x10101010u))
#define Q5HI(W, H) ((((W) >> 4) & 0x0f0f0f0fu) | (__builtin_amdgcn_alignbit((H), (H), r1) & 0x10101010u))
        bytes_to_h2(Q5LO(w.x, qh.x), lo.q[0], lo.q[1]);
        bytes_to_h2(Q5LO(w.y, qh.y), lo.q[2], lo.q[3]);
        bytes_to_h2(Q5LO(w.z, qh.z), lo.q[4], lo.q[5]);
        bytes_to_h2(Q5LO(w.w, qh.w), lo.q[6], lo.q[7]);
        bytes_to_h2(Q5HI(w.x, qh.x), hi.q[0], hi.q[1]);
        bytes_to_h2(Q5HI(w.y, qh.y), hi.q[2], hi.q[3]);
        bytes_to_h2(Q5HI(w.z, qh.z), hi.q[4], hi.q[5]);
        bytes_to_h2(Q5HI(w.w, qh.w), hi.q[6], hi.q[7]);
#undef Q5LO
#undef Q5HI
        hi.qmul = Q24;
    }
    lo.scale = d * sc0; lo.minv = dmin * m0; lo.off = 0.0f; lo.qmul = Q24; lo.xoff = b * 256 + j * 64 + h * 16;
    hi.scale = d * sc1; hi.minv = dmin * m1; hi.off = 0.0f; hi.xoff = lo.xoff + 32;
}

// element offsets of the (up to two) 16-element input groups that chunk c of a row multiplies
template <int KIND>
__device__ __forceinline__ void chunk_xoff(uint32_t c, uint32_t& lo, uint32_t& hi) {
    if (KIND == WRK_MAT_F16) { lo = c * 8; hi = lo; return; }
    if (KIND == WRK_MAT_Q8_0 || KIND == WRK_MAT_INT8) { lo = c * 16; hi = lo; return; }
    if (KIND == WRK_MAT_NF4) { lo = c * 32; hi = lo + 16; return; }
    const uint32_t b = c >> 3, sub = c & 7u;
    if (KIND == WRK_MAT_Q6_K) { lo = b * 256 + (sub >> 2) * 128 + ((sub >> 1) & 1u) * 32 + (sub & 1u) * 16; hi = lo + 64; return; }
    lo = b * 256 + (sub >> 1) * 64 + (sub & 1u) * 16;
    hi = lo + 32;
}

template <int KIND, bool R16, int NB>
__device__ __forceinline__ void dot_raw(const Raw& r, uint32_t c, const f16* xs, const float* xsum, uint32_t kpad, float (&acc)[NB],
                                        const float* __restrict__ levels = nullptr) {
    if (KIND == WRK_MAT_F16) {
        const f16x8 wv = __builtin_bit_cast(f16x8, r.w);
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
    if (KIND == WRK_MAT_NF4) {
        // matmul_vec_nf4.wgsl:47-80: w = level[q] * absmax, f32; nibble i of a dword is element i
        const float amax = f16bits_to_f32(r.a.x);
        const uint32_t wd[4] = {r.w.x, r.w.y, r.w.z, r.w.w};
#pragma unroll
        for (int wi = 0; wi < 4; ++wi) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                wv[i] = levels[(wd[wi] >> (4 * i)) & 15u] * amax;
                if (R16) wv[i] = (float)(f16)wv[i];
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const f16x8 x = *(const f16x8*)(xs + (size_t)n * kpad + c * 32 + wi * 8);
                float a = acc[n];
#pragma unroll
                for (int i = 0; i < 8; ++i) a = fmaf(wv[i], (float)x[i], a);
                acc[n] = a;
            }
        }
        return;
    }
    Group lo, hi;
    decode_raw<KIND>(r, c, lo, hi);
    groups_accumulate<R16, NB>(lo, hi, KIND != WRK_MAT_Q8_0 && KIND != WRK_MAT_INT8, xs, xsum, kpad, acc);
}

// ------------------------------------------------------------------ register-resident inputs (single input vector)
// With one row per wave and lane L owning chunks L, L+64, ... of EVERY row, the inputs a lane multiplies are
// the same for all rows: they are loaded once into registers (no LDS staging, no barrier, no bank conflicts).
struct XRegs {
    f16x8 v[4];     // lo group = v[0..1], hi group = v[2..3]   (F16: v[0] only; Q8_0: v[0..1])
    float s[2];     // sum of the 16 inputs of each group (the K-quant "min" term)
};

__device__ __forceinline__ float sum8(f16x8 a) {
    const f16x2 one = {(f16)1.0f, (f16)1.0f};
    float s = 0.0f;
    s = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 0, 1), one, s, false);
    s = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 2, 3), one, s, false);
    s = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 4, 5), one, s, false);
    s = __builtin_amdgcn_fdot2(__builtin_shufflevector(a, a, 6, 7), one, s, false);
    return s;
}

// only REQUEST the input vectors of chunk c (sums are taken later, by x_sums, so that the weight loads can be issued
// behind these small L2-resident loads and waited for separately: memory returns in issue order)
template <int KIND>
__device__ __forceinline__ XRegs load_x(const f16* __restrict__ x, uint32_t c, bool valid) {
    XRegs r;
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = z;
    r.s[0] = r.s[1] = 0.0f;
    if (!valid) return r;
    uint32_t lo, hi;
    chunk_xoff<KIND>(c, lo, hi);
    r.v[0] = *(const f16x8*)(x + lo);
    if (KIND != WRK_MAT_F16) r.v[1] = *(const f16x8*)(x + lo + 8);
    if (KIND != WRK_MAT_F16 && KIND != WRK_MAT_Q8_0 && KIND != WRK_MAT_INT8) { r.v[2] = *(const f16x8*)(x + hi); r.v[3] = *(const f16x8*)(x + hi + 8); }
    return r;
}

template <int KIND>
__device__ __forceinline__ void x_sums(XRegs& r) {
    if (KIND != WRK_MAT_F16) r.s[0] = sum8(r.v[0]) + sum8(r.v[1]);
    if (KIND != WRK_MAT_F16 && KIND != WRK_MAT_Q8_0 && KIND != WRK_MAT_INT8) r.s[1] = sum8(r.v[2]) + sum8(r.v[3]);
}

__device__ __forceinline__ float dot16r(const f16x2 (&q)[8], const f16x8 xa, const f16x8 xb) {
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

template <bool R16>
__device__ __forceinline__ float group_dot_r(Group& g, const f16x8 xa, const f16x8 xb, float xs16) {
    if (R16) {
        round_group(g);
        return dot16r(g.q, xa, xb);
    }
    return g.scale * (g.qmul * dot16r(g.q, xa, xb) - g.off * xs16) - g.minv * xs16;
}

template <int KIND, bool R16>
__device__ __forceinline__ float dot_raw_reg(const Raw& r, uint32_t c, const XRegs& x) {
    if (KIND == WRK_MAT_F16) {
        const f16x8 wv = __builtin_bit_cast(f16x8, r.w);
        float a = 0.0f;
        a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 0, 1), __builtin_shufflevector(x.v[0], x.v[0], 0, 1), a, false);
        a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 2, 3), __builtin_shufflevector(x.v[0], x.v[0], 2, 3), a, false);
        a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 4, 5), __builtin_shufflevector(x.v[0], x.v[0], 4, 5), a, false);
        a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 6, 7), __builtin_shufflevector(x.v[0], x.v[0], 6, 7), a, false);
        return a;
    }
    Group lo, hi;
    decode_raw<KIND>(r, c, lo, hi);
    float a = group_dot_r<R16>(lo, x.v[0], x.v[1], x.s[0]);
    if (KIND != WRK_MAT_Q8_0 && KIND != WRK_MAT_INT8) a += group_dot_r<R16>(hi, x.v[2], x.v[3], x.s[1]);
    return a;
}

// ------------------------------------------------------------------ the kernel
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

    // rows of this wave: r0 + wave + 4 * ri, processed RB at a time so that RB independent 16-byte loads per
    // lane (RB KiB per wave) are in flight together; chunk c of a row = lane + 64 * ci
    constexpr int RB = NB == 1 ? 4 : (NB == 2 ? 2 : 1);
    const uint32_t nch = num_chunks<KIND>(K, kpad);
    const uint32_t iters = (nch + 63) >> 6;
    const uint32_t nrows = r0 + wave < r1 ? (r1 - r0 - wave + 3) >> 2 : 0;

    Raw raw[RB];
    auto issue = [&](uint32_t ri0, uint32_t ci) {
        const uint32_t c = lane + 64 * ci;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            if (ri0 + rb < nrows && c < nch) {
                const uint32_t r = r0 + wave + 4 * (ri0 + rb);
                // Int8 blocks run over the flattened matrix: the row starts `phase` 16-element chunks into a block
                const uint32_t phase = KIND == WRK_MAT_INT8 ? (uint32_t)((((size_t)r * K) >> 4) & 7u) : 0u;
                raw[rb] = load_raw<KIND>(J.w + (size_t)r * J.row_bytes, K, c, phase);
            }
    };
    // request the first chunks before touching the inputs: weights do not depend on activations
    issue(0, 0);

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
    // NF4 / SF4: the 16 f32 levels go to LDS once per workgroup (a per-element table read from global memory made this
    // kernel 27x slower than Int8: 140 us for a 4096 x 4096 matrix)
    float* lut = xsum + (size_t)NB * (kpad >> 4);
    if (KIND == WRK_MAT_NF4 && tid < 16) lut[tid] = J.aux[tid];
    __syncthreads();

    float best_v[NB];
    uint32_t best_i[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) { best_v[n] = -3.0e38f; best_i[n] = 0xffffffffu; }
    for (uint32_t ri0 = 0; ri0 < nrows; ri0 += RB) {
        float acc[RB][NB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[rb][n] = 0.0f;
        for (uint32_t ci = 0; ci < iters; ++ci) {
            if (ri0 != 0 || ci != 0) issue(ri0, ci);
            const uint32_t c = lane + 64 * ci;
            if (c < nch) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    if (ri0 + rb < nrows) dot_raw<KIND, R16, NB>(raw[rb], c, xs, xsum, kpad, acc[rb], lut);
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (ri0 + rb >= nrows) break;
            const uint32_t r = r0 + wave + 4 * (ri0 + rb);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const float v = wave_sum(acc[rb][n]);
                const uint32_t tk = tok0 + n;
                if (lane == 0 && tk < ntok) {
                    const uint32_t t = tk % J.in.shape[1], b = tk / J.in.shape[1];
                    float o = act_apply(J.act, v * J.scale);
                    if (J.has_res) o = dt_round(J.out, o) + dt_load(J.res, dt_index(J.res, r, t, b));
                    dt_store(J.out, dt_index(J.out, r, t, b), o);
                    if (o > best_v[n]) { best_v[n] = o; best_i[n] = r; }
                }
            }
        }
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

// Decode body: ONE input vector, inputs in registers, RB rows in flight per wave.  No input staging in LDS.
//   KS == 1: a wave owns rows (r0 + wave + 4*i) and walks XI chunk iterations per row (K <= 2048*XI for block kinds)
//   KS == 4: the 4 waves of the workgroup split K (wave w owns chunks 64w..64w+63, XI == 1) and every wave walks
//            ALL rows of the workgroup; the four partial sums per row meet in LDS once, at the end.
template <int KIND, bool R16, int XI, int KS>
__device__ __forceinline__ void matvec_body_reg(const JobDev& J, unsigned char* smem) {
    // rows in flight per wave: everything a wave owns (<= 4 rows for <= 16 rows per workgroup) goes out in ONE round
    // trip -- the in-kernel timeline (WRK_TIMING) showed a second trip costs 1.2 us, a third of the kernel
    constexpr int RB = 4;       // (2 rows for the XI = 2 kernels: fewer VGPRs, 3 waves per SIMD -- measured 2.5 % slower)
    static_assert(KS == 1 || XI <= 2, "K-split: one or two chunk iterations per wave (K <= 16384 for the block kinds)");
    constexpr uint32_t CSTEP = KS == 1 ? 64u : 256u;       // chunk stride between a lane's iterations
    const uint32_t K = J.k;
    const uint32_t kpad = (K + 15u) & ~15u;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r0 = (blockIdx.x - J.wg_begin) * J.rows_per_wg;
    const uint32_t r1 = min(r0 + J.rows_per_wg, J.m);
    const uint32_t nch = num_chunks<KIND>(K, kpad);
    const uint32_t nrows = KS == 1 ? (r0 + wave < r1 ? (r1 - r0 - wave + 3) >> 2 : 0) : (r1 - r0);
    const uint32_t cbase = KS == 1 ? lane : lane + 64 * wave;
    auto row_of = [&](uint32_t ri) { return KS == 1 ? r0 + wave + 4 * ri : r0 + ri; };

    // (tried, round 1: refilling half a batch while the other half is multiplied, for waves that own more than 4 rows.
    // It made every launch slower -- RWKV-6 7B decode 3.00 -> 3.30 ms -- so a batch is issued as a whole.)
    Raw raw[RB][XI];
    auto issue = [&](uint32_t ri0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) {
                const uint32_t c = cbase + CSTEP * ci;
                if (ri0 + rb < nrows && c < nch) {
                    const uint32_t rr = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_of(ri0 + rb));     // wave-uniform: scalar base
                    // Int8 blocks run over the flattened matrix: the row starts `phase` 16-element chunks into a block
                    const uint32_t phase = KIND == WRK_MAT_INT8 ? (uint32_t)((((size_t)rr * K) >> 4) & 7u) : 0u;
                    raw[rb][ci] = load_raw<KIND>(J.w + (size_t)rr * J.row_bytes, K, c, phase);
                }
            }
    };
    WRK_STAMP(J.dbg, 0);
    // Issue order = return order: the small L2-resident operands (input row, LN vectors, residuals) go out FIRST and
    // the weight rows right behind them, so the prologue / input sums run while the weight burst is still arriving
    // and the first row's dot product can start as soon as that row is in (WRK_TIMING: weights-first cost ~1 us).
    const f16* xin = (const f16*)J.in.p + dt_index(J.in, 0, 0, 0);
    XRegs x[XI];
    // epilogue operands: the thread that will finish a row fetches that row's residual / carry value now.
    // KS == 1: lane rb (< RB) of a wave finishes the wave's rb-th row; KS == 4: thread tid finishes row r0 + tid.
    float res_pre = 0.0f, carry_pre = 0.0f, gate_pre = 0.0f;
    {
        const uint32_t r = KS == 1 ? row_of(lane) : r0 + tid;
        const bool mine = KS == 1 ? (lane < (uint32_t)RB && lane < nrows) : tid < nrows;
        if (mine && J.has_res) res_pre = dt_load(J.res, dt_index(J.res, r, 0, 0));
        if (mine && J.carry_dst) carry_pre = (float)J.carry_src[r];
        if (mine && J.gate) gate_pre = (float)J.gate[r];
    }
    if (J.pro) {
        // Fused layer_norm + token_shift prologue, computed ONCE per workgroup: thread t owns elements 8t..8t+7 (and
        // +2048), every global vector (row, LN weight/bias, shift state, mix factor) is requested up front, the two
        // statistics are block reductions, and the shifted f16 input is handed to the waves through LDS.
        constexpr int VPT = 2;                                  // K <= 4096
        f16* xs = (f16*)(smem + 576);
        float* red = (float*)(smem + 544);     // 8 floats
        const uint32_t nvec = K >> 3;
        f16x8 xv[VPT], wv[VPT], bv[VPT], mv[VPT];
        f32x4 pv[VPT][2];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t i = tid + 256 * v;
            if (i < nvec) {
                xv[v] = *(const f16x8*)(xin + i * 8);
                wv[v] = *(const f16x8*)(J.ln_w + i * 8);
                bv[v] = *(const f16x8*)(J.ln_b + i * 8);
                mv[v] = *(const f16x8*)(J.mixw + i * 8);
                pv[v][0] = *(const f32x4*)(J.prev + i * 8);
                pv[v][1] = *(const f32x4*)(J.prev + i * 8 + 4);
            }
        }
        const float c0 = (float)xin[0];
        issue(0);
        // one pass, one block reduction: sums of (x - c) and (x - c)^2 around c = x[0] (any constant is exact in
        // infinite precision; a value of the row keeps the cancellation of E[d^2] - E[d]^2 harmless)
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            if (tid + 256 * v < nvec)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dl = (float)xv[v][e] - c0; s1 += dl; s2 = __builtin_fmaf(dl, dl, s2); }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
        __syncthreads();
        s1 = (red[0] + red[1]) + (red[2] + red[3]);
        s2 = (red[4] + red[5]) + (red[6] + red[7]);
        WRK_STAMP(J.dbg, 4);                    // row arrived, statistics reduced
        const float md = s1 / (float)K;
        const float mean = c0 + md;
        const float dev = 1.0f / sqrtf(fmaxf(s2 / (float)K - md * md, 0.0f) + J.pro_eps);
        const bool publish = J.ln_out && blockIdx.x == J.wg_begin;      // one workgroup stores LN(x) for the state carry
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t i = tid + 256 * v;
            if (i >= nvec) continue;
            f16x8 yv, o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yv[e] = (f16)__builtin_fmaf(((float)xv[v][e] - mean) * dev, (float)wv[v][e], (float)bv[v][e]);
                o[e] = (f16)wgsl_mix((float)yv[e], pv[v][e >> 2][e & 3], (float)mv[v][e]);
            }
            *(f16x8*)(xs + i * 8) = o;
            if (publish) *(f16x8*)(J.ln_out + i * 8) = yv;
        }
        for (uint32_t i = K + tid; i < kpad; i += 256) xs[i] = (f16)0.0f;
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < XI; ++ci) x[ci] = load_x<KIND>(xs, cbase + CSTEP * ci, cbase + CSTEP * ci < nch);
    } else {
#pragma unroll
        for (int ci = 0; ci < XI; ++ci) x[ci] = load_x<KIND>(xin, cbase + CSTEP * ci, cbase + CSTEP * ci < nch);
        issue(0);
    }
#pragma unroll
    for (int ci = 0; ci < XI; ++ci) x_sums<KIND>(x[ci]);

    WRK_STAMP(J.dbg, 1);                        // inputs (and prologue) done
    float* part = (float*)smem;                 // KS == 4: [32 rows][4 waves]
    float best_v = -3.0e38f;
    uint32_t best_i = 0xffffffffu;
    auto finish = [&](uint32_t r, float v, float resv, float carryv, float gatev) {    // activation, fused residual, store, running arg-max
        float o = act_apply(J.act, v * J.scale);
        if (J.gate) o = act_sigmoid(gatev) * dt_round(J.out, o);
        if (J.has_res) o = dt_round(J.out, o) + resv;
        dt_store(J.out, dt_index(J.out, r, 0, 0), o);
        if (J.carry_dst) J.carry_dst[r] = carryv;
        if (o > best_v || (o == best_v && r < best_i)) { best_v = o; best_i = r; }
    };
    for (uint32_t ri0 = 0; ri0 < nrows; ri0 += RB) {
        if (ri0 != 0) issue(ri0);
        float acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            acc[rb] = 0.0f;
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) {
                const uint32_t c = cbase + CSTEP * ci;
                if (ri0 + rb < nrows && c < nch) acc[rb] += dot_raw_reg<KIND, R16>(raw[rb][ci], c, x[ci]);
            }
        }
        WRK_STAMP(J.dbg, 2);                    // weights arrived, dots done
        // the RB reductions are independent chains; then lane rb finishes row rb, so the (long, dependent) epilogue of
        // activation, rounding, residual, store runs ONCE per batch instead of once per row (0.25 us each, WRK_TIMING)
        float mine_v = 0.0f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float v = wave_sum(acc[rb]);
            if (lane == (uint32_t)rb) mine_v = v;
        }
        if (KS == 1) {
            if (lane < (uint32_t)RB && ri0 + lane < nrows) {
                const uint32_t r = row_of(ri0 + lane);
                float resv = res_pre, carryv = carry_pre, gatev = gate_pre;
                if (ri0 != 0) {     // rows beyond the first batch (more than 4 per wave): load on demand
                    if (J.has_res) resv = dt_load(J.res, dt_index(J.res, r, 0, 0));
                    if (J.carry_dst) carryv = (float)J.carry_src[r];
                    if (J.gate) gatev = (float)J.gate[r];
                }
                finish(r, mine_v, resv, carryv, gatev);
            }
        } else if (lane < (uint32_t)RB && ri0 + lane < nrows) part[(ri0 + lane) * 4 + wave] = mine_v;
    }
    if (KS == 4) {
        __syncthreads();
        if (tid < nrows) finish(r0 + tid, (part[tid * 4] + part[tid * 4 + 1]) + (part[tid * 4 + 2] + part[tid * 4 + 3]), res_pre, carry_pre, gate_pre);
    }
    WRK_STAMP(J.dbg, 3);                        // rows reduced and stored
    if (J.amax_val) {       // fused greedy sampling, stage 1 (uniform branch: every wave of the launch takes it)
        float* sv = (float*)(smem + 512);
        uint32_t* si = (uint32_t*)(smem + 528);
        // candidates sit in lanes 0..RB-1 of every wave (KS == 1) or lanes 0..nrows-1 of wave 0 (KS == 4)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best_v, o, WAVE);
            const uint32_t oi = __shfl_xor(best_i, o, WAVE);
            if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
        }
        if (lane == 0) { sv[wave] = best_v; si[wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = sv[0];
            uint32_t bi = si[0];
            for (int w = 1; w < 4; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            J.amax_val[blockIdx.x - J.wg_begin] = bv;
            J.amax_idx[blockIdx.x - J.wg_begin] = bi;
        }
    }
}

template <int KA, int KB, bool R16, int XI, int KS>
__global__ void __launch_bounds__(256) matvec_reg_kernel(const MatvecParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[576 + 2 * 4096];   // partials | arg-max | LN scratch | staged input (prologue)
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_JOBS; ++q)
        if (q < P.njobs && blockIdx.x >= P.jobs[q].wg_begin) ji = q;
    const JobDev& J = P.jobs[ji];
    if (KA == KB || J.kind == (uint32_t)KA) matvec_body_reg<KA, (KA != WRK_MAT_F16) && R16, XI, KS>(J, smem);
    else matvec_body_reg<KB, (KB != WRK_MAT_F16) && R16, (KB == WRK_MAT_F16 ? 4 * XI : XI), 1>(J, smem);
}

// Three kinds in one launch: a K4 kind, Q6_K and F16 -- the r, k, v + LoRA stage of a real Q4_K_M / Q5_K_M file, whose
// attn value is Q6_K in about half of the layers.  (KS == 1 only.)
template <int KA, bool R16, int XI>
__global__ void __launch_bounds__(256) matvec_reg3_kernel(const MatvecParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[576 + 2 * 4096];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_JOBS; ++q)
        if (q < P.njobs && blockIdx.x >= P.jobs[q].wg_begin) ji = q;
    const JobDev& J = P.jobs[ji];
    if (J.kind == (uint32_t)KA) matvec_body_reg<KA, R16, XI, 1>(J, smem);
    else if (J.kind == WRK_MAT_Q6_K) matvec_body_reg<WRK_MAT_Q6_K, R16, XI, 1>(J, smem);
    else matvec_body_reg<WRK_MAT_F16, false, 4 * XI, 1>(J, smem);
}

// One kernel per (inputs-per-pass, kind pair, rounding mode): register allocation is the maximum over
// the code paths a kernel contains, so a launch only carries the decoders its jobs need (a quantised
// kind plus F16 for the LoRA matrices of the same launch).  KA == KB for single-kind launches;
// KA == -1 is the catch-all used when one launch mixes more than two kinds.
template <int NB, int KA, int KB, bool R16>
__global__ void __launch_bounds__(256) matvec_kernel(const MatvecParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_JOBS; ++q)
        if (q < P.njobs && blockIdx.x >= P.jobs[q].wg_begin) ji = q;
    const JobDev& J = P.jobs[ji];
    if (KA >= 0) {
        if (KA == KB || J.kind == (uint32_t)KA) matvec_body<KA, (KA != WRK_MAT_F16) && R16, NB>(J, smem);
        else matvec_body<KB, (KB != WRK_MAT_F16) && R16, NB>(J, smem);
        return;
    }
    const bool r16w = (J.flags & WRK_MATRIX_ROUND_F16) != 0;
    switch (J.kind) {
        case WRK_MAT_Q4_K: if (r16w) matvec_body<WRK_MAT_Q4_K, true, NB>(J, smem); else matvec_body<WRK_MAT_Q4_K, false, NB>(J, smem); break;
        case WRK_MAT_Q5_K: if (r16w) matvec_body<WRK_MAT_Q5_K, true, NB>(J, smem); else matvec_body<WRK_MAT_Q5_K, false, NB>(J, smem); break;
        case WRK_MAT_Q6_K: if (r16w) matvec_body<WRK_MAT_Q6_K, true, NB>(J, smem); else matvec_body<WRK_MAT_Q6_K, false, NB>(J, smem); break;
        case WRK_MAT_Q8_0: if (r16w) matvec_body<WRK_MAT_Q8_0, true, NB>(J, smem); else matvec_body<WRK_MAT_Q8_0, false, NB>(J, smem); break;
        case WRK_MAT_INT8: matvec_body<WRK_MAT_INT8, false, NB>(J, smem); break;
        case WRK_MAT_NF4: matvec_body<WRK_MAT_NF4, false, NB>(J, smem); break;
        default: matvec_body<WRK_MAT_F16, false, NB>(J, smem); break;
    }
}

// ------------------------------------------------------------------ decode matvec, second generation ("dmv")
// Same arithmetic and work split as matvec_body_reg (one input vector, inputs in registers, a wave owns rows, RB rows per
// round trip, KS == 4 splits K over the waves), rebuilt around what the in-kernel timeline and the ISA of the first
// generation showed (round 2):
//   * every global load of the kernel's start-up -- the LN / shift operands, RB x XI weight chunks per lane, the residual /
//     carry / gate operands of the epilogue -- is UNCONDITIONAL (row and chunk indices are clamped, invalid lanes multiply
//     zeros).  Exec-masked loads made the compiler lose count of the outstanding loads and wait `vmcnt(0)`, i.e. for the
//     WHOLE weight burst (~2.2 us), before the layer-norm statistics of the prologue could start, and put a full
//     round trip (`global_load_ushort; s_waitcnt vmcnt(0); v_cvt`) in front of everything for each epilogue operand;
//   * the job's parameters are one compact struct read with a single burst of scalar loads (the first generation re-read
//     pointer and stride per row inside branches: four dependent scalar round trips before the weight loads went out), and
//     the job lookup uses the leading scalar kernel arguments, which gfx950 preloads into SGPRs (amdgpu-kernarg-preload-count);
//   * the prologue is a template parameter (vectors per thread), so launches without one carry no prologue code.
enum { DJ_RES = 1, DJ_RES32 = 2, DJ_CARRY = 4, DJ_GATE = 8, DJ_AMAX = 16, DJ_OUT32 = 32, DJ_PUBLISH = 64 };

struct DJob {
    const uint8_t* w;
    const f16* x;               // dense f16 input [K]
    void* out;                  // dense output, element `row`
    const void* res;            // DJ_RES: residual, element `row` (f16; f32 with DJ_RES32)
    const f16* carry_src;       // DJ_CARRY: carry_dst[row] = carry_src[row]
    float* carry_dst;
    const f16* gate;            // DJ_GATE
    const f16 *ln_w, *ln_b, *mixw;      // prologue: x_in = mix(LN(x), prev, mixw)
    const float* prev;
    f16* ln_out;                // DJ_PUBLISH: the job's first workgroup stores LN(x) here
    float* amax_val;            // DJ_AMAX
    uint32_t* amax_idx;
    unsigned long long* dbg;
    uint32_t k, m, row_bytes, rows_per_wg, wg_begin, act, flags, kind;
    float scale, eps;
};

struct DParams {
    DJob jobs[MAX_JOBS];
};

template <int KIND, bool R16, int XI, int KS, int PRO>
__device__ __forceinline__ void dmv_body(const DJob J, unsigned char* smem) {
    constexpr int RB = 4;
    constexpr uint32_t CSTEP = KS == 1 ? 64u : 256u;
    const uint32_t K = J.k, kpad = (K + 15u) & ~15u;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t r0 = (blockIdx.x - J.wg_begin) * J.rows_per_wg;
    const uint32_t r1 = min(r0 + J.rows_per_wg, J.m);                     // r1 > r0: the host never launches an empty workgroup
    const uint32_t nch = num_chunks<KIND>(K, kpad);
    const uint32_t nrows = KS == 1 ? (r0 + wave < r1 ? (r1 - r0 - wave + 3) >> 2 : 0) : (r1 - r0);
    const uint32_t cbase = KS == 1 ? lane : lane + 64 * wave;
    auto row_of = [&](uint32_t ri) { return KS == 1 ? r0 + wave + 4 * ri : r0 + ri; };
    const uint8_t* __restrict__ W = J.w;
    const uint32_t RBY = J.row_bytes;

    Raw raw[RB][XI];
    auto issue = [&](uint32_t ri0) {        // unconditional: rows / chunks beyond the end are clamped and multiply zeros
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const uint32_t rr = min(row_of(ri0 + rb), r1 - 1);          // wave-uniform -> scalar base
            const uint32_t phase = KIND == WRK_MAT_INT8 ? (uint32_t)((((size_t)rr * K) >> 4) & 7u) : 0u;
            const uint8_t* rowp = W + (size_t)rr * RBY;
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) raw[rb][ci] = load_raw<KIND>(rowp, K, min(cbase + CSTEP * ci, nch - 1), phase);
        }
    };
    WRK_STAMP(J.dbg, 0);

    // ---- (1) every load of the start-up goes out back to back
    const f16* __restrict__ xin = J.x;
    // epilogue operands of the row this thread will finish: KS == 1: lane rb (< RB) finishes the wave's rb-th row of a batch;
    // KS == 4: thread tid finishes row r0 + tid.  Raw bits now, conversion at use.
    const uint32_t fin_row = min(KS == 1 ? row_of(lane & 3u) : r0 + (tid & 31u), r1 - 1);
    const uint32_t fl = J.flags;
    uint32_t res_bits = 0, carry_bits = 0, gate_bits = 0;
    // PRO: 0 none | 1, 2: layer norm + token shift, 1 / 2 vectors per thread (K <= 2048 / 4096) | 3, 4: the post-WKV stage of a
    // split head (group norm over 64-channel heads + time_first bonus + gate), 1 / 2 vectors per thread
    constexpr int VPT = PRO == 0 ? 1 : ((PRO - 1) % 2 + 1);
    constexpr bool GN = PRO >= 3;
    f16x8 xv[VPT], wv[VPT], bv[VPT], mv[VPT];
    f32x4 pv[VPT][2];
    f16 c0h = (f16)0.0f;
    XRegs x[XI];
    if (PRO > 0) {
        const uint32_t nvec = K >> 3;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t i = min(tid + 256u * v, nvec - 1);
            xv[v] = *(const f16x8*)(xin + i * 8);
            wv[v] = *(const f16x8*)(J.ln_w + i * 8);
            bv[v] = *(const f16x8*)(J.ln_b + i * 8);
            mv[v] = *(const f16x8*)(J.mixw + i * 8);
            pv[v][0] = *(const f32x4*)(J.prev + i * 8);
            pv[v][1] = *(const f32x4*)(J.prev + i * 8 + 4);
        }
        if (!GN) c0h = xin[0];
    } else {
#pragma unroll
        for (int ci = 0; ci < XI; ++ci) x[ci] = load_x<KIND>(xin, min(cbase + CSTEP * ci, nch - 1), true);
    }
    issue(0);
    {
        const bool has_res = (fl & DJ_RES) != 0, has_carry = (fl & DJ_CARRY) != 0, has_gate = (fl & DJ_GATE) != 0;
        // absent operands read element 0 of the input vector: always mapped, never used
        const uint16_t* rp = has_res ? (const uint16_t*)J.res : (const uint16_t*)xin;
        const uint32_t ri = has_res ? ((fl & DJ_RES32) ? 2u * fin_row : fin_row) : 0u;
        if (fl & DJ_RES32) res_bits = *(const uint32_t*)(rp + ri);        // uniform branch, one load on either side
        else res_bits = rp[ri];
        carry_bits = (has_carry ? (const uint16_t*)J.carry_src : (const uint16_t*)xin)[has_carry ? fin_row : 0u];
        gate_bits = (has_gate ? (const uint16_t*)J.gate : (const uint16_t*)xin)[has_gate ? fin_row : 0u];
    }

    // ---- (2a) split-head prologue (K3): x_in = g * r16(r16(GN(y)) + tt)  with y = WKV output (f16), tt = (sum_j r_k k r) * v (f32),
    //      g = gate (f16); a head is 64 channels = 8 threads of 8 channels, so the statistics are three DPP steps -- no barrier
    if (GN) {
        f16* xs = (f16*)(smem + 576);
        const uint32_t nvec = K >> 3;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            float y[8], s1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { y[e] = (float)xv[v][e]; s1 += y[e]; }
            s1 += dpp_f32<0xB1>(s1); s1 += dpp_f32<0x4E>(s1); s1 += dpp_f32<0x141>(s1);      // 8-lane sum (quad, quad pair)
            const float mean = s1 * (1.0f / 64.0f);
            float s2 = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { y[e] -= mean; s2 = __builtin_fmaf(y[e], y[e], s2); }
            s2 += dpp_f32<0xB1>(s2); s2 += dpp_f32<0x4E>(s2); s2 += dpp_f32<0x141>(s2);
            const float dev = 1.0f / sqrtf(s2 * (1.0f / 64.0f) + J.eps);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = r16(__builtin_fmaf(y[e] * dev, (float)wv[v][e], (float)bv[v][e]));      // group_norm
                t = r16(t + pv[v][e >> 2][e & 3]);                                                  // time_first_v7
                o[e] = (f16)((float)mv[v][e] * t);                                                  // mul(g, x)
            }
            const uint32_t i = tid + 256u * v;
            if (i < nvec) *(f16x8*)(xs + i * 8) = o;
        }
        for (uint32_t i = K + tid; i < kpad; i += 256) xs[i] = (f16)0.0f;
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < XI; ++ci) x[ci] = load_x<KIND>(xs, min(cbase + CSTEP * ci, nch - 1), true);
    }
    // ---- (2) prologue: layer norm + token shift of the input, once per workgroup, handed to the waves through LDS
    if (PRO > 0 && !GN) {
        f16* xs = (f16*)(smem + 576);
        float* red = (float*)(smem + 544);
        const uint32_t nvec = K >> 3;
        const float c0 = (float)c0h;
        // one pass, one block reduction: sums of (x - c) and (x - c)^2 around c = x[0]
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int v = 0; v < VPT; ++v)
            if (tid + 256u * v < nvec)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dl = (float)xv[v][e] - c0; s1 += dl; s2 = __builtin_fmaf(dl, dl, s2); }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
        __syncthreads();
        s1 = (red[0] + red[1]) + (red[2] + red[3]);
        s2 = (red[4] + red[5]) + (red[6] + red[7]);
        WRK_STAMP(J.dbg, 4);
        const float md = s1 / (float)K;
        const float mean = c0 + md;
        const float dev = 1.0f / sqrtf(fmaxf(s2 / (float)K - md * md, 0.0f) + J.eps);
        const bool publish = (fl & DJ_PUBLISH) && blockIdx.x == J.wg_begin;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t i = tid + 256u * v;
            if (i >= nvec) continue;
            f16x8 yv, o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                yv[e] = (f16)__builtin_fmaf(((float)xv[v][e] - mean) * dev, (float)wv[v][e], (float)bv[v][e]);
                o[e] = (f16)wgsl_mix((float)yv[e], pv[v][e >> 2][e & 3], (float)mv[v][e]);
            }
            *(f16x8*)(xs + i * 8) = o;
            if (publish) *(f16x8*)(J.ln_out + i * 8) = yv;
        }
        for (uint32_t i = K + tid; i < kpad; i += 256) xs[i] = (f16)0.0f;
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < XI; ++ci) x[ci] = load_x<KIND>(xs, min(cbase + CSTEP * ci, nch - 1), true);
    }
    // chunks beyond the row multiply zeros
#pragma unroll
    for (int ci = 0; ci < XI; ++ci)
        if (cbase + CSTEP * ci >= nch) { const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; x[ci].v[0] = x[ci].v[1] = x[ci].v[2] = x[ci].v[3] = z; }
#pragma unroll
    for (int ci = 0; ci < XI; ++ci) x_sums<KIND>(x[ci]);
    WRK_STAMP(J.dbg, 1);

    // ---- (3) dot products, reduction, epilogue
    float* part = (float*)smem;                 // KS == 4: [32 rows][4 waves]
    float best_v = -3.0e38f;
    uint32_t best_i = 0xffffffffu;
    auto finish = [&](uint32_t r, float v, uint32_t rbits, uint32_t cbits, uint32_t gbits) {
        float o = act_apply(J.act, v * J.scale);
        const bool o32 = (fl & DJ_OUT32) != 0;
        if (fl & DJ_GATE) o = act_sigmoid(f16bits_to_f32(gbits)) * (o32 ? o : r16(o));
        if (fl & DJ_RES) o = (o32 ? o : r16(o)) + ((fl & DJ_RES32) ? __builtin_bit_cast(float, rbits) : f16bits_to_f32(rbits));
        if (o32) ((float*)J.out)[r] = o; else ((f16*)J.out)[r] = (f16)o;
        if (fl & DJ_CARRY) J.carry_dst[r] = f16bits_to_f32(cbits);
        if (o > best_v || (o == best_v && r < best_i)) { best_v = o; best_i = r; }
    };
    auto operand_bits = [&](uint32_t r, uint32_t& rbits, uint32_t& cbits, uint32_t& gbits) {      // rows beyond the first batch
        if (fl & DJ_RES) rbits = (fl & DJ_RES32) ? ((const uint32_t*)J.res)[r] : (uint32_t)((const uint16_t*)J.res)[r];
        if (fl & DJ_CARRY) cbits = ((const uint16_t*)J.carry_src)[r];
        if (fl & DJ_GATE) gbits = ((const uint16_t*)J.gate)[r];
    };
    for (uint32_t ri0 = 0; ri0 < nrows; ri0 += RB) {
        if (ri0 != 0) issue(ri0);
        float acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            acc[rb] = 0.0f;
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) acc[rb] += dot_raw_reg<KIND, R16>(raw[rb][ci], min(cbase + CSTEP * ci, nch - 1), x[ci]);
        }
        WRK_STAMP(J.dbg, 2);
        float mine_v = 0.0f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const float v = wave_sum(acc[rb]);
            if (lane == (uint32_t)rb) mine_v = v;
        }
        if (KS == 1) {
            if (lane < (uint32_t)RB && ri0 + lane < nrows) {
                const uint32_t r = row_of(ri0 + lane);
                uint32_t rbits = res_bits, cbits = carry_bits, gbits = gate_bits;
                if (ri0 != 0) operand_bits(r, rbits, cbits, gbits);
                finish(r, mine_v, rbits, cbits, gbits);
            }
        } else if (lane < (uint32_t)RB && ri0 + lane < nrows) part[(ri0 + lane) * 4 + wave] = mine_v;
    }
    if (KS == 4) {
        __syncthreads();
        if (tid < nrows) {
            uint32_t rbits = res_bits, cbits = carry_bits, gbits = gate_bits;
            if (tid >= 32) operand_bits(r0 + tid, rbits, cbits, gbits);
            finish(r0 + tid, (part[tid * 4] + part[tid * 4 + 1]) + (part[tid * 4 + 2] + part[tid * 4 + 3]), rbits, cbits, gbits);
        }
    }
    WRK_STAMP(J.dbg, 3);
    if (fl & DJ_AMAX) {     // fused greedy sampling, stage 1 (uniform branch)
        float* sv = (float*)(smem + 512);
        uint32_t* si = (uint32_t*)(smem + 528);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best_v, o, WAVE);
            const uint32_t oi = __shfl_xor(best_i, o, WAVE);
            if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
        }
        if (lane == 0) { sv[wave] = best_v; si[wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = sv[0];
            uint32_t bi = si[0];
            for (int w = 1; w < 4; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            J.amax_val[blockIdx.x - J.wg_begin] = bv;
            J.amax_idx[blockIdx.x - J.wg_begin] = bi;
        }
    }
}

// b1 .. b7: first workgroup of jobs 1 .. 7 (0xffffffff beyond the last job): LEADING SCALAR arguments, preloaded into SGPRs at wave
// launch, so the job lookup costs no memory access and the job's parameters are the kernel's first (and only) scalar round trip
template <int KA, int KB, bool R16, int XI, int KS, int PRO>
__global__ void __launch_bounds__(256) dmv_kernel(uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7,
                                                  uint32_t kind_b_mask, const DParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[576 + (PRO > 0 ? ((PRO - 1) % 2 + 1) * 4096 : 16)];
    const uint32_t b = blockIdx.x;
    const uint32_t ji = (b >= b1) + (b >= b2) + (b >= b3) + (b >= b4) + (b >= b5) + (b >= b6) + (b >= b7);
    const DJob J = P.jobs[ji];
    if (KA == KB || !((kind_b_mask >> ji) & 1u)) dmv_body<KA, (KA != WRK_MAT_F16) && R16, XI, KS, PRO>(J, smem);
    else dmv_body<KB, (KB != WRK_MAT_F16) && R16, (KB == WRK_MAT_F16 ? 4 * XI : XI), 1, PRO>(J, smem);
}

typedef void (*dmv_fn)(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const DParams);

template <int KA, int KB, int XI, int KS>
static dmv_fn pick_dmv_pro(bool r16, int pro) {
#define DMV_P(PRO_) (r16 ? (dmv_fn)dmv_kernel<KA, KB, true, XI, KS, PRO_> : (dmv_fn)dmv_kernel<KA, KB, false, XI, KS, PRO_>)
    if (KS == 4 || pro == 0) return DMV_P(0);
    if (pro == 1) return DMV_P(1);
    if (pro == 2) return DMV_P(2);
    // the split-head prologue only precedes W_o: a single-kind launch (the F16 pairing is not instantiated for it)
    if (KA != KB) return nullptr;
    return pro == 3 ? DMV_P(3) : DMV_P(4);
#undef DMV_P
}

template <int XI, int KS>
static dmv_fn pick_dmv_kind(int ka, bool has_f16, bool r16, int pro) {
#define DMV_K(A)                                                                              \
    if (ka == A) return (has_f16 && KS == 1) ? pick_dmv_pro<A, WRK_MAT_F16, XI, KS>(r16, pro) : pick_dmv_pro<A, A, XI, KS>(r16, pro);
    DMV_K(WRK_MAT_Q4_K)
    DMV_K(WRK_MAT_Q5_K)
    DMV_K(WRK_MAT_Q6_K)
    DMV_K(WRK_MAT_Q8_0)
    DMV_K(WRK_MAT_INT8)
#undef DMV_K
    if (ka == WRK_MAT_F16) return pick_dmv_pro<WRK_MAT_F16, WRK_MAT_F16, XI, KS>(false, pro);
    return nullptr;
}

// Host side of the dmv kernels: 0 = launched (or would be, dry), -1 = not eligible (the caller falls back to the first-generation kernels)
static int launch_dmv(hipStream_t s, const MatvecParams& P, uint32_t total_wg, int quant, bool has_f16, bool r16, bool dry) {
    static const bool enabled = [] { const char* e = getenv("WRK_DMV"); return !(e && e[0] == '0'); }();
    if (!enabled) return -1;
    DParams D;
    uint32_t bounds[7], bmask = 0, xi = 1;
    int pro = -1;
    bool small_wg = true;
    for (int j = 0; j < 7; ++j) bounds[j] = 0xffffffffu;
    auto dense_base = [](const DTensor& t) { return ((size_t)t.offset[2] * t.stride[1] + t.offset[1]) * t.stride[0] + t.offset[0]; };
    for (int j = 0; j < P.njobs; ++j) {
        const JobDev& J = P.jobs[j];
        if (J.in.dtype != WRK_F16 || (J.k & 7u) || J.in.shape[1] * J.in.shape[2] != 1 || J.kind == WRK_MAT_NF4) return -1;
        if ((J.out.dtype != WRK_F16 && J.out.dtype != WRK_F32) || J.m == 0 || J.rows_per_wg == 0) return -1;
        const size_t ib = dense_base(J.in);
        if (ib & 7u) return -1;
        // MatJob::pro: 1 = layer norm + token shift, 2 = split-head post-WKV stage (group norm + time_first + gate)
        const int jp = J.pro ? (J.k <= 2048 ? 1 : (J.k <= 4096 ? 2 : 9)) + (J.pro == 2 ? 2 : 0) : 0;
        if (jp >= 9 || (pro >= 0 && jp != pro)) return -1;        // one prologue shape per launch
        if (J.pro == 2 && (J.k & 63u)) return -1;
        pro = jp;
        if (J.rows_per_wg > 32) small_wg = false;
        const uint32_t kpad = (J.k + 15u) & ~15u;
        uint32_t nch, need;
        if (J.kind == WRK_MAT_F16) { nch = kpad >> 3; need = quant < 0 ? (nch + 63) / 64 : (nch + 255) / 256; bmask |= (quant < 0 ? 0u : 1u << j); }
        else { nch = (J.kind == WRK_MAT_Q8_0 || J.kind == WRK_MAT_INT8) ? (J.k >> 4) : (J.k >> 8) * 8; need = (nch + 63) / 64; }
        xi = need > xi ? need : xi;
        DJob& d = D.jobs[j];
        memset(&d, 0, sizeof d);
        d.w = J.w;
        d.x = (const f16*)J.in.p + ib;
        const size_t esz = J.out.dtype == WRK_F32 ? 4 : 2;
        d.out = (char*)J.out.p + dense_base(J.out) * esz;
        d.flags = J.out.dtype == WRK_F32 ? DJ_OUT32 : 0u;
        if (J.has_res) {
            if (J.res.dtype != WRK_F16 && J.res.dtype != WRK_F32) return -1;
            d.res = (const char*)J.res.p + dense_base(J.res) * (J.res.dtype == WRK_F32 ? 4 : 2);
            d.flags |= DJ_RES | (J.res.dtype == WRK_F32 ? DJ_RES32 : 0u);
        }
        if (J.carry_dst) { d.carry_src = J.carry_src; d.carry_dst = J.carry_dst; d.flags |= DJ_CARRY; }
        if (J.gate) { d.gate = J.gate; d.flags |= DJ_GATE; }
        if (J.amax_val) { d.amax_val = J.amax_val; d.amax_idx = J.amax_idx; d.flags |= DJ_AMAX; }
        if (J.pro) { d.ln_w = J.ln_w; d.ln_b = J.ln_b; d.mixw = J.mixw; d.prev = J.prev; d.ln_out = J.ln_out; d.eps = J.pro_eps; if (J.ln_out) d.flags |= DJ_PUBLISH; }
        d.dbg = J.dbg;
        d.k = J.k; d.m = J.m; d.row_bytes = J.row_bytes; d.rows_per_wg = J.rows_per_wg; d.wg_begin = J.wg_begin; d.act = J.act; d.kind = J.kind;
        d.scale = J.scale;
        if (j > 0) bounds[j - 1] = J.wg_begin;
    }
    if (xi > 8) return -1;
    // Launches that stream tens of MB are bound by their steady state, not their start-up: there the first-generation kernels are
    // ~3 % faster (RWKV-6 7B Q5_K_M, 40-57 MB per launch: 2.89 vs 2.99 ms per token; the 2.9B model's 12-15 MB launches are 16 %
    // faster on the dmv kernels, round 2).  WRK_DMV_MAXMB moves the line.
    static const size_t max_bytes = [] { const char* e = getenv("WRK_DMV_MAXMB"); return (size_t)(e ? atoi(e) : 24) << 20; }();
    size_t launch_bytes = 0;
    for (int j = 0; j < P.njobs; ++j) launch_bytes += (size_t)P.jobs[j].m * P.jobs[j].row_bytes;
    // (rows of one chunk iteration -- the 1.5B model's 110 MB Q6_K head -- stay: 21.1 vs 27.8 us; the split-head prologue exists here only)
    if (launch_bytes > max_bytes && xi > 1 && pro != 3 && pro != 4) return -1;
    dmv_fn fn = nullptr;
    const int ka = quant < 0 ? WRK_MAT_F16 : quant;
    const bool mixf = has_f16 && quant >= 0;
    if (xi == 1) fn = pick_dmv_kind<1, 1>(ka, mixf, r16, pro);
    else if (xi == 2) fn = pick_dmv_kind<2, 1>(ka, mixf, r16, pro);
    else if (quant < 0 && xi <= 4) fn = pick_dmv_kind<4, 1>(ka, false, false, pro);          // F16 rows up to 2048 elements
    else if (!mixf && small_wg && pro == 0) fn = xi <= 4 ? pick_dmv_kind<1, 4>(ka, false, r16, 0) : pick_dmv_kind<2, 4>(ka, false, r16, 0);   // K over the 4 waves
    if (!fn) return -1;
    if (!dry) hipLaunchKernelGGL(fn, dim3(total_wg, 1), dim3(256), 0, s, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4], bounds[5], bounds[6], bmask, D);
    return 0;
}

// ------------------------------------------------------------------ f32 activation frames (Bundle::<f32>)
// The reference's runtime is generic over the activation type F (v7.rs:281-320, `Runtime<F>`); with F = f32 its matmul
// shaders read f32 inputs (IN_FP32) and multiply them with the decoded weight in f32.  This is that arithmetic: the weight
// is decoded exactly as gguf.rs does (scale * (code - off) - min, each product / difference rounded once; rounded to f16 first
// under WRK_MATRIX_ROUND_F16) and accumulated with fma in f32.  A correctness path (parity tests without f16 stores), not a
// tuned one: one wave per row, inputs re-read from L2 per row.
template <int KIND>
__device__ __forceinline__ float row_dot_f32(const JobDev& J, const uint8_t* __restrict__ row, uint32_t r, size_t xbase, uint32_t lane) {
    const uint32_t K = J.k, kpad = (K + 15u) & ~15u;
    const uint32_t nch = num_chunks<KIND>(K, kpad);
    const bool r16w = (J.flags & WRK_MATRIX_ROUND_F16) != 0;
    const uint32_t phase = KIND == WRK_MAT_INT8 ? (uint32_t)((((size_t)r * K) >> 4) & 7u) : 0u;
    float acc = 0.0f;
    for (uint32_t c = lane; c < nch; c += 64) {
        const Raw raw = load_raw<KIND>(row, K, c, phase);
        if (KIND == WRK_MAT_F16) {
            const f16x8 wv = __builtin_bit_cast(f16x8, raw.w);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c * 8 + e < K) acc = __builtin_fmaf((float)wv[e], dt_load(J.in, xbase + c * 8 + e), acc);
        } else if (KIND == WRK_MAT_NF4) {
            const float amax = f16bits_to_f32(raw.a.x);
            const uint32_t wd[4] = {raw.w.x, raw.w.y, raw.w.z, raw.w.w};
#pragma unroll
            for (int wi = 0; wi < 4; ++wi)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc = __builtin_fmaf(J.aux[(wd[wi] >> (4 * i)) & 15u] * amax, dt_load(J.in, xbase + c * 32 + wi * 8 + i), acc);
        } else {
            Group lo, hi;
            decode_raw<KIND>(raw, c, lo, hi);
            const bool two = KIND != WRK_MAT_Q8_0 && KIND != WRK_MAT_INT8;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (g == 1 && !two) break;
                const Group& G = g ? hi : lo;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float w = ((float)G.q[i][h] * G.qmul - G.off) * G.scale - G.minv;
                        if (r16w) w = r16(w);
                        acc = __builtin_fmaf(w, dt_load(J.in, xbase + G.xoff + 2 * i + h), acc);
                    }
            }
        }
    }
    return wave_sum(acc);
}

__global__ void __launch_bounds__(256) matvec_f32in_kernel(const MatvecParams P) {
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_JOBS; ++q)
        if (q < P.njobs && blockIdx.x >= P.jobs[q].wg_begin) ji = q;
    const JobDev& J = P.jobs[ji];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = (blockIdx.x - J.wg_begin) * J.rows_per_wg, r1 = min(r0 + J.rows_per_wg, J.m);
    const uint32_t tk = blockIdx.y, t = tk % J.in.shape[1], b = tk / J.in.shape[1];
    const size_t xbase = dt_index(J.in, 0, t, b);
    for (uint32_t r = r0 + wave; r < r1; r += 4) {
        const uint8_t* row = J.w + (size_t)r * J.row_bytes;
        float v;
        switch (J.kind) {
            case WRK_MAT_Q4_K: v = row_dot_f32<WRK_MAT_Q4_K>(J, row, r, xbase, lane); break;
            case WRK_MAT_Q5_K: v = row_dot_f32<WRK_MAT_Q5_K>(J, row, r, xbase, lane); break;
            case WRK_MAT_Q6_K: v = row_dot_f32<WRK_MAT_Q6_K>(J, row, r, xbase, lane); break;
            case WRK_MAT_Q8_0: v = row_dot_f32<WRK_MAT_Q8_0>(J, row, r, xbase, lane); break;
            case WRK_MAT_INT8: v = row_dot_f32<WRK_MAT_INT8>(J, row, r, xbase, lane); break;
            case WRK_MAT_NF4: v = row_dot_f32<WRK_MAT_NF4>(J, row, r, xbase, lane); break;
            default: v = row_dot_f32<WRK_MAT_F16>(J, row, r, xbase, lane); break;
        }
        if (lane == 0) {
            float o = act_apply(J.act, v * J.scale);
            if (J.has_res) o = dt_round(J.out, o) + dt_load(J.res, dt_index(J.res, r, t, b));
            dt_store(J.out, dt_index(J.out, r, t, b), o);
        }
    }
}

typedef void (*matvec_fn)(const MatvecParams);

template <int NB, int KA, int KB>
static matvec_fn pick_r16(bool r16) {
    return r16 ? (matvec_fn)matvec_kernel<NB, KA, KB, true> : (matvec_fn)matvec_kernel<NB, KA, KB, false>;
}

template <int NB>
static matvec_fn pick_kernel(int ka, int kb, bool r16) {
    // ka: the quantised kind of the launch (or F16 if none); kb: F16 when LoRA/F16 jobs ride along, else == ka
#define PAIR(A)                                                                     \
    if (ka == A) return kb == A ? pick_r16<NB, A, A>(r16) : pick_r16<NB, A, WRK_MAT_F16>(r16);
    PAIR(WRK_MAT_Q4_K)
    PAIR(WRK_MAT_Q5_K)
    PAIR(WRK_MAT_Q6_K)
    PAIR(WRK_MAT_Q8_0)
#undef PAIR
    // web-rwkv's own formats never round to f16; a dedicated kernel per kind (the all-kinds catch-all allocates registers
    // for every decoder: NF4 ran at 65 GB/s in it)
    if (ka == WRK_MAT_NF4) return kb == ka ? (matvec_fn)matvec_kernel<NB, WRK_MAT_NF4, WRK_MAT_NF4, false> : (matvec_fn)matvec_kernel<NB, WRK_MAT_NF4, WRK_MAT_F16, false>;
    if (ka == WRK_MAT_INT8) return kb == ka ? (matvec_fn)matvec_kernel<NB, WRK_MAT_INT8, WRK_MAT_INT8, false> : (matvec_fn)matvec_kernel<NB, WRK_MAT_INT8, WRK_MAT_F16, false>;
    if (ka == WRK_MAT_F16) return pick_r16<NB, WRK_MAT_F16, WRK_MAT_F16>(false);
    return (matvec_fn)matvec_kernel<NB, -1, -1, false>;
}

template <int KA, int KB, int XI, int KS>
static matvec_fn pick_reg_r16(bool r16) {
    return r16 ? (matvec_fn)matvec_reg_kernel<KA, KB, true, XI, KS> : (matvec_fn)matvec_reg_kernel<KA, KB, false, XI, KS>;
}

template <int XI, int KS>
static matvec_fn pick_reg_kernel(int ka, int kb, bool r16) {
#define PAIR(A)                                                                     \
    if (ka == A) return (kb == A || KS == 4) ? pick_reg_r16<A, A, XI, KS>(r16) : pick_reg_r16<A, WRK_MAT_F16, XI, 1>(r16);
    PAIR(WRK_MAT_Q4_K)
    PAIR(WRK_MAT_Q5_K)
    PAIR(WRK_MAT_Q6_K)
    PAIR(WRK_MAT_Q8_0)
    PAIR(WRK_MAT_INT8)
#undef PAIR
    return nullptr;
}

// register-input decode kernel: one input vector, dense f16 input rows
static matvec_fn pick_reg(const MatvecParams& P, int quant, bool has_f16, bool r16, int quant2 = -1) {
    uint32_t xi = 1;
    for (int j = 0; j < P.njobs; ++j) {
        const JobDev& J = P.jobs[j];
        if (J.in.dtype != WRK_F16 || (J.k & 7u) || J.in.shape[1] * J.in.shape[2] != 1) return nullptr;
        if (J.pro && J.k > 4096) return nullptr;        // the staged input of the LN prologue is sized for K <= 4096
        if (J.pro == 2) return nullptr;                 // the split-head prologue exists in the dmv kernels only
        const size_t base = ((size_t)J.in.offset[2] * J.in.stride[1] + J.in.offset[1]) * J.in.stride[0] + J.in.offset[0];
        if (base & 7u) return nullptr;
        const uint32_t kpad = (J.k + 15u) & ~15u;
        // chunk iterations per row, in units of the QUANTISED kind's count (F16 rows carry 4x the chunks per element)
        uint32_t nch, need;
        if (J.kind == WRK_MAT_F16) { nch = kpad >> 3; need = quant < 0 ? (nch + 63) / 64 : (nch + 255) / 256; }
        else if (J.kind == WRK_MAT_NF4) return nullptr;      // level-table decode lives in the LDS-staged kernel only
        else { nch = (J.kind == WRK_MAT_Q8_0 || J.kind == WRK_MAT_INT8) ? (J.k >> 4) : (J.k >> 8) * 8; need = (nch + 63) / 64; }
        xi = need > xi ? need : xi;
    }
    if (xi > 8) return nullptr;
    bool small_wg = true;       // the K-split kernels combine <= 32 rows per workgroup in LDS
    for (int j = 0; j < P.njobs; ++j) if (P.jobs[j].rows_per_wg > 32) small_wg = false;
    if (quant < 0) {    // F16-only launch
        if (xi == 1) return (matvec_fn)matvec_reg_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 1, 1>;
        if (xi == 2) return (matvec_fn)matvec_reg_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 2, 1>;
        if (xi <= 4) return (matvec_fn)matvec_reg_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 4, 1>;
        return small_wg ? (matvec_fn)matvec_reg_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 2, 4> : nullptr;     // 2048 < K <= 4096 (F16)
    }
    if (quant2 >= 0) {      // (Q4_K | Q5_K) + Q6_K (+ F16), short rows only
        const int ka = quant == WRK_MAT_Q6_K ? quant2 : quant;
        if ((quant != WRK_MAT_Q6_K && quant2 != WRK_MAT_Q6_K) || (ka != WRK_MAT_Q4_K && ka != WRK_MAT_Q5_K) || xi > 2) return nullptr;
        if (ka == WRK_MAT_Q4_K) {
            if (xi == 1) return r16 ? (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q4_K, true, 1> : (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q4_K, false, 1>;
            return r16 ? (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q4_K, true, 2> : (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q4_K, false, 2>;
        }
        if (xi == 1) return r16 ? (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q5_K, true, 1> : (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q5_K, false, 1>;
        return r16 ? (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q5_K, true, 2> : (matvec_fn)matvec_reg3_kernel<WRK_MAT_Q5_K, false, 2>;
    }
    const int kb = has_f16 ? WRK_MAT_F16 : quant;
    if (xi == 1) return pick_reg_kernel<1, 1>(quant, kb, r16);
    if (xi == 2) return pick_reg_kernel<2, 1>(quant, kb, r16);
    if (has_f16 || !small_wg) return nullptr;       // K-split kernels are single-kind
    if (xi <= 4) return pick_reg_kernel<1, 4>(quant, quant, r16);     // 4096 < K <= 8192 for the block kinds: K over the 4 waves
    return pick_reg_kernel<2, 4>(quant, quant, r16);                  // 8192 < K <= 16384
}

template <int NB>
static int launch_matvec(hipStream_t s, const MatvecParams& P, uint32_t total_wg, uint32_t tok_groups, size_t smem, bool dry, bool no_catchall) {
    // classify the kinds of this launch
    int quant = -1, quant2 = -1, nquant = 0;
    bool has_f16 = false, r16 = false, mixed_r16 = false;
    for (int j = 0; j < P.njobs; ++j) {
        const int k = (int)P.jobs[j].kind;
        if (k == WRK_MAT_F16) { has_f16 = true; continue; }
        const bool jr = (P.jobs[j].flags & WRK_MATRIX_ROUND_F16) != 0;
        if (nquant == 0) { quant = k; r16 = jr; nquant = 1; }
        else {
            if (k != quant && k != quant2) { quant2 = quant2 < 0 ? k : quant2; nquant = (k == quant2) ? 2 : 3; }
            if (jr != r16) mixed_r16 = true;
        }
    }
    matvec_fn fn = nullptr;
    bool needs_reg = false;     // fused prologue / state carry exist only in the register-input decode kernel
    for (int j = 0; j < P.njobs; ++j) needs_reg = needs_reg || P.jobs[j].pro || P.jobs[j].carry_dst || P.jobs[j].gate;
    if (NB == 1 && tok_groups == 1 && nquant <= 1 && !mixed_r16 && launch_dmv(s, P, total_wg, nquant ? quant : -1, has_f16, r16, dry) == 0) return 0;
    if (NB == 1 && tok_groups == 1 && nquant <= 2 && !mixed_r16) {
        fn = pick_reg(P, nquant ? quant : -1, has_f16, r16, nquant == 2 ? quant2 : -1);
        if (fn) {
            if (!dry) hipLaunchKernelGGL(fn, dim3(total_wg, 1), dim3(256), 0, s, P);
            return 0;
        }
    }
    if (needs_reg) return -3;
    if (no_catchall && (nquant >= 2 || mixed_r16)) return -4;      // caller splits the jobs by kind instead
    if (dry) return 0;
    if (nquant >= 2 || mixed_r16) fn = (matvec_fn)matvec_kernel<NB, -1, -1, false>;
    else if (nquant == 0) fn = pick_kernel<NB>(WRK_MAT_F16, WRK_MAT_F16, false);
    else fn = pick_kernel<NB>(quant, has_f16 ? WRK_MAT_F16 : quant, r16);
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
    }
    hipLaunchKernelGGL(fn, dim3(total_wg, tok_groups), dim3(256), smem, s, P);
    return 0;
}

// Rows per workgroup of job j: `rpw` for the lightest rows of the launch, fewer for heavier rows so that every
// workgroup moves about the same bytes (an F16 LoRA row is 3.5x a Q4_K row of the same K: with equal row counts the
// F16 workgroups were the stragglers of the launch, WRK_TIMING round 1).
static uint32_t matvec_job_rpw(const MatJob* jobs, int njobs, int j, uint32_t rpw) {
    uint32_t min_rb = jobs[0].row_bytes;
    for (int q = 1; q < njobs; ++q) min_rb = jobs[q].row_bytes < min_rb ? jobs[q].row_bytes : min_rb;
    uint32_t r = (uint32_t)((uint64_t)rpw * min_rb / (jobs[j].row_bytes ? jobs[j].row_bytes : 1)) & ~3u;
    return r < 4 ? 4 : r;
}

uint32_t matvec_num_wg(const MatJob* jobs, int njobs, int num_cu, uint32_t* rows_per_wg) {
    uint32_t total_rows = 0;
    for (int j = 0; j < njobs; ++j) total_rows += jobs[j].m;
    // rows per workgroup: aim at >= 4 workgroups per CU, 4..32 rows (1..8 per wave)
    static const uint32_t per_cu = [] { const char* e = getenv("WRK_WG_PER_CU"); const int v = e ? atoi(e) : 4; return (uint32_t)(v < 1 ? 1 : (v > 16 ? 16 : v)); }();
    uint32_t rpw = (total_rows + (uint32_t)num_cu * per_cu - 1) / ((uint32_t)num_cu * per_cu);
    rpw = (rpw + 3) & ~3u;
    static const uint32_t rpw_min = [] { const char* e = getenv("WRK_RPW_MIN"); const int v = e ? atoi(e) : 4; return (uint32_t)(v < 4 ? 4 : (v > 32 ? 32 : v)) & ~3u; }();
    rpw = rpw < rpw_min ? rpw_min : (rpw > 32 ? 32 : rpw);
    // a launch with the LN prologue pays ~6 vector loads + two block reductions per workgroup: amortise over more rows
    static const uint32_t pro_rpw = [] { const char* e = getenv("WRK_PRO_RPW"); const int v = e ? atoi(e) : 16; return (uint32_t)(v < 4 ? 4 : (v > 32 ? 32 : v)) & ~3u; }();
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].pro && rpw < pro_rpw) rpw = pro_rpw;
    uint32_t wg = 0;
    for (int j = 0; j < njobs; ++j) { const uint32_t r = matvec_job_rpw(jobs, njobs, j, rpw); wg += (jobs[j].m + r - 1) / r; }
    if (rows_per_wg) *rows_per_wg = rpw;
    return wg;
}

// All jobs of one call must have the same number of input vectors (T*B); they run in ONE launch.
int matvec(hipStream_t s, const MatJob* jobs, int njobs, int num_cu, bool dry_run, bool no_catchall) {
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
        d.w = jobs[j].w; d.aux = (const float*)jobs[j].aux; d.kind = jobs[j].kind; d.flags = jobs[j].flags; d.k = jobs[j].k; d.m = jobs[j].m;
        d.row_bytes = jobs[j].row_bytes; d.act = jobs[j].act; d.rows_per_wg = matvec_job_rpw(jobs, njobs, j, rpw); d.wg_begin = wg;
        d.in = jobs[j].in; d.out = jobs[j].out; d.res = jobs[j].res; d.has_res = jobs[j].has_res;
        d.amax_val = jobs[j].amax_val; d.amax_idx = jobs[j].amax_idx;
        d.pro = jobs[j].pro; d.pro_eps = jobs[j].pro_eps; d.ln_w = (const f16*)jobs[j].ln_w; d.ln_b = (const f16*)jobs[j].ln_b;
        d.mixw = (const f16*)jobs[j].mixw; d.prev = jobs[j].prev; d.ln_out = (f16*)jobs[j].ln_out;
        d.carry_src = (const f16*)jobs[j].carry_src; d.carry_dst = jobs[j].carry_dst; d.gate = (const f16*)jobs[j].gate; d.scale = jobs[j].scale; d.dbg = jobs[j].dbg;
        wg += (jobs[j].m + d.rows_per_wg - 1) / d.rows_per_wg;
    }
    bool f32in = false;
    for (int j = 0; j < njobs; ++j) f32in = f32in || jobs[j].in.dtype == WRK_F32;
    if (f32in) {    // Bundle::<f32> frames: the f32-input kernel (no fused prologue / carry / arg-max there)
        for (int j = 0; j < njobs; ++j)
            if (jobs[j].pro || jobs[j].carry_dst || jobs[j].gate || jobs[j].amax_val) return -3;
        if (dry_run) return 0;
        hipLaunchKernelGGL(matvec_f32in_kernel, dim3(wg, ntok), dim3(256), 0, s, P);
        return 0;
    }
    const uint32_t kpad = (kmax + 15u) & ~15u;
    // pick inputs-per-pass: LDS budget 144 KiB
    int nb = ntok >= 8 ? 8 : (ntok >= 4 ? 4 : (ntok >= 2 ? 2 : 1));
    while (nb > 1 && (size_t)nb * kpad * 2 + (size_t)nb * (kpad >> 4) * 4 > 144 * 1024) nb >>= 1;
    size_t smem = (size_t)nb * kpad * 2 + (size_t)nb * (kpad >> 4) * 4 + 64;      // inputs | per-16 sums | NF4 level table
    if (smem < 256) smem = 256;
    const uint32_t groups = (ntok + nb - 1) / nb;
    int rc;
    switch (nb) {
        case 8: rc = launch_matvec<8>(s, P, wg, groups, smem, dry_run, no_catchall); break;
        case 4: rc = launch_matvec<4>(s, P, wg, groups, smem, dry_run, no_catchall); break;
        case 2: rc = launch_matvec<2>(s, P, wg, groups, smem, dry_run, no_catchall); break;
        default: rc = launch_matvec<1>(s, P, wg, groups, smem, dry_run, no_catchall); break;
    }
    return rc;
}

// Jobs of several quantised kinds (a real Q4_K_M file keeps attn value / ffn value in Q6_K for half of the layers): one
// launch per kind -- each on that kind's dedicated kernels, F16 jobs riding with the first -- instead of one launch on the
// all-kinds catch-all kernel, whose register allocation is the maximum over every decoder.
int matvec_grouped(hipStream_t s, const MatJob* jobs, int njobs, int num_cu, bool dry_run) {
    if (njobs <= 0 || njobs > MAX_JOBS) return -1;
    uint32_t kinds[MAX_JOBS];
    int nk = 0;
    for (int j = 0; j < njobs; ++j) {
        if (jobs[j].kind == WRK_MAT_F16) continue;
        bool seen = false;
        for (int q = 0; q < nk; ++q) seen = seen || kinds[q] == jobs[j].kind;
        if (!seen) kinds[nk++] = jobs[j].kind;
    }
    if (nk <= 1) return matvec(s, jobs, njobs, num_cu, dry_run);
    {   // a K4 kind + Q6_K (+ F16) has its own three-kind register kernel: one launch
        const int rc = matvec(s, jobs, njobs, num_cu, dry_run, true);
        if (rc == 0) return 0;
    }
    for (int q = 0; q < nk; ++q) {
        MatJob g[MAX_JOBS];
        int n = 0;
        for (int j = 0; j < njobs; ++j)
            if (jobs[j].kind == kinds[q] || (q == 0 && jobs[j].kind == WRK_MAT_F16)) g[n++] = jobs[j];
        // the job that publishes LN(x) must be in the first group (later launches may already consume it)
        const int rc = matvec(s, g, n, num_cu, dry_run);
        if (rc != 0) return rc;
    }
    return 0;
}

}  // namespace wrk
