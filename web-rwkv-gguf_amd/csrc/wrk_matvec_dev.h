// Device helpers shared by the decode matvec translation units (wrk_matvec.hip: first-generation and LDS-staged kernels, f32 frames;
// wrk_dmv.hip: the second-generation decode kernels): code -> f16 conversion, per-kind chunk load / decode, register-resident inputs,
// and the launch parameter structs.  Split out so the two halves compile in parallel (one TU took four minutes).
#pragma once
#include "wrk_device.h"

namespace wrk {

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
    uint32_t tok_prev_stride, tok_mix_stride, tok_carry_src_stride, tok_carry_dst_stride, tok_gate_stride;     // MatJob's
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


// wrk_dmv.hip: host side of the second-generation kernels; 0 = launched (or would be, dry), -1 = not eligible
int launch_dmv(hipStream_t s, const MatvecParams& P, uint32_t total_wg, int quant, bool has_f16, bool r16, bool dry, int quant2 = -1);

}  // namespace wrk
