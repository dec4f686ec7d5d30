// MFMA inline-dequant GEMM for gfx950: Y[M, N] = act(W[M, K] . X[K, N]) for N >= 16 stacked tokens
// (prefill chunks and batched decode).
//
// Replaces (reference file:line)
//   matmul_mat_fp16              ops.rs:999-1061  + shaders/matmul_mat_fp16.wgsl
//   matmul_mat_q4k(_opt)         ops.rs:1332-1536 + shaders/matmul_mat_q4k_opt.wgsl   (32x32 tile, f32 FMA, no tensor cores)
//   matmul_mat_q5k/q6k/q8_0      ops.rs:1543-1948 -- semantics from gguf.rs:11-37,149-274
//
// Design: v_mfma_f32_16x16x32_f16.  A = a 16-row x 32-k weight fragment, B = 32-k x 16-token activation
// fragment, C/D = 16 rows x 16 tokens in f32.  The weights stay EXACT (ggml-canonical, f32-equivalent) although
// the MFMA operands are f16, by feeding the matrix core small integers and applying the floating scales in f32:
//   Q4_K / Q5_K : A = q * sc        (q <= 31, sc <= 63  ->  <= 1953 < 2048: exact in f16), one 32-k step = one
//                 sub-block; per 256-block epilogue  total += d * acc - sum_s (dmin * m_s) * sum_x(s, token)
//   Q8_0        : A = int8 code, one step = one block; per step  total += d * acc
//   Q6_K        : A = (q6 - 32) * sc with sc = 2*s1 + s0 split over two MFMAs (2*c*s1 even <= 4096 and c*s0 are
//                 exact); the 16-element scale groups sit inside the fragment; per 256-block  total += d * acc
//   F16         : A = the weights, accumulated over the whole K
// Products of two f16 are exact in f32 and MFMA accumulates in f32, so each (row, token) is an f32 dot
// product of the canonical dequantised weights, as in the matvec path.
// Integer codes become f16 without arithmetic (code in the low mantissa bits = subnormal code*2^-24, see
// wrk_matvec.hip); one v_pk_mul_f16 per pair both applies the integer sub-scale and moves them to normal range.
//
// Work split: a workgroup owns 16 rows x (16*NT) tokens; its 4 waves split K (every 4th block / step) and their
// partial tiles meet in LDS; grid = (M/16, N/(16*NT)).
// Lane l: A row = l & 15, k-group g = l >> 4 (8 consecutive k); C column (token) = l & 15, rows 4g..4g+3.
#include "wrk_device.h"

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
    DTensor in, out, res;       // [K, T, B], [M, T, B]
    const float* xsum;          // [N][K/32] sums of 32 consecutive inputs (K4 kinds only)
};

// token index -> (t, b) of the [C, T, B] views
__device__ __forceinline__ void tok_tb(const DTensor& d, uint32_t tok, uint32_t& t, uint32_t& b) { t = tok % d.shape[1]; b = tok / d.shape[1]; }

// per-32 sums of the inputs: xsum[n][s] = sum_{e<32} x[n][32 s + e]
__global__ void __launch_bounds__(256) xsum32_kernel(DTensor in, float* __restrict__ xsum, uint32_t k32) {
    const uint32_t tok = blockIdx.y;
    uint32_t t, b;
    tok_tb(in, tok, t, b);
    const size_t base = dt_index(in, 0, t, b);
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < k32; s += gridDim.x * 256) {
        float acc = 0.0f;
        if (in.dtype == WRK_F16 && ((base & 7u) == 0)) {
            const f16* p = (const f16*)in.p + base + (size_t)s * 32;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f16x8 x = *(const f16x8*)(p + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float)x[e];
            }
        } else {
            for (int e = 0; e < 32; ++e) acc += (float)(f16)dt_load(in, base + (size_t)s * 32 + e);
        }
        xsum[(size_t)tok * k32 + s] = acc;
    }
}

template <int KIND, int NT>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams P) {
    __shared__ float sh_scale[4][16][12];       // per wave: [row][d, mn0..mn7] staged by the row lanes (K4 kinds)
    __shared__ float sh_tot[3][NT][4][64];      // K-split partial sums of waves 1..3
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = lane & 15, g = lane >> 4;
    // a workgroup owns 16 rows; its 4 waves split K (wave w takes every 4th block / step) and meet in LDS
    const uint32_t m0 = blockIdx.x * 16;
    const uint32_t row = min(m0 + r, P.m - 1);
    const uint8_t* wrow = P.w + (size_t)row * P.row_bytes;
    const uint32_t n0 = blockIdx.y * 16 * NT;
    const uint32_t K = P.k, nb = K >> 8;

    // B operand rows: token column c = lane & 15 of each tile
    const f16* xrow[NT];
    bool xlive[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        xlive[t] = tok < P.n;
        uint32_t tt, bb;
        tok_tb(P.in, xlive[t] ? tok : 0, tt, bb);
        xrow[t] = (const f16*)P.in.p + dt_index(P.in, 0, tt, bb);
    }
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto loadB = [&](int t, uint32_t koff) -> f16x8 { return xlive[t] ? *(const f16x8*)(xrow[t] + koff + 8 * g) : zero8; };

    f32x4v total[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    if (KIND == WRK_MAT_F16) {
        const f16* wr = (const f16*)wrow;
        for (uint32_t k0 = 32 * wave; k0 < K; k0 += 128) {
            const f16x8 a = (k0 + 8 * g + 8 <= K) ? *(const f16x8*)(wr + k0 + 8 * g) : zero8;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f16x8 bfr = (k0 + 8 * g + 8 <= K) ? loadB(t, k0) : zero8;
                total[t] = mfma16(a, bfr, total[t]);
            }
        }
    } else if (KIND == WRK_MAT_Q8_0) {
        const uint32_t nblk = K >> 5;
        for (uint32_t s = wave; s < nblk; s += 4) {
            const u32x2 q = *(const u32x2*)(wrow + (size_t)s * 32 + 8 * g);
            // int8 -> (u - 128): subnormal u*2^-24, scaled by 2^15 to u*2^-9 (normal), minus 128*2^-9
            const f16x8 a = add8(mul8(codes8(q.x ^ 0x80808080u, q.y ^ 0x80808080u), 32768.0f), -0.25f);
            // d of the C rows this lane owns: rows 4g..4g+3 of the wave
            float dd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t rr = min(m0 + 4 * g + i, P.m - 1);
                dd[i] = (float)*(const f16*)(P.w + (size_t)rr * P.row_bytes + K + (size_t)s * 2) * 512.0f;   // * 2^9
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4v acc = mfma16(a, loadB(t, s * 32), (f32x4v){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int i = 0; i < 4; ++i) total[t][i] = __builtin_fmaf(dd[i], acc[i], total[t][i]);
            }
        }
    } else if (KIND == WRK_MAT_Q6_K) {
        for (uint32_t b = wave; b < nb; b += 4) {
            f32x4v acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n128 = 0; n128 < 2; ++n128) {
                const u32x2 qh = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 64 + n128 * 32 + 8 * g);
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {      // element group 128*n128 + 32*kq + (0..31)
                    const u32x2 ql = *(const u32x2*)(wrow + (size_t)b * 128 + n128 * 64 + (kq & 1) * 32 + 8 * g);
                    const uint32_t sh = 2 * kq;
                    uint32_t c0, c1;
                    if (kq < 2) { c0 = (ql.x & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = (ql.y & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                    else { c0 = ((ql.x >> 4) & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = ((ql.y >> 4) & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                    // c = code - 32 as c * 2^-9 (normal f16, exact)
                    const f16x8 c = add8(mul8(codes8(c0, c1), 32768.0f), -0.0625f);
                    const int sc = (int)*(const int8_t*)(wrow + (size_t)nb * 192 + (size_t)b * 16 + n128 * 8 + (g >> 1) + 2 * kq);
                    const int s1 = sc >> 1, s0 = sc & 1;            // sc = 2*s1 + s0
                    const f16x8 a1 = mul8(c, (float)(2 * s1)), a0 = mul8(c, (float)s0);
                    const uint32_t koff = b * 256 + n128 * 128 + kq * 32;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x8 bfr = loadB(t, koff);
                        acc[t] = mfma16(a1, bfr, acc[t]);
                        acc[t] = mfma16(a0, bfr, acc[t]);
                    }
                }
            }
            float dd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t rr = min(m0 + 4 * g + i, P.m - 1);
                dd[i] = (float)*(const f16*)(P.w + (size_t)rr * P.row_bytes + (size_t)nb * 208 + (size_t)b * 2) * 512.0f;   // * 2^9
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) total[t][i] = __builtin_fmaf(dd[i], acc[t][i], total[t][i]);
        }
    } else {   // Q4_K / Q5_K
        const uint32_t hoff = KIND == WRK_MAT_Q4_K ? nb * 128 : nb * 160;     // (d, dmin) plane
        const uint32_t soff = hoff + nb * 4;                                   // unpacked scales plane
        const uint32_t k32 = K >> 5;
        for (uint32_t b = wave; b < nb; b += 4) {
            // the 16 row lanes of the wave (g == 0) stage d and dmin*m_s of their row for the C-row owners
            const uint32_t dd16 = *(const uint32_t*)(wrow + hoff + (size_t)b * 4);
            const u32x4 sm = *(const u32x4*)(wrow + soff + (size_t)b * 16);
            const float d = (float)__builtin_bit_cast(f16, (uint16_t)(dd16 & 0xffffu)), dmin = (float)__builtin_bit_cast(f16, (uint16_t)(dd16 >> 16));
            if (g == 0) {
                float* o = sh_scale[wave][r];
                o[0] = d * 16384.0f;                                                   // acc is in units of 2^-14
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t v = sm[j];
                    o[1 + 2 * j] = dmin * (float)((v >> 16) & 0xffu);
                    o[2 + 2 * j] = dmin * (float)(v >> 24);
                }
            }
            u32x2 qh = {0u, 0u};
            if (KIND == WRK_MAT_Q5_K) qh = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 32 + 8 * g);
            f32x4v acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x2 q = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
                const uint32_t v = sm[j];
                const float sc0 = (float)(v & 0xffu), sc1 = (float)((v >> 8) & 0xffu);
                f16x8 alo, ahi;
                if (KIND == WRK_MAT_Q4_K) {
                    alo = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), sc0 * 1024.0f);          // q*sc*2^-14
                    ahi = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), sc1 * 64.0f);            // (16q)*sc*2^-18
                } else {
                    const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
                    alo = mul8(codes8((q.x & 0x0f0f0f0fu) | (((qh.x >> s0) & 0x01010101u) << 4), (q.y & 0x0f0f0f0fu) | (((qh.y >> s0) & 0x01010101u) << 4)), sc0 * 1024.0f);
                    ahi = mul8(codes8(((q.x >> 4) & 0x0f0f0f0fu) | (((qh.x >> s1) & 0x01010101u) << 4), ((q.y >> 4) & 0x0f0f0f0fu) | (((qh.y >> s1) & 0x01010101u) << 4)), sc1 * 1024.0f);
                }
                const uint32_t koff = b * 256 + j * 64;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = mfma16(alo, loadB(t, koff), acc[t]);
                    acc[t] = mfma16(ahi, loadB(t, koff + 32), acc[t]);
                }
            }
            // epilogue of the block: rows 4g..4g+3, token column r
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS stores of this wave have landed (same wave, in order)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t tok = n0 + 16 * t + r;
                float xs[8];
                if (tok < P.n) {
                    const f32x4 x0 = *(const f32x4*)(P.xsum + (size_t)tok * k32 + b * 8), x1 = *(const f32x4*)(P.xsum + (size_t)tok * k32 + b * 8 + 4);
                    xs[0] = x0[0]; xs[1] = x0[1]; xs[2] = x0[2]; xs[3] = x0[3]; xs[4] = x1[0]; xs[5] = x1[1]; xs[6] = x1[2]; xs[7] = x1[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xs[e] = 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* o = sh_scale[wave][4 * g + i];
                    float mins = 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) mins = __builtin_fmaf(o[1 + e], xs[e], mins);
                    total[t][i] += o[0] * acc[t][i] - mins;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // combine the four K slices
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) sh_tot[wave - 1][t][i][lane] = total[t][i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) total[t][i] += (sh_tot[0][t][i][lane] + sh_tot[1][t][i][lane]) + sh_tot[2][t][i][lane];

    // store: lane owns rows m0 + 4g + (0..3) of token column r of each tile
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n) continue;
        uint32_t tt, bb;
        tok_tb(P.out, tok, tt, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t mr = m0 + 4 * g + i;
            if (mr >= P.m) continue;
            float o = act_apply(P.act, total[t][i]);
            if (P.has_res) { uint32_t rt, rb; tok_tb(P.res, tok, rt, rb); o = dt_round(P.out, o) + dt_load(P.res, dt_index(P.res, mr, rt, rb)); }
            dt_store(P.out, dt_index(P.out, mr, tt, bb), o);
        }
    }
}

template <int KIND>
static void launch_gemm(hipStream_t s, const GemmParams& P) {
    const uint32_t gx = (P.m + 15) / 16;
    // tokens per wave: enough tiles to amortise the decode, few enough to keep >= ~2 waves per SIMD
    if (P.n > 64) gemm_kernel<KIND, 4><<<dim3(gx, (P.n + 63) / 64), 256, 0, s>>>(P);
    else if (P.n > 16) gemm_kernel<KIND, 2><<<dim3(gx, (P.n + 31) / 32), 256, 0, s>>>(P);
    else gemm_kernel<KIND, 1><<<dim3(gx, (P.n + 15) / 16), 256, 0, s>>>(P);
}

// returns -2 when this job is not for the MFMA path (caller falls back to the matvec kernels)
int matmul_mfma(hipStream_t s, const MatJob& j, int, float* xsum_scratch, size_t xsum_cap) {
    const uint32_t n = j.in.shape[1] * j.in.shape[2];
    if (n < 16) return -2;
    if (j.flags & WRK_MATRIX_ROUND_F16) return -2;          // parity mode: per-element f16 rounding lives in the matvec kernels
    if (j.in.dtype != WRK_F16 || (j.k & 31u)) return -2;
    // rows of the input views must be 16-byte aligned for the B-fragment loads
    if ((j.in.stride[0] & 7u) || (j.in.offset[0] & 7u)) return -2;
    GemmParams P;
    P.w = j.w; P.kind = j.kind; P.k = j.k; P.m = j.m; P.row_bytes = j.row_bytes; P.act = j.act; P.n = n;
    P.has_res = j.has_res; P.in = j.in; P.out = j.out; P.res = j.res; P.xsum = nullptr;
    switch (j.kind) {
        case WRK_MAT_F16: launch_gemm<WRK_MAT_F16>(s, P); return 0;
        case WRK_MAT_Q8_0: launch_gemm<WRK_MAT_Q8_0>(s, P); return 0;
        case WRK_MAT_Q6_K: launch_gemm<WRK_MAT_Q6_K>(s, P); return 0;
        case WRK_MAT_Q4_K:
        case WRK_MAT_Q5_K: {
            const uint32_t k32 = j.k >> 5;
            if (!xsum_scratch || (size_t)n * k32 > xsum_cap) return -2;
            xsum32_kernel<<<dim3((k32 + 255) / 256, n), 256, 0, s>>>(j.in, xsum_scratch, k32);
            P.xsum = xsum_scratch;
            if (j.kind == WRK_MAT_Q4_K) launch_gemm<WRK_MAT_Q4_K>(s, P); else launch_gemm<WRK_MAT_Q5_K>(s, P);
            return 0;
        }
        default: return -2;
    }
}

}  // namespace wrk
