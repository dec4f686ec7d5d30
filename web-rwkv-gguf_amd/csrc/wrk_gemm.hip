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
//                 sub-block; the min term rides the matrix core too: a second MFMA per step with A = m_s (the 6-bit
//                 min, splat over the fragment) accumulates m_s * sum_k x_k, so per 256-block
//                 total += d * acc - dmin * acc_min   (no input-sum pre-pass, no LDS staging of scales)
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
// partial tiles meet in LDS; grid = (sum_jobs M_j/16, N/(16*NT)): several matrices that multiply the same number of
// tokens (r, k, v and the LoRA down-projections of a layer) run in ONE launch.  K4 blocks are software-pipelined:
// all loads of the next 256-block (weights, scales, B fragments) are in flight while the current one is multiplied.
// Lane l: A row = l & 15, k-group g = l >> 4 (8 consecutive k); C column (token) = l & 15, rows 4g..4g+3.
#include <cstdlib>

#include "wrk_device.h"

#include "wrk_gemm_dev.h"

namespace wrk {

// Residual operands of the C elements a lane stores: raw bits, requested with the first loads of the kernel.  UNCONDITIONAL: a launch
// without residual (or with the other element type) reads a mapped dummy (the first activations).  The branchy form
// (`if (has_res) { if (res32) load4 else load2 }`) made the compiler wait `vmcnt(0)` at the join -- in the K-sliced kernel that was the whole
// weight burst before the activations could be staged, in the K-split kernel one extra round trip at the head of W_o / ffn.value.
template <int NT>
__device__ __forceinline__ void load_residual(const GemmParams& P, uint32_t n0, uint32_t m0, uint32_t r, uint32_t g, uint32_t (&resb)[NT][4]) {
    const bool r32 = P.has_res && P.res32, r16b = P.has_res && !P.res32;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const size_t ro = (size_t)min(n0 + 16 * t + r, P.n - 1) * P.rs + min(m0 + 4 * g, P.m - 4u);
        const u32x4 v4 = *(const u32x4*)(r32 ? (const void*)((const float*)P.res_p + ro) : (const void*)P.x);
        const u32x2 v2 = *(const u32x2*)(r16b ? (const void*)((const f16*)P.res_p + ro) : (const void*)P.x);
        resb[t][0] = r32 ? v4.x : (v2.x & 0xffffu);
        resb[t][1] = r32 ? v4.y : (v2.x >> 16);
        resb[t][2] = r32 ? v4.z : (v2.y & 0xffffu);
        resb[t][3] = r32 ? v4.w : (v2.y >> 16);
    }
}

// Round 2: every global load of this kernel is UNCONDITIONAL (token, row, block and k indices are clamped; dead blocks are
// multiplied by zero scales, dead token columns are never stored) and every wave of a workgroup runs the same trip count.
// The first version predicated its loads (`live ? load : zero`, `if (u < nmine) load_w`); the compiler then loses count of
// the outstanding loads and waits `vmcnt(0)` in front of every use -- the ISA had one full memory round trip per 256-block.
// Tensors are dense token stacks (checked on the host): token tok of the input / output / residual is `tok * stride`
// elements from the base.
// NR (round 3): row tiles per workgroup.  With 16 tokens every 16-row workgroup pulls the whole activation stack (64 KB) next to 18 KB of weights;
// NR = 2 multiplies two row tiles with the same B fragments (K4 kinds and F16 only).  An experiment that lost (see the launch site): off by default.
template <int KIND, int NT, int NW, int NR = 1>
__device__ __forceinline__ void gemm_body(const GemmParams& P, float (*sh_tot)[NR * NT][4][64]) {
    static_assert(NR == 1 || KIND == WRK_MAT_Q4_K || KIND == WRK_MAT_Q5_K || KIND == WRK_MAT_F16, "two row tiles: K4 kinds and F16");
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * 16 * NR;
    WRK_STAMP(P.dbg, 0);
    const uint8_t* wrows[NR];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) wrows[rt] = P.w + (size_t)min(m0 + 16u * rt + r, P.m - 1) * P.row_bytes;
    const uint8_t* wrow = wrows[0];
    const uint32_t n0 = blockIdx.y * 16 * NT;
    const uint32_t K = P.k, nb = K >> 8;

    // B operand rows: token column c = lane & 15 of each tile (clamped: a dead column computes garbage that is never stored)
    const f16* xrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xrow[t] = P.x + (size_t)min(n0 + 16 * t + r, P.n - 1) * P.xs + 8 * g;
    auto loadB = [&](int t, uint32_t koff) -> f16x8 { return *(const f16x8*)(xrow[t] + koff); };

    f32x4v totals[NR][NT];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) totals[rt][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    f32x4v (&total)[NT] = totals[0];

    // residual operands of the C elements this lane stores (wave 0 does the epilogue): raw bits, requested with the first weights
    uint32_t resbs[NR][NT][4];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) resbs[rt][t][i] = 0;
        load_residual<NT>(P, n0, m0 + 16u * rt, r, g, resbs[rt]);
    }

    if (KIND == WRK_MAT_F16) {
        // this wave's 32-k steps are wave, wave + NW, ...; FB steps' fragments are requested together.  Steps beyond K are
        // clamped to the last full fragment and their A fragment zeroed (a select, not a branch).
        constexpr int FB = 8;
        const uint32_t nsteps = (K + 31) >> 5, iters = (nsteps + NW - 1) / NW;
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t klast = K - 8;
        for (uint32_t i0 = 0; i0 < iters; i0 += FB) {
            f16x8 a[NR][FB], bf[NT <= 2 ? NT : 1][FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const uint32_t k0 = 32 * (wave + NW * (i0 + u)) + 8 * g;
                const bool ok = i0 + u < iters && k0 + 8 <= K;
                const uint32_t kc = min(k0, klast);
#pragma unroll
                for (int rt = 0; rt < NR; ++rt) {
                    const f16x8 av = *(const f16x8*)((const f16*)wrows[rt] + kc);
                    a[rt][u] = ok ? av : zero8;
                }
                if (NT <= 2) {
#pragma unroll
                    for (int t = 0; t < (NT <= 2 ? NT : 1); ++t) bf[t][u] = *(const f16x8*)(xrow[t] - 8 * g + kc);
                }
            }
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const uint32_t kc = min(32 * (wave + NW * (i0 + u)) + 8 * g, klast);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f16x8 bfr = NT <= 2 ? bf[NT <= 2 ? t : 0][u] : *(const f16x8*)(xrow[t] - 8 * g + kc);
#pragma unroll
                    for (int rt = 0; rt < NR; ++rt) totals[rt][t] = mfma16(a[rt][u], bfr, totals[rt][t]);
                }
            }
        }
    } else if (KIND == WRK_MAT_Q8_0) {
        const uint32_t nblk = K >> 5, iters = (nblk + NW - 1) / NW;
        const uint8_t* drow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) drow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + K;
        for (uint32_t it = 0; it < iters; ++it) {
            const uint32_t s0 = wave + NW * it, s = min(s0, nblk - 1);
            const u32x2 q = *(const u32x2*)(wrow + (size_t)s * 32 + 8 * g);
            uint32_t db[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) db[i] = *(const uint16_t*)(drow[i] + (size_t)s * 2);
            f16x8 bfr[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bfr[t] = loadB(t, s * 32);
            // int8 -> (u - 128): subnormal u*2^-24, scaled by 2^15 to u*2^-9 (normal), minus 128*2^-9
            const f16x8 a = add8(mul8(codes8(q.x ^ 0x80808080u, q.y ^ 0x80808080u), 32768.0f), -0.25f);
            const float livef = s0 < nblk ? 512.0f : 0.0f;                       // * 2^9; a dead step contributes nothing
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4v acc = mfma16(a, bfr[t], (f32x4v){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int i = 0; i < 4; ++i) total[t][i] = __builtin_fmaf((float)__builtin_bit_cast(f16, (uint16_t)db[i]) * livef, acc[i], total[t][i]);
            }
        }
    } else if (KIND == WRK_MAT_INT8) {
        // web-rwkv Int8 (matmul_mat_int8, ops.rs:1072-1146): w = code / 255 * (max - min) + min per 128 elements.  The codes
        // (0..255, exact in f16) and a fragment of ones go through the matrix core; per 128-block
        //   total += (max - min) / 255 * sum(code * x) + min * sum(x).   Rows are block aligned (K % 128 == 0, host-checked).
        const uint32_t nblk = K >> 7, iters = (nblk + NW - 1) / NW;
        const f16 one = (f16)1.0f;
        const f16x8 ones = {one, one, one, one, one, one, one, one};
        const uint8_t* mrow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mrow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + K;
        for (uint32_t it = 0; it < iters; ++it) {
            const uint32_t b0 = wave + NW * it, b = min(b0, nblk - 1);
            f32x4v acc[NT], asum[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; asum[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
            u32x2 q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
            uint32_t mm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) mm[i] = *(const uint32_t*)(mrow[i] + (size_t)b * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x8 a = mul8(codes8(q[j].x, q[j].y), 32768.0f);       // code * 2^-9, exact
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f16x8 bfr = loadB(t, b * 128 + j * 32);
                    acc[t] = mfma16(a, bfr, acc[t]);
                    asum[t] = mfma16(ones, bfr, asum[t]);
                }
            }
            const float livef = b0 < nblk ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mn = (float)__builtin_bit_cast(f16, (uint16_t)(mm[i] & 0xffffu)) * livef, mx = (float)__builtin_bit_cast(f16, (uint16_t)(mm[i] >> 16)) * livef;
                const float sc = (mx - mn) * (512.0f / 255.0f);
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t][i] += sc * acc[t][i] + mn * asum[t][i];
            }
        }
    } else if (KIND == WRK_MAT_NF4) {
        // web-rwkv NF4 / SF4 (matmul_mat_nf4, ops.rs:1150-1222): w = level[q] * absmax per 64 elements, levels = 16 f32 of Matrix::Fp4 { q }.
        // The levels are not integers, so each goes through the matrix core as hi + lo with hi = f16(level), lo = f16(level - hi): two
        // MFMAs per 32-k step, products exact, level error 2^-22 relative; absmax is applied to the f32 sum per 64-block.  The (hi, lo)
        // pairs sit in LDS (one dword per level); a lane looks up its 8 nibbles per step.
        uint32_t* lut = (uint32_t*)&sh_tot[0][0][0][0] + (NW - 1) * NT * 4 * 64;       // 16 dwords behind the K-split partials
        if (threadIdx.x < 16) {
            const float lv = P.levels[threadIdx.x];
            const f16 hi = (f16)lv, lo = (f16)(lv - (float)hi);
            lut[threadIdx.x] = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
        }
        __syncthreads();
        const uint32_t nblk = K >> 6, iters = (nblk + NW - 1) / NW;
        const uint8_t* arow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) arow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + (K >> 1);      // absmax f16 per 64
        for (uint32_t it = 0; it < iters; ++it) {
            const uint32_t b0 = wave + NW * it, b = min(b0, nblk - 1);
            uint32_t qw[2], am[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) qw[h] = *(const uint32_t*)(wrow + (size_t)b * 32 + h * 16 + 4 * g);      // 8 nibbles = this lane's 8 k of step h
#pragma unroll
            for (int i = 0; i < 4; ++i) am[i] = *(const uint16_t*)(arow[i] + (size_t)b * 2);
            f16x8 bfr[2][NT];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t) bfr[h][t] = loadB(t, b * 64 + h * 32);
            f32x4v acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 ahi, alo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t pr = lut[(qw[h] >> (4 * e)) & 15u];
                    ahi[e] = __builtin_bit_cast(f16, (uint16_t)(pr & 0xffffu));
                    alo[e] = __builtin_bit_cast(f16, (uint16_t)(pr >> 16));
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = mfma16(ahi, bfr[h][t], acc[t]);
                    acc[t] = mfma16(alo, bfr[h][t], acc[t]);
                }
            }
            const float livef = b0 < nblk ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = (float)__builtin_bit_cast(f16, (uint16_t)am[i]) * livef;
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t][i] = __builtin_fmaf(a, acc[t][i], total[t][i]);
            }
        }
        __syncthreads();        // the table shares the partial-sum array's allocation tail: done with it before the combine writes
    } else if (KIND == WRK_MAT_Q6_K) {
        const uint32_t iters = (nb + NW - 1) / NW;
        const uint8_t* drow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) drow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + (size_t)nb * 208;
        struct W6 { u32x2 ql[4]; u32x2 qh[2]; u32x4 sc; uint32_t d[4]; };
        auto load_w6 = [&](W6& R, uint32_t b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) R.ql[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);       // j = 2 n128 + (kq & 1)
#pragma unroll
            for (int h = 0; h < 2; ++h) R.qh[h] = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 64 + h * 32 + 8 * g);
            R.sc = *(const u32x4*)(wrow + (size_t)nb * 192 + (size_t)b * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) R.d[i] = *(const uint16_t*)(drow[i] + (size_t)b * 2);
        };
        const uint32_t gsh = 8 * (g >> 1);       // scale byte of this lane's 16-element half of a 32-group
        W6 Wc, Wn;
        load_w6(Wc, min(wave, nb - 1));
        for (uint32_t it = 0; it < iters; ++it) {
            const uint32_t b0 = wave + NW * it, b = min(b0, nb - 1);
            load_w6(Wn, min(b0 + NW, nb - 1));          // next block's weights in flight while this one is multiplied
            f32x4v acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n128 = 0; n128 < 2; ++n128) {
                const u32x2 qh = Wc.qh[n128];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {      // element group 128*n128 + 32*kq + (0..31)
                    const u32x2 ql = Wc.ql[2 * n128 + (kq & 1)];
                    const uint32_t sh = 2 * kq;
                    uint32_t c0, c1;
                    if (kq < 2) { c0 = (ql.x & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = (ql.y & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                    else { c0 = ((ql.x >> 4) & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = ((ql.y >> 4) & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                    // c = code - 32 as c * 2^-9 (normal f16, exact)
                    const f16x8 c = add8(mul8(codes8(c0, c1), 32768.0f), -0.0625f);
                    const uint32_t word = Wc.sc[2 * n128 + (kq >> 1)];
                    const int sc = (int)(int8_t)((word >> (16 * (kq & 1) + gsh)) & 0xffu);
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
            const float livef = b0 < nb ? 512.0f : 0.0f;      // * 2^9
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dd = (float)__builtin_bit_cast(f16, (uint16_t)Wc.d[i]) * livef;
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t][i] = __builtin_fmaf(dd, acc[t][i], total[t][i]);
            }
            Wc = Wn;
        }
    } else {   // Q4_K / Q5_K
        const uint32_t hoff = KIND == WRK_MAT_Q4_K ? nb * 128 : nb * 160;     // (d, dmin) plane
        const uint32_t soff = hoff + nb * 4;                                   // unpacked scales plane
        struct WBlk {
            u32x2 q[4];             // quant bytes of this lane's 8 k per 64-element step
            u32x2 qh;               // Q5_K high bits
            u32x4 sm;               // (sc, sc', m, m') bytes of the row's four 64-element groups
            uint32_t dd[4];         // (d, dmin) of the four C rows this lane owns
        };
        struct BBlk { f16x8 bf[NT <= 2 ? NT : 1][8]; };   // B fragments of the eight 32-k sub-blocks
        const uint8_t* crows[NR][4];
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) crows[rt][i] = P.w + (size_t)min(m0 + 16u * rt + 4 * g + i, P.m - 1) * P.row_bytes + hoff;
        auto load_w1 = [&](WBlk& R, uint32_t b, int rt) {
            const uint8_t* wr = wrows[rt];
#pragma unroll
            for (int j = 0; j < 4; ++j) R.q[j] = *(const u32x2*)(wr + (size_t)b * 128 + j * 32 + 8 * g);
            if (KIND == WRK_MAT_Q5_K) R.qh = *(const u32x2*)(wr + (size_t)nb * 128 + (size_t)b * 32 + 8 * g);
            R.sm = *(const u32x4*)(wr + soff + (size_t)b * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) R.dd[i] = *(const uint32_t*)(crows[rt][i] + (size_t)b * 4);
        };
        auto load_w = [&](WBlk (&R)[NR], uint32_t b) {
#pragma unroll
            for (int rt = 0; rt < NR; ++rt) load_w1(R[rt], b, rt);
        };
        auto load_b = [&](BBlk& R, uint32_t b) {
#pragma unroll
            for (int t = 0; t < (NT <= 2 ? NT : 1); ++t)
#pragma unroll
                for (int sb = 0; sb < 8; ++sb) R.bf[t][sb] = loadB(t, b * 256 + sb * 32);
        };
        // NT > 2 (prefill): B fragments are fetched per 64-k step instead (a whole block of them would be 128 VGPRs)
        auto mul_blk1 = [&](const WBlk& R, const BBlk& X, uint32_t b, bool live, f32x4v (&total)[NT]) {
            f32x4v acc[NT], amin[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; amin[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x2 q = R.q[j];
                const uint32_t v = R.sm[j];
                const float sc0 = (float)(v & 0xffu), sc1 = (float)((v >> 8) & 0xffu);
                f16x8 alo, ahi;
                if (KIND == WRK_MAT_Q4_K) {
                    alo = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), sc0 * 1024.0f);          // q*sc*2^-14
                    ahi = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), sc1 * 64.0f);            // (16q)*sc*2^-18
                } else {
                    const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
                    alo = mul8(codes8((q.x & 0x0f0f0f0fu) | (((R.qh.x >> s0) & 0x01010101u) << 4), (q.y & 0x0f0f0f0fu) | (((R.qh.y >> s0) & 0x01010101u) << 4)), sc0 * 1024.0f);
                    ahi = mul8(codes8(((q.x >> 4) & 0x0f0f0f0fu) | (((R.qh.x >> s1) & 0x01010101u) << 4), ((q.y >> 4) & 0x0f0f0f0fu) | (((R.qh.y >> s1) & 0x01010101u) << 4)), sc1 * 1024.0f);
                }
                // min term on the matrix core: A = m (integer <= 63, exact), so acc_min = m * sum_k x_k in f32
                const f16 m0h = (f16)(float)((v >> 16) & 0xffu), m1h = (f16)(float)(v >> 24);
                const f16x8 mlo = {m0h, m0h, m0h, m0h, m0h, m0h, m0h, m0h}, mhi = {m1h, m1h, m1h, m1h, m1h, m1h, m1h, m1h};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f16x8 b0 = NT <= 2 ? X.bf[t][2 * j] : loadB(t, b * 256 + j * 64);
                    const f16x8 b1 = NT <= 2 ? X.bf[t][2 * j + 1] : loadB(t, b * 256 + j * 64 + 32);
                    acc[t] = mfma16(alo, b0, acc[t]);
                    acc[t] = mfma16(ahi, b1, acc[t]);
                    amin[t] = mfma16(mlo, b0, amin[t]);
                    amin[t] = mfma16(mhi, b1, amin[t]);
                }
            }
            // block epilogue for C rows 4g..4g+3: total += d * acc * 2^14 - dmin * acc_min  (a dead block: zero scales)
            const float livef = live ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] & 0xffffu)) * (16384.0f * livef);
                const float dmin = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] >> 16)) * livef;
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t][i] += d * acc[t][i] - dmin * amin[t][i];
            }
        };
        // every wave runs ceil(nb / NW) blocks: wave, wave + NW, ... clamped to the last block (dead ones count zero)
        const uint32_t iters = (nb + NW - 1) / NW;
        auto blk = [&](uint32_t i) { return min(wave + NW * i, nb - 1); };
        WRK_STAMP(P.dbg, 1);
        // weights TWO blocks ahead, B fragments one block ahead (16-token regime) or with the block (more tokens)
        // (four blocks ahead measured slower at 16 tokens: 18.3 vs 15.9 us for ffn.value -- register pressure, round 2)
        auto mul_blk = [&](const WBlk (&R)[NR], const BBlk& X, uint32_t b, bool live) {
#pragma unroll
            for (int rt = 0; rt < NR; ++rt) mul_blk1(R[rt], X, b, live, totals[rt]);
        };
        WBlk W0[NR], W1[NR];
        load_w(W0, blk(0));
        load_w(W1, blk(1));
        if (NT == 1) {
            BBlk X0, X1;
            load_b(X0, blk(0));
            for (uint32_t i = 0; i < iters; i += 2) {
                load_b(X1, blk(i + 1));
                mul_blk(W0, X0, blk(i), wave + NW * i < nb);
                load_w(W0, blk(i + 2));
                if (i + 1 >= iters) break;              // uniform over the workgroup
                load_b(X0, blk(i + 2));
                mul_blk(W1, X1, blk(i + 1), wave + NW * (i + 1) < nb);
                load_w(W1, blk(i + 3));
            }
        } else {
            BBlk X;     // NT == 2: one block of B fragments, fetched with the block; NT > 2: unused
            for (uint32_t i = 0; i < iters; i += 2) {
                if (NT == 2) load_b(X, blk(i));
                mul_blk(W0, X, blk(i), wave + NW * i < nb);
                load_w(W0, blk(i + 2));
                if (i + 1 >= iters) break;
                if (NT == 2) load_b(X, blk(i + 1));
                mul_blk(W1, X, blk(i + 1), wave + NW * (i + 1) < nb);
                load_w(W1, blk(i + 3));
            }
        }
    }

    WRK_STAMP(P.dbg, 2);                // this wave's blocks are multiplied
    // combine the K slices
    if (wave > 0) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) sh_tot[wave - 1][rt * NT + t][i][lane] = totals[rt][t][i];
    }
    __syncthreads();
    if (wave > 0) return;
    WRK_STAMP(P.dbg, 3);                // partial tiles met in LDS
#pragma unroll
    for (int rt = 0; rt < NR; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = 0.0f;
#pragma unroll
                for (int w = 0; w < NW - 1; ++w) a += sh_tot[w][rt * NT + t][i][lane];
                totals[rt][t][i] += a;
            }

    // store: lane owns rows m0 + 16 rt + 4g + (0..3) of token column r of each tile
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) {
        const uint32_t mr = m0 + 16u * rt;
        if (mr >= P.m) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint32_t tok = n0 + 16 * t + r;
            if (tok >= P.n) continue;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = act_apply(P.act, totals[rt][t][i] * P.scale);
                if (P.has_res) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? __builtin_bit_cast(float, resbs[rt][t][i]) : (float)__builtin_bit_cast(f16, (uint16_t)resbs[rt][t][i]));
            }
            const size_t oo = (size_t)tok * P.os + mr + 4 * g;
            if (mr + 4 * g + 4 <= P.m) {        // the common case: four consecutive rows in one store
                if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
                else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (mr + 4 * g + i < P.m) { if (P.out32) ((float*)P.out_p)[oo + i] = o[i]; else ((f16*)P.out_p)[oo + i] = (f16)o[i]; }
            }
        }
    }
    WRK_STAMP(P.dbg, 4);
}

// One launch, several matrices: blockIdx.x -> job (like the matvec launches), kind dispatched at run time.
template <int NT, int NW>
__global__ void __launch_bounds__(64 * NW) gemm_kernel(const GemmBatch B) {
    __shared__ float sh_tot[NW - 1 + 1][NT][4][64];     // K-split partial sums of waves 1..NW-1 (+ one slab: the NF4 level table lives in its head)
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.njobs && blockIdx.x >= B.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.jobs[ji];
    switch (P.kind) {
        case WRK_MAT_Q4_K: gemm_body<WRK_MAT_Q4_K, NT, NW>(P, sh_tot); break;
        case WRK_MAT_Q5_K: gemm_body<WRK_MAT_Q5_K, NT, NW>(P, sh_tot); break;
        case WRK_MAT_Q6_K: gemm_body<WRK_MAT_Q6_K, NT, NW>(P, sh_tot); break;
        case WRK_MAT_Q8_0: gemm_body<WRK_MAT_Q8_0, NT, NW>(P, sh_tot); break;
        case WRK_MAT_INT8: gemm_body<WRK_MAT_INT8, NT, NW>(P, sh_tot); break;
        case WRK_MAT_NF4: gemm_body<WRK_MAT_NF4, NT, NW>(P, sh_tot); break;
        default: gemm_body<WRK_MAT_F16, NT, NW>(P, sh_tot); break;
    }
}

// two row tiles per workgroup (<= 16 tokens, K4 kinds + F16 LoRA rows): see gemm_body
template <int NW>
__global__ void __launch_bounds__(64 * NW) gemm_pair_kernel(const GemmBatch B) {
    __shared__ float sh_tot[NW][2][4][64];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.njobs && blockIdx.x >= B.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.jobs[ji];
    switch (P.kind) {
        case WRK_MAT_Q4_K: gemm_body<WRK_MAT_Q4_K, 1, NW, 2>(P, sh_tot); break;
        case WRK_MAT_Q5_K: gemm_body<WRK_MAT_Q5_K, 1, NW, 2>(P, sh_tot); break;
        default: gemm_body<WRK_MAT_F16, 1, NW, 2>(P, sh_tot); break;
    }
}

// ------------------------------------------------------------------ prefill tile kernel (Q4_K / Q5_K, >= 96 tokens)
// The K-split kernel above re-reads the whole activation tile from L2 for every 16 rows (rocprof: 1 GB of L2->L1
// traffic for an 8192x2048 matrix x 512 tokens -- the limiter, not the matrix core).  Here a workgroup owns
// 64 rows x 64 tokens: its 4 waves take 16 rows each over the WHOLE K and share the activation tile through LDS
// ([64 tokens][256 k] f16 per 256-block, double buffered: 2 x 33 KB, two workgroups per CU; rows padded by 16 B so the B-fragment reads of a
// 16-lane phase hit 16 distinct bank groups).  Per block and wave: 64 MFMAs (4 steps x 4 token tiles x {lo, hi,
// min-lo, min-hi}) against 32 ds_read_b128 -- MFMA and LDS time are balanced.  The next block's activations are
// fetched into registers (and its weights requested) while the current one is multiplied.
constexpr int TILE_TOK = 64, TILE_TT = TILE_TOK / 16, TILE_ROWS = 64, TILE_LDS_ROW = 256 + 8;     // f16 elements per staged token row
constexpr int TILE_STAGE = TILE_TOK * 32 / 256;      // 16-byte chunks each thread stages per block

template <int KIND>
__device__ __forceinline__ void gemm_tile_body(const GemmParams& P, f16* __restrict__ lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;     // this wave's 16 rows
    const uint32_t n0 = blockIdx.y * TILE_TOK;
    const uint32_t K = P.k, nb = K >> 8;
    const uint32_t row = min(m0 + r, P.m - 1);
    const uint8_t* wrow = P.w + (size_t)row * P.row_bytes;
    const uint32_t hoff = KIND == WRK_MAT_Q4_K ? nb * 128 : nb * 160, soff = hoff + nb * 4;
    const uint8_t* crow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) crow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + hoff;

    // activation staging: 64 tokens x 32 chunks of 8 f16 per block, 8 per thread: chunk q -> token (tid >> 5) + 8 q,
    // columns 8 (tid & 31).  The input is a dense [K, T, 1] stack (checked on the host): token rows are xs0 apart.
    const size_t xs0 = P.in.stride[0];
    const f16* xbase = (const f16*)P.in.p + dt_index(P.in, 0, 0, 0) + (size_t)(n0 + (tid >> 5)) * xs0 + (tid & 31u) * 8;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 stage[TILE_STAGE];
    auto fetch_x = [&](uint32_t b) {
#pragma unroll
        for (int q = 0; q < TILE_STAGE; ++q)
            stage[q] = (n0 + (tid >> 5) + 8 * q < P.n && b * 256 + (tid & 31u) * 8 < K) ? *(const f16x8*)(xbase + (size_t)8 * q * xs0 + (size_t)b * 256) : zero8;
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * TILE_TOK * TILE_LDS_ROW;
#pragma unroll
        for (int q = 0; q < TILE_STAGE; ++q) *(f16x8*)(base + ((tid >> 5) + 8 * q) * TILE_LDS_ROW + (tid & 31u) * 8) = stage[q];
    };

    struct WBlk { u32x2 q[4]; u32x2 qh; u32x4 sm; uint32_t dd[4]; };
    auto load_w = [&](WBlk& R, uint32_t b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) R.q[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
        if (KIND == WRK_MAT_Q5_K) R.qh = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 32 + 8 * g);
        R.sm = *(const u32x4*)(wrow + soff + (size_t)b * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.dd[i] = *(const uint32_t*)(crow[i] + (size_t)b * 4);
    };

    f32x4v total[TILE_TT];
#pragma unroll
    for (int t = 0; t < TILE_TT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    WBlk W0, W1;
    load_w(W0, 0);
    fetch_x(0);
    store_x(0);
    __syncthreads();
    // one 256-block: multiply block b out of LDS buffer `buf` with weights R while the next block's weights (into Rn)
    // and activations (into `stage`) are in flight; buffers alternate by unrolling, so no register array is indexed
    auto step = [&](uint32_t b, uint32_t buf, const WBlk& R, WBlk& Rn) {
        const bool more = b + 1 < nb;
        if (more) { load_w(Rn, b + 1); fetch_x(b + 1); }
        const f16* xt = lds + (size_t)buf * TILE_TOK * TILE_LDS_ROW + r * TILE_LDS_ROW + 8 * g;
        f32x4v acc[TILE_TT], amin[TILE_TT];
#pragma unroll
        for (int t = 0; t < TILE_TT; ++t) { acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; amin[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x2 q = R.q[j];
            const uint32_t v = R.sm[j];
            const float sc0 = (float)(v & 0xffu), sc1 = (float)((v >> 8) & 0xffu);
            f16x8 alo, ahi;
            if (KIND == WRK_MAT_Q4_K) {
                alo = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), sc0 * 1024.0f);
                ahi = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), sc1 * 64.0f);
            } else {
                const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
                alo = mul8(codes8((q.x & 0x0f0f0f0fu) | (((R.qh.x >> s0) & 0x01010101u) << 4), (q.y & 0x0f0f0f0fu) | (((R.qh.y >> s0) & 0x01010101u) << 4)), sc0 * 1024.0f);
                ahi = mul8(codes8(((q.x >> 4) & 0x0f0f0f0fu) | (((R.qh.x >> s1) & 0x01010101u) << 4), ((q.y >> 4) & 0x0f0f0f0fu) | (((R.qh.y >> s1) & 0x01010101u) << 4)), sc1 * 1024.0f);
            }
            const f16 m0h = (f16)(float)((v >> 16) & 0xffu), m1h = (f16)(float)(v >> 24);
            const f16x8 mlo = {m0h, m0h, m0h, m0h, m0h, m0h, m0h, m0h}, mhi = {m1h, m1h, m1h, m1h, m1h, m1h, m1h, m1h};
#pragma unroll
            for (int t = 0; t < TILE_TT; ++t) {
                const f16x8 b0 = *(const f16x8*)(xt + (size_t)t * 16 * TILE_LDS_ROW + j * 64);
                const f16x8 b1 = *(const f16x8*)(xt + (size_t)t * 16 * TILE_LDS_ROW + j * 64 + 32);
                acc[t] = mfma16(alo, b0, acc[t]);
                acc[t] = mfma16(ahi, b1, acc[t]);
                amin[t] = mfma16(mlo, b0, amin[t]);
                amin[t] = mfma16(mhi, b1, amin[t]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] & 0xffffu)) * 16384.0f;
            const float dmin = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] >> 16));
#pragma unroll
            for (int t = 0; t < TILE_TT; ++t) total[t][i] += d * acc[t][i] - dmin * amin[t][i];
        }
        if (more) store_x(buf ^ 1u);        // the other buffer was last read in iteration b - 1 (the barrier orders it)
        __syncthreads();
    };
    for (uint32_t b = 0; b < nb; b += 2) {
        step(b, 0u, W0, W1);
        if (b + 1 < nb) step(b + 1, 1u, W1, W0);
    }

    // store: lane owns rows m0 + 4g + (0..3) of token column r of each 16-token tile
#pragma unroll
    for (int t = 0; t < TILE_TT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n) continue;
        uint32_t tt, bb;
        tok_tb(P.out, tok, tt, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t mr = m0 + 4 * g + i;
            if (mr >= P.m) continue;
            float o = act_apply(P.act, total[t][i] * P.scale);
            if (P.has_res) { uint32_t rt, rb; tok_tb(P.res, tok, rt, rb); o = dt_round(P.out, o) + dt_load(P.res, dt_index(P.res, mr, rt, rb)); }
            dt_store(P.out, dt_index(P.out, mr, tt, bb), o);
        }
    }
}

// F16 matrices (the LoRA projections: 2048 -> 96..256 and back) on the same tiling: A fragments are plain loads, K need
// not be a multiple of 256 (fragments and staged activations beyond K are zero).
__device__ __forceinline__ void gemm_tile_body_f16(const GemmParams& P, f16* __restrict__ lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;
    const uint32_t n0 = blockIdx.y * TILE_TOK;
    const uint32_t K = P.k, nb = (K + 255) >> 8;
    const f16* wr = (const f16*)(P.w + (size_t)min(m0 + r, P.m - 1) * P.row_bytes);
    const size_t xs0 = P.in.stride[0];
    const f16* xbase = (const f16*)P.in.p + dt_index(P.in, 0, 0, 0) + (size_t)(n0 + (tid >> 5)) * xs0 + (tid & 31u) * 8;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 stage[TILE_STAGE];
    auto fetch_x = [&](uint32_t b) {
#pragma unroll
        for (int q = 0; q < TILE_STAGE; ++q)
            stage[q] = (n0 + (tid >> 5) + 8 * q < P.n && b * 256 + (tid & 31u) * 8 < K) ? *(const f16x8*)(xbase + (size_t)8 * q * xs0 + (size_t)b * 256) : zero8;
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * TILE_TOK * TILE_LDS_ROW;
#pragma unroll
        for (int q = 0; q < TILE_STAGE; ++q) *(f16x8*)(base + ((tid >> 5) + 8 * q) * TILE_LDS_ROW + (tid & 31u) * 8) = stage[q];
    };
    struct FBlk { f16x8 a[8]; };
    auto load_w = [&](FBlk& R, uint32_t b) {
#pragma unroll
        for (int sb = 0; sb < 8; ++sb) { const uint32_t k = b * 256 + sb * 32 + 8 * g; R.a[sb] = (k + 8 <= K) ? *(const f16x8*)(wr + k) : zero8; }
    };
    f32x4v total[TILE_TT];
#pragma unroll
    for (int t = 0; t < TILE_TT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    FBlk W0, W1;
    load_w(W0, 0);
    fetch_x(0);
    store_x(0);
    __syncthreads();
    auto step = [&](uint32_t b, uint32_t buf, const FBlk& R, FBlk& Rn) {
        const bool more = b + 1 < nb;
        if (more) { load_w(Rn, b + 1); fetch_x(b + 1); }
        const f16* xt = lds + (size_t)buf * TILE_TOK * TILE_LDS_ROW + r * TILE_LDS_ROW + 8 * g;
#pragma unroll
        for (int sb = 0; sb < 8; ++sb)
#pragma unroll
            for (int t = 0; t < TILE_TT; ++t) total[t] = mfma16(R.a[sb], *(const f16x8*)(xt + (size_t)t * 16 * TILE_LDS_ROW + sb * 32), total[t]);
        if (more) store_x(buf ^ 1u);
        __syncthreads();
    };
    for (uint32_t b = 0; b < nb; b += 2) {
        step(b, 0u, W0, W1);
        if (b + 1 < nb) step(b + 1, 1u, W1, W0);
    }
#pragma unroll
    for (int t = 0; t < TILE_TT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n) continue;
        uint32_t tt, bb;
        tok_tb(P.out, tok, tt, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t mr = m0 + 4 * g + i;
            if (mr >= P.m) continue;
            float o = act_apply(P.act, total[t][i] * P.scale);
            if (P.has_res) { uint32_t rt, rb; tok_tb(P.res, tok, rt, rb); o = dt_round(P.out, o) + dt_load(P.res, dt_index(P.res, mr, rt, rb)); }
            dt_store(P.out, dt_index(P.out, mr, tt, bb), o);
        }
    }
}

// Q6_K on the same tiling (the real llama.cpp Q4_K_M mix keeps half the value matrices and the head in Q6_K; on the
// K-split kernel they made the mixed 32 x 128 prefill 23 % slower than pure Q4_K).  Row layout [ql 128 nb][qh 64 nb]
// [int8 scales 16 nb][d f16 nb]; codes - 32 and the scale split sc = 2 s1 + s0 stay exact in f16 as in gemm_body.
__device__ __forceinline__ void gemm_tile_body_q6k(const GemmParams& P, f16* __restrict__ lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;
    const uint32_t n0 = blockIdx.y * TILE_TOK;
    const uint32_t K = P.k, nb = K >> 8;
    const uint8_t* wrow = P.w + (size_t)min(m0 + r, P.m - 1) * P.row_bytes;
    const uint8_t* drow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) drow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + (size_t)nb * 208;
    const size_t xs0 = P.in.stride[0];
    const f16* xbase = (const f16*)P.in.p + dt_index(P.in, 0, 0, 0) + (size_t)(n0 + (tid >> 5)) * xs0 + (tid & 31u) * 8;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 stage[TILE_STAGE];
    auto fetch_x = [&](uint32_t b) {
#pragma unroll
        for (int q = 0; q < TILE_STAGE; ++q)
            stage[q] = (n0 + (tid >> 5) + 8 * q < P.n && b * 256 + (tid & 31u) * 8 < K) ? *(const f16x8*)(xbase + (size_t)8 * q * xs0 + (size_t)b * 256) : zero8;
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * TILE_TOK * TILE_LDS_ROW;
#pragma unroll
        for (int q = 0; q < TILE_STAGE; ++q) *(f16x8*)(base + ((tid >> 5) + 8 * q) * TILE_LDS_ROW + (tid & 31u) * 8) = stage[q];
    };
    struct WBlk { u32x2 ql[4]; u32x2 qh[2]; u32x4 sc; uint16_t d[4]; };
    auto load_w = [&](WBlk& R, uint32_t b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) R.ql[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);       // j = 2 n128 + (kq & 1)
#pragma unroll
        for (int h = 0; h < 2; ++h) R.qh[h] = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 64 + h * 32 + 8 * g);
        R.sc = *(const u32x4*)(wrow + (size_t)nb * 192 + (size_t)b * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.d[i] = *(const uint16_t*)(drow[i] + (size_t)b * 2);
    };
    f32x4v total[TILE_TT];
#pragma unroll
    for (int t = 0; t < TILE_TT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    WBlk W0, W1;
    load_w(W0, 0);
    fetch_x(0);
    store_x(0);
    __syncthreads();
    const uint32_t gsh = 8 * (g >> 1);       // scale byte of this lane's 16-element half of a 32-group
    auto step = [&](uint32_t b, uint32_t buf, const WBlk& R, WBlk& Rn) {
        const bool more = b + 1 < nb;
        if (more) { load_w(Rn, b + 1); fetch_x(b + 1); }
        const f16* xt = lds + (size_t)buf * TILE_TOK * TILE_LDS_ROW + r * TILE_LDS_ROW + 8 * g;
        f32x4v acc[TILE_TT];
#pragma unroll
        for (int t = 0; t < TILE_TT; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n128 = 0; n128 < 2; ++n128) {
            const u32x2 qh = R.qh[n128];
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {      // element group 128 n128 + 32 kq + (0..31); scale index 8 n128 + 2 kq + (g >> 1)
                const u32x2 ql = R.ql[2 * n128 + (kq & 1)];
                const uint32_t sh = 2 * kq;
                uint32_t c0, c1;
                if (kq < 2) { c0 = (ql.x & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = (ql.y & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                else { c0 = ((ql.x >> 4) & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = ((ql.y >> 4) & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                const f16x8 c = add8(mul8(codes8(c0, c1), 32768.0f), -0.0625f);       // (code - 32) * 2^-9, exact
                const uint32_t word = R.sc[2 * n128 + (kq >> 1)];
                const int sc = (int)(int8_t)((word >> (16 * (kq & 1) + gsh)) & 0xffu);
                const int s1 = sc >> 1, s0 = sc & 1;
                const f16x8 a1 = mul8(c, (float)(2 * s1)), a0 = mul8(c, (float)s0);
#pragma unroll
                for (int t = 0; t < TILE_TT; ++t) {
                    const f16x8 bfr = *(const f16x8*)(xt + (size_t)t * 16 * TILE_LDS_ROW + n128 * 128 + kq * 32);
                    acc[t] = mfma16(a1, bfr, acc[t]);
                    acc[t] = mfma16(a0, bfr, acc[t]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = (float)__builtin_bit_cast(f16, R.d[i]) * 512.0f;
#pragma unroll
            for (int t = 0; t < TILE_TT; ++t) total[t][i] = __builtin_fmaf(d, acc[t][i], total[t][i]);
        }
        if (more) store_x(buf ^ 1u);
        __syncthreads();
    };
    for (uint32_t b = 0; b < nb; b += 2) {
        step(b, 0u, W0, W1);
        if (b + 1 < nb) step(b + 1, 1u, W1, W0);
    }
#pragma unroll
    for (int t = 0; t < TILE_TT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n) continue;
        uint32_t tt, bb;
        tok_tb(P.out, tok, tt, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t mr = m0 + 4 * g + i;
            if (mr >= P.m) continue;
            float o = act_apply(P.act, total[t][i] * P.scale);
            if (P.has_res) { uint32_t rt, rb; tok_tb(P.res, tok, rt, rb); o = dt_round(P.out, o) + dt_load(P.res, dt_index(P.res, mr, rt, rb)); }
            dt_store(P.out, dt_index(P.out, mr, tt, bb), o);
        }
    }
}

// ------------------------------------------------------------------ prefill tile kernel, second generation (Q4_K / Q5_K)
// Round 2.  What bounded the first tile kernel (12-15 % of the dense f16 MFMA peak): (a) one dequantised 16 x 32 fragment fed only
// TILE_TT = 4 MFMAs, so the VALU dequant (~35 instructions per 64-k step) cost as much issue time as the matrix work; (b) its
// predicated loads (`more ? load : zero`) made the compiler wait vmcnt(0) in front of every LDS store, so nothing was in flight
// across a block.  Here a workgroup owns 64 rows x 16*TT tokens with TT = 8 (128 tokens: a fragment feeds 8 MFMAs; TT = 4 when
// the launch would otherwise have too few workgroups), the activations are staged per HALF block (128 k: 2 x 34 KB of LDS at TT = 8,
// still two workgroups per CU, a barrier every 8*TT MFMAs per wave as before), every global load is unconditional (clamped
// indices; dead tokens are never stored), and a half-stage issues the NEXT half's activation loads before its weights so the LDS
// store waits with a counted vmcnt.
constexpr int T2_KH = 128, T2_ROW = T2_KH + 8;      // staged k per half block; LDS row stride in f16 (+16 B: conflict-free fragment reads)

template <int KIND, int TT>
__device__ __forceinline__ void gemm_tile2_body(const GemmParams& P, f16* __restrict__ lds) {
    constexpr int TOK = 16 * TT, NST = TOK * 16 / 256;       // tokens per workgroup; 16-byte chunks a thread stages per half block
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;     // this wave's 16 rows
    const uint32_t n0 = blockIdx.y * TOK;
    const uint32_t K = P.k, nb = K >> 8;
    const uint8_t* wrow = P.w + (size_t)min(m0 + r, P.m - 1) * P.row_bytes;
    const uint32_t hoff = KIND == WRK_MAT_Q4_K ? nb * 128 : nb * 160, soff = hoff + nb * 4;
    const uint8_t* crow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) crow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + hoff;

    // staging: TOK tokens x 16 chunks of 8 f16 per half block; chunk q of a thread -> token (tid >> 4) + 16 q, columns 8 (tid & 15)
    const f16* xsrc[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) xsrc[q] = P.x + (size_t)min(n0 + (tid >> 4) + 16 * q, P.n - 1) * P.xs + (tid & 15u) * 8;
    f16x8 stage[NST];
    const uint32_t nhalf = 2 * nb;
    auto fetch_x = [&](uint32_t h) {            // half block h (clamped: the extra fetch after the last half is never stored)
        const uint32_t hc = min(h, nhalf - 1);
#pragma unroll
        for (int q = 0; q < NST; ++q) stage[q] = *(const f16x8*)(xsrc[q] + (size_t)hc * T2_KH);
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * TOK * T2_ROW;
#pragma unroll
        for (int q = 0; q < NST; ++q) *(f16x8*)(base + ((tid >> 4) + 16 * q) * T2_ROW + (tid & 15u) * 8) = stage[q];
    };
    struct WBlk { u32x2 q[4]; u32x2 qh; u32x4 sm; uint32_t dd[4]; };
    auto load_w = [&](WBlk& R, uint32_t b0) {
        const uint32_t b = min(b0, nb - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) R.q[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
        if (KIND == WRK_MAT_Q5_K) R.qh = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 32 + 8 * g);
        R.sm = *(const u32x4*)(wrow + soff + (size_t)b * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.dd[i] = *(const uint32_t*)(crow[i] + (size_t)b * 4);
    };

    f32x4v total[TT], acc[TT], amin[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    WBlk W0, W1;
    fetch_x(0);
    load_w(W0, 0);
    store_x(0);
    __syncthreads();
    // one half block: j = 2 hf, 2 hf + 1 of block b out of LDS buffer `buf`
    auto half = [&](uint32_t b, int hf, uint32_t buf, const WBlk& R, WBlk& Rn) {
        fetch_x(2 * b + hf + 1);                    // next half's activations first, then (once per block) the next block's weights
        if (hf == 0) load_w(Rn, b + 1);
        const f16* xt = lds + (size_t)buf * TOK * T2_ROW + r * T2_ROW + 8 * g;
        if (hf == 0) {
#pragma unroll
            for (int t = 0; t < TT; ++t) { acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; amin[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * hf + jj;
            const u32x2 q = R.q[j];
            const uint32_t v = R.sm[j];
            const float sc0 = (float)(v & 0xffu), sc1 = (float)((v >> 8) & 0xffu);
            f16x8 alo, ahi;
            if (KIND == WRK_MAT_Q4_K) {
                alo = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), sc0 * 1024.0f);
                ahi = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), sc1 * 64.0f);
            } else {
                const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
                alo = mul8(codes8((q.x & 0x0f0f0f0fu) | (((R.qh.x >> s0) & 0x01010101u) << 4), (q.y & 0x0f0f0f0fu) | (((R.qh.y >> s0) & 0x01010101u) << 4)), sc0 * 1024.0f);
                ahi = mul8(codes8(((q.x >> 4) & 0x0f0f0f0fu) | (((R.qh.x >> s1) & 0x01010101u) << 4), ((q.y >> 4) & 0x0f0f0f0fu) | (((R.qh.y >> s1) & 0x01010101u) << 4)), sc1 * 1024.0f);
            }
            const f16 m0h = (f16)(float)((v >> 16) & 0xffu), m1h = (f16)(float)(v >> 24);
            const f16x8 mlo = {m0h, m0h, m0h, m0h, m0h, m0h, m0h, m0h}, mhi = {m1h, m1h, m1h, m1h, m1h, m1h, m1h, m1h};
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const f16x8 b0 = *(const f16x8*)(xt + (size_t)t * 16 * T2_ROW + jj * 64);
                const f16x8 b1 = *(const f16x8*)(xt + (size_t)t * 16 * T2_ROW + jj * 64 + 32);
                acc[t] = mfma16(alo, b0, acc[t]);
                acc[t] = mfma16(ahi, b1, acc[t]);
                amin[t] = mfma16(mlo, b0, amin[t]);
                amin[t] = mfma16(mhi, b1, amin[t]);
            }
        }
        if (hf == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] & 0xffffu)) * 16384.0f;
                const float dmin = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] >> 16));
#pragma unroll
                for (int t = 0; t < TT; ++t) total[t][i] += d * acc[t][i] - dmin * amin[t][i];
            }
        }
        store_x(buf ^ 1u);          // the other buffer was last read one half ago (the barrier below orders it)
        __syncthreads();
    };
    for (uint32_t b = 0; b < nb; b += 2) {          // uniform over the workgroup
        half(b, 0, 0u, W0, W1);
        half(b, 1, 1u, W0, W1);
        if (b + 1 >= nb) break;
        half(b + 1, 0, 0u, W1, W0);
        half(b + 1, 1, 1u, W1, W0);
    }

    // store: lane owns rows m0 + 4g + (0..3) of token column r of each 16-token tile (four consecutive rows: one vector)
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n || m0 + 4 * g >= P.m) continue;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act_apply(P.act, total[t][i] * P.scale);
        const size_t oo = (size_t)tok * P.os + m0 + 4 * g;
        if (P.has_res) {
            const size_t ro = (size_t)tok * P.rs + m0 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? ((const float*)P.res_p)[ro + i] : (float)((const f16*)P.res_p)[ro + i]);
        }
        if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
        else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
    }
}

// Q8_0 on the same tile (round 2; cfg 5's 14B file): 32-element blocks with one f16 scale each, so every 32-k step has its own row scale:
// per step  total += d * (A . B)  with A = the int8 codes (exact in f16), as in the K-split kernel.  Rows are [K int8][K / 32 f16 scales];
// K % 128 == 0 (host-checked).  A half block = four steps; its 32 code bytes per lane-row and the four scales of each C row are requested
// one half ahead.
template <int TT>
__device__ __forceinline__ void gemm_tile2_body_q8(const GemmParams& P, f16* __restrict__ lds) {
    constexpr int TOK = 16 * TT, NST = TOK * 16 / 256;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;
    const uint32_t n0 = blockIdx.y * TOK;
    const uint32_t K = P.k, nhalf = K >> 7;
    const uint8_t* wrow = P.w + (size_t)min(m0 + r, P.m - 1) * P.row_bytes;
    const uint8_t* drow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) drow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + K;
    const f16* xsrc[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) xsrc[q] = P.x + (size_t)min(n0 + (tid >> 4) + 16 * q, P.n - 1) * P.xs + (tid & 15u) * 8;
    f16x8 stage[NST];
    auto fetch_x = [&](uint32_t h) {
        const uint32_t hc = min(h, nhalf - 1);
#pragma unroll
        for (int q = 0; q < NST; ++q) stage[q] = *(const f16x8*)(xsrc[q] + (size_t)hc * T2_KH);
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * TOK * T2_ROW;
#pragma unroll
        for (int q = 0; q < NST; ++q) *(f16x8*)(base + ((tid >> 4) + 16 * q) * T2_ROW + (tid & 15u) * 8) = stage[q];
    };
    struct W8 { u32x2 q[4]; u32x2 d[4]; };        // codes of the four steps; the four step scales (f16) of each of the lane's four C rows
    auto load_w = [&](W8& R, uint32_t h) {
        const uint32_t hc = min(h, nhalf - 1);
#pragma unroll
        for (int st = 0; st < 4; ++st) R.q[st] = *(const u32x2*)(wrow + (size_t)hc * 128 + st * 32 + 8 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.d[i] = *(const u32x2*)(drow[i] + (size_t)hc * 8);
    };
    f32x4v total[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    W8 W0, W1;
    fetch_x(0);
    load_w(W0, 0);
    store_x(0);
    __syncthreads();
    auto half = [&](uint32_t h, uint32_t buf, const W8& R, W8& Rn) {
        fetch_x(h + 1);
        load_w(Rn, h + 1);
        const f16* xt = lds + (size_t)buf * TOK * T2_ROW + r * T2_ROW + 8 * g;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            // int8 -> (u - 128): subnormal u * 2^-24, scaled by 2^15 to u * 2^-9 (normal), minus 128 * 2^-9
            const f16x8 a = add8(mul8(codes8(R.q[st].x ^ 0x80808080u, R.q[st].y ^ 0x80808080u), 32768.0f), -0.25f);
            float dsc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t wd = st < 2 ? R.d[i].x : R.d[i].y;
                dsc[i] = (float)__builtin_bit_cast(f16, (uint16_t)((st & 1) ? (wd >> 16) : (wd & 0xffffu))) * 512.0f;     // * 2^9
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const f16x8 bq = *(const f16x8*)(xt + (size_t)t * 16 * T2_ROW + st * 32);
                const f32x4v acc = mfma16(a, bq, (f32x4v){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int i = 0; i < 4; ++i) total[t][i] = __builtin_fmaf(dsc[i], acc[i], total[t][i]);
            }
        }
        store_x(buf ^ 1u);
        __syncthreads();
    };
    for (uint32_t h = 0; h < nhalf; h += 2) {       // uniform over the workgroup
        half(h, 0u, W0, W1);
        if (h + 1 >= nhalf) break;
        half(h + 1, 1u, W1, W0);
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n || m0 + 4 * g >= P.m) continue;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act_apply(P.act, total[t][i] * P.scale);
        const size_t oo = (size_t)tok * P.os + m0 + 4 * g;
        if (P.has_res) {
            const size_t ro = (size_t)tok * P.rs + m0 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? ((const float*)P.res_p)[ro + i] : (float)((const f16*)P.res_p)[ro + i]);
        }
        if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
        else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
    }
}

// F16 rows on the same tile (LoRA down-projections in prefill: K = D; any F16 matrix with K % 128 == 0): the A fragments come straight
// from memory, one half block (four 32-k steps) ahead; accumulation over the whole K in the matrix core.
template <int TT>
__device__ __forceinline__ void gemm_tile2_body_f16(const GemmParams& P, f16* __restrict__ lds) {
    constexpr int TOK = 16 * TT, NST = TOK * 16 / 256;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;
    const uint32_t n0 = blockIdx.y * TOK;
    const uint32_t K = P.k, nhalf = K >> 7;
    const f16* wrow = (const f16*)(P.w + (size_t)min(m0 + r, P.m - 1) * P.row_bytes) + 8 * g;
    const f16* xsrc[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) xsrc[q] = P.x + (size_t)min(n0 + (tid >> 4) + 16 * q, P.n - 1) * P.xs + (tid & 15u) * 8;
    f16x8 stage[NST];
    auto fetch_x = [&](uint32_t h) {
        const uint32_t hc = min(h, nhalf - 1);
#pragma unroll
        for (int q = 0; q < NST; ++q) stage[q] = *(const f16x8*)(xsrc[q] + (size_t)hc * T2_KH);
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * TOK * T2_ROW;
#pragma unroll
        for (int q = 0; q < NST; ++q) *(f16x8*)(base + ((tid >> 4) + 16 * q) * T2_ROW + (tid & 15u) * 8) = stage[q];
    };
    struct WF { f16x8 a[4]; };
    auto load_w = [&](WF& R, uint32_t h) {
        const uint32_t hc = min(h, nhalf - 1);
#pragma unroll
        for (int st = 0; st < 4; ++st) R.a[st] = *(const f16x8*)(wrow + (size_t)hc * 128 + st * 32);
    };
    f32x4v total[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    WF W0, W1;
    fetch_x(0);
    load_w(W0, 0);
    store_x(0);
    __syncthreads();
    auto half = [&](uint32_t h, uint32_t buf, const WF& R, WF& Rn) {
        fetch_x(h + 1);
        load_w(Rn, h + 1);
        const f16* xt = lds + (size_t)buf * TOK * T2_ROW + r * T2_ROW + 8 * g;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int t = 0; t < TT; ++t) total[t] = mfma16(R.a[st], *(const f16x8*)(xt + (size_t)t * 16 * T2_ROW + st * 32), total[t]);
        store_x(buf ^ 1u);
        __syncthreads();
    };
    for (uint32_t h = 0; h < nhalf; h += 2) {       // uniform over the workgroup
        half(h, 0u, W0, W1);
        if (h + 1 >= nhalf) break;
        half(h + 1, 1u, W1, W0);
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const uint32_t tok = n0 + 16 * t + r;
        if (tok >= P.n || m0 + 4 * g >= P.m) continue;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = act_apply(P.act, total[t][i] * P.scale);
        const size_t oo = (size_t)tok * P.os + m0 + 4 * g;
        if (P.has_res) {
            const size_t ro = (size_t)tok * P.rs + m0 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? ((const float*)P.res_p)[ro + i] : (float)((const f16*)P.res_p)[ro + i]);
        }
        if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
        else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
    }
}

template <int TT>
__global__ void __launch_bounds__(256) gemm_tile2_kernel(const GemmBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile2_smem[];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.njobs && blockIdx.x >= B.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.jobs[ji];
    if (P.kind == WRK_MAT_Q4_K) gemm_tile2_body<WRK_MAT_Q4_K, TT>(P, (f16*)tile2_smem);
    else if (P.kind == WRK_MAT_Q5_K) gemm_tile2_body<WRK_MAT_Q5_K, TT>(P, (f16*)tile2_smem);
    else if (P.kind == WRK_MAT_Q8_0) gemm_tile2_body_q8<TT>(P, (f16*)tile2_smem);
    else gemm_tile2_body_f16<TT>(P, (f16*)tile2_smem);
}

__global__ void __launch_bounds__(256) gemm_tile_kernel(const GemmBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.njobs && blockIdx.x >= B.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.jobs[ji];
    if (P.kind == WRK_MAT_Q4_K) gemm_tile_body<WRK_MAT_Q4_K>(P, (f16*)tile_smem);
    else if (P.kind == WRK_MAT_Q5_K) gemm_tile_body<WRK_MAT_Q5_K>(P, (f16*)tile_smem);
    else if (P.kind == WRK_MAT_Q6_K) gemm_tile_body_q6k(P, (f16*)tile_smem);
    else gemm_tile_body_f16(P, (f16*)tile_smem);
}

// ------------------------------------------------------------------ decode-batch tile kernel (Q4_K / Q5_K, <= 16 tokens)
// Same idea as the prefill tile for the few-token regime of batched decode: 64 rows x 16 tokens per workgroup, every
// wave walks the whole K for its 16 rows, the 16 x 256 activation block is shared through LDS (8 KB per block, double
// buffered).  Against the K-split kernel this (a) reads the activations once per 64 rows instead of once per 16 (at
// D = 4096 the K-split kernel moves 128 KB of activations per 45 KB of weights), and (b) takes the B fragments off the
// vector-memory queue (LDS has its own counter), so the weight rows can be prefetched FOUR blocks ahead without the
// in-order return of loads stalling the multiply.
template <int KIND>
__device__ __forceinline__ void gemm_dec_body(const GemmParams& P, f16* __restrict__ lds) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * TILE_ROWS + wave * 16;
    const uint32_t K = P.k, nb = K >> 8;
    const uint32_t row = min(m0 + r, P.m - 1);
    const uint8_t* wrow = P.w + (size_t)row * P.row_bytes;
    const uint32_t hoff = KIND == WRK_MAT_Q4_K ? nb * 128 : nb * 160, soff = hoff + nb * 4;
    const uint8_t* crow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) crow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + hoff;
    // staging: 16 tokens x 32 chunks of 8 f16 per block, 2 per thread: chunk q -> token (tid >> 5) + 8 q
    const size_t xs0 = P.in.stride[0];
    const f16* xbase = (const f16*)P.in.p + dt_index(P.in, 0, 0, 0) + (size_t)(tid >> 5) * xs0 + (tid & 31u) * 8;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 stage[2];
    auto fetch_x = [&](uint32_t b) {
#pragma unroll
        for (int q = 0; q < 2; ++q) stage[q] = ((tid >> 5) + 8 * q < P.n) ? *(const f16x8*)(xbase + (size_t)8 * q * xs0 + (size_t)b * 256) : zero8;
    };
    auto store_x = [&](uint32_t buf) {
        f16* base = lds + (size_t)buf * 16 * TILE_LDS_ROW;
#pragma unroll
        for (int q = 0; q < 2; ++q) *(f16x8*)(base + ((tid >> 5) + 8 * q) * TILE_LDS_ROW + (tid & 31u) * 8) = stage[q];
    };
    struct WBlk { u32x2 q[4]; u32x2 qh; u32x4 sm; uint32_t dd[4]; };
    auto load_w = [&](WBlk& R, uint32_t b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) R.q[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
        if (KIND == WRK_MAT_Q5_K) R.qh = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 32 + 8 * g);
        R.sm = *(const u32x4*)(wrow + soff + (size_t)b * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.dd[i] = *(const uint32_t*)(crow[i] + (size_t)b * 4);
    };
    f32x4v total = {0.f, 0.f, 0.f, 0.f};
    auto mul_blk = [&](const WBlk& R, uint32_t buf) {
        const f16* xt = lds + (size_t)buf * 16 * TILE_LDS_ROW + r * TILE_LDS_ROW + 8 * g;
        f32x4v acc = {0.f, 0.f, 0.f, 0.f}, amin = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x2 q = R.q[j];
            const uint32_t v = R.sm[j];
            const float sc0 = (float)(v & 0xffu), sc1 = (float)((v >> 8) & 0xffu);
            f16x8 alo, ahi;
            if (KIND == WRK_MAT_Q4_K) {
                alo = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), sc0 * 1024.0f);
                ahi = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), sc1 * 64.0f);
            } else {
                const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
                alo = mul8(codes8((q.x & 0x0f0f0f0fu) | (((R.qh.x >> s0) & 0x01010101u) << 4), (q.y & 0x0f0f0f0fu) | (((R.qh.y >> s0) & 0x01010101u) << 4)), sc0 * 1024.0f);
                ahi = mul8(codes8(((q.x >> 4) & 0x0f0f0f0fu) | (((R.qh.x >> s1) & 0x01010101u) << 4), ((q.y >> 4) & 0x0f0f0f0fu) | (((R.qh.y >> s1) & 0x01010101u) << 4)), sc1 * 1024.0f);
            }
            const f16 m0h = (f16)(float)((v >> 16) & 0xffu), m1h = (f16)(float)(v >> 24);
            const f16x8 mlo = {m0h, m0h, m0h, m0h, m0h, m0h, m0h, m0h}, mhi = {m1h, m1h, m1h, m1h, m1h, m1h, m1h, m1h};
            const f16x8 b0 = *(const f16x8*)(xt + j * 64), b1 = *(const f16x8*)(xt + j * 64 + 32);
            acc = mfma16(alo, b0, acc);
            acc = mfma16(ahi, b1, acc);
            amin = mfma16(mlo, b0, amin);
            amin = mfma16(mhi, b1, amin);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] & 0xffffu)) * 16384.0f;
            const float dmin = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] >> 16));
            total[i] += d * acc[i] - dmin * amin[i];
        }
    };
    WBlk W[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if ((uint32_t)u < nb) load_w(W[u], u);
    fetch_x(0);
    store_x(0);
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {           // unrolled by the ring depth: every buffer index is a compile-time constant
            const uint32_t b = b0 + u;
            if (b >= nb) break;
            const bool more = b + 1 < nb;
            if (more) fetch_x(b + 1);           // L2-resident, issued BEFORE the far weight prefetch below (loads return in order)
            mul_blk(W[u], u & 1);
            if (b + 4 < nb) load_w(W[u], b + 4);
            if (more) store_x((u & 1) ^ 1);
            __syncthreads();
        }
    }
    const uint32_t tok = r;
    if (tok < P.n) {
        uint32_t tt, bb;
        tok_tb(P.out, tok, tt, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t mr = m0 + 4 * g + i;
            if (mr >= P.m) continue;
            float o = act_apply(P.act, total[i] * P.scale);
            if (P.has_res) { uint32_t rt, rb; tok_tb(P.res, tok, rt, rb); o = dt_round(P.out, o) + dt_load(P.res, dt_index(P.res, mr, rt, rb)); }
            dt_store(P.out, dt_index(P.out, mr, tt, bb), o);
        }
    }
}

__global__ void __launch_bounds__(256) gemm_dec_kernel(const GemmBatch B) {
    __shared__ __attribute__((aligned(16))) f16 dec_smem[2 * 16 * TILE_LDS_ROW];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.njobs && blockIdx.x >= B.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.jobs[ji];
    if (P.kind == WRK_MAT_Q4_K) gemm_dec_body<WRK_MAT_Q4_K>(P, dec_smem);
    else gemm_dec_body<WRK_MAT_Q5_K>(P, dec_smem);
}

// ------------------------------------------------------------------ K-sliced GEMM for decode batches (5 .. 32 stacked tokens; round 2)
// What the in-kernel timelines of batched decode showed: a CU ingests only ~30 GB/s when its loads are one
// dependent burst (outstanding misses x latency), and the kernels above make every workgroup pull the WHOLE activation stack -- 64 KB at
// 16 tokens, 256 KB at 64 -- next to 5-18 KB of weights; ffn.value (2048 rows, K = 8192) ran 13.8 us at 16 tokens and 23 us at 32 / 64 on
// HALF the chip.  Here a workgroup owns 64 rows x a SLICE of K (BPS 256-blocks): its 4 waves take 16 rows each, the slice of the
// activations (tokens x BPS*256) is staged ONCE through LDS with coalesced loads, every global load of the kernel is issued up front and
// unconditionally.  K slices of a row group meet through f32 partial tiles in a scratch buffer: each slice stores its tile, releases,
// and bumps the row group's counter; the LAST arriver acquires, adds the tiles in slice order (so the sum does not depend on who is
// last), applies scale / activation / residual and stores, and resets the counter for the next launch.  Per workgroup ingest at 16 tokens,
// K = 2048, BPS = 2: 18 KB of weights + 16 KB of activations.
struct KsJob {
    uint32_t nslices, rg_begin;     // K slices per row group; first row group of this job in the counter / partial space
};
struct KsBatch {
    GemmBatch g;
    KsJob ks[GEMM_MAX_JOBS];
    float* part;                    // [row group][slice][token (ntp)][64 rows] f32
    uint32_t* counters;             // [row group], zero between launches
    uint32_t ntp;                   // tokens padded to the tile: 16 * NT
};

template <int KIND, int NT, int BPS>
__device__ __forceinline__ void gemm_ks_body(const GemmParams& P, const KsJob& E, float* __restrict__ part, uint32_t* __restrict__ counters,
                                             f16* __restrict__ lds, uint32_t* sh_flag) {
    constexpr uint32_t ROWF = BPS * 256 + 8;            // f16 per staged token row (+16 B: the 16 token rows of a fragment read hit distinct banks)
    constexpr uint32_t CPT = BPS * 32;                  // 16-byte chunks per token and slice
    constexpr int NST = 2 * NT * BPS;                   // chunks each thread stages
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t r = lane & 15, g = lane >> 4;
    const uint32_t local = blockIdx.x - P.wg_begin;
    const uint32_t rg = local / E.nslices, z = local - rg * E.nslices;
    const uint32_t m0 = rg * 64 + wave * 16;
    const uint32_t K = P.k, nb = K >> 8;
    const uint32_t b0 = z * BPS;                        // host: nslices * BPS == nb
    WRK_STAMP(P.dbg, 0);
    const uint32_t row = min(m0 + r, P.m - 1);
    const uint8_t* wrow = P.w + (size_t)row * P.row_bytes;

    // ---- every global load of the kernel, back to back: activations of the slice, weights of the slice, residual operands
    f16x8 stage[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const uint32_t c = tid + 256u * i, tok = c / CPT, ch = c % CPT;
        stage[i] = *(const f16x8*)(P.x + (size_t)min(tok, P.n - 1) * P.xs + (size_t)b0 * 256 + ch * 8);
    }
    struct WBlk { u32x2 q[4]; u32x2 qh; u32x4 sm; uint32_t dd[4]; };
    WBlk W[(KIND == WRK_MAT_F16 || KIND == WRK_MAT_Q6_K || KIND == WRK_MAT_Q8_0) ? 1 : BPS];
    struct W8 { u32x2 q[8]; u32x4 d[4]; };             // Q8_0 (round 3): this lane's 8 codes of each 32-block; the eight block scales of its four C rows
    W8 V8[KIND == WRK_MAT_Q8_0 ? BPS : 1];
    f16x8 WF[KIND == WRK_MAT_F16 ? BPS : 1][8];
    struct W6 { u32x2 ql[4]; u32x2 qh[2]; u32x4 sc; uint32_t d[4]; };
    W6 V6[KIND == WRK_MAT_Q6_K ? BPS : 1];
    if (KIND == WRK_MAT_Q6_K) {
        const uint8_t* drow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) drow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + (size_t)nb * 208;
#pragma unroll
        for (int u = 0; u < BPS; ++u) {
            const uint32_t b = b0 + u;
#pragma unroll
            for (int j = 0; j < 4; ++j) V6[u].ql[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);       // j = 2 n128 + (kq & 1)
#pragma unroll
            for (int h = 0; h < 2; ++h) V6[u].qh[h] = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 64 + h * 32 + 8 * g);
            V6[u].sc = *(const u32x4*)(wrow + (size_t)nb * 192 + (size_t)b * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) V6[u].d[i] = *(const uint16_t*)(drow[i] + (size_t)b * 2);
        }
    } else if (KIND == WRK_MAT_F16) {
        const f16* wr = (const f16*)wrow;
#pragma unroll
        for (int u = 0; u < BPS; ++u)
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) WF[u][sb] = *(const f16x8*)(wr + (size_t)(b0 + u) * 256 + sb * 32 + 8 * g);
    } else if (KIND == WRK_MAT_Q8_0) {
        // device row: K int8 codes, then one f16 scale per 32-block (K % 256 == 0 here: the scales of a 256-slice are 16 aligned bytes)
        const uint8_t* drow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) drow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + K;
#pragma unroll
        for (int u = 0; u < BPS; ++u) {
            const uint32_t b = b0 + u;
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) V8[u].q[sb] = *(const u32x2*)(wrow + (size_t)b * 256 + sb * 32 + 8 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) V8[u].d[i] = *(const u32x4*)(drow[i] + (size_t)b * 16);
        }
    } else {
        const uint32_t hoff = KIND == WRK_MAT_Q4_K ? nb * 128 : nb * 160, soff = hoff + nb * 4;
        const uint8_t* crow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) crow[i] = P.w + (size_t)min(m0 + 4 * g + i, P.m - 1) * P.row_bytes + hoff;
#pragma unroll
        for (int u = 0; u < BPS; ++u) {
            const uint32_t b = b0 + u;
#pragma unroll
            for (int j = 0; j < 4; ++j) W[u].q[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
            if (KIND == WRK_MAT_Q5_K) W[u].qh = *(const u32x2*)(wrow + (size_t)nb * 128 + (size_t)b * 32 + 8 * g);
            W[u].sm = *(const u32x4*)(wrow + soff + (size_t)b * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) W[u].dd[i] = *(const uint32_t*)(crow[i] + (size_t)b * 4);
        }
    }
    uint32_t resb[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) resb[t][i] = 0;
    load_residual<NT>(P, 0, m0, r, g, resb);
    // ---- the activation slice, once per workgroup, through LDS
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const uint32_t c = tid + 256u * i, tok = c / CPT, ch = c % CPT;
        *(f16x8*)(lds + tok * ROWF + ch * 8) = stage[i];
    }
    __syncthreads();
    WRK_STAMP(P.dbg, 1);

    f32x4v total[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const f16* xt = lds + r * ROWF + 8 * g;             // B fragment of token tile t, block u, sub-block sb: xt + 16 t ROWF + 256 u + 32 sb
#pragma unroll
    for (int u = 0; u < BPS; ++u) {
        if (KIND == WRK_MAT_F16) {
#pragma unroll
            for (int sb = 0; sb < 8; ++sb)
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t] = mfma16(WF[u][sb], *(const f16x8*)(xt + 16 * t * ROWF + 256 * u + 32 * sb), total[t]);
        } else if (KIND == WRK_MAT_Q8_0) {      // as gemm_body: A = the int8 code, one MFMA per 32-block, its scale applied to the f32 sum
            const W8& R = V8[KIND == WRK_MAT_Q8_0 ? u : 0];
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) {
                const u32x2 q = R.q[sb];
                const f16x8 a = add8(mul8(codes8(q.x ^ 0x80808080u, q.y ^ 0x80808080u), 32768.0f), -0.25f);       // (code + 128) * 2^-9 - 128 * 2^-9
                float dsc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dsc[i] = (float)__builtin_bit_cast(f16, (uint16_t)(R.d[i][sb >> 1] >> (16 * (sb & 1)))) * 512.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4v acc = mfma16(a, *(const f16x8*)(xt + 16 * t * ROWF + 256 * u + 32 * sb), (f32x4v){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int i = 0; i < 4; ++i) total[t][i] = __builtin_fmaf(dsc[i], acc[i], total[t][i]);
                }
            }
        } else if (KIND == WRK_MAT_Q6_K) {      // as gemm_body: A = (q6 - 32) * sc with sc = 2 s1 + s0 over two MFMAs, d per 256-block in f32
            const W6& R = V6[KIND == WRK_MAT_Q6_K ? u : 0];
            const uint32_t gsh = 8 * (g >> 1);
            f32x4v acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n128 = 0; n128 < 2; ++n128) {
                const u32x2 qh = R.qh[n128];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const u32x2 ql = R.ql[2 * n128 + (kq & 1)];
                    const uint32_t sh = 2 * kq;
                    uint32_t c0, c1;
                    if (kq < 2) { c0 = (ql.x & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = (ql.y & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                    else { c0 = ((ql.x >> 4) & 0x0f0f0f0fu) | (((qh.x >> sh) & 0x03030303u) << 4); c1 = ((ql.y >> 4) & 0x0f0f0f0fu) | (((qh.y >> sh) & 0x03030303u) << 4); }
                    const f16x8 c = add8(mul8(codes8(c0, c1), 32768.0f), -0.0625f);
                    const uint32_t word = R.sc[2 * n128 + (kq >> 1)];
                    const int sc = (int)(int8_t)((word >> (16 * (kq & 1) + gsh)) & 0xffu);
                    const int s1 = sc >> 1, s0 = sc & 1;
                    const f16x8 a1 = mul8(c, (float)(2 * s1)), a0 = mul8(c, (float)s0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x8 bfr = *(const f16x8*)(xt + 16 * t * ROWF + 256 * u + n128 * 128 + kq * 32);
                        acc[t] = mfma16(a1, bfr, acc[t]);
                        acc[t] = mfma16(a0, bfr, acc[t]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dd = (float)__builtin_bit_cast(f16, (uint16_t)R.d[i]) * 512.0f;      // * 2^9
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t][i] = __builtin_fmaf(dd, acc[t][i], total[t][i]);
            }
        } else {
            const WBlk& R = W[(KIND == WRK_MAT_F16 || KIND == WRK_MAT_Q6_K || KIND == WRK_MAT_Q8_0) ? 0 : u];
            f32x4v acc[NT], amin[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; amin[t] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x2 q = R.q[j];
                const uint32_t v = R.sm[j];
                const float sc0 = (float)(v & 0xffu), sc1 = (float)((v >> 8) & 0xffu);
                f16x8 alo, ahi;
                if (KIND == WRK_MAT_Q4_K) {
                    alo = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), sc0 * 1024.0f);          // q*sc*2^-14
                    ahi = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), sc1 * 64.0f);            // (16q)*sc*2^-18
                } else {
                    const uint32_t s0 = 2 * j, s1 = 2 * j + 1;
                    alo = mul8(codes8((q.x & 0x0f0f0f0fu) | (((R.qh.x >> s0) & 0x01010101u) << 4), (q.y & 0x0f0f0f0fu) | (((R.qh.y >> s0) & 0x01010101u) << 4)), sc0 * 1024.0f);
                    ahi = mul8(codes8(((q.x >> 4) & 0x0f0f0f0fu) | (((R.qh.x >> s1) & 0x01010101u) << 4), ((q.y >> 4) & 0x0f0f0f0fu) | (((R.qh.y >> s1) & 0x01010101u) << 4)), sc1 * 1024.0f);
                }
                const f16 m0h = (f16)(float)((v >> 16) & 0xffu), m1h = (f16)(float)(v >> 24);
                const f16x8 mlo = {m0h, m0h, m0h, m0h, m0h, m0h, m0h, m0h}, mhi = {m1h, m1h, m1h, m1h, m1h, m1h, m1h, m1h};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f16x8 bq0 = *(const f16x8*)(xt + 16 * t * ROWF + 256 * u + 64 * j);
                    const f16x8 bq1 = *(const f16x8*)(xt + 16 * t * ROWF + 256 * u + 64 * j + 32);
                    acc[t] = mfma16(alo, bq0, acc[t]);
                    acc[t] = mfma16(ahi, bq1, acc[t]);
                    amin[t] = mfma16(mlo, bq0, amin[t]);
                    amin[t] = mfma16(mhi, bq1, amin[t]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] & 0xffffu)) * 16384.0f;
                const float dmin = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd[i] >> 16));
#pragma unroll
                for (int t = 0; t < NT; ++t) total[t][i] += d * acc[t][i] - dmin * amin[t][i];
            }
        }
    }
    WRK_STAMP(P.dbg, 2);

    // ---- K slices meet: partial tile out, count, the last arriver adds them up in slice order
    if (E.nslices > 1) {
        const size_t tile = (size_t)16 * NT * 64;                              // floats per (row group, slice)
        float* p0 = part + (size_t)(E.rg_begin + rg) * E.nslices * tile;
        // Agent-scope coherence PER ACCESS (sc1 write-through stores, sc1 loads, an sc1 counter), not cache-wide fences: the first version
        // used release / acquire fences (buffer_wbl2 + buffer_inv per workgroup) and ran 4x SLOWER than the kernels it replaces -- every
        // workgroup wrote back and invalidated its XCD's whole L2, activations included.  Order: this wave's tile stores are performed
        // (vmcnt(0)) before the workgroup barrier, the counter is bumped after it; the last arriver's loads are issued after it has seen
        // the count.  ISA assumptions (gfx942 / gfx950): `sc1` on a global store = write-through to agent (device) scope, `sc1` on a load =
        // served at agent scope (not from this XCD's L2 copy); `s_waitcnt vmcnt(0)` returns only when the wave's stores have been acknowledged
        // at that scope (loads and stores share the counter and return in order).  Nothing here orders OTHER addresses: only the tiles and the
        // counter are exchanged.  WRK_GEMM_KS=0 switches the kernel off; the step program zeroes the counters before every step.
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float* pp = p0 + (size_t)z * tile + (size_t)(16 * t + r) * 64 + wave * 16 + 4 * g;
            // one 16-byte write-through store per tile.  The s_nop covers the VMEM-store-data hazard (a store of more than 8 bytes followed
            // by a VALU write of its data registers): the compiler's hazard pass does not look inside an asm statement
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(pp), "v"(total[t]) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *sh_flag = __hip_atomic_fetch_add(counters + E.rg_begin + rg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*sh_flag != E.nslices - 1) return;                                 // uniform over the workgroup
        if (tid == 0) __hip_atomic_store(counters + E.rg_begin + rg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        WRK_STAMP(P.dbg, 3);
        // all partial tiles of the row group in ONE round trip: up to PF slices x NT vector loads in flight, added in slice order
        constexpr int PF = NT == 1 ? 8 : 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) total[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        for (uint32_t z0 = 0; z0 < E.nslices; z0 += PF) {
            f32x4v pv[PF][NT];
            const float* pa[PF][NT];
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) pa[u][t] = p0 + (size_t)min(z0 + u, E.nslices - 1) * tile + (size_t)(16 * t + r) * 64 + wave * 16 + 4 * g;
            // eight agent-scope vector loads and their wait in ONE statement: the compiler cannot see that an asm load completes later, so
            // the values may only leave the statement once they have arrived (separate load / wait statements let it copy a register early)
            static_assert(PF * NT == 8, "eight loads per statement");
            f32x4v(&pf)[8] = *reinterpret_cast<f32x4v(*)[8]>(&pv[0][0]);
            const float* const(&af)[8] = *reinterpret_cast<const float* const(*)[8]>(&pa[0][0]);
            asm volatile(
                "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
                "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                : "=&v"(pf[0]), "=&v"(pf[1]), "=&v"(pf[2]), "=&v"(pf[3]), "=&v"(pf[4]), "=&v"(pf[5]), "=&v"(pf[6]), "=&v"(pf[7])
                : "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(af[4]), "v"(af[5]), "v"(af[6]), "v"(af[7])
                : "memory");
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (z0 + u < E.nslices)
#pragma unroll
                    for (int t = 0; t < NT; ++t) total[t] += pv[u][t];
        }
    }

    // store: lane owns rows m0 + 4g + (0..3) of token column r of each tile
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t tok = 16 * t + r;
        if (tok >= P.n) continue;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = act_apply(P.act, total[t][i] * P.scale);
            if (P.has_res) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? __builtin_bit_cast(float, resb[t][i]) : (float)__builtin_bit_cast(f16, (uint16_t)resb[t][i]));
        }
        const size_t oo = (size_t)tok * P.os + m0 + 4 * g;
        if (m0 + 4 * g + 4 <= P.m) {
            if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
            else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m0 + 4 * g + i < P.m) { if (P.out32) ((float*)P.out_p)[oo + i] = o[i]; else ((f16*)P.out_p)[oo + i] = (f16)o[i]; }
        }
    }
    WRK_STAMP(P.dbg, 4);
}

template <int NT, int BPS>
__global__ void __launch_bounds__(256) gemm_ks_kernel(const KsBatch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    __shared__ uint32_t sh_flag;
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.g.njobs && blockIdx.x >= B.g.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.g.jobs[ji];
    f16* lds = (f16*)ks_smem;
    switch (P.kind) {
        case WRK_MAT_Q4_K: gemm_ks_body<WRK_MAT_Q4_K, NT, BPS>(P, B.ks[ji], B.part, B.counters, lds, &sh_flag); break;
        case WRK_MAT_Q5_K: gemm_ks_body<WRK_MAT_Q5_K, NT, BPS>(P, B.ks[ji], B.part, B.counters, lds, &sh_flag); break;
        case WRK_MAT_Q6_K: gemm_ks_body<WRK_MAT_Q6_K, NT, BPS>(P, B.ks[ji], B.part, B.counters, lds, &sh_flag); break;
        case WRK_MAT_Q8_0: gemm_ks_body<WRK_MAT_Q8_0, NT, BPS>(P, B.ks[ji], B.part, B.counters, lds, &sh_flag); break;
        default: gemm_ks_body<WRK_MAT_F16, NT, BPS>(P, B.ks[ji], B.part, B.counters, lds, &sh_flag); break;
    }
}

// fewest stacked tokens sent to the matrix cores (tiles are padded to 16 tokens; below this the matvec kernels run)
uint32_t gemm_min_tokens() {
    static const uint32_t v = [] { const char* e = getenv("WRK_GEMM_MIN"); const int x = e ? atoi(e) : 2; return (uint32_t)(x < 2 ? 2 : x); }();     // measured (1.5B decode): 2 sequences break even, 3: 1.33 vs 1.79 ms, 8: 2x in favour of MFMA
    return v;
}

// token tok = t + b * shape[1] of a [C, T, B] view sits at base + tok * stride[0] when the view covers the whole T extent of
// its parent (or has a single batch)
static bool dense_stack(const DTensor& d) { return d.shape[2] == 1 || (d.shape[1] == d.stride[1] && d.offset[1] == 0); }
static size_t stack_base(const DTensor& d) { return ((size_t)d.offset[2] * d.stride[1] + d.offset[1]) * d.stride[0] + d.offset[0]; }

static bool gemm_ok(const MatJob& j, uint32_t n) {
    if (j.in.shape[1] * j.in.shape[2] != n || n < gemm_min_tokens()) return false;
    if (!dense_stack(j.in) || !dense_stack(j.out) || (j.has_res && !dense_stack(j.res))) return false;
    if (j.m < 4 || (j.m & 3u) || (j.out.dtype != WRK_F16 && j.out.dtype != WRK_F32) || (j.has_res && j.res.dtype != WRK_F16 && j.res.dtype != WRK_F32)) return false;
    // four consecutive output rows are stored (and residual rows loaded) as one vector
    if (((stack_base(j.out) | j.out.stride[0]) & 3u) || (j.has_res && ((stack_base(j.res) | j.res.stride[0]) & 3u))) return false;
    if (j.k < 32) return false;
    if (j.flags & WRK_MATRIX_ROUND_F16) return false;       // parity mode: per-element f16 rounding lives in the matvec kernels
    if (j.in.dtype != WRK_F16 || (j.k & 31u)) return false;
    // rows of the input views must be 16-byte aligned for the B-fragment loads
    if ((j.in.stride[0] & 7u) || (j.in.offset[0] & 7u)) return false;
    switch (j.kind) {
        case WRK_MAT_F16: case WRK_MAT_Q8_0: case WRK_MAT_Q6_K: case WRK_MAT_Q4_K: case WRK_MAT_Q5_K: return true;
        case WRK_MAT_INT8: return (j.k & 127u) == 0;      // rows aligned to the 128-element blocks
        case WRK_MAT_NF4: return (j.k & 63u) == 0 && j.aux != nullptr;
        default: return false;
    }
}

// All jobs multiply the same number of tokens; they run in ONE launch.  Returns -2 when any job is not for the MFMA
// path (n < 16, ROUND_F16, non-f16 / unaligned input, K % 32, Int8 / NF4): the caller falls back to the matvec kernels.
static void fill_job(GemmParams& P, const MatJob& j, uint32_t n, uint32_t wg_begin) {
    P.w = j.w; P.kind = j.kind; P.k = j.k; P.m = j.m; P.row_bytes = j.row_bytes; P.act = j.act; P.n = n;
    P.has_res = j.has_res; P.in = j.in; P.out = j.out; P.res = j.res; P.wg_begin = wg_begin; P.scale = j.scale; P.dbg = j.dbg;
    P.x = (const f16*)j.in.p + stack_base(j.in); P.xs = j.in.stride[0];
    P.out32 = j.out.dtype == WRK_F32; P.os = j.out.stride[0];
    P.out_p = (char*)j.out.p + stack_base(j.out) * (P.out32 ? 4 : 2);
    P.res32 = j.has_res && j.res.dtype == WRK_F32; P.rs = j.has_res ? j.res.stride[0] : 0;
    P.res_p = j.has_res ? (const char*)j.res.p + stack_base(j.res) * (P.res32 ? 4 : 2) : nullptr;
    P.levels = (const float*)j.aux;
}

// the K-sliced kernel: 0 = launched, -1 = not applicable (the caller continues with the other kernels)
static int launch_ks(hipStream_t s, const MatJob* jobs, int njobs, uint32_t n) {
    // Measured (1.5B Q4_K_M decode, ms per step: K-split kernels | this kernel for ffn.value only | for every launch; profiles/r02_ks_modes.txt):
    //   5 seq 1.265 | 1.222 | 1.382    8: 1.319 | 1.251 | 1.391    16: 1.498 | 1.390 | 1.468    24: 1.962 | 1.899 | 1.877
    //   32: 2.158 | 2.044 | 1.967      48: 3.150 | 3.981 | 3.803   64: 3.474 | 4.295 | 4.011
    // -> up to 16 tokens only the launch with few row tiles and long rows (ffn.value: 2048 rows x K = 8192, 13.8 -> 8.7 us at 16 tokens),
    // 17 .. 32 tokens every eligible launch, beyond that none (a 64-token slice is 32 KB per 256-block: the ingest problem again).
    // Each phase of the kernel is one memory round trip of ~2 us under load (stage | weights | tile out + count | tiles in), which is why it
    // only wins where the K-split kernel needs four dependent block iterations.  WRK_GEMM_KS: 0 off | 1 as above | 2 every eligible launch.
    // (read per call, not once per process: the tests switch it; launches are captured into graphs, so this is off the replay path)
    const char* mode_env = getenv("WRK_GEMM_KS");
    const int mode = mode_env ? atoi(mode_env) : 1;
    if (mode <= 0 || !jobs[0].ks_part || !jobs[0].ks_cnt || n > 32) return -1;
    const int nt = n <= 16 ? 1 : 2;
    bool all_q8 = false, any_other = false;
    for (int q = 0; q < njobs; ++q) {
        if (jobs[q].kind == WRK_MAT_Q8_0) all_q8 = true;
        else if (jobs[q].kind != WRK_MAT_F16) any_other = true;
    }
    all_q8 = all_q8 && !any_other;
    if (mode == 1 && n <= 16) {
        uint32_t tiles16 = 0, kmax = 0;
        for (int q = 0; q < njobs; ++q) { tiles16 += (jobs[q].m + 15) / 16; kmax = jobs[q].k > kmax ? jobs[q].k : kmax; }
        // round 3 (tools/ks_v6_sweep.sh, three RWKV-6 layers with D = 4096, ms per step, K-split kernels | this kernel with 2 | 4 blocks per slice):
        //   Q5_K x 16 streams 0.508 | 0.515 | 0.570, x 8 0.429 | - | 0.538        -> the K4 kinds keep the rule above
        //   Q8_0 x 16 streams 0.808 | 0.542 | 0.631, x 8 0.725 | - | 0.605        -> Q8_0 launches always come here (the K-split kernel's Q8_0
        //   body loads four row scales per 32-block and lane; here a slice's scales are one 16-byte load per row), two blocks per slice
        if (!(all_q8 || (tiles16 < 256 && kmax >= 4096))) return -1;
    }
    for (int q = 0; q < njobs; ++q) {
        const MatJob& j = jobs[q];
        if (j.kind != WRK_MAT_Q4_K && j.kind != WRK_MAT_Q5_K && j.kind != WRK_MAT_Q6_K && j.kind != WRK_MAT_Q8_0 && j.kind != WRK_MAT_F16) return -1;
        if (j.kind == WRK_MAT_Q8_0 && (j.row_bytes & 15u)) return -1;       // the block scales of a slice are loaded as 16-byte vectors
        if ((j.k & 255u) || j.k < 256) return -1;
    }
    // blocks per slice: as many as still give every CU a workgroup (fewer slices = fewer partial tiles to add); the staged activation
    // slice (16 nt tokens x bps x 256) has to fit the default 64 KB of dynamic LDS
    static const int force_bps = [] { const char* e = getenv("WRK_KS_BPS"); return e ? atoi(e) : 0; }();
    int bps = 0;
    for (int cand = (all_q8 && !force_bps) ? 2 : 4; cand >= 1 && !bps; cand >>= 1) {
        if (nt * cand > 4) continue;
        if (force_bps && cand != force_bps) continue;
        bool div = true;
        uint32_t wgs = 0;
        for (int q = 0; q < njobs; ++q) { const uint32_t nb = jobs[q].k >> 8; div = div && nb % cand == 0; wgs += ((jobs[q].m + 63) / 64) * (nb / cand); }
        if (div && (wgs >= 256 || cand == 1 || force_bps)) bps = cand;
    }
    if (!bps) return -1;
    KsBatch B;
    B.g.njobs = njobs;
    B.part = jobs[0].ks_part; B.counters = jobs[0].ks_cnt; B.ntp = 16u * nt;
    uint32_t wg = 0, rg0 = 0;
    size_t floats = 0;
    for (int q = 0; q < njobs; ++q) {
        const uint32_t nsl = (jobs[q].k >> 8) / bps, nrg = (jobs[q].m + 63) / 64;
        if (nsl > 64) return -1;
        fill_job(B.g.jobs[q], jobs[q], n, wg);
        B.ks[q].nslices = nsl; B.ks[q].rg_begin = rg0;
        wg += nrg * nsl; rg0 += nrg;
        floats += (size_t)nrg * nsl * 64 * 16 * nt;
    }
    // (row groups of one job are contiguous in the partial space: rg_begin counts row groups, the slices of a group sit side by side,
    // so the partial offset of job q is rg_begin x nslices_q tiles only when all jobs share nslices -- they do: one bps, and K differs
    // only between launches; checked here)
    for (int q = 1; q < njobs; ++q)
        if (B.ks[q].nslices != B.ks[0].nslices) return -1;
    if (floats > jobs[0].ks_part_cap || rg0 > jobs[0].ks_cnt_cap) return -1;
    const size_t smem = (size_t)16 * nt * (bps * 256 + 8) * sizeof(f16);
#define KS_LAUNCH(NT_, BPS_)                                                                                                              \
    do {                                                                                                                                  \
        if (smem > 64 * 1024) {                                                                                                           \
            if (!lds_attr_once((const void*)gemm_ks_kernel<NT_, BPS_>, smem)) return -1;                                                  \
        }                                                                                                                                 \
        gemm_ks_kernel<NT_, BPS_><<<dim3(wg), 256, smem, s>>>(B);                                                                         \
    } while (0)
    if (nt == 1) { if (bps == 4) KS_LAUNCH(1, 4); else if (bps == 2) KS_LAUNCH(1, 2); else KS_LAUNCH(1, 1); }
    else { if (bps == 2) KS_LAUNCH(2, 2); else KS_LAUNCH(2, 1); }
#undef KS_LAUNCH
    return 0;
}

int matmul_mfma_multi(hipStream_t s, const MatJob* jobs, int njobs, int) {
    if (njobs <= 0 || njobs > GEMM_MAX_JOBS) return -2;
    const uint32_t n = jobs[0].in.shape[1] * jobs[0].in.shape[2];
    for (int j = 0; j < njobs; ++j)
        if (!gemm_ok(jobs[j], n)) return -2;
    if (launch_ks(s, jobs, njobs, n) == 0) return 0;
    // prefill regime: Q4_K / Q5_K / Q6_K / F16 matrices with >= 64 rows go to the LDS-tiled kernel, the rest (Q8_0, Int8, short)
    // to the K-split kernel, each group in one launch
    static const bool use_tile = [] { const char* e = getenv("WRK_GEMM_TILE"); return !(e && e[0] == '0'); }();
    // measured (round 1): SLOWER than the K-split kernel -- 12.2 vs 8.9 us for 8192 x 2048 x 16 tokens, batch-16 decode 2.35 vs
    // 1.76 ms -- 128 workgroups leave half the CUs idle and every 256-block costs a workgroup barrier.  Off unless WRK_GEMM_DEC=1.
    static const bool use_dec = [] { const char* e = getenv("WRK_GEMM_DEC"); return e && e[0] == '1'; }();
    GemmBatch T, B, Dq, T2, T3;
    T.njobs = B.njobs = Dq.njobs = T2.njobs = T3.njobs = 0;
    uint32_t twg = 0, wg = 0, kmax = 0, dwg = 0, t2wg = 0, t3wg = 0;
    // third-generation prefill tile (wrk_gemm3.hip): Q4_K, >= 128 stacked tokens, the launch's sum scratch at hand.  WRK_GEMM_TILE3=0: off
    // (read per call: the tests compare the kernels)
    const char* t3e = getenv("WRK_GEMM_TILE3");
    const bool use_tile3 = !(t3e && t3e[0] == '0') && jobs[0].xsum != nullptr && n >= 128;        // (round 3: from 128 tokens on, K split over workgroups for one or two token tiles)
    static const bool use_tile2 = [] { const char* e = getenv("WRK_GEMM_TILE2"); return !(e && e[0] == '0'); }();
    for (int q = 0; q < njobs; ++q) {
        const MatJob& j = jobs[q];
        // below 512 tokens the tile runs K-split (wrk_gemm3.hip) and pays a sum pre-pass and a reduce launch: measured (1.5B, tokens/s, this tile |
        // the kernels below): 128 tokens 21.5 k | 21.6 k with every matrix on it -- only the long rows gain (ffn value, K = 8192: 49.5 us -> 16 + 6.5 + 5);
        // 256 tokens 38.9 k | 35.8 k; 384 tokens (three token tiles, no split) 37.8 k | 46.6 k
        const bool t3_n = n >= 512 || (n > 128 && n <= 256) || (n == 128 && j.k >= 4096);
        if (use_tile3 && t3_n && (j.kind == WRK_MAT_Q4_K || j.kind == WRK_MAT_Q5_K) && j.m >= 128 && j.in.shape[2] == 1 && (j.k & 255u) == 0 && T3.njobs < GEMM_MAX_JOBS) {
            fill_job(T3.jobs[T3.njobs++], j, n, t3wg);
            t3wg += (j.m + 127) / 128;
            continue;
        }
        // decode batches (<= 16 tokens): rows up to 24 blocks long go to the 64-row LDS tile; longer rows keep the K split
        if (use_dec && n <= 16 && (j.kind == WRK_MAT_Q4_K || j.kind == WRK_MAT_Q5_K) && j.m >= 64 && j.in.shape[2] == 1 && (j.k >> 8) <= 24) {
            fill_job(Dq.jobs[Dq.njobs++], j, n, dwg);
            dwg += (j.m + TILE_ROWS - 1) / TILE_ROWS;
            continue;
        }
        // the tile kernel walks the whole K in every wave: it needs enough workgroups to fill the chip (a 2048 x 8192
        // matrix x 128 tokens has only 64 tiles and is faster on the K-split kernel: 44 vs 74 us)
        const uint32_t tiles = ((j.m + TILE_ROWS - 1) / TILE_ROWS) * ((n + TILE_TOK - 1) / TILE_TOK);
        // (Q8_0 has a body in the second-generation tile kernel only: >= 512 tokens, K % 128 == 0, rows in fours)
        // (round 3: from 48 tokens on -- the first-generation tile has no Q8_0 body, and the K-split kernel re-reads the activations per 16 rows)
        const bool q8_tile2 = j.kind == WRK_MAT_Q8_0 && use_tile2 && n >= 48 && (j.k & 127u) == 0 && (j.m & 3u) == 0;
        const bool tile = use_tile && n >= 48 && (j.kind == WRK_MAT_Q4_K || j.kind == WRK_MAT_Q5_K || j.kind == WRK_MAT_Q6_K || j.kind == WRK_MAT_F16 || q8_tile2) && j.m >= 64 &&
                          j.in.shape[2] == 1 && (tiles >= 96 || (j.k <= 2560 && tiles >= 64) ||    // enough workgroups, or a short serial walk,
                                                 (j.kind == WRK_MAT_F16 && j.k <= 2560));          // or a LoRA down-projection (64..320 rows): a handful of
        // tiles that ride along with the big matrices of their stage; on the K-split kernel they were a 32-us launch of 64 workgroups per layer
        // K4 kinds whose output rows come in fours (one vector store per lane) take the second-generation tile kernel
        // (with few workgroups -- a single 128-token chunk -- the first-generation kernel's longer stages are 4 % faster)
        static const bool f16_tile2_on = [] { const char* e = getenv("WRK_GEMM_TILE2_F16"); return !(e && e[0] == '0'); }();
        const bool f16_tile2 = f16_tile2_on && j.kind == WRK_MAT_F16 && (j.k & 127u) == 0 && (j.row_bytes & 15u) == 0;
        if (tile && use_tile2 && (n >= 512 || q8_tile2) && (j.kind == WRK_MAT_Q4_K || j.kind == WRK_MAT_Q5_K || q8_tile2 || f16_tile2) && (j.m & 3u) == 0) { fill_job(T2.jobs[T2.njobs++], j, n, t2wg); t2wg += (j.m + TILE_ROWS - 1) / TILE_ROWS; }
        else if (tile) { fill_job(T.jobs[T.njobs++], j, n, twg); twg += (j.m + TILE_ROWS - 1) / TILE_ROWS; }
        else { fill_job(B.jobs[B.njobs++], j, n, wg); wg += (j.m + 15) / 16; kmax = j.k > kmax ? j.k : kmax; }
    }
    if (T3.njobs && gemm_tile3_launch(s, T3, t3wg, n, jobs[0].xsum, jobs[0].xsum_cap) != 0) {
        // (scratch too small for this launch: the second-generation tile takes the jobs)
        for (int q = 0; q < T3.njobs; ++q) { T2.jobs[T2.njobs] = T3.jobs[q]; T2.jobs[T2.njobs].wg_begin = t2wg; t2wg += (T3.jobs[q].m + TILE_ROWS - 1) / TILE_ROWS; ++T2.njobs; }
    }
    if (T2.njobs) {
        // 64-token tiles (TT = 4).  Measured on MI355X (round 2, 32 x 128-token prefill, tokens/s, 1.5B | 2.9B): first-generation tile
        // 86.8 k | 44.1 k; this kernel TT = 4 96.2 k | 49.0 k, TT = 6 96.8 k | 48.6 k, TT = 8 83.3 k | 42.1 k (308 registers: one
        // workgroup per CU, or spills at 256) -- more MFMAs per dequantised fragment buy nothing, the LDS fragment reads (1 KB per two
        // MFMAs) are the co-limiter; the gain is the unconditional loads and the shorter stages.
        const size_t smem2 = (size_t)2 * 64 * T2_ROW * sizeof(f16);       // 34 816 B
        gemm_tile2_kernel<4><<<dim3(t2wg, (n + 63) / 64), 256, smem2, s>>>(T2);
    }
    if (T.njobs) {
        const size_t smem = (size_t)2 * TILE_TOK * TILE_LDS_ROW * sizeof(f16);       // 67 584 B
        if (!lds_attr_once((const void*)gemm_tile_kernel, smem)) return -1;
        gemm_tile_kernel<<<dim3(twg, (n + TILE_TOK - 1) / TILE_TOK), 256, smem, s>>>(T);
    }
    if (Dq.njobs) gemm_dec_kernel<<<dim3(dwg), 256, 0, s>>>(Dq);
    if (B.njobs) {
        // tokens per wave: enough tiles to amortise the decode, few enough to keep >= ~2 waves per SIMD;
        // few row tiles x long rows (decode batches through ffn.value): 8 waves split K so a wave's serial chain is short
        const bool deep = wg < 256 && kmax >= 4096;
        // (measured, round 1: 16 waves x 2 blocks for K = 8192 is SLOWER, 23 vs 12 us at 2..16 tokens: 128 VGPRs per lane at 1024
        // threads spill the all-kinds kernel to scratch)
        if (n > 64) gemm_kernel<4, 4><<<dim3(wg, (n + 63) / 64), 256, 0, s>>>(B);
        else if (n > 16) { if (deep) gemm_kernel<2, 8><<<dim3(wg, (n + 31) / 32), 512, 0, s>>>(B); else gemm_kernel<2, 4><<<dim3(wg, (n + 31) / 32), 256, 0, s>>>(B); }
        else {
            // two row tiles per workgroup where that still leaves >= ~200 workgroups (r, k, v + LoRA, ffn key at 16 tokens).  MEASURED SLOWER
            // (round 3, 1.5B decode, ms per step, pairs | single tiles: 8 sequences 1.339 | 1.244, 16 sequences 1.456 | 1.377): halving the
            // activation re-reads does not pay for twice the serial blocks per wave -- these launches are bound by the dependent chain of a
            // workgroup, not by the L2 bandwidth of the shared stack.  Off unless WRK_GEMM_PAIR=1 (kept for the A/B and its test).
            const char* n8 = getenv("WRK_GEMM_NW8");           // A/B: eight waves split K for every launch of <= 16 tokens
            const bool nw8 = n8 && n8[0] == '1';
            const char* pe = getenv("WRK_GEMM_PAIR");
            bool pair = pe && pe[0] == '1' && !deep && wg >= 400;
            for (int q = 0; q < B.njobs; ++q) pair = pair && (B.jobs[q].kind == WRK_MAT_Q4_K || B.jobs[q].kind == WRK_MAT_Q5_K || B.jobs[q].kind == WRK_MAT_F16);
            if (pair) {
                uint32_t wg2 = 0;
                for (int q = 0; q < B.njobs; ++q) { B.jobs[q].wg_begin = wg2; wg2 += (B.jobs[q].m + 31) / 32; }
                gemm_pair_kernel<4><<<dim3(wg2, (n + 15) / 16), 256, 0, s>>>(B);
            }
            else if (deep || nw8) gemm_kernel<1, 8><<<dim3(wg, (n + 15) / 16), 512, 0, s>>>(B);
            else gemm_kernel<1, 4><<<dim3(wg, (n + 15) / 16), 256, 0, s>>>(B);
        }
    }
    return 0;
}

int matmul_mfma(hipStream_t s, const MatJob& j, int num_cu) { return matmul_mfma_multi(s, &j, 1, num_cu); }

}  // namespace wrk
