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

#include "wrk_matvec_dev.h"

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

// the kinds of a launch
struct KindMix { int quant = -1, quant2 = -1, nquant = 0; bool has_f16 = false, r16 = false, mixed_r16 = false; };
static KindMix classify_kinds(const MatvecParams& P) {
    KindMix c;
    for (int j = 0; j < P.njobs; ++j) {
        const int k = (int)P.jobs[j].kind;
        if (k == WRK_MAT_F16) { c.has_f16 = true; continue; }
        const bool jr = (P.jobs[j].flags & WRK_MATRIX_ROUND_F16) != 0;
        if (c.nquant == 0) { c.quant = k; c.r16 = jr; c.nquant = 1; }
        else {
            if (k != c.quant && k != c.quant2) { c.quant2 = c.quant2 < 0 ? k : c.quant2; c.nquant = (k == c.quant2) ? 2 : 3; }
            if (jr != c.r16) c.mixed_r16 = true;
        }
    }
    return c;
}

template <int NB>
static int launch_matvec(hipStream_t s, const MatvecParams& P, uint32_t total_wg, uint32_t tok_groups, size_t smem, bool dry, bool no_catchall) {
    const KindMix c = classify_kinds(P);
    const int quant = c.quant, quant2 = c.quant2, nquant = c.nquant;
    const bool has_f16 = c.has_f16, r16 = c.r16, mixed_r16 = c.mixed_r16;
    matvec_fn fn = nullptr;
    bool needs_reg = false;     // fused prologue / state carry exist only in the register-input decode kernel
    for (int j = 0; j < P.njobs; ++j) needs_reg = needs_reg || P.jobs[j].pro || P.jobs[j].carry_dst || P.jobs[j].gate;
    if (NB == 1 && tok_groups == 1 && nquant <= 2 && !mixed_r16 && launch_dmv(s, P, total_wg, nquant ? quant : -1, has_f16, r16, dry, nquant == 2 ? quant2 : -1) == 0) return 0;
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
int matvec(hipStream_t s, const MatJob* jobs, int njobs, int num_cu, bool dry_run, bool no_catchall, bool dmv_only) {
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
        d.tok_prev_stride = jobs[j].tok_prev_stride; d.tok_mix_stride = jobs[j].tok_mix_stride; d.tok_carry_src_stride = jobs[j].tok_carry_src_stride;
        d.tok_carry_dst_stride = jobs[j].tok_carry_dst_stride; d.tok_gate_stride = jobs[j].tok_gate_stride;
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
    if (ntok >= 2 && ntok <= 4) {       // a few sequences decoding together: the dmv kernels with 2 / 4 tokens per launch
        const KindMix c = classify_kinds(P);
        if (c.nquant <= 2 && !c.mixed_r16 && launch_dmv(s, P, wg, c.nquant ? c.quant : -1, c.has_f16, c.r16, dry_run, c.nquant == 2 ? c.quant2 : -1) == 0) return 0;
    }
    if (dmv_only && ntok > 1) return -5;
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
