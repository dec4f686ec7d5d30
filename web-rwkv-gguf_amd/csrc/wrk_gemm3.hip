// Third-generation prefill tile GEMM for gfx950 (round 3): Y[M, N] = act(W[M, K] . X[K, N]) for Q4_K / Q5_K matrices and >= 128 stacked tokens.
//
// Replaces matmul_mat_q4k(_opt) (ops.rs:1332-1536, shaders/matmul_mat_q4k_opt.wgsl:86-229: 32 x 32 tile, f32 FMA).  Semantics: gguf.rs:95-143.
//
// What bounded the second-generation tile (wrk_gemm.hip gemm_tile2_kernel; 36.5 % matrix-pipe busy, HALF of it min-term MFMAs, 10 % of the
// dense f16 peak end to end -- profiles/r02_prefill_mfma_util.json):
//   * every wave dequantised its own 16 rows and fed each fragment to only 4 token tiles: 1 KB of LDS fragment reads per two MFMAs and
//     ~20 vector instructions of unpacking per 4 useful MFMAs;
//   * the K-quant min term  -dmin * m_s * sum_k x_k  rode the matrix core as a second MFMA per step (A = m_s splat).
// Here:
//   * a workgroup of 8 waves owns 128 rows x 128 tokens.  Each wave dequantises ONE 16-row tile per half block into an LDS A-tile that all
//     eight waves read; a wave multiplies 64 rows x 32 tokens (4 x 2 register blocking): per 32-k step 4 A + 2 B fragment reads feed 8 MFMAs
//     (0.75 reads per MFMA instead of 2), and the unpacking is done once per weight instead of once per token tile;
//   * the min term is ONE small GEMM over the sub-block input sums: S[tok][s] = sum of the 32 inputs of sub-block s (a pre-pass over X,
//     xsum_kernel, f32 split into hi + lo f16), A_min[row][s] = -dmin * m_s (17 significant bits: exact as hi + lo f16), and
//     total += A_min . S as three MFMAs (hi.hi, hi.lo, lo.hi) per 1024 k and C tile: +9 % MFMAs instead of +100 %;
//   * the main term is unchanged in value: A = q * sc (exact small integers in f16), accumulated per 256-block, total += d * acc in f32.
// LDS: A 2 x 128 x 128 f16 + X 2 x 128 x 128 f16 (swizzled, see t3_off) + per-row d 2 x 128 f32 + A_min hi / lo 2 x 128 x 40 f16 = 149 KB: one workgroup (16 waves would
// not fit the registers: 64 accumulator registers per wave) per CU, two waves per SIMD.
#include <cstdlib>
#include <type_traits>

#include "wrk_gemm_dev.h"

namespace wrk {

// LDS images of a half block, [128 rows][128 k] f16 = 256-byte rows WITHOUT padding: 16-byte chunk c of row r sits at chunk c ^ (r & 15).
// ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- so a fragment read
// (lane = row r + 16 g, chunk 4 ks + g) puts rows {0-3, 12-15} at chunk 4 ks and rows 4-11 at chunk 4 ks + 1 into one group: with the
// swizzle their slots are {r ^ 4 ks} and {(r ^ 1) ^ 4 ks}: all sixteen distinct.  The first build padded the rows by 16 bytes instead
// (slot = r + g + 4 ks): rows 11 (g = 1) and 12 (g = 0) collide in every group -- 37 % of its LDS cycles were conflict cycles.
constexpr int T3_ROWS = 128, T3_TOK = 128, T3_KH = 128, T3_LR = T3_KH, T3_MR = 32 + 8;
__device__ __forceinline__ uint32_t t3_off(uint32_t row, uint32_t chunk) { return row * (uint32_t)T3_LR + ((chunk ^ (row & 15u)) << 3); }      // f16 elements
constexpr size_t T3_LDS = (size_t)(2 * 128 * T3_LR + 2 * 128 * T3_LR) * 2 + 2 * 128 * 4 + (size_t)2 * 128 * T3_MR * 2;

struct T3Batch {
    GemmBatch g;
    const f16* sh[GEMM_MAX_JOBS];       // input sums, high parts: [token][K / 32]
    const f16* sl[GEMM_MAX_JOBS];       // low parts
    // K split over blockIdx.z (round 3: chunks of 128 .. 256 tokens -- one or two token tiles -- leave most CUs without a workgroup otherwise):
    // slice z multiplies blocks [z bps, (z + 1) bps) and leaves an f32 partial tile [z][token][row]; t3_reduce_kernel adds them in slice order
    uint32_t bps, kslices;              // kslices == 1: the whole K, epilogue in this kernel
    float* part;
    size_t poff[GEMM_MAX_JOBS];         // floats
};

// sums of the 32 inputs of every sub-block, f32, stored as hi + lo f16 (hi = round(s), lo = round(s - hi): 22 significant bits)
__global__ void __launch_bounds__(256) xsum_kernel(const f16* __restrict__ x, uint32_t xs, uint32_t n, uint32_t nsub, f16* __restrict__ sh, f16* __restrict__ sl) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= n * nsub) return;
    const uint32_t tok = idx / nsub, sub = idx - tok * nsub;
    const f16x8* p = (const f16x8*)(x + (size_t)tok * xs + sub * 32u);
    const f16x2 one = {(f16)1.0f, (f16)1.0f};
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f16x8 v = p[q];
        s = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 0, 1), one, s, false);
        s = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 2, 3), one, s, false);
        s = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 4, 5), one, s, false);
        s = __builtin_amdgcn_fdot2(__builtin_shufflevector(v, v, 6, 7), one, s, false);
    }
    const f16 h = (f16)s;
    sh[idx] = h;
    sl[idx] = (f16)(s - (float)h);
}

struct T3W { u32x2 q[4]; u32x4 sm; uint32_t dd; u32x2 qh; };        // qh: Q5_K high bits of this lane's 8 columns (bit s = sub-block s)

// Dequantise one 32-k piece (PC = 0 .. 3) of half HF (k = 128 HF .. +127) of a block of ONE 16-row tile (rows 16 * tile + r of the workgroup: a wave's own
// tile) into the A image.  Pieces are issued BETWEEN the MFMA groups of the half that is being multiplied (round 3, second build: the unpacking
// and the LDS stores of the next half run while the matrix pipe works through the current one instead of after it).
template <int HF, int PC>
__device__ __forceinline__ void t3_dequant_piece(const T3W& R, f16* __restrict__ dst, bool q5) {     // dst: As + t3_off(16 tile + r, 4 PC + g); q5: uniform
    constexpr int j = 2 * HF + (PC >> 1);
    const u32x2 q = R.q[j];
    const uint32_t v = R.sm[j];
    if (q5) {       // Q5_K (round 3, as gemm_body): the fifth bit of sub-block s = 2 j + (PC & 1) is bit s of the lane's qh bytes; q <= 31, q * sc <= 1953: exact
        constexpr uint32_t sb = 2 * j + (PC & 1);
        const uint32_t lo0 = (PC & 1) ? ((q.x >> 4) & 0x0f0f0f0fu) : (q.x & 0x0f0f0f0fu), lo1 = (PC & 1) ? ((q.y >> 4) & 0x0f0f0f0fu) : (q.y & 0x0f0f0f0fu);
        const uint32_t c0 = lo0 | (((R.qh.x >> sb) & 0x01010101u) << 4), c1 = lo1 | (((R.qh.y >> sb) & 0x01010101u) << 4);
        *(f16x8*)dst = mul8(codes8(c0, c1), (float)((PC & 1) ? ((v >> 8) & 0xffu) : (v & 0xffu)) * 1024.0f);
        return;
    }
    if ((PC & 1) == 0) *(f16x8*)dst = mul8(codes8(q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu), (float)(v & 0xffu) * 1024.0f);            // q * sc * 2^-14, exact
    else *(f16x8*)dst = mul8(codes8(q.x & 0xf0f0f0f0u, q.y & 0xf0f0f0f0u), (float)((v >> 8) & 0xffu) * 64.0f);
}
// the row's d and the block's min products (with the first half of a block)
__device__ __forceinline__ void t3_dequant_meta(const T3W& R, uint32_t b, uint32_t nb /* end of the slice */, float* __restrict__ Dd, f16* __restrict__ Amh, f16* __restrict__ Aml,
                                                uint32_t tile, uint32_t r, uint32_t g) {
    const float d = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd & 0xffffu)) * 16384.0f;
    const float dmin = (float)__builtin_bit_cast(f16, (uint16_t)(R.dd >> 16));
    if (g == 0) Dd[(b & 1u) * 128u + 16u * tile + r] = d;
    // lane g: sub-blocks 2g, 2g + 1 of the block (the mins of 64-element group g)
    const uint32_t v = g == 0 ? R.sm[0] : g == 1 ? R.sm[1] : g == 2 ? R.sm[2] : R.sm[3];
    const float p0 = -(dmin * (float)((v >> 16) & 0xffu)), p1 = -(dmin * (float)(v >> 24));
    const f16 h0 = (f16)p0, h1 = (f16)p1;
    const f16x2 hh = {h0, h1}, ll = {(f16)(p0 - (float)h0), (f16)(p1 - (float)h1)};
    const uint32_t col = (b & 3u) * 8u + 2u * g, rowo = (16u * tile + r) * T3_MR;
    *(f16x2*)(Amh + rowo + col) = hh;
    *(f16x2*)(Aml + rowo + col) = ll;
    if (b + 1 == nb) {                      // last block: the rest of its group of four multiplies zeros
        const f16x2 z = {(f16)0.0f, (f16)0.0f};
        for (uint32_t bb = (b & 3u) + 1; bb < 4; ++bb) { *(f16x2*)(Amh + rowo + bb * 8u + 2u * g) = z; *(f16x2*)(Aml + rowo + bb * 8u + 2u * g) = z; }
    }
}
template <int HF>
__device__ __forceinline__ void t3_dequant(const T3W& R, uint32_t b, uint32_t nb, f16* __restrict__ As, float* __restrict__ Dd, f16* __restrict__ Amh,
                                           f16* __restrict__ Aml, uint32_t tile, uint32_t r, uint32_t g, bool q5) {
    t3_dequant_piece<HF, 0>(R, As + t3_off(16u * tile + r, 0u + g), q5); t3_dequant_piece<HF, 1>(R, As + t3_off(16u * tile + r, 4u + g), q5);
    t3_dequant_piece<HF, 2>(R, As + t3_off(16u * tile + r, 8u + g), q5); t3_dequant_piece<HF, 3>(R, As + t3_off(16u * tile + r, 12u + g), q5);
    if (HF == 0) t3_dequant_meta(R, b, nb, Dd, Amh, Aml, tile, r, g);
}

// Builds of round 3 and what WRK_T3_DIAG=4 (cycle counters of wave 0, summed over a workgroup's halves) says about them -- K = 8192, per 128-k half:
//   first build (padded rows, unpacking + LDS stores behind the MFMAs)      32 x 128 prefill 110 k tok/s (with the quad WKV kernel)
//   swizzled images + unpacking pieces between the MFMA groups               127.7 k: issue + first fragments 865 | MFMAs + pieces 1 273 | min term / block scales 758 |
//                                                                            drain 156 | waiting at the barrier 1 330  = 4 380 cycles (the matrix pipe needs 1 024)
//   + global loads behind the first MFMA group, C = 0 on a block's first step  127.0 k: 331 | 1 860 | 765 | 305 | 1 257: the time moved, the half did not get shorter
//   + activations by LDS-DMA (no staging registers, no LDS stores for them)  124.9 k: 290 | 1 853 | 726 | 235 | 1 310 (225 registers instead of 253; not kept)
//   + waves 4 - 7 produce first and multiply afterwards, waves 0 - 3 the other way round (SIMD partners out of phase)   120.7 k vs 121.8 k same box: nothing (not kept)
// So a half costs ~4 400 cycles whatever is moved or removed: the two waves of a SIMD run the same phases at the same time (the barrier every
// half re-aligns them), so one wave's vector work does not fill the other's matrix-pipe gaps, and wave 0 spends 30 % of a half waiting for
// its SIMD partner.  What is left to try: block scales applied while the NEXT block multiplies (needs a second accumulator set: 257 registers
// with the DMA build), or 256-k stages with single-buffered images (half as many barriers).
// (A second build gave the waves ROLES -- waves 0-3 multiply 64 x 64 each, waves 4-7 only load / unpack / store the next half -- so that a
// SIMD's matrix pipe and its vector pipe would be busy at the same time.  It was SLOWER: 304 vs 215 us per launch.  What bounds this kernel is
// not the split of the issue slots but what a CU can take in: 41 KB of activations + weights per 128-k half per CU for 4.2 MFLOP, and the
// eight waves keep only ~64 KB of loads in flight against ~2 us of latency under load = ~18 GB/s per CU (the second-generation tile, with
// three workgroups per CU, takes in 28 GB/s per CU at half the arithmetic intensity).  See DESIGN.md 4.2.)
template <int DIAG>     // 0: product; diagnostics (WRK_T3_DIAG; 1-3 wrong results): 1 no activation loads, 2 no weight loads / unpacking, 3 no MFMA, 4 cycle breakdown (printf)
__global__ void __launch_bounds__(512) gemm_tile3_kernel(const T3Batch B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t3_smem[];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < GEMM_MAX_JOBS; ++q)
        if (q < B.g.njobs && blockIdx.x >= B.g.jobs[q].wg_begin) ji = q;
    const GemmParams& P = B.g.jobs[ji];
    const f16* __restrict__ SH = B.sh[ji];
    const f16* __restrict__ SL = B.sl[ji];
    f16* As = (f16*)t3_smem;
    f16* Xs = As + 2 * 128 * T3_LR;
    float* Dd = (float*)(Xs + 2 * 128 * T3_LR);
    f16* Amh = (f16*)(Dd + 256);
    f16* Aml = Amh + 128 * T3_MR;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t r = lane & 15, g = lane >> 4, wy = wave >> 2, wx = wave & 3u;
    const uint32_t m0 = (blockIdx.x - P.wg_begin) * T3_ROWS, n0 = blockIdx.y * T3_TOK;
    const uint32_t K = P.k, nb = K >> 8, nsub = K >> 5;
    const uint32_t b_begin = B.kslices > 1 ? blockIdx.z * B.bps : 0u, b_end = B.kslices > 1 ? min(nb, b_begin + B.bps) : nb;
    // dequant role: rows m0 + 16 wave + r
    const uint8_t* wrow = P.w + (size_t)min(m0 + 16u * wave + r, P.m - 1) * P.row_bytes;
    const bool q5 = P.kind == WRK_MAT_Q5_K;                 // uniform over the workgroup
    const uint32_t hoff = q5 ? nb * 160 : nb * 128, soff = hoff + nb * 4;
    auto load_w = [&](T3W& R, uint32_t b0) {
        const uint32_t b = min(b0, nb - 1);
        if (DIAG == 2) { for (int j = 0; j < 4; ++j) R.q[j] = (u32x2){b0, b0}; R.sm = (u32x4){1, 1, 1, 1}; R.dd = 1; R.qh = (u32x2){0, 0}; return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) R.q[j] = *(const u32x2*)(wrow + (size_t)b * 128 + j * 32 + 8 * g);
        R.qh = *(const u32x2*)(q5 ? wrow + (size_t)nb * 128 + (size_t)b * 32 + 8 * g : wrow);      // unconditional (Q4_K: a dummy inside the row)
        R.sm = *(const u32x4*)(wrow + soff + (size_t)b * 16);
        R.dd = *(const uint32_t*)(wrow + hoff + (size_t)b * 4);
    };
    // activation staging: 128 tokens x 16 chunks of 8 f16 per half block; chunk c of a thread -> token (tid >> 4) + 32 c, columns 8 (tid & 15)
    const f16* xsrc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xsrc[c] = P.x + (size_t)min(n0 + (tid >> 4) + 32u * c, P.n - 1) * P.xs + (tid & 15u) * 8;
    // the activations of a half are requested TWO halves before they are multiplied (one workgroup per CU: nothing else hides a global
    // load's ~2 us under load)
    f16x8 stage0[4], stage1[4];
    const uint32_t nhalf = 2 * nb;
    auto fetch_x = [&](f16x8 (&stage)[4], uint32_t h) {
        const uint32_t hc = min(h, nhalf - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) stage[c] = DIAG == 1 ? (f16x8){1, 1, 1, 1, 1, 1, 1, 1} : *(const f16x8*)(xsrc[c] + (size_t)hc * T3_KH);
    };
    // swizzled LDS offsets (f16 elements), per lane: chunk (4 ks + g) ^ (r & 15) = 4 (ks ^ (r >> 2)) + (g ^ (r & 3)), and every row a lane touches
    // in one image has the same r & 15 -- so an access is base + kof[ks] + an immediate
    const uint32_t kof[4] = {((0u ^ (r >> 2)) << 5), ((1u ^ (r >> 2)) << 5), ((2u ^ (r >> 2)) << 5), ((3u ^ (r >> 2)) << 5)};
    const uint32_t gl8 = (g ^ (r & 3u)) << 3;
    const uint32_t a_base = (64u * wy + r) * T3_LR + gl8, x_base = (32u * wx + r) * T3_LR + gl8, d_base = (16u * wave + r) * T3_LR + gl8;
    const uint32_t s_base = (tid >> 4) * T3_LR + (((tid & 15u) ^ ((tid >> 4) & 15u)) << 3);        // rows (tid >> 4) + 32 c: same row & 15
    auto store_x1 = [&](const f16x8 (&stage)[4], uint32_t buf, int c) {
        *(f16x8*)(Xs + (size_t)buf * 128 * T3_LR + s_base + (uint32_t)c * 32u * T3_LR) = stage[c];
    };
    auto store_x = [&](const f16x8 (&stage)[4], uint32_t buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) store_x1(stage, buf, c);
    };

    f32x4v total[4][2], acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { total[rt][tt] = (f32x4v){0.f, 0.f, 0.f, 0.f}; acc[rt][tt] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
    // input sums of this lane's tokens for the group of four blocks in flight (lane g: block 4 bq + g)
    f16x8 shf[2], slf[2];
    auto load_sums = [&](uint32_t bq) {
        const uint32_t blk = 4 * bq + g;
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const size_t o = (size_t)min(n0 + 32u * wx + 16u * tt + r, P.n - 1) * nsub + (size_t)min(blk, nb - 1) * 8;
            const f16x8 a = *(const f16x8*)(SH + o), c = *(const f16x8*)(SL + o);
            shf[tt] = blk < nb ? a : z;
            slf[tt] = blk < nb ? c : z;
        }
    };

    T3W W0, W1;
    load_w(W0, b_begin);
    fetch_x(stage0, 2 * b_begin);
    fetch_x(stage1, 2 * b_begin + 1);
    load_sums(b_begin >> 2);
    if (b_begin & 3u) {         // a slice that starts inside a group of four blocks: the blocks of the group in front of it multiply zeros
        const f16x2 z = {(f16)0.0f, (f16)0.0f};
        const uint32_t rowo = (16u * wave + r) * T3_MR;
        for (uint32_t bb = 0; bb < (b_begin & 3u); ++bb) { *(f16x2*)(Amh + rowo + bb * 8u + 2u * g) = z; *(f16x2*)(Aml + rowo + bb * 8u + 2u * g) = z; }
    }
    t3_dequant<0>(W0, b_begin, b_end, As, Dd, Amh, Aml, wave, r, g, q5);
    store_x(stage0, 0);
    __syncthreads();

    // one half block: compute half (b, hf) out of buffers hf, produce the next half into buffers hf ^ 1
    // (stage registers: half h + 1 sits in stage[hf ^ 1]; half h + 2 is requested into stage[hf], free since half h was stored)
    // WRK_T3_DIAG=4: where a half goes, in shader cycles summed over the halves of one workgroup's wave 0 (printed by workgroup (0, 0))
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto tick = [&](int k) {
        if (DIAG != 4) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (k >= 0) tacc[k] += now - tprev;
        tprev = now;
    };
    tick(-1);
    auto half = [&](uint32_t b, int hf, const T3W& Rc, T3W& Rn) {
        const f16* ab = As + (size_t)hf * 128 * T3_LR;
        const f16* xb = Xs + (size_t)hf * 128 * T3_LR;
        // the fragments of step ks + 1 are read while step ks multiplies (two register sets; the compiler barriers keep the reads of a step
        // together and one step ahead -- left alone the scheduler waits for each step's reads right before its MFMAs: 23 waits per half)
        f16x8 fa[2][4], fb[2][2];
        auto frags = [&](int set, int ks) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) fa[set][rt] = *(const f16x8*)(ab + a_base + kof[ks] + rt * 16 * T3_LR);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) fb[set][tt] = *(const f16x8*)(xb + x_base + kof[ks] + tt * 16 * T3_LR);
        };
        // one 32-k piece of the NEXT half (this wave's 16-row tile of weights, a quarter of its share of the activations) behind each MFMA group
        const bool more = hf == 0 || b + 1 < b_end;
        auto produce = [&](auto pc) {
            constexpr int PC = decltype(pc)::value;
            if (hf == 0) t3_dequant_piece<1, PC>(Rc, As + (size_t)128 * T3_LR + d_base + kof[PC], q5);
            else if (more) t3_dequant_piece<0, PC>(Rn, As + d_base + kof[PC], q5);
            if (hf == 0) store_x1(stage1, 1u, PC); else store_x1(stage0, 0u, PC);
        };
        frags(0, 0);
        tick(0);                // global loads and the first fragment reads issued (and, with the diagnostic's wait, the fragments arrived)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            asm volatile("" ::: "memory");
            if (ks < 3) frags((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if (DIAG == 3) acc[rt][tt][0] += (float)fa[ks & 1][rt][0] * (float)fb[ks & 1][tt][0];
                    else if (hf == 0 && ks == 0) acc[rt][tt] = mfma16(fa[0][rt], fb[0][tt], (f32x4v){0.f, 0.f, 0.f, 0.f});     // a block's first step: C = 0 (no zeroing pass)
                    else acc[rt][tt] = mfma16(fa[ks & 1][rt], fb[ks & 1][tt], acc[rt][tt]);
                }
            if (ks == 0) {
                // the global loads of the halves to come go out behind the first MFMA group (the first build issued them in front of the first
                // fragment reads: ~800 cycles per half before the matrix pipe had anything to do)
                if (hf == 0) load_w(Rn, b + 1);
                if (hf == 0) fetch_x(stage0, 2 * b + 2); else fetch_x(stage1, 2 * b + 3);
            }
            if (ks == 0) produce(std::integral_constant<int, 0>{});
            else if (ks == 1) produce(std::integral_constant<int, 1>{});
            else if (ks == 2) produce(std::integral_constant<int, 2>{});
            else produce(std::integral_constant<int, 3>{});
        }
        asm volatile("" ::: "memory");
        tick(1);                // MFMA groups + the pieces of the next half
        if (hf == 0 && ((b & 3u) == 3u || b + 1 == b_end)) {
            // the min term of this group of (up to) four blocks: A_min (LDS, complete since the last barrier) x the input sums
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const uint32_t ro = (64u * wy + 16u * rt + r) * T3_MR + 8u * g;
                const f16x8 ah = *(const f16x8*)(Amh + ro), al = *(const f16x8*)(Aml + ro);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    total[rt][tt] = mfma16(ah, shf[tt], total[rt][tt]);
                    total[rt][tt] = mfma16(ah, slf[tt], total[rt][tt]);
                    total[rt][tt] = mfma16(al, shf[tt], total[rt][tt]);
                }
            }
            load_sums((b >> 2) + 1);            // the next group's sums (clamped / zeroed beyond the end)
        }
        if (hf == 1) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const f32x4v dv = *(const f32x4v*)(Dd + (b & 1u) * 128u + 64u * wy + 16u * rt + 4u * g);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) total[rt][tt][i] = __builtin_fmaf(dv[i], acc[rt][tt][i], total[rt][tt][i]);
                    if (DIAG == 3) acc[rt][tt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        // the next block's row scales and min products (its weights were unpacked piece by piece above)
        if (hf == 1 && b + 1 < b_end) t3_dequant_meta(Rn, b + 1, b_end, Dd, Amh, Aml, wave, r, g);
        tick(2);                // min term, block scales, meta
        if (DIAG == 4) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tick(3); }     // everything this wave has in flight
        __syncthreads();
        tick(4);                // waiting for the other waves
    };
    for (uint32_t b = b_begin; b < b_end; b += 2) {          // uniform over the workgroup
        half(b, 0, W0, W1);
        half(b, 1, W0, W1);
        if (b + 1 >= b_end) break;
        half(b + 1, 0, W1, W0);
        half(b + 1, 1, W1, W0);
    }

    if (DIAG == 4 && blockIdx.x == 0 && blockIdx.y == 1 && tid == 0)
        printf("[t3 diag] K %u halves %u | issue+first frags %llu | mfma+pieces %llu | min/scales %llu | drain %llu | barrier %llu (cycles, wave 0)\n", K, nhalf,
               tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
    // store: lane owns rows m0 + 64 wy + 16 rt + 4g + (0..3) of token n0 + 32 wx + 16 tt + r
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const uint32_t tok = n0 + 32u * wx + 16u * tt + r, row = m0 + 64u * wy + 16u * rt + 4u * g;
            if (tok >= P.n || row >= P.m) continue;
            if (B.kslices > 1) {        // this slice's share, f32, added up by t3_reduce_kernel
                *(f32x4v*)(B.part + B.poff[ji] + ((size_t)blockIdx.z * P.n + tok) * P.m + row) = total[rt][tt];
                continue;
            }
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = act_apply(P.act, total[rt][tt][i] * P.scale);
            const size_t oo = (size_t)tok * P.os + row;
            if (P.has_res) {
                const size_t ro = (size_t)tok * P.rs + row;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? ((const float*)P.res_p)[ro + i] : (float)((const f16*)P.res_p)[ro + i]);
            }
            if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
            else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
        }
}

// The slices of a split launch, added in slice order (deterministic), then the epilogue of the product: scale, activation, residual, store.
// Thread = four rows of one token.
__global__ void __launch_bounds__(256) t3_reduce_kernel(const T3Batch B, uint32_t total_quads) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total_quads) return;
    int ji = 0;
    uint32_t base = 0;
#pragma unroll
    for (int q = 0; q + 1 < GEMM_MAX_JOBS; ++q) {
        const uint32_t cnt = q < B.g.njobs ? (B.g.jobs[q].m >> 2) * B.g.jobs[q].n : 0u;
        if (q + 1 < B.g.njobs && idx >= base + cnt && ji == q) { base += cnt; ji = q + 1; }
    }
    const GemmParams& P = B.g.jobs[ji];
    const uint32_t local = idx - base, mq = P.m >> 2, tok = local / mq, row = (local - tok * mq) << 2;
    const float* p = B.part + B.poff[ji] + (size_t)tok * P.m + row;
    const size_t zs = (size_t)P.n * P.m;
    f32x4v t = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (uint32_t z = 0; z < B.kslices; ++z) t += *(const f32x4v*)(p + z * zs);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = act_apply(P.act, t[i] * P.scale);
    const size_t oo = (size_t)tok * P.os + row;
    if (P.has_res) {
        const size_t ro = (size_t)tok * P.rs + row;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (P.out32 ? o[i] : r16(o[i])) + (P.res32 ? ((const float*)P.res_p)[ro + i] : (float)((const f16*)P.res_p)[ro + i]);
    }
    if (P.out32) *(f32x4v*)((float*)P.out_p + oo) = (f32x4v){o[0], o[1], o[2], o[3]};
    else { typedef _Float16 f16x4 __attribute__((ext_vector_type(4))); *(f16x4*)((f16*)P.out_p + oo) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]}; }
}

int gemm_tile3_launch(hipStream_t s, const GemmBatch& T3, uint32_t row_tiles, uint32_t n, void* xsum, size_t xsum_cap) {
    if (T3.njobs <= 0 || !xsum) return -1;
    T3Batch B;
    B.g = T3;
    // one pair of sum arrays per DISTINCT input (r, k, v of a layer read three different shifted inputs; a repeated one is summed once)
    size_t used = 0;
    const f16* seen_x[GEMM_MAX_JOBS];
    int nseen = 0;
    for (int q = 0; q < T3.njobs; ++q) {
        const GemmParams& P = T3.jobs[q];
        int hit = -1;
        for (int u = 0; u < q; ++u)
            if (T3.jobs[u].x == P.x && T3.jobs[u].xs == P.xs && T3.jobs[u].k == P.k) { hit = u; break; }
        if (hit >= 0) { B.sh[q] = B.sh[hit]; B.sl[q] = B.sl[hit]; continue; }
        const size_t cnt = (size_t)n * (P.k >> 5), bytes = (cnt * 2 + 255) & ~(size_t)255;
        if (used + 2 * bytes > xsum_cap) return -1;
        f16* sh = (f16*)((char*)xsum + used);
        f16* sl = (f16*)((char*)xsum + used + bytes);
        used += 2 * bytes;
        B.sh[q] = sh; B.sl[q] = sl;
        seen_x[nseen++] = P.x;
        xsum_kernel<<<dim3((uint32_t)((cnt + 255) / 256)), 256, 0, s>>>(P.x, P.xs, n, P.k >> 5, sh, sl);
    }
    (void)seen_x;
    // K split: one or two token tiles and fewer than ~160 workgroups -> slices of 4, 2 or 1 blocks (the first that fills the chip), all jobs of the
    // launch the same K.  Partial tiles behind the sum arrays in the launch's scratch.
    const uint32_t ttiles = (n + T3_TOK - 1) / T3_TOK;
    B.bps = 0; B.kslices = 1; B.part = nullptr;
    {
        bool same_k = true;
        for (int q = 1; q < T3.njobs; ++q) same_k = same_k && T3.jobs[q].k == T3.jobs[0].k;
        const uint32_t nb = T3.jobs[0].k >> 8;
        const char* se = getenv("WRK_T3_SPLIT");        // 0: off (A/B)
        if (same_k && ttiles <= 2 && row_tiles * ttiles < 160 && nb >= 2 && !(se && se[0] == '0')) {
            uint32_t bps = 1;
            for (uint32_t cand = 4; cand >= 1; cand >>= 1)
                if (nb % cand == 0 && nb / cand >= 2 && row_tiles * ttiles * (nb / cand) >= 160) { bps = cand; break; }
            const uint32_t ks = (nb + bps - 1) / bps;
            size_t floats = 0, at = (used + 255) & ~(size_t)255;
            for (int q = 0; q < T3.njobs; ++q) { B.poff[q] = floats; floats += (size_t)ks * n * T3.jobs[q].m; }
            if (ks >= 2 && at + floats * 4 <= xsum_cap) { B.bps = bps; B.kslices = ks; B.part = (float*)((char*)xsum + at); }
        }
    }
    const char* de = getenv("WRK_T3_DIAG");
    const int diag = de ? atoi(de) : 0;
    const dim3 grid(row_tiles, ttiles, B.kslices);
#define T3_GO(D) do { if (!lds_attr_once((const void*)gemm_tile3_kernel<D>, T3_LDS)) return -1; gemm_tile3_kernel<D><<<grid, 512, T3_LDS, s>>>(B); } while (0)
    if (diag == 1) T3_GO(1); else if (diag == 2) T3_GO(2); else if (diag == 3) T3_GO(3); else if (diag == 4) T3_GO(4); else T3_GO(0);
#undef T3_GO
    if (B.kslices > 1) {
        uint32_t quads = 0;
        for (int q = 0; q < T3.njobs; ++q) quads += (T3.jobs[q].m >> 2) * n;
        t3_reduce_kernel<<<dim3((quads + 255) / 256), 256, 0, s>>>(B, quads);
    }
    return 0;
}

}  // namespace wrk
