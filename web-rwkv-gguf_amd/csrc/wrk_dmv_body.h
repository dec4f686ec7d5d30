// Second-generation decode matvec kernels ("dmv"): device side, shared by wrk_dmv.hip (one input vector) and wrk_dmvt.hip
// (2 .. 4 input vectors: a few sequences decoding together).  See DESIGN.md 4.1.
#pragma once
#include "wrk_matvec_dev.h"

namespace wrk {

// ------------------------------------------------------------------ decode matvec, second generation ("dmv")
// Same arithmetic and work split as matvec_body_reg (inputs in registers, a wave owns rows, RB rows per round trip, KS == 4
// splits K over the waves), rebuilt around what the in-kernel timeline and the ISA of the first generation showed (round 2,
// DESIGN.md section 5):
//   * every global load of the kernel's start-up -- the LN / shift operands, RB x XI weight chunks per lane, the residual /
//     carry / gate operands of the epilogue -- is UNCONDITIONAL (row, chunk and token indices are clamped, invalid lanes
//     multiply zeros).  Exec-masked loads made the compiler lose count of the outstanding loads and wait `vmcnt(0)`, i.e. for
//     the WHOLE weight burst (~2.2 us), before the layer-norm statistics of the prologue could start, and put a full
//     round trip (`global_load_ushort; s_waitcnt vmcnt(0); v_cvt`) in front of everything for each epilogue operand;
//   * the job's parameters are one compact struct read with a single burst of scalar loads (the first generation re-read
//     pointer and stride per row inside branches: four dependent scalar round trips before the weight loads went out), and
//     the job lookup uses the leading scalar kernel arguments, which gfx950 preloads into SGPRs (amdgpu-kernarg-preload-count);
//   * the prologue is a template parameter (vectors per thread), so launches without one carry no prologue code.
// NT > 1 (round 2): NT input vectors (tokens of NT sequences) per launch.  A weight chunk is requested and decoded ONCE and
// multiplied with every token's register-resident inputs; the prologues run per token; lane rb + 4 t finishes row rb of token t.
// The matrix-core GEMM needs 16-token tiles and pays 7 launches per layer; up to 4 tokens this kernel keeps the 5-launch layer.
enum { DJ_RES = 1, DJ_RES32 = 2, DJ_CARRY = 4, DJ_GATE = 8, DJ_AMAX = 16, DJ_OUT32 = 32, DJ_PUBLISH = 64 };

// The job of a workgroup.  Two PLAIN structs with the same leading fields (no base class: with `struct DJobT : DJob` clang no longer
// treated the pointer members of the by-value kernel argument as global pointers and emitted flat_load for every operand -- the
// 2-token decode step went from 0.757 to 0.839 ms, same ISA size, same registers; found by diffing the assembly).
// (Field order: measured.  Grouping "what every launch reads" into the leading 96 bytes made the batch-1 step 0.9 % SLOWER, 0.6213 vs
// 0.6158 ms same-box; this order stays.)
#define WRK_DJOB_FIELDS                                                                                                          \
    const uint8_t* w;                                                                                                            \
    const f16* x;          /* dense f16 inputs: token t at x + t * xs */                                                          \
    void* out;             /* dense outputs: element `row` of token t at t * os + row */                                          \
    const void* res;       /* DJ_RES: residual, element t * rs + row (f16; f32 with DJ_RES32) */                                  \
    const f16* carry_src;  /* DJ_CARRY: carry_dst[t * cdst_s + row] = carry_src[t * csrc_s + row] */                              \
    float* carry_dst;                                                                                                            \
    const f16* gate;       /* DJ_GATE, element t * gate_s + row */                                                                \
    const f16 *ln_w, *ln_b, *mixw; /* prologue: x_in = mix(LN(x), prev, mixw); mixw of token t at t * mix_s (0: shared) */        \
    const float* prev;     /* token t at t * prev_s */                                                                            \
    f16* ln_out;           /* DJ_PUBLISH: the job's first workgroup stores LN(x) of token t at t * K */                           \
    float* amax_val;       /* DJ_AMAX: [workgroup][token] */                                                                      \
    uint32_t* amax_idx;                                                                                                          \
    unsigned long long* dbg;                                                                                                     \
    uint32_t k, m, row_bytes, rows_per_wg, wg_begin, act, flags, kind;                                                           \
    float scale, eps;
struct DJob {                   // one input vector (the batch-1 decode kernels): 160 bytes, three scalar-cache lines
    WRK_DJOB_FIELDS
};
// several input vectors: the per-token strides ride behind the common part.  (They were members of DJob at first: the 196-byte struct
// cost the ONE-token kernels 0.6 % of the decode step -- same-box A/B 0.6244 vs 0.6204 ms -- one more scalar-cache line per job read
// at the head of every launch; so the one-token instantiations keep the 160-byte job.)
struct DJobT {
    WRK_DJOB_FIELDS
    uint32_t ntok, xs, os, rs, mix_s, prev_s, csrc_s, cdst_s, gate_s;
};

struct DParams {
    DJob jobs[MAX_JOBS];
};
struct DParamsT {
    DJobT jobs[MAX_JOBS];
};
template <int NT> struct DParamsOf { typedef DParamsT type; };
template <> struct DParamsOf<1> { typedef DParams type; };

// per-token fields of a job: constants for the one-token job
template <class JT> struct TokF {
    static __device__ __forceinline__ uint32_t ntok(const JT&) { return 1u; }
    static __device__ __forceinline__ uint32_t xs(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t os(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t rs(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t mix_s(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t prev_s(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t csrc_s(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t cdst_s(const JT&) { return 0u; }
    static __device__ __forceinline__ uint32_t gate_s(const JT&) { return 0u; }
};
template <> struct TokF<DJobT> {
    static __device__ __forceinline__ uint32_t ntok(const DJobT& j) { return j.ntok; }
    static __device__ __forceinline__ uint32_t xs(const DJobT& j) { return j.xs; }
    static __device__ __forceinline__ uint32_t os(const DJobT& j) { return j.os; }
    static __device__ __forceinline__ uint32_t rs(const DJobT& j) { return j.rs; }
    static __device__ __forceinline__ uint32_t mix_s(const DJobT& j) { return j.mix_s; }
    static __device__ __forceinline__ uint32_t prev_s(const DJobT& j) { return j.prev_s; }
    static __device__ __forceinline__ uint32_t csrc_s(const DJobT& j) { return j.csrc_s; }
    static __device__ __forceinline__ uint32_t cdst_s(const DJobT& j) { return j.cdst_s; }
    static __device__ __forceinline__ uint32_t gate_s(const DJobT& j) { return j.gate_s; }
};

// decode a chunk once, multiply it with the inputs of every token (arithmetic per token as dot_raw_reg)
template <int KIND, bool R16, int NT, int XI>
__device__ __forceinline__ void dot_raw_tokens(const Raw& r, uint32_t c, const XRegs (&x)[NT][XI], int ci, float (&acc)[NT]) {
    if (KIND == WRK_MAT_F16) {
        const f16x8 wv = __builtin_bit_cast(f16x8, r.w);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f16x8 xv = x[t][ci].v[0];
            float a = 0.0f;
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 0, 1), __builtin_shufflevector(xv, xv, 0, 1), a, false);
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 2, 3), __builtin_shufflevector(xv, xv, 2, 3), a, false);
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 4, 5), __builtin_shufflevector(xv, xv, 4, 5), a, false);
            a = __builtin_amdgcn_fdot2(__builtin_shufflevector(wv, wv, 6, 7), __builtin_shufflevector(xv, xv, 6, 7), a, false);
            acc[t] += a;
        }
        return;
    }
    constexpr bool TWO = KIND != WRK_MAT_Q8_0 && KIND != WRK_MAT_INT8;
    Group lo, hi;
    decode_raw<KIND>(r, c, lo, hi);
    if (R16) { round_group(lo); if (TWO) round_group(hi); }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const XRegs& X = x[t][ci];
        float a;
        if (R16) a = dot16r(lo.q, X.v[0], X.v[1]);
        else a = lo.scale * (lo.qmul * dot16r(lo.q, X.v[0], X.v[1]) - lo.off * X.s[0]) - lo.minv * X.s[0];
        if (TWO) {
            if (R16) a += dot16r(hi.q, X.v[2], X.v[3]);
            else a += hi.scale * (hi.qmul * dot16r(hi.q, X.v[2], X.v[3]) - hi.off * X.s[1]) - hi.minv * X.s[1];
        }
        acc[t] += a;
    }
}

// LDS: [0, 512 NT) K-split partials [NT][32 rows][4 waves] | [512 NT, 544 NT) arg-max [NT][4] values, [NT][4] indices |
//      [544 NT, 576 NT) LN statistics [NT][8] | [576 NT, ...) the prologue's inputs [NT][kpad] f16
template <int KIND, bool R16, int XI, int KS, int PRO, int NT, class JT>
__device__ __forceinline__ void dmv_body(const JT J, unsigned char* smem) {
    typedef TokF<JT> TF;
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
    const uint32_t ntok = NT == 1 ? 1u : TF::ntok(J);                          // 1 <= ntok <= NT; tokens beyond it are clamped and never stored
    auto row_of = [&](uint32_t ri) { return KS == 1 ? r0 + wave + 4 * ri : r0 + ri; };
    auto tok_of = [&](uint32_t t) { return NT == 1 ? 0u : min(t, ntok - 1); };
    const uint8_t* __restrict__ W = J.w;
    const uint32_t RBY = J.row_bytes;
    float* part = (float*)smem;
    float* sv = (float*)(smem + 512 * NT);
    uint32_t* si = (uint32_t*)(smem + 528 * NT);
    float* red = (float*)(smem + 544 * NT);
    f16* xs = (f16*)(smem + 576 * NT);

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
    // epilogue operands of the (row, token) this thread will finish: KS == 1: lane rb + 4 t finishes the wave's rb-th row of a
    // batch for token t; KS == 4: thread ri + 32 t finishes row r0 + ri of token t.  Raw bits now, conversion at use.
    const uint32_t fin_row = min(KS == 1 ? row_of(lane & 3u) : r0 + (tid & 31u), r1 - 1);
    const uint32_t fin_t = NT == 1 ? 0u : (KS == 1 ? (lane >> 2) : (tid >> 5));     // (one token: everything per-token below folds away)
    const uint32_t fin_tok = tok_of(fin_t);
    const uint32_t fl = J.flags;
    uint32_t res_bits = 0, carry_bits = 0, gate_bits = 0;
    // PRO: 0 none | 1, 2: layer norm + token shift, 1 / 2 vectors per thread (K <= 2048 / 4096) | 3, 4: the post-WKV stage of a
    // split head (group norm over 64-channel heads + time_first bonus + gate), 1 / 2 vectors per thread
    constexpr int VPT = PRO == 0 ? 1 : ((PRO - 1) % 2 + 1);
    constexpr bool GN = PRO >= 3;
    f16x8 xv[NT][VPT], wv[VPT], bv[VPT], mv[GN ? NT : 1][VPT];
    f32x4 pv[NT][VPT][2];
    f16 c0h[NT];
    XRegs x[NT][XI];
    if (PRO > 0) {
        const uint32_t nvec = K >> 3;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const uint32_t i = min(tid + 256u * v, nvec - 1);
            wv[v] = *(const f16x8*)(J.ln_w + i * 8);
            bv[v] = *(const f16x8*)(J.ln_b + i * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t tt = tok_of(t);
                xv[t][v] = *(const f16x8*)(xin + (size_t)tt * TF::xs(J) + i * 8);
                if (GN || t == 0) mv[GN ? t : 0][v] = *(const f16x8*)(J.mixw + (size_t)tt * TF::mix_s(J) + i * 8);
                pv[t][v][0] = *(const f32x4*)(J.prev + (size_t)tt * TF::prev_s(J) + i * 8);
                pv[t][v][1] = *(const f32x4*)(J.prev + (size_t)tt * TF::prev_s(J) + i * 8 + 4);
            }
        }
        if (!GN) {
#pragma unroll
            for (int t = 0; t < NT; ++t) c0h[t] = xin[(size_t)tok_of(t) * TF::xs(J)];
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) x[t][ci] = load_x<KIND>(xin + (size_t)tok_of(t) * TF::xs(J), min(cbase + CSTEP * ci, nch - 1), true);
    }
    issue(0);
    {
        const bool has_res = (fl & DJ_RES) != 0, has_carry = (fl & DJ_CARRY) != 0, has_gate = (fl & DJ_GATE) != 0;
        // absent operands read element 0 of the input vector: always mapped, never used
        const uint16_t* rp = has_res ? (const uint16_t*)J.res : (const uint16_t*)xin;
        const uint32_t re = fin_tok * TF::rs(J) + fin_row;
        const uint32_t ri = has_res ? ((fl & DJ_RES32) ? 2u * re : re) : 0u;
        if (fl & DJ_RES32) res_bits = *(const uint32_t*)(rp + ri);        // uniform branch, one load on either side
        else res_bits = rp[ri];
        carry_bits = (has_carry ? (const uint16_t*)J.carry_src : (const uint16_t*)xin)[has_carry ? fin_tok * TF::csrc_s(J) + fin_row : 0u];
        gate_bits = (has_gate ? (const uint16_t*)J.gate : (const uint16_t*)xin)[has_gate ? fin_tok * TF::gate_s(J) + fin_row : 0u];
    }

    // ---- (2a) split-head prologue (K3): x_in = g * r16(r16(GN(y)) + tt)  with y = WKV output (f16), tt = (sum_j r_k k r) * v (f32),
    //      g = gate (f16); a head is 64 channels = 8 threads of 8 channels, so the statistics are three DPP steps -- no barrier
    if (GN) {
        const uint32_t nvec = K >> 3;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                float y[8], s1 = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { y[e] = (float)xv[t][v][e]; s1 += y[e]; }
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
                    float u = r16(__builtin_fmaf(y[e] * dev, (float)wv[v][e], (float)bv[v][e]));       // group_norm
                    u = r16(u + pv[t][v][e >> 2][e & 3]);                                               // time_first_v7
                    o[e] = (f16)((float)mv[GN ? t : 0][v][e] * u);                                      // mul(g, x)
                }
                const uint32_t i = tid + 256u * v;
                if (i < nvec) *(f16x8*)(xs + (size_t)t * kpad + i * 8) = o;
            }
        for (uint32_t i = K + tid; i < kpad; i += 256)
#pragma unroll
            for (int t = 0; t < NT; ++t) xs[(size_t)t * kpad + i] = (f16)0.0f;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) x[t][ci] = load_x<KIND>(xs + (size_t)t * kpad, min(cbase + CSTEP * ci, nch - 1), true);
    }
    // ---- (2) prologue: layer norm + token shift of the input, once per workgroup, handed to the waves through LDS
    if (PRO > 0 && !GN) {
        const uint32_t nvec = K >> 3;
        // one pass, one block reduction: sums of (x - c) and (x - c)^2 around c = x[0]
        float s1[NT], s2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float c0 = (float)c0h[t];
            s1[t] = 0.0f; s2[t] = 0.0f;
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (tid + 256u * v < nvec)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dl = (float)xv[t][v][e] - c0; s1[t] += dl; s2[t] = __builtin_fmaf(dl, dl, s2[t]); }
            s1[t] = wave_sum(s1[t]);
            s2[t] = wave_sum(s2[t]);
            if (lane == 0) { red[8 * t + wave] = s1[t]; red[8 * t + 4 + wave] = s2[t]; }
        }
        __syncthreads();
        WRK_STAMP(J.dbg, 4);
        const bool publish = (fl & DJ_PUBLISH) && blockIdx.x == J.wg_begin;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float c0 = (float)c0h[t];
            const float a1 = (red[8 * t + 0] + red[8 * t + 1]) + (red[8 * t + 2] + red[8 * t + 3]);
            const float a2 = (red[8 * t + 4] + red[8 * t + 5]) + (red[8 * t + 6] + red[8 * t + 7]);
            const float md = a1 / (float)K;
            const float mean = c0 + md;
            const float dev = 1.0f / sqrtf(fmaxf(a2 / (float)K - md * md, 0.0f) + J.eps);
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const uint32_t i = tid + 256u * v;
                if (i >= nvec) continue;
                f16x8 yv, o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    yv[e] = (f16)__builtin_fmaf(((float)xv[t][v][e] - mean) * dev, (float)wv[v][e], (float)bv[v][e]);
                    o[e] = (f16)wgsl_mix((float)yv[e], pv[t][v][e >> 2][e & 3], (float)mv[0][v][e]);
                }
                *(f16x8*)(xs + (size_t)t * kpad + i * 8) = o;
                if (publish && (uint32_t)t < ntok) *(f16x8*)(J.ln_out + (size_t)t * K + i * 8) = yv;
            }
        }
        for (uint32_t i = K + tid; i < kpad; i += 256)
#pragma unroll
            for (int t = 0; t < NT; ++t) xs[(size_t)t * kpad + i] = (f16)0.0f;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) x[t][ci] = load_x<KIND>(xs + (size_t)t * kpad, min(cbase + CSTEP * ci, nch - 1), true);
    }
    // chunks beyond the row multiply zeros
#pragma unroll
    for (int ci = 0; ci < XI; ++ci)
        if (cbase + CSTEP * ci >= nch) {
            const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < NT; ++t) x[t][ci].v[0] = x[t][ci].v[1] = x[t][ci].v[2] = x[t][ci].v[3] = z;
        }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ci = 0; ci < XI; ++ci) x_sums<KIND>(x[t][ci]);
    WRK_STAMP(J.dbg, 1);

    // ---- (3) dot products, reduction, epilogue
    float best_v = -3.0e38f;
    uint32_t best_i = 0xffffffffu;
    auto finish = [&](uint32_t r, uint32_t tok, float v, uint32_t rbits, uint32_t cbits, uint32_t gbits) {
        float o = act_apply(J.act, v * J.scale);
        const bool o32 = (fl & DJ_OUT32) != 0;
        if (fl & DJ_GATE) o = act_sigmoid(f16bits_to_f32(gbits)) * (o32 ? o : r16(o));
        if (fl & DJ_RES) o = (o32 ? o : r16(o)) + ((fl & DJ_RES32) ? __builtin_bit_cast(float, rbits) : f16bits_to_f32(rbits));
        const size_t oo = (size_t)tok * TF::os(J) + r;
        if (o32) ((float*)J.out)[oo] = o; else ((f16*)J.out)[oo] = (f16)o;
        if (fl & DJ_CARRY) J.carry_dst[(size_t)tok * TF::cdst_s(J) + r] = f16bits_to_f32(cbits);
        if (o > best_v || (o == best_v && r < best_i)) { best_v = o; best_i = r; }
    };
    auto operand_bits = [&](uint32_t r, uint32_t tok, uint32_t& rbits, uint32_t& cbits, uint32_t& gbits) {      // rows beyond the first batch
        if (fl & DJ_RES) rbits = (fl & DJ_RES32) ? ((const uint32_t*)J.res)[(size_t)tok * TF::rs(J) + r] : (uint32_t)((const uint16_t*)J.res)[(size_t)tok * TF::rs(J) + r];
        if (fl & DJ_CARRY) cbits = ((const uint16_t*)J.carry_src)[(size_t)tok * TF::csrc_s(J) + r];
        if (fl & DJ_GATE) gbits = ((const uint16_t*)J.gate)[(size_t)tok * TF::gate_s(J) + r];
    };
    for (uint32_t ri0 = 0; ri0 < nrows; ri0 += RB) {
        if (ri0 != 0) issue(ri0);
        float acc[RB][NT];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[rb][t] = 0.0f;
#pragma unroll
            for (int ci = 0; ci < XI; ++ci) dot_raw_tokens<KIND, R16, NT, XI>(raw[rb][ci], min(cbase + CSTEP * ci, nch - 1), x, ci, acc[rb]);
        }
        WRK_STAMP(J.dbg, 2);
        float mine_v = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float v = wave_sum(acc[rb][t]);
                if (lane == (uint32_t)(rb + RB * t)) mine_v = v;
            }
        const uint32_t frb = lane & 3u;
        if (KS == 1) {
            if (lane < (uint32_t)(RB * NT) && (NT == 1 || fin_t < ntok) && ri0 + frb < nrows) {
                const uint32_t r = row_of(ri0 + frb);
                uint32_t rbits = res_bits, cbits = carry_bits, gbits = gate_bits;
                if (ri0 != 0) operand_bits(r, fin_t, rbits, cbits, gbits);
                finish(r, fin_t, mine_v, rbits, cbits, gbits);
            }
        } else if (lane < (uint32_t)(RB * NT) && ri0 + frb < nrows) part[(((NT == 1 ? 0u : lane >> 2) * 32u) + ri0 + frb) * 4 + wave] = mine_v;
    }
    if (KS == 4) {
        __syncthreads();
        const uint32_t ri = tid & 31u;
        if (NT == 1 ? tid < nrows : (ri < nrows && fin_t < ntok)) {
            const uint32_t p = NT == 1 ? tid : (fin_t * 32u + ri);
            finish(r0 + ri, fin_t, (part[p * 4] + part[p * 4 + 1]) + (part[p * 4 + 2] + part[p * 4 + 3]), res_bits, carry_bits, gate_bits);
        }
    }
    WRK_STAMP(J.dbg, 3);
    if (fl & DJ_AMAX) {     // fused greedy sampling, stage 1 (uniform branch).  KS == 1: lanes 4 t .. 4 t + 3 hold token t's candidates;
                            // KS == 4 (NT == 1 only, host-checked): every thread holds token 0's
#pragma unroll
        for (int o = (KS == 1 ? 2 : 32); o > 0; o >>= 1) {
            const float ov = __shfl_xor(best_v, o, WAVE);
            const uint32_t oi = __shfl_xor(best_i, o, WAVE);
            if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
        }
        if (KS == 1 ? ((lane & 3u) == 0 && lane < 4u * NT) : lane == 0) { sv[4 * (KS == 1 ? lane >> 2 : 0) + wave] = best_v; si[4 * (KS == 1 ? lane >> 2 : 0) + wave] = best_i; }
        __syncthreads();
        if (tid < ntok) {
            float bv2 = sv[4 * tid];
            uint32_t bi = si[4 * tid];
            for (int w = 1; w < 4; ++w)
                if (sv[4 * tid + w] > bv2 || (sv[4 * tid + w] == bv2 && si[4 * tid + w] < bi)) { bv2 = sv[4 * tid + w]; bi = si[4 * tid + w]; }
            J.amax_val[(size_t)(blockIdx.x - J.wg_begin) * ntok + tid] = bv2;
            J.amax_idx[(size_t)(blockIdx.x - J.wg_begin) * ntok + tid] = bi;
        }
    }
}

template <int PRO, int NT>
constexpr unsigned dmv_smem_bytes() { return 576u * NT + (PRO > 0 ? (unsigned)((PRO - 1) % 2 + 1) * 4096u * NT : 16u); }

// b1 .. b7: first workgroup of jobs 1 .. 7 (0xffffffff beyond the last job): LEADING SCALAR arguments, preloaded into SGPRs at wave
// launch, so the job lookup costs no memory access and the job's parameters are the kernel's first (and only) scalar round trip
template <int KA, int KB, bool R16, int XI, int KS, int PRO, int NT>
__global__ void __launch_bounds__(256) dmv_kernel(uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7,
                                                  uint32_t kind_b_mask, const typename DParamsOf<NT>::type P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[dmv_smem_bytes<PRO, NT>()];
    const uint32_t b = blockIdx.x;
    const uint32_t ji = (b >= b1) + (b >= b2) + (b >= b3) + (b >= b4) + (b >= b5) + (b >= b6) + (b >= b7);
    const auto J = P.jobs[ji];
    if (KA == KB || !((kind_b_mask >> ji) & 1u)) dmv_body<KA, (KA != WRK_MAT_F16) && R16, XI, KS, PRO, NT>(J, smem);
    else dmv_body<KB, (KB != WRK_MAT_F16) && R16, (KB == WRK_MAT_F16 ? 4 * XI : XI), 1, PRO, NT>(J, smem);
}

// Three kinds in one launch: a K4 kind, Q6_K and F16 -- the r, k, v + LoRA stage of a real llama.cpp Q4_K_M / Q5_K_M file, whose attn
// value is Q6_K in about half of the layers (KS == 1; the job's own kind field selects the body)
template <int KA, bool R16, int XI, int PRO, int NT>
__global__ void __launch_bounds__(256) dmv3_kernel(uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7,
                                                   uint32_t, const typename DParamsOf<NT>::type P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[dmv_smem_bytes<PRO, NT>()];
    const uint32_t b = blockIdx.x;
    const uint32_t ji = (b >= b1) + (b >= b2) + (b >= b3) + (b >= b4) + (b >= b5) + (b >= b6) + (b >= b7);
    const auto J = P.jobs[ji];
    if (J.kind == (uint32_t)KA) dmv_body<KA, R16, XI, 1, PRO, NT>(J, smem);
    else if (J.kind == WRK_MAT_Q6_K) dmv_body<WRK_MAT_Q6_K, R16, XI, 1, PRO, NT>(J, smem);
    else dmv_body<WRK_MAT_F16, false, 4 * XI, 1, PRO, NT>(J, smem);
}

typedef void (*dmv_fn)(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const DParams);
typedef void (*dmvt_fn)(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const DParamsT);

// wrk_dmvt.hip: the kernel for NT = 2 or 4 tokens (nullptr: combination not instantiated)
dmvt_fn pick_dmv_tokens(int nt, int ka, int quant2, bool has_f16, bool f16_only, uint32_t xi, int ks, int pro);

}  // namespace wrk
