// Second-generation decode matvec kernels ("dmv") and their host launcher; see the section comment below and DESIGN.md 4.1.
#include <cstdlib>

#include "wrk_matvec_dev.h"

namespace wrk {

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

// Three kinds in one launch: a K4 kind, Q6_K and F16 -- the r, k, v + LoRA stage of a real llama.cpp Q4_K_M / Q5_K_M file, whose attn
// value is Q6_K in about half of the layers (KS == 1; the job's own kind field selects the body)
template <int KA, bool R16, int XI, int PRO>
__global__ void __launch_bounds__(256) dmv3_kernel(uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7,
                                                   uint32_t, const DParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[576 + (PRO > 0 ? ((PRO - 1) % 2 + 1) * 4096 : 16)];
    const uint32_t b = blockIdx.x;
    const uint32_t ji = (b >= b1) + (b >= b2) + (b >= b3) + (b >= b4) + (b >= b5) + (b >= b6) + (b >= b7);
    const DJob J = P.jobs[ji];
    if (J.kind == (uint32_t)KA) dmv_body<KA, R16, XI, 1, PRO>(J, smem);
    else if (J.kind == WRK_MAT_Q6_K) dmv_body<WRK_MAT_Q6_K, R16, XI, 1, PRO>(J, smem);
    else dmv_body<WRK_MAT_F16, false, 4 * XI, 1, PRO>(J, smem);
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
int launch_dmv(hipStream_t s, const MatvecParams& P, uint32_t total_wg, int quant, bool has_f16, bool r16, bool dry, int quant2) {
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
    if (quant2 >= 0) {      // (Q4_K | Q5_K) + Q6_K (+ F16): short rows, LN prologue or none
        const int k4 = quant == WRK_MAT_Q6_K ? quant2 : quant;
        if ((quant != WRK_MAT_Q6_K && quant2 != WRK_MAT_Q6_K) || (k4 != WRK_MAT_Q4_K && k4 != WRK_MAT_Q5_K) || xi != 1 || pro > 1) return -1;
#define DMV3(A) (pro == 1 ? (r16 ? (dmv_fn)dmv3_kernel<A, true, 1, 1> : (dmv_fn)dmv3_kernel<A, false, 1, 1>) : (r16 ? (dmv_fn)dmv3_kernel<A, true, 1, 0> : (dmv_fn)dmv3_kernel<A, false, 1, 0>))
        fn = k4 == WRK_MAT_Q4_K ? DMV3(WRK_MAT_Q4_K) : DMV3(WRK_MAT_Q5_K);
#undef DMV3
    }
    else if (xi == 1) fn = pick_dmv_kind<1, 1>(ka, mixf, r16, pro);
    else if (xi == 2) fn = pick_dmv_kind<2, 1>(ka, mixf, r16, pro);
    else if (quant < 0 && xi <= 4) fn = pick_dmv_kind<4, 1>(ka, false, false, pro);          // F16 rows up to 2048 elements
    else if (!mixf && small_wg && pro == 0) fn = xi <= 4 ? pick_dmv_kind<1, 4>(ka, false, r16, 0) : pick_dmv_kind<2, 4>(ka, false, r16, 0);   // K over the 4 waves
    if (!fn) return -1;
    if (!dry) hipLaunchKernelGGL(fn, dim3(total_wg, 1), dim3(256), 0, s, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4], bounds[5], bounds[6], bmask, D);
    return 0;
}

}  // namespace wrk
