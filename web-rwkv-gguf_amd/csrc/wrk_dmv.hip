// Second-generation decode matvec kernels ("dmv"): the one-token instantiations and the host launcher; the kernel bodies live in
// wrk_dmv_body.h, the 2 .. 8-token instantiations in wrk_dmvt{2,4}.hip.  See DESIGN.md 4.1.
#include <cstdlib>

#include "wrk_dmv_body.h"

namespace wrk {

template <int KA, int KB, int XI, int KS>
static dmv_fn pick_dmv_pro(bool r16, int pro) {
#define DMV_P(PRO_) (r16 ? (dmv_fn)dmv_kernel<KA, KB, true, XI, KS, PRO_, 1> : (dmv_fn)dmv_kernel<KA, KB, false, XI, KS, PRO_, 1>)
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

dmvt_fn pick_dmv_nt2(int ka, int quant2, bool has_f16, bool f16_only, uint32_t xi, int ks, int pro);
dmvt_fn pick_dmv_nt4(int ka, int quant2, bool has_f16, bool f16_only, uint32_t xi, int ks, int pro);
dmvt_fn pick_dmv_tokens(int nt, int ka, int quant2, bool has_f16, bool f16_only, uint32_t xi, int ks, int pro) {
    if (nt == 2) return pick_dmv_nt2(ka, quant2, has_f16, f16_only, xi, ks, pro);
    if (nt == 4) return pick_dmv_nt4(ka, quant2, has_f16, f16_only, xi, ks, pro);
    return nullptr;
}

// Host side of the dmv kernels: 0 = launched (or would be, dry), -1 = not eligible (the caller falls back to the first-generation kernels)
int launch_dmv(hipStream_t s, const MatvecParams& P, uint32_t total_wg, int quant, bool has_f16, bool r16, bool dry, int quant2) {
    static const bool enabled = [] { const char* e = getenv("WRK_DMV"); return !(e && e[0] == '0'); }();
    if (!enabled) return -1;
    DParams D;
    DParamsT DT;
    uint32_t bounds[7], bmask = 0, xi = 1;
    int pro = -1;
    bool small_wg = true;
    for (int j = 0; j < 7; ++j) bounds[j] = 0xffffffffu;
    auto dense_base = [](const DTensor& t) { return ((size_t)t.offset[2] * t.stride[1] + t.offset[1]) * t.stride[0] + t.offset[0]; };
    // token tok = t + b * shape[1] of a [C, T, B] view sits at base + tok * stride[0] when the view covers the whole T extent of its parent
    auto dense_stack = [](const DTensor& t) { return t.shape[2] == 1 || (t.shape[1] == t.stride[1] && t.offset[1] == 0); };
    const uint32_t ntok = P.jobs[0].in.shape[1] * P.jobs[0].in.shape[2];
    // 2 / 4-token instantiations exist.  Measured (1.5B Q4_K_M, ms per step, this path | MFMA path): 2 sequences 0.759 | 1.214, 3: 1.113 | 1.231,
    // 4: 1.142 | 1.242; an 8-token instantiation lost (2.21 | 1.30: 300 registers, one workgroup per CU, and every workgroup pulls 12 KB
    // of prologue operands PER TOKEN through a CU that sustains ~30 GB/s): dropped.  WRK_DMV_TOKENS=1 restores the MFMA path.
    if (ntok == 0 || ntok > 4) return -1;
    const int nt = ntok == 1 ? 1 : (ntok <= 2 ? 2 : 4);
    static const uint32_t max_tok = [] { const char* e = getenv("WRK_DMV_TOKENS"); const int v = e ? atoi(e) : 4; return (uint32_t)(v < 1 ? 1 : v); }();
    if (ntok > max_tok) return -1;
    if (nt > 1 && r16) return -1;           // the per-element rounding mode stays on the one-token kernels
    for (int j = 0; j < P.njobs; ++j) {
        const JobDev& J = P.jobs[j];
        if (J.in.dtype != WRK_F16 || (J.k & 7u) || J.in.shape[1] * J.in.shape[2] != ntok || J.kind == WRK_MAT_NF4) return -1;
        if (nt > 1) {
            if (!dense_stack(J.in) || !dense_stack(J.out) || (J.has_res && !dense_stack(J.res)) || (J.in.stride[0] & 7u)) return -1;
            if (J.kind != WRK_MAT_F16 && J.kind != WRK_MAT_Q4_K && J.kind != WRK_MAT_Q5_K && J.kind != WRK_MAT_Q6_K) return -1;
        }
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
        const size_t esz = J.out.dtype == WRK_F32 ? 4 : 2;
        if (J.has_res && J.res.dtype != WRK_F16 && J.res.dtype != WRK_F32) return -1;
        auto fill = [&](auto& d) {          // the fields the one-token and the several-token job share
            memset(&d, 0, sizeof d);
            d.w = J.w;
            d.x = (const f16*)J.in.p + ib;
            d.out = (char*)J.out.p + dense_base(J.out) * esz;
            d.flags = J.out.dtype == WRK_F32 ? DJ_OUT32 : 0u;
            if (J.has_res) {
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
        };
        if (nt == 1) fill(D.jobs[j]);
        else {
            DJobT& dt = DT.jobs[j];
            fill(dt);
            dt.ntok = ntok; dt.xs = J.in.stride[0]; dt.os = J.out.stride[0]; dt.rs = J.has_res ? J.res.stride[0] : 0;
            dt.mix_s = J.tok_mix_stride; dt.prev_s = J.tok_prev_stride; dt.csrc_s = J.tok_carry_src_stride; dt.cdst_s = J.tok_carry_dst_stride; dt.gate_s = J.tok_gate_stride;
        }
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
    if (nt == 1 && launch_bytes > max_bytes && xi > 1 && pro != 3 && pro != 4) return -1;
    dmv_fn fn = nullptr;
    const int ka = quant < 0 ? WRK_MAT_F16 : quant;
    const bool mixf = has_f16 && quant >= 0;
    if (nt > 1) {
        // 2 .. 8 tokens: the instantiated subset (wrk_dmvt_inst.h); K over the 4 waves exists without prologue / arg-max only
        int ks = 1;
        uint32_t xk = xi;
        if (quant2 < 0 && !(quant < 0 && xi <= 4) && xi > 2) {
            bool amax = false;
            for (int j = 0; j < P.njobs; ++j) amax = amax || P.jobs[j].amax_val;
            if (mixf || !small_wg || pro != 0 || amax || xi > 8) return -1;
            ks = 4; xk = xi <= 4 ? 1 : 2;
        }
        const dmvt_fn ft = pick_dmv_tokens(nt, ka, quant2, mixf, quant < 0, xk, ks, pro);
        if (!ft) return -1;
        if (!dry) hipLaunchKernelGGL(ft, dim3(total_wg, 1), dim3(256), 0, s, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4], bounds[5], bounds[6], bmask, DT);
        return 0;
    }
    else if (quant2 >= 0) {      // (Q4_K | Q5_K) + Q6_K (+ F16): LN prologue or none; rows up to 4096 elements (round 3: the 2.9B model's K = 2560)
        const int k4 = quant == WRK_MAT_Q6_K ? quant2 : quant;
        const int lnpro = xi == 1 ? 1 : 2;
        if ((quant != WRK_MAT_Q6_K && quant2 != WRK_MAT_Q6_K) || (k4 != WRK_MAT_Q4_K && k4 != WRK_MAT_Q5_K) || xi > 2 || (pro != 0 && pro != lnpro)) return -1;
#define DMV3X(A, X, PR) (r16 ? (dmv_fn)dmv3_kernel<A, true, X, PR, 1> : (dmv_fn)dmv3_kernel<A, false, X, PR, 1>)
#define DMV3(A) (xi == 1 ? (pro ? DMV3X(A, 1, 1) : DMV3X(A, 1, 0)) : (pro ? DMV3X(A, 2, 2) : DMV3X(A, 2, 0)))
        fn = k4 == WRK_MAT_Q4_K ? DMV3(WRK_MAT_Q4_K) : DMV3(WRK_MAT_Q5_K);
#undef DMV3
#undef DMV3X
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
