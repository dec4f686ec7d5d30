// Instantiations of the dmv kernels for DMVT_NT tokens (included by wrk_dmvt2.hip / wrk_dmvt4.hip: one translation
// unit per token count so they compile in parallel).  The subset: Q4_K / Q5_K / Q6_K (+ F16 LoRA rows in the same launch), F16 alone,
// the three-kind launch of real Q4_K_M / Q5_K_M tensor mixes (rows up to 4096 elements); no per-element rounding mode; XI = 1 with the K <= 2048 prologues,
// XI = 2 with the K <= 4096 ones; K over the waves without prologue.
#include "wrk_dmv_body.h"

namespace wrk {

#define NTK DMVT_NT

template <int KA, int KB, int XI>
static dmvt_fn pick_pro(int pro) {
    constexpr int P1 = XI == 1 ? 1 : 2, P3 = XI == 1 ? 3 : 4;
    if (pro == 0) return (dmvt_fn)dmv_kernel<KA, KB, false, XI, 1, 0, NTK>;
    if (pro == P1) return (dmvt_fn)dmv_kernel<KA, KB, false, XI, 1, P1, NTK>;
    if (pro == P3 && KA == KB) return (dmvt_fn)dmv_kernel<KA, KA, false, XI, 1, P3, NTK>;
    return nullptr;
}

template <int KA>
static dmvt_fn pick_kind(bool has_f16, uint32_t xi, int ks, int pro) {
    if (ks == 4) {
        if (has_f16 || pro != 0) return nullptr;
        if (xi == 1) return (dmvt_fn)dmv_kernel<KA, KA, false, 1, 4, 0, NTK>;
        if (xi == 2 && NTK < 8) return (dmvt_fn)dmv_kernel<KA, KA, false, (NTK < 8 ? 2 : 1), 4, 0, NTK>;
        return nullptr;
    }
    if (xi == 1) return has_f16 ? pick_pro<KA, WRK_MAT_F16, 1>(pro) : pick_pro<KA, KA, 1>(pro);
    if (xi == 2 && NTK < 8) return has_f16 ? pick_pro<KA, WRK_MAT_F16, (NTK < 8 ? 2 : 1)>(pro) : pick_pro<KA, KA, (NTK < 8 ? 2 : 1)>(pro);
    return nullptr;
}

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
dmvt_fn CAT(pick_dmv_nt, DMVT_NT)(int ka, int quant2, bool has_f16, bool f16_only, uint32_t xi, int ks, int pro) {
    if (quant2 >= 0) {
        const int k4 = ka == WRK_MAT_Q6_K ? quant2 : ka;
        const int lnpro = xi == 1 ? 1 : 2;
        if ((ka != WRK_MAT_Q6_K && quant2 != WRK_MAT_Q6_K) || xi > 2 || (xi == 2 && NTK >= 4) || ks != 1 || (pro != 0 && pro != lnpro)) return nullptr;
        // rows of 2049 .. 4096 elements (round 3: the 2.9B model's real tensor mix): two sequences only -- measured, 2.9B Q4_K_M mix, ms per
        // step, this launch | the MFMA layer: 2 sequences 2.01 | 2.20, 4 sequences 3.64 | 2.25 (four tokens x two chunk iterations x three decoders)
        constexpr int X2 = NTK < 4 ? 2 : 1;
#define DMV3T(A) (xi == 1 ? (pro ? (dmvt_fn)dmv3_kernel<A, false, 1, 1, NTK> : (dmvt_fn)dmv3_kernel<A, false, 1, 0, NTK>) \
                          : (pro ? (dmvt_fn)dmv3_kernel<A, false, X2, (X2 == 2 ? 2 : 1), NTK> : (dmvt_fn)dmv3_kernel<A, false, X2, 0, NTK>))
        if (k4 == WRK_MAT_Q4_K) return DMV3T(WRK_MAT_Q4_K);
        if (k4 == WRK_MAT_Q5_K) return DMV3T(WRK_MAT_Q5_K);
#undef DMV3T
        return nullptr;
    }
    if (f16_only) {     // F16 rows alone (a head kept in f16, LoRA rows): no prologue
        if (ks != 1 || pro != 0) return nullptr;
        if (xi == 1) return (dmvt_fn)dmv_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 1, 1, 0, NTK>;
        if (xi == 2) return (dmvt_fn)dmv_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 2, 1, 0, NTK>;
        if (xi <= 4) return (dmvt_fn)dmv_kernel<WRK_MAT_F16, WRK_MAT_F16, false, 4, 1, 0, NTK>;
        return nullptr;
    }
    switch (ka) {
        case WRK_MAT_Q4_K: return pick_kind<WRK_MAT_Q4_K>(has_f16, xi, ks, pro);
        case WRK_MAT_Q5_K: return pick_kind<WRK_MAT_Q5_K>(has_f16, xi, ks, pro);
        case WRK_MAT_Q6_K: return pick_kind<WRK_MAT_Q6_K>(has_f16, xi, ks, pro);
        default: return nullptr;
    }
}

}  // namespace wrk
