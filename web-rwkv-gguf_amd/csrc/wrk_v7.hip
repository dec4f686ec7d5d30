// RWKV-7 model runner behind wrk_v7_* (include/wrk_hip.h).
//
// mode 0 ("op-by-op") enqueues exactly the TensorOp list that v7::Bundle::dispatch builds
// (src/runtime/v7.rs:598-713, dispatch_layer :716-1007, dispatch_header :1009-1036), one kernel per
// reference op, on scratch buffers laid out like `Runtime<f16>` (v7.rs:281-364).
// mode 1 ("fused") runs the decode fast path of wrk_v7_fused.hip when every sequence of the chunk
// contributes exactly one token, and falls back to mode 0 otherwise (prefill chunks).
#include "wrk_internal.h"
#include "wrk_v7.h"
#include "wrk_v7_engine.h"

#define LOCK(ctx) std::lock_guard<std::recursive_mutex> _lk((ctx)->mu)

static constexpr float LN_EPS = 1.0e-5f;    // v7.rs:47
static constexpr float GN_EPS = 64.0e-5f;   // v7.rs:48
static constexpr float L2_EPS = 1.0e-12f;   // v7.rs:46

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
static wrk::MatJob mj(const wrk_matrix* m, DTensor in, DTensor out, uint32_t act);

// ------------------------------------------------------------------ scratch ("Runtime<f16>" + "Header<f16>")
int32_t wrk_v7_model::ensure_scratch(uint32_t T, uint32_t NH) {
    if (T <= scratch_tokens && NH <= scratch_headers && scratch) return WRK_OK;
    if (ctx->capturing_here()) return wrk_fail(ctx, WRK_E_ARG, "scratch must be sized before capture");
    const uint32_t nt = T > scratch_tokens ? T : scratch_tokens;
    const uint32_t nh = NH > scratch_headers ? NH : scratch_headers;
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_graphs();                                   // captured graphs hold the old pointers
    if (scratch) hipFree(scratch);
    scratch = nullptr;
    const size_t D = d.num_emb, F = d.num_hidden, V = d.num_vocab;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up256(bytes); return o; };
    const size_t esz = act_dtype == WRK_F32 ? 4 : 2;        // Runtime<F>: every buffer but `input` holds F
    const size_t vecT = D * nt * esz;
    size_t o_named[32];
    int ni = 0;
    for (int i = 0; i < 23; ++i) o_named[ni++] = take(vecT);          // input, x, att_x, att_v0, 6 shifted, r w k v a g o, kk, vv, ffn_x, ffn_kx, ffn_v, spare
    const size_t o_n = take(vecT * 4);
    const size_t o_auxw = take((size_t)d.lora_w * nt * esz), o_auxa = take((size_t)d.lora_a * nt * esz);
    const size_t o_auxg = take((size_t)d.lora_g * nt * esz), o_auxv = take((size_t)d.lora_v * nt * esz);
    const size_t o_ffnk = take(F * nt * esz);
    const size_t o_headx = take(D * nh * esz), o_heado = take(V * nh * 4);
    const size_t o_cur = take((size_t)nt * 4), o_tok = take((size_t)nt * 4), o_hdr = take((size_t)nh * 4), o_arg = take((size_t)nh * 4);
    const size_t o_cnt = take(256);
    // K-sliced GEMM (decode batches of 2 .. 64 tokens): f32 partial tiles [row group][K slice][token][64 rows] of the largest launch of a
    // layer (at most one slice per 256-block) and one arrival counter per row group
    size_t ks_floats = 0, ks_groups = 0, o_ksp = 0, o_ksc = 0;
    if (nt >= 2 && act_dtype == WRK_F16) {       // (a frame sized for longer chunks still serves decode batches of up to 64)
        const size_t ntp = nt <= 16 ? 16 : (nt <= 32 ? 32 : 64);
        auto tiles = [](size_t m, size_t k) { return ((m + 63) / 64) * (k / 256 ? k / 256 : 1); };
        auto groups = [](size_t m) { return (m + 63) / 64; };
        const size_t lora = tiles(d.lora_w, D) + tiles(d.lora_a, D) + tiles(d.lora_g, D) + tiles(d.lora_v, D);
        const size_t k1 = 3 * tiles(D, D) + lora, k5 = tiles(F, D), k6 = tiles(D, F);
        const size_t most = k1 > k5 ? (k1 > k6 ? k1 : k6) : (k5 > k6 ? k5 : k6);
        ks_floats = most * 64 * ntp;
        ks_groups = 3 * groups(D) + groups(d.lora_w) + groups(d.lora_a) + groups(d.lora_g) + groups(d.lora_v) + groups(F) + 8;
        o_ksp = take(ks_floats * 4);
        o_ksc = take(ks_groups * 4);
    }
    WRK_HIP(ctx, hipMalloc(&scratch, off));
    WRK_HIP(ctx, hipMemsetAsync(scratch, 0, off, ctx->stream));
    if (nt >= 128 && act_dtype == WRK_F16) {
        // prefill GEMM (wrk_gemm3.hip): sub-block input sums of the stacked tokens, hi + lo f16 per 32 inputs, for the up to three distinct
        // inputs of a launch (r, k, v) or the F-wide ffn vector; + the f32 partial tiles of K-split launches (chunks of up to 256 tokens:
        // the widest launch, r / k / v or the ffn key, in two to four slices)
        const size_t widest = std::max<size_t>(3 * D, F);
        const size_t part = (size_t)256 * widest * 4 * 4;
        const int32_t rs = wrk_ctx_reserve_gemm_scratch(ctx, (size_t)nt * (widest / 32) * 4 + 8 * 1024 + part);
        if (rs != WRK_OK) return rs;
    }
    char* b = (char*)scratch;
    ni = 0;
    s.input = b + o_named[ni++]; s.x = b + o_named[ni++]; s.att_x = b + o_named[ni++]; s.att_v0 = b + o_named[ni++];
    s.rx = b + o_named[ni++]; s.wx = b + o_named[ni++]; s.kx = b + o_named[ni++]; s.vx = b + o_named[ni++];
    s.ax = b + o_named[ni++]; s.gx = b + o_named[ni++];
    s.r = b + o_named[ni++]; s.w = b + o_named[ni++]; s.k = b + o_named[ni++]; s.v = b + o_named[ni++];
    s.a = b + o_named[ni++]; s.g = b + o_named[ni++]; s.o = b + o_named[ni++];
    s.kk = b + o_named[ni++]; s.vv = b + o_named[ni++];
    s.ffn_x = b + o_named[ni++]; s.ffn_kx = b + o_named[ni++]; s.ffn_v = b + o_named[ni++]; s.ln_tmp = b + o_named[ni++];
    s.n = b + o_n;
    s.aux_w = b + o_auxw; s.aux_a = b + o_auxa; s.aux_g = b + o_auxg; s.aux_v = b + o_auxv;
    s.ffn_k = b + o_ffnk;
    s.head_x = b + o_headx; s.head_o = (float*)(b + o_heado);
    s.cursors = (uint32_t*)(b + o_cur); s.tokens = (uint32_t*)(b + o_tok); s.headers = (uint32_t*)(b + o_hdr); s.argmax = (uint32_t*)(b + o_arg);
    s.counter = (uint32_t*)(b + o_cnt);
    s.ks_part = ks_floats ? (float*)(b + o_ksp) : nullptr; s.ks_cnt = ks_floats ? (uint32_t*)(b + o_ksc) : nullptr;
    s.ks_part_cap = ks_floats; s.ks_cnt_cap = (uint32_t)ks_groups;
    scratch_tokens = nt;
    scratch_headers = nh;
    // arg-max partials of the head matvec: one (value, index) per workgroup and header row
    {
        wrk::MatJob hj = mj(head, make_dense(s.head_x, WRK_F16, d.num_emb, nh), make_dense(s.head_o, WRK_F32, d.num_vocab, nh), 0);
        const size_t need = (size_t)wrk::matvec_num_wg(&hj, 1, ctx->num_cu, nullptr) * nh;
        if (need > amax_cap) {
            free_fused();
            WRK_HIP(ctx, hipMalloc((void**)&amax_val, need * 4));
            WRK_HIP(ctx, hipMalloc((void**)&amax_idx, need * 4));
            amax_cap = need;
        }
    }
    return WRK_OK;
}

static wrk::MatJob mj(const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    wrk::MatJob j{m->data, m->aux, m->kind, m->flags, m->k, m->m, (uint32_t)m->row_bytes, in, out, act, 0};
    j.scale = m->out_scale;
    return j;
}

static int32_t mm(wrk_ctx* ctx, const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    wrk::MatJob j = mj(m, in, out, act);
    j.xsum = ctx->gemm_scratch; j.xsum_cap = ctx->gemm_scratch_cap;
    int rc = -2;
    if (in.shape[1] * in.shape[2] >= wrk::gemm_min_tokens()) rc = wrk::matmul_mfma(ctx->op_stream(), j, ctx->num_cu);
    if (rc == -2) rc = wrk::matvec(ctx->op_stream(), &j, 1, ctx->num_cu);
    if (rc != 0) return wrk_fail(ctx, WRK_E_ARG, "matmul launch rejected (K=%u M=%u)", m->k, m->m);
    return WRK_OK;
}
// several matrices x the same token count in one MFMA launch per kernel family; per-matrix launches when the GEMM declines
static int32_t mm_group(wrk_ctx* ctx, wrk::MatJob* jobs, int n) {
    const uint32_t T = jobs[0].in.shape[1] * jobs[0].in.shape[2];
    jobs[0].xsum = ctx->gemm_scratch; jobs[0].xsum_cap = ctx->gemm_scratch_cap;
    if (T >= wrk::gemm_min_tokens() && wrk::matmul_mfma_multi(ctx->op_stream(), jobs, n, ctx->num_cu) == 0) return WRK_OK;
    for (int i = 0; i < n; ++i) {
        jobs[i].xsum = ctx->gemm_scratch; jobs[i].xsum_cap = ctx->gemm_scratch_cap;
        int rc = -2;
        if (T >= wrk::gemm_min_tokens()) rc = wrk::matmul_mfma(ctx->op_stream(), jobs[i], ctx->num_cu);
        if (rc == -2) rc = wrk::matvec(ctx->op_stream(), &jobs[i], 1, ctx->num_cu);
        if (rc != 0) return wrk_fail(ctx, WRK_E_ARG, "matmul launch rejected (K=%u M=%u)", jobs[i].k, jobs[i].m);
    }
    return WRK_OK;
}
#define MM(...)                                   \
    do {                                          \
        int32_t _r = mm(ctx, __VA_ARGS__);        \
        if (_r != WRK_OK) return _r;              \
    } while (0)

// ------------------------------------------------------------------ mode 0: the reference op list
int32_t wrk_v7_model::enqueue_ops(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity_headers, bool merged) {
    hipStream_t q = ctx->op_stream();
    const uint32_t D = d.num_emb, F = d.num_hidden, H = d.num_head, S = D / H, V = d.num_vocab;
    const uint32_t AT = act_dtype;      // F of Runtime<F>
    if (AT != WRK_F16) merged = false;  // the merged launches are written for dense f16 rows
    auto vec = [&](void* p, uint32_t c = 0) { return make_dense(p, AT, c ? c : D, T); };
    auto heads = [&](void* p) { return make_dense(p, AT, S, H, T); };
    DTensor x = vec(s.x), att_x = vec(s.att_x), v0 = vec(s.att_v0);
    DTensor rx = vec(s.rx), wx = vec(s.wx), kx = vec(s.kx), vx = vec(s.vx), ax = vec(s.ax), gx = vec(s.gx);
    DTensor r = vec(s.r), w = vec(s.w), k = vec(s.k), v = vec(s.v), a = vec(s.a), g = vec(s.g), o = vec(s.o);
    DTensor kk = vec(s.kk), vv = vec(s.vv);
    DTensor ffn_x = vec(s.ffn_x), ffn_kx = vec(s.ffn_kx), ffn_v = vec(s.ffn_v), ffn_k = vec(s.ffn_k, F);
    DTensor aux_w = vec(s.aux_w, d.lora_w), aux_a = vec(s.aux_a, d.lora_a), aux_g = vec(s.aux_g, d.lora_g), aux_v = vec(s.aux_v, d.lora_v);
    DTensor n4 = make_dense(s.n, AT, S, H, T, 4);
    auto nslice = [&](uint32_t i) { DTensor t = make_dense(s.n, AT, D, T, 4); t.shape[2] = 1; t.offset[2] = i; return t; };
    auto bvec = [&](const wrk_buf* b) { return make_dense(b->ptr, WRK_F16, D, 1, 1); };
    // WRK_MERGE_MASK (debug): 1 shifts in one pass, 2 grouped projections, 4 pre-WKV stage, 8 post-WKV stage, 16 W_o add in the epilogue,
    // 32 blit + layer_norm in one pass, 64 ffn.value's add in the epilogue
    const char* mm_env = getenv("WRK_MERGE_MASK");
    const unsigned mask = mm_env ? (unsigned)atoi(mm_env) : 127u;
    const bool wide = merged && S == 64 && T > 0;     // merged element-wise stages (one wave per head and token)
    const bool m_shift = merged && (mask & 1u), m_group = merged && (mask & 2u), m_pre = wide && (mask & 4u), m_post = wide && (mask & 8u),
               m_res = merged && (mask & 16u), m_ln = merged && (mask & 32u), m_tail = merged && (mask & 64u);

    // embed: LN(ln0) in place on the gathered rows, blit to x (v7.rs:649-659)
    if (!skip_embed) {
        DTensor input = make_dense(s.input, WRK_F16, D, T);     // Runtime::input is f16 whatever F is (v7.rs:283)
        wrk::layer_norm(q, ln0_w->ptr, ln0_b->ptr, input, LN_EPS);
        wrk::blit(q, input, x);
    }

    for (uint32_t li = layer_begin; li < d.num_layer && li < layer_end; ++li) {
        const wrk_v7_layer_desc& L = layers[li];
        // state views (v7.rs:198-208): att = rows 0..S, ffn = row S+1 of [D, S+2, B]
        DTensor st_att = make_dense(st->layer_ptr(li), WRK_F32, D, S + 2, st->num_batch);
        st_att.shape[1] = S + 1;
        DTensor st_row0 = st_att; st_row0.shape[1] = 1;
        DTensor st_ffn = make_dense(st->layer_ptr(li), WRK_F32, D, S + 2, st->num_batch);
        st_ffn.shape[1] = 1; st_ffn.offset[1] = S + 1;

        if (m_ln) wrk::layer_norm_from(q, L.ln1_w->ptr, L.ln1_b->ptr, x, att_x, LN_EPS);     // 1-2 in one pass
        else {
            wrk::blit(q, x, att_x);                                                      // 1
            wrk::layer_norm(q, L.ln1_w->ptr, L.ln1_b->ptr, att_x, LN_EPS);               // 2
        }
        // Mode 1 (merged): same arithmetic, fewer launches for multi-token chunks -- the six shifts read LN(x) once, the projections
        // that are ready together share a launch (r, k, v, LoRA downs | LoRA ups), the element-wise chains run as one kernel.
        if (m_shift) {
            const DTensor mixes[6] = {bvec(L.x_r), bvec(L.x_w), bvec(L.x_k), bvec(L.x_v), bvec(L.x_a), bvec(L.x_g)};
            const DTensor outs[6] = {rx, wx, kx, vx, ax, gx};
            wrk::token_shift_multi(q, s.cursors, mixes, outs, 6, st_row0, att_x, 1);
        } else {
            wrk::token_shift(q, s.cursors, bvec(L.x_r), st_row0, att_x, rx, 1);               // 3
            wrk::token_shift(q, s.cursors, bvec(L.x_w), st_row0, att_x, wx, 1);
            wrk::token_shift(q, s.cursors, bvec(L.x_k), st_row0, att_x, kx, 1);
            wrk::token_shift(q, s.cursors, bvec(L.x_v), st_row0, att_x, vx, 1);
            wrk::token_shift(q, s.cursors, bvec(L.x_a), st_row0, att_x, ax, 1);
            wrk::token_shift(q, s.cursors, bvec(L.x_g), st_row0, att_x, gx, 1);
        }
        if (m_group) {
            wrk::MatJob ja[7] = {mj(L.w_r, rx, r, WRK_ACT_NONE), mj(L.w_k, kx, k, WRK_ACT_NONE), mj(L.w_v, vx, v, WRK_ACT_NONE),
                                 mj(L.w1, wx, aux_w, WRK_ACT_TANH), mj(L.a1, ax, aux_a, WRK_ACT_NONE), mj(L.g1, gx, aux_g, WRK_ACT_SIGMOID),
                                 mj(li ? L.v1 : L.a1, li ? vx : ax, li ? aux_v : aux_a, WRK_ACT_NONE)};
            int32_t rg = mm_group(ctx, ja, li ? 7 : 6);
            if (rg != WRK_OK) return rg;
            wrk::MatJob jb[4] = {mj(L.w2, aux_w, w, WRK_ACT_NONE), mj(L.a2, aux_a, a, WRK_ACT_NONE), mj(L.g2, aux_g, g, WRK_ACT_NONE),
                                 mj(li ? L.v2 : L.a2, li ? aux_v : aux_a, li ? vv : a, WRK_ACT_NONE)};
            rg = mm_group(ctx, jb, li ? 4 : 3);
            if (rg != WRK_OK) return rg;
        } else {
            MM(L.w_r, rx, r, WRK_ACT_NONE);                                                  // 4
            MM(L.w_k, kx, k, WRK_ACT_NONE);
            MM(L.w_v, vx, v, WRK_ACT_NONE);
            MM(L.w1, wx, aux_w, WRK_ACT_TANH);                                               // 5
            MM(L.w2, aux_w, w, WRK_ACT_NONE);
            MM(L.a1, ax, aux_a, WRK_ACT_NONE);                                               // 6
            MM(L.a2, aux_a, a, WRK_ACT_NONE);
            MM(L.g1, gx, aux_g, WRK_ACT_SIGMOID);                                            // 7
            MM(L.g2, aux_g, g, WRK_ACT_NONE);
            if (li) {                                                                        // 10 (projections)
                MM(L.v1, vx, aux_v, WRK_ACT_NONE);
                MM(L.v2, aux_v, vv, WRK_ACT_NONE);
            }
        }
        // (not with many sequences: the one-wave-per-head kernel prepares its decays itself, and 4 D bytes per token would be written for nothing)
        float* wdec = ((size_t)F * (act_dtype == WRK_F32 ? 4 : 2) >= (size_t)4 * D && (size_t)wkv_nseq * H < 768) ? (float*)s.ffn_k : nullptr;
        if (m_pre) {     // steps 5-11's element-wise ops in one launch, bit-identical (wrk_ops.hip: pre_wkv_v7)
            // (the decays of the chunk kernel ride along, f32 [D, T], in the ffn key buffer: 4 D values per token, free until the ffn key GEMM)
            wrk::pre_wkv_v7(q, s.w, s.a, s.k, s.v, s.vv, s.att_v0, s.n, L.w0->ptr, L.a0->ptr, L.k_k->ptr, L.k_a->ptr, li ? L.v0->ptr : L.a0->ptr,
                            D, T, li == 0, L2_EPS, wdec);
        } else {
            wrk::binary(q, 0, bvec(L.w0), w, 0, 0, 0);                                       // 5
            wrk::binary(q, 0, bvec(L.a0), a, 0, 0, WRK_ACT_SIGMOID);                         // 6
            wrk::blit(q, k, kk);                                                             // 8
            wrk::binary(q, 1, bvec(L.k_k), kk, 0, 0, 0);
            wrk::l2_norm(q, heads(s.kk), L2_EPS);
            wrk::control_k_v7(q, L.k_a->ptr, a, k);                                          // 9
            if (li == 0) {                                                                   // 10
                wrk::blit(q, v, v0);
            } else {
                wrk::binary(q, 0, bvec(L.v0), vv, 0, 0, WRK_ACT_SIGMOID);
                wrk::lerp(q, v0, v, vv, 1);
            }
            wrk::blit(q, k, nslice(0));                                                      // 11
            wrk::blit(q, v, nslice(1));
            wrk::blit(q, a, nslice(2));
            wrk::blit(q, kk, nslice(3));
        }
        wrk::time_mix_v7(q, s.cursors, st_att, heads(s.r), heads(s.w), n4, heads(s.att_x), wkv_nseq, m_pre ? wdec : nullptr);   // 12
        if (m_post) wrk::post_wkv_v7(q, s.att_x, s.r, s.g, s.n, L.gn_w->ptr, L.gn_b->ptr, L.r_k->ptr, D, T, GN_EPS);    // 13-15 in one launch
        else {
            wrk::group_norm(q, L.gn_w->ptr, L.gn_b->ptr, heads(s.att_x), GN_EPS);        // 13
            wrk::time_first_v7(q, L.r_k->ptr, heads(s.r), n4, heads(s.att_x));           // 14
            wrk::binary(q, 1, g, att_x, 0, 0, 0);                                        // 15
        }
        if (m_res) {        // the add rides the projection's epilogue: x = round(W_o att_x) + x
            wrk::MatJob jo = mj(L.w_o, att_x, x, WRK_ACT_NONE);
            jo.has_res = 1;
            jo.res = x;
            const int32_t rg = mm_group(ctx, &jo, 1);
            if (rg != WRK_OK) return rg;
        } else {
            MM(L.w_o, att_x, o, WRK_ACT_NONE);                                           // 16
            wrk::binary(q, 0, o, x, 0, 0, 0);
        }
        if (m_ln) wrk::layer_norm_from(q, L.ln2_w->ptr, L.ln2_b->ptr, x, ffn_x, LN_EPS);
        else {
            wrk::blit(q, x, ffn_x);                                                      // 17
            wrk::layer_norm(q, L.ln2_w->ptr, L.ln2_b->ptr, ffn_x, LN_EPS);
        }
        wrk::token_shift(q, s.cursors, bvec(L.ffn_x_k), st_ffn, ffn_x, ffn_kx, 1);        // 18
        MM(L.ffn_w_k, ffn_kx, ffn_k, WRK_ACT_SQUARED_RELU);                              // 19
        if (m_tail) {       // x = round(W_v k) + x in the epilogue; channel_mix still saves the ffn shift state (its copy is dead)
            wrk::MatJob jv = mj(L.ffn_w_v, ffn_k, x, WRK_ACT_NONE);
            jv.has_res = 1;
            jv.res = x;
            const int32_t rg = mm_group(ctx, &jv, 1);
            if (rg != WRK_OK) return rg;
            wrk::channel_mix_v7(q, s.cursors, st_ffn, ffn_v, ffn_x);
        } else {
            MM(L.ffn_w_v, ffn_k, ffn_v, WRK_ACT_NONE);                                   // 20
            wrk::channel_mix_v7(q, s.cursors, st_ffn, ffn_v, ffn_x);                     // 21
            wrk::binary(q, 0, ffn_x, x, 0, 0, 0);                                        // 22
        }
        if ((li + 1) % d.rescale == 0) wrk::affine(q, x, 0.5f, 0.0f);                    // 23
    }
    // header (v7.rs:1009-1036): gather header rows, LN(ln_out), head matmul into f32 logits
    if (NH > 0) {
        DTensor head_x = make_dense(s.head_x, AT, D, NH);
        if (identity_headers) wrk::blit(q, make_dense(s.x, AT, D, NH), head_x);
        else wrk::gather_rows_any(q, x, s.headers, head_x, NH);
        wrk::layer_norm(q, ln_out_w->ptr, ln_out_b->ptr, head_x, LN_EPS);
        MM(head, head_x, make_dense(s.head_o, WRK_F32, V, NH), WRK_ACT_NONE);
    }
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

// ------------------------------------------------------------------ C ABI
bool split_head_env_on() { const char* e = getenv("WRK_SPLIT_HEAD"); return !(e && e[0] == '0'); }

extern "C" {

int32_t wrk_v7_model_create(wrk_ctx* ctx, const wrk_v7_model_desc* desc, wrk_v7_model** out) {
    if (!ctx || !desc || !out) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_ARG(ctx, desc->num_layer >= 1 && desc->num_head >= 1 && desc->num_emb % desc->num_head == 0, "bad model dims");
    WRK_ARG(ctx, desc->num_emb / desc->num_head == 64, "head size must be 64");
    WRK_ARG(ctx, desc->layers && desc->head && desc->ln0_w && desc->ln0_b && desc->ln_out_w && desc->ln_out_b, "missing tensors");
    WRK_ARG(ctx, desc->head->k == desc->num_emb && desc->head->m >= desc->num_vocab, "head matrix shape mismatch");
    const uint32_t D = desc->num_emb, F = desc->num_hidden;
    for (uint32_t l = 0; l < desc->num_layer; ++l) {
        const wrk_v7_layer_desc& L = desc->layers[l];
        const wrk_buf* vecs[] = {L.ln1_w, L.ln1_b, L.ln2_w, L.ln2_b, L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g, L.w0, L.a0,
                                 L.r_k, L.k_k, L.k_a, L.gn_w, L.gn_b, L.ffn_x_k};
        for (const wrk_buf* b : vecs) WRK_ARG(ctx, b && b->bytes >= (size_t)D * 2, "layer %u: vector missing or shorter than D f16", l);
        struct { const wrk_matrix* m; uint32_t k, mm; } mats[] = {
            {L.w1, D, desc->lora_w}, {L.w2, desc->lora_w, D}, {L.a1, D, desc->lora_a}, {L.a2, desc->lora_a, D},
            {L.g1, D, desc->lora_g}, {L.g2, desc->lora_g, D}, {L.w_k, D, D}, {L.w_v, D, D}, {L.w_r, D, D}, {L.w_o, D, D},
            {L.ffn_w_k, D, F}, {L.ffn_w_v, F, D}};
        for (auto& e : mats) WRK_ARG(ctx, e.m && e.m->k == e.k && e.m->m == e.mm, "layer %u: matrix missing or wrong shape (want K=%u M=%u)", l, e.k, e.mm);
        if (l > 0) {
            WRK_ARG(ctx, L.v0 && L.v0->bytes >= (size_t)D * 2, "layer %u: v0 missing", l);
            WRK_ARG(ctx, L.v1 && L.v1->k == D && L.v1->m == desc->lora_v && L.v2 && L.v2->k == desc->lora_v && L.v2->m == D, "layer %u: v1/v2 wrong shape", l);
        }
    }
    wrk_v7_model* m = new wrk_v7_model();
    m->ctx = ctx;
    m->d = *desc;
    m->d.rescale = desc->rescale ? desc->rescale : 1024;
    m->layers.assign(desc->layers, desc->layers + desc->num_layer);
    m->d.layers = m->layers.data();
    m->ln0_w = desc->ln0_w; m->ln0_b = desc->ln0_b; m->ln_out_w = desc->ln_out_w; m->ln_out_b = desc->ln_out_b;
    m->emb = desc->emb_f16; m->head = desc->head;
    // retain every handle (the Rust side keeps Arc clones inside v7::Model)
    auto rb = [](const wrk_buf* b) { if (b) const_cast<wrk_buf*>(b)->refs.fetch_add(1); };
    auto rm = [](const wrk_matrix* x) { if (x) const_cast<wrk_matrix*>(x)->refs.fetch_add(1); };
    rb(m->ln0_w); rb(m->ln0_b); rb(m->ln_out_w); rb(m->ln_out_b); rb(m->emb); rm(m->head);
    for (auto& L : m->layers) {
        const wrk_buf* vecs[] = {L.ln1_w, L.ln1_b, L.ln2_w, L.ln2_b, L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g, L.w0, L.a0, L.v0,
                                 L.r_k, L.k_k, L.k_a, L.gn_w, L.gn_b, L.ffn_x_k};
        for (const wrk_buf* b : vecs) rb(b);
        const wrk_matrix* mats[] = {L.w1, L.w2, L.a1, L.a2, L.g1, L.g2, L.v1, L.v2, L.w_k, L.w_v, L.w_r, L.w_o, L.ffn_w_k, L.ffn_w_v};
        for (const wrk_matrix* x : mats) rm(x);
    }
    *out = m;
    return WRK_OK;
}

int32_t wrk_v7_model_destroy(wrk_v7_model* m) {
    if (!m) return WRK_E_ARG;
    wrk_ctx* ctx = m->ctx;
    {
        LOCK(ctx);
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
        for (auto& kv : m->graphs) wrk_program_destroy(kv.second);
        if (m->scratch) hipFree(m->scratch);
        m->free_fused();
        wrk_v7_engine_destroy(m->engine);
        m->engine = nullptr;
        for (hipStream_t s : m->lane_streams) { hipStreamSynchronize(s); hipStreamDestroy(s); }
        for (hipEvent_t e : m->lane_events) hipEventDestroy(e);
        m->lane_streams.clear(); m->lane_events.clear();
    }
    for (wrk_v7_model* lane : m->lanes) wrk_v7_model_destroy(lane);
    m->lanes.clear();
    if (m->history) { LOCK(ctx); hipFree(m->history); m->history = nullptr; }
    auto fb = [](const wrk_buf* b) { if (b) wrk_buf_release(const_cast<wrk_buf*>(b)); };
    auto fm = [](const wrk_matrix* x) { if (x) wrk_matrix_release(const_cast<wrk_matrix*>(x)); };
    fb(m->ln0_w); fb(m->ln0_b); fb(m->ln_out_w); fb(m->ln_out_b); fb(m->emb); fm(m->head);
    for (auto& L : m->layers) {
        const wrk_buf* vecs[] = {L.ln1_w, L.ln1_b, L.ln2_w, L.ln2_b, L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g, L.w0, L.a0, L.v0,
                                 L.r_k, L.k_k, L.k_a, L.gn_w, L.gn_b, L.ffn_x_k};
        for (const wrk_buf* b : vecs) fb(b);
        const wrk_matrix* mats[] = {L.w1, L.w2, L.a1, L.a2, L.g1, L.g2, L.v1, L.v2, L.w_k, L.w_v, L.w_r, L.w_o, L.ffn_w_k, L.ffn_w_v};
        for (const wrk_matrix* x : mats) fm(x);
    }
    delete m;
    return WRK_OK;
}

// A = sum(weight tensors read per token) + 2*L*D*(S+2)*4*B + B*(D*2 + V*4)
size_t wrk_v7_model_token_bytes(const wrk_v7_model* m, uint32_t B) {
    if (!m) return 0;
    const size_t D = m->d.num_emb, S = D / m->d.num_head, V = m->d.num_vocab;
    size_t w = wrk_matrix_stream_bytes(m->head) + 4 * D * 2;     // head + ln0/ln_out vectors
    for (uint32_t l = 0; l < m->d.num_layer; ++l) {
        const wrk_v7_layer_desc& L = m->layers[l];
        const wrk_matrix* mats[] = {L.w1, L.w2, L.a1, L.a2, L.g1, L.g2, L.w_k, L.w_v, L.w_r, L.w_o, L.ffn_w_k, L.ffn_w_v};
        for (const wrk_matrix* x : mats) w += wrk_matrix_stream_bytes(x);
        if (l > 0) w += wrk_matrix_stream_bytes(L.v1) + wrk_matrix_stream_bytes(L.v2) + D * 2;
        w += 18 * D * 2;                                         // f16 vectors of the layer
    }
    return w + 2 * (size_t)m->d.num_layer * D * (S + 2) * 4 * B + (size_t)B * (D * 2 + V * 4);
}

int32_t wrk_v7_state_create(wrk_ctx* ctx, const wrk_v7_model* model, uint32_t num_batch, wrk_v7_state** out) {
    if (!ctx || !model || !out) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_ARG(ctx, num_batch >= 1 && num_batch <= 255, "num_batch must be 1..255 (cursor batch is u8, tensor/mod.rs:53-60)");
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    wrk_v7_state* st = new wrk_v7_state();
    st->ctx = ctx;
    st->num_layer = model->d.num_layer;
    st->num_emb = model->d.num_emb;
    st->head_size = model->d.num_emb / model->d.num_head;
    st->num_batch = num_batch;
    const size_t bytes = st->layer_elems() * st->num_layer * 4;
    hipError_t e = hipMalloc((void**)&st->data, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(st->data, 0, bytes, ctx->stream);
    if (e != hipSuccess) { delete st; return wrk_fail(ctx, e == hipErrorOutOfMemory ? WRK_E_OOM : WRK_E_HIP, "state alloc: %s", hipGetErrorString(e)); }
    *out = st;
    return WRK_OK;
}

int32_t wrk_v7_state_destroy(wrk_v7_state* st) {
    if (!st) return WRK_E_ARG;
    {
        LOCK(st->ctx);
        hipSetDevice(st->ctx->device);
        hipStreamSynchronize(st->ctx->stream);
        hipFree(st->data);
    }
    delete st;
    return WRK_OK;
}

// host layout [L][S+2][D] (reference shape [D, S+2, L, 1], x fastest); device [L][B][S+2][D]
int32_t wrk_v7_state_load(wrk_ctx* ctx, wrk_v7_state* st, uint32_t batch, const float* src) {
    if (!ctx || !st || !src) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, batch < st->num_batch, "batch %u out of range", batch);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t per = (size_t)(st->head_size + 2) * st->num_emb;
    for (uint32_t l = 0; l < st->num_layer; ++l)
        WRK_HIP(ctx, hipMemcpyAsync(st->layer_ptr(l) + batch * per, src + l * per, per * 4, hipMemcpyHostToDevice, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WRK_OK;
}

int32_t wrk_v7_state_back(wrk_ctx* ctx, const wrk_v7_state* st, uint32_t batch, float* dst) {
    if (!ctx || !st || !dst) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, batch < st->num_batch, "batch %u out of range", batch);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t per = (size_t)(st->head_size + 2) * st->num_emb;
    for (uint32_t l = 0; l < st->num_layer; ++l)
        WRK_HIP(ctx, hipMemcpyAsync(dst + l * per, st->layer_ptr(l) + batch * per, per * 4, hipMemcpyDeviceToHost, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WRK_OK;
}

static int32_t state_d2d(wrk_ctx* ctx, const wrk_v7_state* st, uint32_t batch, const wrk_buf* buf, bool to_state) {
    if (!ctx || !st || !buf) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, batch < st->num_batch, "batch %u out of range", batch);
    const size_t per = (size_t)(st->head_size + 2) * st->num_emb;
    WRK_ARG(ctx, buf->bytes >= per * st->num_layer * 4, "snapshot buffer holds %zu bytes, a state needs %zu", buf->bytes, per * st->num_layer * 4);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t l = 0; l < st->num_layer; ++l) {
        float* slot = st->layer_ptr(l) + batch * per;
        float* snap = (float*)buf->ptr + l * per;
        WRK_HIP(ctx, hipMemcpyAsync(to_state ? slot : snap, to_state ? snap : slot, per * 4, hipMemcpyDeviceToDevice, ctx->op_stream()));
    }
    return WRK_OK;
}

int32_t wrk_v7_state_read(wrk_ctx* ctx, const wrk_v7_state* st, uint32_t batch, wrk_buf* buf) { return state_d2d(ctx, st, batch, buf, false); }
int32_t wrk_v7_state_write(wrk_ctx* ctx, wrk_v7_state* st, uint32_t batch, const wrk_buf* buf) { return state_d2d(ctx, st, batch, buf, true); }

int32_t wrk_v7_infer(wrk_ctx* ctx, wrk_v7_model* m, wrk_v7_state* st, const uint32_t* tokens, const uint16_t* emb_rows,
                     const uint32_t* cursors, uint32_t T, const uint32_t* headers, uint32_t NH, float* logits, uint32_t* argmax,
                     uint32_t mode) {
    if (!ctx || !m || !st) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    if (T == 0) return WRK_OK;                                  // v7.rs:626-635: empty job
    WRK_ARG(ctx, cursors, "cursors required");
    WRK_ARG(ctx, tokens || emb_rows, "either token ids or gathered embedding rows are required");
    WRK_ARG(ctx, !tokens || m->emb, "token ids given but the model has no device embedding table");
    WRK_ARG(ctx, NH == 0 || headers, "headers required");
    WRK_ARG(ctx, st->num_emb == m->d.num_emb && st->num_layer == m->d.num_layer, "state does not belong to this model");
    const uint32_t D = m->d.num_emb, V = m->d.num_vocab;
    // validate cursors / tokens / headers on the host: a bad index would fault the GPU
    uint32_t nseq = 0;
    bool one_token_each = true;
    std::vector<uint8_t> seen(256, 0);
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t c = cursors[t], b = c & 0xff, tok = (c >> 8) & 0xffff, len = c >> 24;
        WRK_ARG(ctx, b < st->num_batch, "cursor %u: batch %u >= %u", t, b, st->num_batch);
        WRK_ARG(ctx, len >= 1 && tok <= t && t < tok + len && tok + len <= T, "cursor %u: bad range (token %u len %u)", t, tok, len);
        if (tok == t) {
            ++nseq;
            WRK_ARG(ctx, !seen[b], "cursor %u: batch %u appears in two chunks of one dispatch", t, b);   // two writers of one state slice
            seen[b] = 1;
        }
        if (len != 1) one_token_each = false;
        if (tokens) WRK_ARG(ctx, tokens[t] < V, "token %u: id %u >= vocab %u", t, tokens[t], V);
    }
    // batches of consecutive tokens are consecutive: the few-sequence decode kernels address per-sequence state rows by a stride
    bool contiguous = true;
    for (uint32_t t = 1; t < T; ++t) contiguous = contiguous && (cursors[t] & 0xff) == (cursors[0] & 0xff) + t;
    bool identity = (NH == T);
    for (uint32_t h = 0; h < NH; ++h) {
        WRK_ARG(ctx, headers[h] < T, "header %u: row %u >= %u tokens", h, headers[h], T);
        if (headers[h] != h) identity = false;
    }
    int32_t rc = m->ensure_scratch(T, NH ? NH : 1);
    if (rc != WRK_OK) return rc;
    if (T == 1 && mode == 1) { rc = m->ensure_engine(); if (rc != WRK_OK) return rc; }
    rc = wrk_buf_write_raw(ctx, m->s.cursors, cursors, (size_t)T * 4);
    if (rc != WRK_OK) return rc;
    if (NH) { rc = wrk_buf_write_raw(ctx, m->s.headers, headers, (size_t)NH * 4); if (rc != WRK_OK) return rc; }
    const bool fused = (mode == 1 && one_token_each && nseq == T && m->act_dtype == WRK_F16);
    if (tokens) {
        rc = wrk_buf_write_raw(ctx, m->s.tokens, tokens, (size_t)T * 4);
        if (rc != WRK_OK) return rc;
        if (!fused) wrk::gather_rows_f16(ctx->op_stream(), m->emb->ptr, m->s.tokens, m->s.input, D, T);
    } else {
        rc = wrk_buf_write_raw(ctx, m->s.input, emb_rows, (size_t)T * D * 2);
        if (rc != WRK_OK) return rc;
    }
    // The launches of a job depend only on its shape (token count, header rows, flags): cursors, tokens and header rows
    // are device data.  Capture once per shape and replay (the reference caches the RnnJob of a repeated RnnInfo);
    // a 128-token chunk of one sequence is ~1 400 small launches, which replay at graph rate.
    m->wkv_nseq = nseq;       // chunk kernel of the WKV state: one wave or four per head (part of the graph key below)
    auto enqueue_job = [&]() -> int32_t {
        int32_t r;
        if (fused) r = m->enqueue_fused_decode(st, T, NH, identity, tokens != nullptr, NH && argmax, false, cursors[0] & 0xff, contiguous);
        else {
            r = m->enqueue_ops(st, T, NH, identity, mode == 1);
            if (r == WRK_OK && NH && argmax) wrk::argmax_rows(ctx->op_stream(), m->s.head_o, V, V, NH, m->s.argmax);
        }
        return r;
    };
    static const bool no_graph = [] { const char* e = getenv("WRK_NO_GRAPH"); return e && e[0] == '1'; }();
    if (no_graph || ctx->capturing_here()) rc = enqueue_job();
    else {
        // bit 5: a non-fused job enqueues the merged launch list in mode 1 and the reference op list in mode 0 -- two graphs
        const uint32_t flags = 16u | (m->act_dtype == WRK_F32 ? 64u : 0u) | (fused ? 1u : 0u) | (identity ? 2u : 0u) | (tokens ? 4u : 0u) | ((NH && argmax) ? 8u : 0u) |
                               ((!fused && mode == 1) ? 32u : 0u) | (fused ? (cursors[0] & 0xffu) << 8 : 0u) | ((fused && contiguous) ? 1u << 16 : 0u) |
                               ((fused && T == 1 && m->engine_on()) ? 1u << 17 : 0u) | (split_head_env_on() ? 0u : 1u << 18) |
                               ((!fused && (size_t)nseq * m->d.num_head >= 768) ? 1u << 19 : 0u);
        const wrk_v7_model::GraphKey key{st->uid, T, flags, NH};
        wrk_program* prog = nullptr;
        auto it = m->graphs.find(key);
        if (it != m->graphs.end()) prog = it->second;
        else {
            if (m->graphs.size() > 64) { WRK_HIP(ctx, hipStreamSynchronize(ctx->stream)); m->drop_graphs(); }    // bound the cache
            rc = wrk_capture_begin(ctx);
            if (rc != WRK_OK) return rc;
            rc = enqueue_job();
            wrk_program* p = nullptr;
            const int32_t rc2 = wrk_capture_end(ctx, &p);
            if (rc != WRK_OK) { if (p) wrk_program_destroy(p); return rc; }
            if (rc2 != WRK_OK) return rc2;
            prog = p;
            m->graphs[key] = prog;
        }
        WRK_HIP(ctx, hipGraphLaunch(prog->exec, ctx->stream));
        rc = WRK_OK;
    }
    if (rc != WRK_OK) return rc;
    WRK_LAUNCH_CHECK(ctx);
    if (ctx->capturing_here()) return WRK_OK;   // recorded into the caller's program: results exist after it has been launched
    if (NH && logits) WRK_HIP(ctx, hipMemcpyAsync(logits, m->s.head_o, (size_t)NH * V * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (NH && argmax) WRK_HIP(ctx, hipMemcpyAsync(argmax, m->s.argmax, (size_t)NH * 4, hipMemcpyDeviceToHost, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return wrk_v7_engine_check(m->engine);
}

// Bundle::<F>::new (v7.rs:514-536): the activation type of the frame.  F16 is the reference's default build
// (`Bundle::<f16>`); F32 stores every Runtime<F> buffer in f32 and takes the op-by-op path with f32-input matmuls.
int32_t wrk_v7_model_set_frame_dtype(wrk_ctx* ctx, wrk_v7_model* m, uint32_t dtype) {
    if (!ctx || !m) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, dtype == WRK_F16 || dtype == WRK_F32, "frame dtype must be WRK_F16 or WRK_F32");
    WRK_ARG(ctx, !ctx->capturing_here(), "cannot change the frame inside a capture");
    if (dtype == m->act_dtype) return WRK_OK;
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->drop_graphs();
    if (m->scratch) hipFree(m->scratch);
    m->scratch = nullptr;
    m->scratch_tokens = m->scratch_headers = 0;
    m->act_dtype = dtype;
    // the concurrent-pipeline lanes carry the frame type they were created with (and programs captured for it): drop them, the next
    // generate_greedy(groups > 1) rebuilds them with the new type
    for (hipStream_t ls : m->lane_streams) WRK_HIP(ctx, hipStreamSynchronize(ls));
    std::vector<wrk_v7_model*> old_lanes;
    old_lanes.swap(m->lanes);
    for (wrk_v7_model* lane : old_lanes) wrk_v7_model_destroy(lane);
    return WRK_OK;
}

int32_t wrk_v7_model_engine_status(wrk_ctx* ctx, wrk_v7_model* m, char* why, size_t capacity) {
    if (!ctx || !m) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_ARG(ctx, !ctx->capturing_here(), "not inside a capture");
    const int32_t rc = m->ensure_engine();
    if (rc != WRK_OK) return rc;
    if (why && capacity) {
        const std::string& w = m->engine ? std::string() : (m->engine_tried ? m->engine_why : std::string("disabled (WRK_ENGINE=0 or f32 frames)"));
        snprintf(why, capacity, "%s", w.c_str());
    }
    return m->engine ? 1 : 0;
}

// Teacher-forced run of ONE layer (parity tests; the reference reaches the same buffers through v7::HookMap closures over
// `Frame`, v7.rs:386-421, examples/inspect.rs:100-248): the layer's op list (mode 0), or its fused / merged launches (mode 1),
// on a caller-supplied layer input.  Every Runtime<F> buffer of the frame stays readable through wrk_v7_frame_read.
int32_t wrk_v7_infer_layer(wrk_ctx* ctx, wrk_v7_model* m, wrk_v7_state* st, uint32_t layer, const void* x, const void* v_first,
                           const uint32_t* cursors, uint32_t T, uint32_t mode) {
    if (!ctx || !m || !st || !x || !cursors) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_ARG(ctx, !ctx->capturing_here(), "wrk_v7_infer_layer cannot be captured");
    WRK_ARG(ctx, layer < m->d.num_layer, "layer %u out of range", layer);
    WRK_ARG(ctx, T >= 1, "no tokens");
    WRK_ARG(ctx, layer == 0 || v_first, "layers above 0 read the layer-0 value (att_v0)");
    WRK_ARG(ctx, st->num_emb == m->d.num_emb && st->num_layer == m->d.num_layer, "state does not belong to this model");
    bool one_token_each = true;
    uint32_t nseq = 0;
    std::vector<uint8_t> seen(256, 0);
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t c = cursors[t], b = c & 0xff, tok = (c >> 8) & 0xffff, len = c >> 24;
        WRK_ARG(ctx, b < st->num_batch, "cursor %u: batch %u >= %u", t, b, st->num_batch);
        WRK_ARG(ctx, len >= 1 && tok <= t && t < tok + len && tok + len <= T, "cursor %u: bad range", t);
        if (tok == t) { ++nseq; WRK_ARG(ctx, !seen[b], "cursor %u: batch %u twice", t, b); seen[b] = 1; }
        if (len != 1) one_token_each = false;
    }
    bool contiguous = true;
    for (uint32_t t = 1; t < T; ++t) contiguous = contiguous && (cursors[t] & 0xff) == (cursors[0] & 0xff) + t;
    int32_t rc = m->ensure_scratch(T, 1);
    if (rc != WRK_OK) return rc;
    if (T == 1 && mode == 1) { rc = m->ensure_engine(); if (rc != WRK_OK) return rc; }
    const size_t esz = m->act_dtype == WRK_F32 ? 4 : 2;
    rc = wrk_buf_write_raw(ctx, m->s.cursors, cursors, (size_t)T * 4);
    if (rc == WRK_OK) rc = wrk_buf_write_raw(ctx, m->s.x, x, (size_t)T * m->d.num_emb * esz);
    if (rc == WRK_OK && v_first) rc = wrk_buf_write_raw(ctx, m->s.att_v0, v_first, (size_t)T * m->d.num_emb * esz);
    if (rc != WRK_OK) return rc;
    m->layer_begin = layer; m->layer_end = layer + 1; m->skip_embed = true;
    m->wkv_nseq = nseq;
    // this entry point exists to read the frame buffers of a layer back (wrk_v7_frame_read): the engine keeps them in LDS and granules,
    // so the launches run here; WRK_ENGINE_INSPECT=1 (tests/test_gpu_engine.py) runs the engine's layer instead
    { const char* ei = getenv("WRK_ENGINE_INSPECT"); m->engine_skip_once = !(ei && ei[0] == '1'); }
    if (mode == 1 && one_token_each && nseq == T && m->act_dtype == WRK_F16) rc = m->enqueue_fused_decode(st, T, 0, true, false, false, false, cursors[0] & 0xff, contiguous);
    else rc = m->enqueue_ops(st, T, 0, true, mode == 1);
    m->layer_begin = 0; m->layer_end = 0xffffffffu; m->skip_embed = false; m->engine_skip_once = false;
    if (rc != WRK_OK) return rc;
    WRK_LAUNCH_CHECK(ctx);
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return wrk_v7_engine_check(m->engine);
}

// TensorGpu::back on one buffer of the frame (names as examples/inspect.rs:208-248: x, att_x, att_r, ..., ffn_v; plus head_x, head_o)
int32_t wrk_v7_frame_read(wrk_ctx* ctx, wrk_v7_model* m, const char* name, uint32_t T, void* dst, size_t capacity, size_t* bytes) {
    if (!ctx || !m || !name || !bytes) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, m->scratch && T >= 1 && T <= m->scratch_tokens, "no frame of %u tokens has been run", T);
    const uint32_t D = m->d.num_emb, F = m->d.num_hidden;
    const V7Scratch& s = m->s;
    struct { const char* n; const void* p; uint32_t c; } tab[] = {
        {"x", s.x, D}, {"att_x", s.att_x, D}, {"att_v0", s.att_v0, D}, {"att_rx", s.rx, D}, {"att_wx", s.wx, D}, {"att_kx", s.kx, D},
        {"att_vx", s.vx, D}, {"att_ax", s.ax, D}, {"att_gx", s.gx, D}, {"att_r", s.r, D}, {"att_w", s.w, D}, {"att_k", s.k, D}, {"att_v", s.v, D},
        {"att_a", s.a, D}, {"att_g", s.g, D}, {"att_o", s.o, D}, {"att_kk", s.kk, D}, {"att_vv", s.vv, D}, {"att_n", s.n, 4 * D},
        {"aux_w", s.aux_w, m->d.lora_w}, {"aux_a", s.aux_a, m->d.lora_a}, {"aux_g", s.aux_g, m->d.lora_g}, {"aux_v", s.aux_v, m->d.lora_v},
        {"ffn_x", s.ffn_x, D}, {"ffn_kx", s.ffn_kx, D}, {"ffn_k", s.ffn_k, F}, {"ffn_v", s.ffn_v, D}, {"att_x_ln", s.ln_tmp, D}};
    const size_t esz = m->act_dtype == WRK_F32 ? 4 : 2;
    for (auto& e : tab)
        if (strcmp(e.n, name) == 0) {
            const size_t n = (size_t)e.c * T * esz;
            *bytes = n;
            if (!dst) return WRK_OK;
            WRK_ARG(ctx, capacity >= n, "frame buffer %s needs %zu bytes, capacity %zu", name, n, capacity);
            WRK_HIP(ctx, hipSetDevice(ctx->device));
            WRK_HIP(ctx, hipMemcpyAsync(dst, e.p, n, hipMemcpyDeviceToHost, ctx->stream));
            WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return WRK_OK;
        }
    return wrk_fail(ctx, WRK_E_ARG, "no frame buffer named %s", name);
}

// generate_greedy, part 1: frame, token / cursor upload and the (cached) decode-step program of sequences [b0, b0 + B) on model frame `m`
static int32_t greedy_prepare(wrk_ctx* ctx, wrk_v7_model* m, wrk_v7_state* st, const uint32_t* first_tokens, uint32_t b0, uint32_t B,
                              uint32_t steps, uint32_t mode, bool eager, wrk_program** prog_out) {
    const uint32_t D = m->d.num_emb, V = m->d.num_vocab;
    int32_t rc = m->ensure_scratch(B, B);
    if (rc != WRK_OK) return rc;
    if (B == 1 && mode == 1) { rc = m->ensure_engine(); if (rc != WRK_OK) return rc; }
    rc = m->ensure_history((size_t)steps * B);
    if (rc != WRK_OK) return rc;
    std::vector<uint32_t> cur(B), hdr(B);
    for (uint32_t b = 0; b < B; ++b) { cur[b] = (b0 + b) | (b << 8) | (1u << 24); hdr[b] = b; }
    rc = wrk_buf_write_raw(ctx, m->s.cursors, cur.data(), (size_t)B * 4);
    if (rc == WRK_OK) rc = wrk_buf_write_raw(ctx, m->s.headers, hdr.data(), (size_t)B * 4);
    if (rc == WRK_OK) rc = wrk_buf_write_raw(ctx, m->s.tokens, first_tokens, (size_t)B * 4);
    if (rc != WRK_OK) return rc;
    WRK_HIP(ctx, hipMemsetAsync(m->s.counter, 0, 4, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *prog_out = nullptr;
    if (eager) return WRK_OK;
    // one graph per (state, first sequence, B, mode): the analogue of the reference's cached RnnJob for a repeated RnnInfo
    const wrk_v7_model::GraphKey key{st->uid, B | (b0 << 16), mode | (m->act_dtype == WRK_F32 ? 4u : 0u) | ((B == 1 && mode == 1 && m->engine_on()) ? 8u : 0u) |
                                                             (split_head_env_on() ? 0u : 16u)};
    auto it = m->graphs.find(key);
    if (it != m->graphs.end()) { *prog_out = it->second; return WRK_OK; }
    rc = wrk_capture_begin(ctx);
    if (rc != WRK_OK) return rc;
    if (mode == 1 && m->act_dtype == WRK_F16) rc = m->enqueue_fused_decode(st, B, B, true, true, true, true, b0, true);
    else {
        wrk::gather_rows_f16(ctx->op_stream(), m->emb->ptr, m->s.tokens, m->s.input, D, B);
        rc = m->enqueue_ops(st, B, B, true);
        if (rc == WRK_OK) {
            wrk::argmax_rows(ctx->op_stream(), m->s.head_o, V, V, B, m->s.argmax);
            wrk::advance_tokens(ctx->op_stream(), m->s.argmax, m->s.tokens, m->history, m->s.counter, B);
        }
    }
    wrk_program* p = nullptr;
    const int32_t rc2 = wrk_capture_end(ctx, &p);
    if (rc != WRK_OK) { if (p) wrk_program_destroy(p); return rc; }
    if (rc2 != WRK_OK) return rc2;
    m->graphs[key] = p;
    *prog_out = p;
    return WRK_OK;
}

int32_t wrk_v7_generate_greedy(wrk_ctx* ctx, wrk_v7_model* m, wrk_v7_state* st, const uint32_t* first_tokens, uint32_t B,
                               uint32_t steps, uint32_t* out_tokens, float* last_logits, float* elapsed_ms, uint32_t mode_arg) {
    if (!ctx || !m || !st || !first_tokens) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_ARG(ctx, m->emb, "generate_greedy needs the device embedding table");
    WRK_ARG(ctx, B >= 1 && B <= st->num_batch, "num_batch %u exceeds the state's %u", B, st->num_batch);
    WRK_ARG(ctx, st->num_emb == m->d.num_emb && st->num_layer == m->d.num_layer, "state does not belong to this model");
    const uint32_t V = m->d.num_vocab;
    for (uint32_t b = 0; b < B; ++b) WRK_ARG(ctx, first_tokens[b] < V, "first token %u out of vocab", first_tokens[b]);
    if (elapsed_ms) *elapsed_ms = 0.0f;
    if (steps == 0) return WRK_OK;
    // mode: bits 0-7 = 0 op-by-op / 1 fused; bits 8-15 = number of concurrent pipelines the sequences are dealt over (0, 1: one)
    const uint32_t mode = mode_arg & 0xffu;
    uint32_t groups = (mode_arg >> 8) & 0xffu;
    if (groups < 1) groups = 1;
    if (groups > B) groups = B;
    const char* ng = getenv("WRK_NO_GRAPH");
    const bool eager = ng && ng[0] == '1';
    WRK_ARG(ctx, groups == 1 || !eager, "concurrent pipelines replay captured programs: not with WRK_NO_GRAPH=1");
    wrk::timing_slot(ctx, nullptr);     // WRK_TIMING=1: allocate the stamp buffer outside the capture

    // lanes: lane 0 is this model's own frame; lanes 1.. are clones sharing the weight handles
    while (m->lanes.size() + 1 < groups) {
        wrk_v7_model* lane = nullptr;
        const int32_t rc = wrk_v7_model_create(ctx, &m->d, &lane);
        if (rc != WRK_OK) return rc;
        lane->act_dtype = m->act_dtype;
        m->lanes.push_back(lane);
    }
    while (m->lane_streams.size() < groups) {
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        WRK_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        WRK_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->lane_streams.push_back(s);
        m->lane_events.push_back(e);
    }
    struct Lane { wrk_v7_model* mdl; uint32_t b0, nb; wrk_program* prog; };
    std::vector<Lane> L(groups);
    for (uint32_t g = 0; g < groups; ++g) {
        L[g].mdl = g == 0 ? m : m->lanes[g - 1];
        // the persistent engine needs every CU for itself: two of them side by side (one per lane) would each hold part of the chip
        // and wait for the rest forever (until their bounded spins give up) -- concurrent pipelines keep the five-launch layer
        L[g].mdl->engine_blocked = groups > 1;
        L[g].b0 = (uint32_t)((uint64_t)B * g / groups);
        L[g].nb = (uint32_t)((uint64_t)B * (g + 1) / groups) - L[g].b0;
        const int32_t rc = greedy_prepare(ctx, L[g].mdl, st, first_tokens + L[g].b0, L[g].b0, L[g].nb, steps, mode, eager, &L[g].prog);
        if (rc != WRK_OK) return rc;
    }
    // every early return below leaves through this guard: the timing events are destroyed and, after an error, the lane streams are
    // drained (a lane's queued step programs must not outlive a frame that the next call may reallocate) -- ADVICE r02
    struct Guard {
        wrk_v7_model* m; hipEvent_t e0 = nullptr, e1 = nullptr; bool ok = false;
        ~Guard() {
            if (!ok) { for (hipStream_t ls : m->lane_streams) hipStreamSynchronize(ls); hipStreamSynchronize(m->ctx->stream); }
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
        }
    } guard{m};
    WRK_HIP(ctx, hipEventCreate(&guard.e0));
    WRK_HIP(ctx, hipEventCreate(&guard.e1));
    hipEvent_t e0 = guard.e0, e1 = guard.e1;
    WRK_HIP(ctx, hipEventRecord(e0, ctx->stream));
    if (groups == 1) {
        for (uint32_t i = 0; i < steps; ++i) {
            if (eager) {
                int32_t rc;
                if (mode == 1 && m->act_dtype == WRK_F16) rc = m->enqueue_fused_decode(st, B, B, true, true, true, true, 0, true);
                else {
                    wrk::gather_rows_f16(ctx->op_stream(), m->emb->ptr, m->s.tokens, m->s.input, m->d.num_emb, B);
                    rc = m->enqueue_ops(st, B, B, true);
                    if (rc == WRK_OK) {
                        wrk::argmax_rows(ctx->op_stream(), m->s.head_o, V, V, B, m->s.argmax);
                        wrk::advance_tokens(ctx->op_stream(), m->s.argmax, m->s.tokens, m->history, m->s.counter, B);
                    }
                }
                if (rc != WRK_OK) return rc;
            } else WRK_HIP(ctx, hipGraphLaunch(L[0].prog->exec, ctx->stream));
        }
    } else {
        // every lane replays its own step program on its own stream; the lanes start together behind e0 and the submission
        // stream joins them all before e1
        for (uint32_t g = 0; g < groups; ++g) WRK_HIP(ctx, hipStreamWaitEvent(m->lane_streams[g], e0, 0));
        for (uint32_t i = 0; i < steps; ++i)
            for (uint32_t g = 0; g < groups; ++g) WRK_HIP(ctx, hipGraphLaunch(L[g].prog->exec, m->lane_streams[g]));
        for (uint32_t g = 0; g < groups; ++g) {
            WRK_HIP(ctx, hipEventRecord(m->lane_events[g], m->lane_streams[g]));
            WRK_HIP(ctx, hipStreamWaitEvent(ctx->stream, m->lane_events[g], 0));
        }
    }
    WRK_HIP(ctx, hipEventRecord(e1, ctx->stream));
    WRK_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.0f;
    WRK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (elapsed_ms) *elapsed_ms = ms;
    for (uint32_t g = 0; g < groups; ++g) {
        const Lane& ln = L[g];
        if (out_tokens) {
            if (groups == 1) WRK_HIP(ctx, hipMemcpyAsync(out_tokens, ln.mdl->history, (size_t)steps * B * 4, hipMemcpyDeviceToHost, ctx->stream));
            else WRK_HIP(ctx, hipMemcpy2DAsync(out_tokens + ln.b0, (size_t)B * 4, ln.mdl->history, (size_t)ln.nb * 4, (size_t)ln.nb * 4, steps,
                                               hipMemcpyDeviceToHost, ctx->stream));
        }
        if (last_logits) WRK_HIP(ctx, hipMemcpyAsync(last_logits + (size_t)ln.b0 * V, ln.mdl->s.head_o, (size_t)ln.nb * V * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    wrk::timing_report(ctx);
    for (uint32_t g = 0; g < groups; ++g) {
        wrk_v7_engine_report(L[g].mdl->engine);
        const int32_t rc = wrk_v7_engine_check(L[g].mdl->engine);
        if (rc != WRK_OK) return rc;
    }
    guard.ok = true;
    return WRK_OK;
}

}  // extern "C"

// Persistent decode engine: built once, outside captures.  WRK_ENGINE=0 keeps the five-launch layer (read per call, part of the
// graph key: tests compare the two paths in one process).
static bool engine_env_on() { const char* e = getenv("WRK_ENGINE"); return !(e && e[0] == '0'); }
bool engine_env_on_public() { return engine_env_on(); }
bool wrk_v7_model::engine_on() const { return engine != nullptr && !engine_blocked && engine_env_on() && act_dtype == WRK_F16; }
int32_t wrk_v7_model::ensure_engine() {
    if (engine_tried || !engine_env_on() || act_dtype != WRK_F16) return WRK_OK;
    if (ctx->capturing_here()) return WRK_OK;       // allocations are not capturable: the caller's program keeps the launches
    engine_tried = true;
    const std::string keep = ctx->err;
    const int32_t rc = wrk_v7_engine_create(this, &engine);
    if (rc != WRK_OK) { engine = nullptr; engine_why = ctx->err; ctx->err = keep; }
    if (rc != WRK_OK && rc != WRK_E_UNSUPPORTED) return rc;
    return WRK_OK;
}

int32_t wrk_v7_model::ensure_history(size_t n) {
    if (n <= history_cap && history) return WRK_OK;
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_graphs();
    if (history) hipFree(history);
    history = nullptr;
    WRK_HIP(ctx, hipMalloc((void**)&history, n * 4 + 256));
    history_cap = n;
    return WRK_OK;
}

int32_t wrk_buf_write_raw(wrk_ctx* ctx, void* dst, const void* src, size_t bytes) {
    wrk_buf tmp{ctx, dst, bytes, {1}};
    return wrk_buf_write(ctx, &tmp, 0, src, bytes);
}
