// RWKV-6 model runner behind wrk_v6_* (include/wrk_hip.h): the TensorOp list of v6::Bundle::dispatch
// (src/runtime/v6.rs:590-699, dispatch_layer :701-958, dispatch_header :960-993), one kernel per reference op,
// matrices through the inline-dequant matvec / MFMA GEMM.  Buffer dtypes follow Runtime<f16> (v6.rs:265-300):
// att_k, att_v, att_r and time_decay are f32, everything else f16.
#include "wrk_internal.h"
#include "wrk_v7.h"

#define LOCK(ctx) std::lock_guard<std::recursive_mutex> _lk((ctx)->mu)

static constexpr float LN_EPS = 1.0e-5f;    // v6.rs:46
static constexpr float GN_EPS = 64.0e-5f;   // v6.rs:47

struct V6Scratch {
    void *input, *x, *aux_x, *att_x, *att_xx, *att_sx, *att_w, *att_g, *att_o, *tmx, *tmt, *tm, *ffn_x, *ffn_kx, *ffn_rx, *ffn_k, *ffn_v, *ffn_r, *head_x;
    float *att_k, *att_v, *att_r, *time_decay, *head_o;
    uint32_t *cursors, *tokens, *headers, *argmax, *counter;
    float* ks_part; uint32_t* ks_cnt; size_t ks_part_cap; uint32_t ks_cnt_cap;     // K-sliced GEMM scratch (2 .. 32 sequences), see MatJob
};

struct wrk_v6_model {
    wrk_ctx* ctx = nullptr;
    wrk_v6_model_desc d{};
    std::vector<wrk_v6_layer_desc> layers;
    void* scratch = nullptr;
    uint32_t scratch_tokens = 0, scratch_headers = 0;
    V6Scratch s{};
    uint32_t* history = nullptr;
    size_t history_cap = 0;
    uint32_t wkv_nseq = 0;      // sequences of the job being enqueued (0: unknown): picks the WKV chunk kernel (wrk::time_mix_v6)
    std::map<std::tuple<const void*, uint32_t, uint32_t>, wrk_program*> graphs;      // (state, sequences, mode)

    void drop_graphs() { for (auto& kv : graphs) wrk_program_destroy(kv.second); graphs.clear(); }
    int32_t ensure_scratch(uint32_t T, uint32_t NH);
    int32_t ensure_history(size_t n);
    int32_t enqueue_ops(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity, bool merged = false);
    int32_t enqueue_fused_decode(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity, uint32_t batch0);
};

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

int32_t wrk_v6_model::ensure_scratch(uint32_t T, uint32_t NH) {
    if (T <= scratch_tokens && NH <= scratch_headers && scratch) return WRK_OK;
    if (ctx->capturing_here()) return wrk_fail(ctx, WRK_E_ARG, "scratch must be sized before capture");
    const uint32_t nt = T > scratch_tokens ? T : scratch_tokens, nh = NH > scratch_headers ? NH : scratch_headers;
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_graphs();
    if (scratch) hipFree(scratch);
    scratch = nullptr;
    const size_t D = d.num_emb, F = d.num_hidden, V = d.num_vocab, R = d.time_mix, W = d.time_decay;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up256(bytes); return o; };
    const size_t v16 = D * nt * 2, v32 = D * nt * 4;
    const size_t o_input = take(v16), o_x = take(v16), o_aux = take(v16), o_attx = take(v16), o_attxx = take(v16), o_sx = take(v16 * 5);
    const size_t o_w = take(W * nt * 2), o_g = take(v16), o_o = take(v16), o_tmx = take(R * 5 * nt * 2), o_tmt = take(R * 5 * nt * 2), o_tm = take(v16 * 5);
    const size_t o_fx = take(v16), o_fkx = take(v16), o_frx = take(v16), o_fk = take(F * nt * 2), o_fv = take(v16), o_fr = take(v16);
    const size_t o_k = take(v32), o_v = take(v32), o_r = take(v32), o_td = take(v32);
    const size_t o_hx = take(D * nh * 2), o_ho = take(V * nh * 4);
    const size_t o_cur = take((size_t)nt * 4), o_tok = take((size_t)nt * 4), o_hdr = take((size_t)nh * 4), o_arg = take((size_t)nh * 4), o_cnt = take(256);
    // K-sliced GEMM (decode batches; round 3, as wrk_v7_model::ensure_scratch): f32 partial tiles [row group][K slice][token][64 rows] of the
    // largest launch of a layer (at most one slice per 256-block) and one arrival counter per row group
    size_t ks_floats = 0, ks_groups = 0, o_ksp = 0, o_ksc = 0;
    if (nt >= 2) {
        const size_t ntp = nt <= 16 ? 16 : (nt <= 32 ? 32 : 64);
        auto tiles = [](size_t m, size_t k) { return ((m + 63) / 64) * (k / 256 ? k / 256 : 1); };
        auto groups = [](size_t m) { return (m + 63) / 64; };
        const size_t k3 = 4 * tiles(D, D) + tiles(W, D), k6 = tiles(F, D) + tiles(D, D), k7 = tiles(D, F), k1 = tiles(5 * R, D);
        const size_t most = std::max(std::max(k3, k6), std::max(k7, k1));
        ks_floats = most * 64 * ntp;
        ks_groups = 4 * groups(D) + groups(W) + groups(F) + groups(5 * R) + 8;
        o_ksp = take(ks_floats * 4);
        o_ksc = take(ks_groups * 4);
    }
    WRK_HIP(ctx, hipMalloc(&scratch, off));
    WRK_HIP(ctx, hipMemsetAsync(scratch, 0, off, ctx->stream));
    if (nt >= 128) {
        // prefill GEMM (wrk_gemm3.hip, Q4_K / Q5_K): sub-block input sums of the stacked tokens for the up to four distinct inputs of a launch (the
        // k, v, r, g projections read four shifted inputs) or the F-wide ffn vector, + the f32 partial tiles of K-split launches (chunks <= 256 tokens)
        const size_t widest = std::max<size_t>(4 * D, F);
        const int32_t rs = wrk_ctx_reserve_gemm_scratch(ctx, (size_t)nt * (widest / 32) * 4 + 16 * 1024 + (size_t)256 * widest * 4 * 4);
        if (rs != WRK_OK) return rs;
    }
    char* b = (char*)scratch;
    s.ks_part = ks_floats ? (float*)(b + o_ksp) : nullptr; s.ks_cnt = ks_floats ? (uint32_t*)(b + o_ksc) : nullptr;
    s.ks_part_cap = ks_floats; s.ks_cnt_cap = (uint32_t)ks_groups;
    s.input = b + o_input; s.x = b + o_x; s.aux_x = b + o_aux; s.att_x = b + o_attx; s.att_xx = b + o_attxx; s.att_sx = b + o_sx;
    s.att_w = b + o_w; s.att_g = b + o_g; s.att_o = b + o_o; s.tmx = b + o_tmx; s.tmt = b + o_tmt; s.tm = b + o_tm;
    s.ffn_x = b + o_fx; s.ffn_kx = b + o_fkx; s.ffn_rx = b + o_frx; s.ffn_k = b + o_fk; s.ffn_v = b + o_fv; s.ffn_r = b + o_fr;
    s.att_k = (float*)(b + o_k); s.att_v = (float*)(b + o_v); s.att_r = (float*)(b + o_r); s.time_decay = (float*)(b + o_td);
    s.head_x = b + o_hx; s.head_o = (float*)(b + o_ho);
    s.cursors = (uint32_t*)(b + o_cur); s.tokens = (uint32_t*)(b + o_tok); s.headers = (uint32_t*)(b + o_hdr); s.argmax = (uint32_t*)(b + o_arg);
    s.counter = (uint32_t*)(b + o_cnt);
    scratch_tokens = nt; scratch_headers = nh;
    return WRK_OK;
}

int32_t wrk_v6_model::ensure_history(size_t n) {
    if (n <= history_cap && history) return WRK_OK;
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_graphs();
    if (history) hipFree(history);
    history = nullptr;
    WRK_HIP(ctx, hipMalloc((void**)&history, n * 4 + 256));
    history_cap = n;
    return WRK_OK;
}

static int32_t mm6(wrk_ctx* ctx, const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    wrk::MatJob j{m->data, m->aux, m->kind, m->flags, m->k, m->m, (uint32_t)m->row_bytes, in, out, act, 0};
    j.scale = m->out_scale;
    j.xsum = ctx->gemm_scratch; j.xsum_cap = ctx->gemm_scratch_cap;        // third-generation prefill tile (Q4_K / Q5_K chunks)
    int rc = -2;
    if (in.shape[1] * in.shape[2] >= wrk::gemm_min_tokens()) rc = wrk::matmul_mfma(ctx->op_stream(), j, ctx->num_cu);
    if (rc == -2) rc = wrk::matvec(ctx->op_stream(), &j, 1, ctx->num_cu);
    if (rc != 0) return wrk_fail(ctx, WRK_E_ARG, "matmul launch rejected (K=%u M=%u rc=%d)", m->k, m->m, rc);
    return WRK_OK;
}
#define MM(...) do { int32_t _r = mm6(ctx, __VA_ARGS__); if (_r != WRK_OK) return _r; } while (0)
static wrk::MatJob job6m(const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    wrk::MatJob j{m->data, m->aux, m->kind, m->flags, m->k, m->m, (uint32_t)m->row_bytes, in, out, act, 0};
    j.scale = m->out_scale;
    return j;
}
// several matrices x the same token count in one MFMA launch per kernel family; per-matrix launches when the GEMM declines
static int32_t mm6_group(wrk_ctx* ctx, wrk::MatJob* jobs, int n) {
    const uint32_t T = jobs[0].in.shape[1] * jobs[0].in.shape[2];
    for (int i = 0; i < n; ++i) { jobs[i].xsum = ctx->gemm_scratch; jobs[i].xsum_cap = ctx->gemm_scratch_cap; }
    if (T >= wrk::gemm_min_tokens() && wrk::matmul_mfma_multi(ctx->op_stream(), jobs, n, ctx->num_cu) == 0) return WRK_OK;
    for (int i = 0; i < n; ++i) {
        int rc = -2;
        if (T >= wrk::gemm_min_tokens()) rc = wrk::matmul_mfma(ctx->op_stream(), jobs[i], ctx->num_cu);
        if (rc == -2) rc = wrk::matvec(ctx->op_stream(), &jobs[i], 1, ctx->num_cu);
        if (rc != 0) return wrk_fail(ctx, WRK_E_ARG, "matmul launch rejected (K=%u M=%u rc=%d)", jobs[i].k, jobs[i].m, rc);
    }
    return WRK_OK;
}
#define MMG(jobs, n) do { int32_t _r = mm6_group(ctx, jobs, n); if (_r != WRK_OK) return _r; } while (0)

int32_t wrk_v6_model::enqueue_ops(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity, bool merged) {
    hipStream_t q = ctx->op_stream();
    const uint32_t D = d.num_emb, F = d.num_hidden, H = d.num_head, S = D / H, V = d.num_vocab, R = d.time_mix, W = d.time_decay;
    auto vec = [&](void* p, uint32_t c = 0, uint32_t dt = WRK_F16) { return make_dense(p, dt, c ? c : D, T); };
    auto heads = [&](void* p, uint32_t dt = WRK_F16) { return make_dense(p, dt, S, H, T); };
    auto bvec = [&](const wrk_buf* b) { return make_dense(b->ptr, WRK_F16, D, 1, 1); };
    DTensor x = vec(s.x), att_x = vec(s.att_x), att_xx = vec(s.att_xx), aux_x = vec(s.aux_x), att_g = vec(s.att_g), att_o = vec(s.att_o);
    DTensor att_k = vec(s.att_k, 0, WRK_F32), att_v = vec(s.att_v, 0, WRK_F32), att_r = vec(s.att_r, 0, WRK_F32), tdec = vec(s.time_decay, 0, WRK_F32);
    DTensor att_w = vec(s.att_w, W), ffn_x = vec(s.ffn_x), ffn_kx = vec(s.ffn_kx), ffn_rx = vec(s.ffn_rx), ffn_k = vec(s.ffn_k, F), ffn_v = vec(s.ffn_v), ffn_r = vec(s.ffn_r);
    DTensor sx5 = make_dense(s.att_sx, WRK_F16, D, T, 5), tm5 = make_dense(s.tm, WRK_F16, D, T, 5);
    auto slice = [&](DTensor t, uint32_t i) { t.shape[2] = 1; t.offset[2] = i; return t; };

    DTensor input = vec(s.input);
    wrk::layer_norm(q, d.ln0_w->ptr, d.ln0_b->ptr, input, LN_EPS);
    wrk::blit(q, input, x);
    for (uint32_t li = 0; li < d.num_layer; ++li) {
        const wrk_v6_layer_desc& L = layers[li];
        DTensor st_att = make_dense(st->layer_ptr(li), WRK_F32, D, S + 2, st->num_batch);
        st_att.shape[1] = S + 1;
        DTensor st_row0 = st_att; st_row0.shape[1] = 1;
        DTensor st_ffn = make_dense(st->layer_ptr(li), WRK_F32, D, S + 2, st->num_batch);
        st_ffn.shape[1] = 1; st_ffn.offset[1] = S + 1;

        // merged (mode 1, multi-token chunks): the same arithmetic in fewer launches -- blit + LN in one pass, the projections that
        // are ready together in one MFMA launch, W_o's add in the epilogue; results are bit-identical to the op list
        if (merged) wrk::layer_norm_from(q, L.ln1_w->ptr, L.ln1_b->ptr, x, att_x, LN_EPS);
        else {
            wrk::blit(q, x, att_x);
            wrk::layer_norm(q, L.ln1_w->ptr, L.ln1_b->ptr, att_x, LN_EPS);
        }
        wrk::token_shift(q, s.cursors, bvec(L.time_mix_x), st_row0, att_x, att_xx, 1);
        MM(L.time_mix_w1, att_xx, make_dense(s.tmx, WRK_F16, R * 5, T), WRK_ACT_TANH);                     // time_mix_x [R, 5, T] seen as [5R, T]
        wrk::transpose(q, make_dense(s.tmx, WRK_F16, R, 5, T), make_dense(s.tmt, WRK_F16, R, T, 5));
        if (merged) {
            wrk::MatJob jw[5];
            for (uint32_t i = 0; i < 5; ++i) jw[i] = job6m(L.time_mix_w2[i], slice(make_dense(s.tmt, WRK_F16, R, T, 5), i), slice(tm5, i), WRK_ACT_NONE);
            MMG(jw, 5);
        } else
            for (uint32_t i = 0; i < 5; ++i)                                                              // batched [R, D, 5] matmul
                MM(L.time_mix_w2[i], slice(make_dense(s.tmt, WRK_F16, R, T, 5), i), slice(tm5, i), WRK_ACT_NONE);
        wrk::binary(q, 0, make_dense(L.time_mix->ptr, WRK_F16, D, 1, 5), tm5, 0, 0, 0);                    // add(time_mix, buffer.time_mix)
        wrk::token_shift(q, s.cursors, tm5, st_row0, att_x, sx5, 1);
        if (merged) {
            wrk::MatJob jp[5] = {job6m(L.w_k, slice(sx5, 1), att_k, WRK_ACT_NONE), job6m(L.w_v, slice(sx5, 2), att_v, WRK_ACT_NONE),
                                 job6m(L.w_r, slice(sx5, 3), att_r, WRK_ACT_NONE), job6m(L.w_g, slice(sx5, 4), att_g, WRK_ACT_NONE),
                                 job6m(L.time_decay_w1, slice(sx5, 0), att_w, WRK_ACT_TANH)};
            MMG(jp, 5);
        } else {
            MM(L.w_k, slice(sx5, 1), att_k, WRK_ACT_NONE);
            MM(L.w_v, slice(sx5, 2), att_v, WRK_ACT_NONE);
            MM(L.w_r, slice(sx5, 3), att_r, WRK_ACT_NONE);
            MM(L.w_g, slice(sx5, 4), att_g, WRK_ACT_NONE);
            MM(L.time_decay_w1, slice(sx5, 0), att_w, WRK_ACT_TANH);
        }
        MM(L.time_decay_w2, att_w, tdec, WRK_ACT_NONE);
        wrk::binary(q, 0, bvec(L.time_decay), tdec, 0, 0, 0);
        wrk::activate(q, tdec, WRK_ACT_STABLE_EXP);
        wrk::blit(q, att_x, aux_x);
        wrk::time_mix_v6(q, s.cursors, heads(s.time_decay, WRK_F32), L.time_first->ptr, st_att, heads(s.att_k, WRK_F32), heads(s.att_v, WRK_F32),
                         heads(s.att_r, WRK_F32), heads(s.aux_x), wkv_nseq);
        wrk::group_norm(q, L.gn_w->ptr, L.gn_b->ptr, heads(s.aux_x), GN_EPS);
        wrk::blit(q, aux_x, att_x);
        wrk::binary(q, 1, att_g, att_x, WRK_ACT_SILU, 0, 0);                                              // mul_activate(att_g Silu, att_x)
        if (merged) {
            wrk::MatJob jo = job6m(L.w_o, att_x, x, WRK_ACT_NONE);       // x = round(W_o att_x) + x
            jo.has_res = 1;
            jo.res = x;
            MMG(&jo, 1);
            wrk::layer_norm_from(q, L.ln2_w->ptr, L.ln2_b->ptr, x, ffn_x, LN_EPS);
            const DTensor fm[2] = {bvec(L.ffn_mix_k), bvec(L.ffn_mix_r)}, fo[2] = {ffn_kx, ffn_rx};
            wrk::token_shift_multi(q, s.cursors, fm, fo, 2, st_ffn, ffn_x, 1);
            wrk::MatJob jf[2] = {job6m(L.ffn_w_k, ffn_kx, ffn_k, WRK_ACT_SQUARED_RELU), job6m(L.ffn_w_r, ffn_rx, ffn_r, WRK_ACT_NONE)};
            MMG(jf, 2);
            MM(L.ffn_w_v, ffn_k, ffn_v, WRK_ACT_NONE);
        } else {
            MM(L.w_o, att_x, att_o, WRK_ACT_NONE);
            wrk::binary(q, 0, att_o, x, 0, 0, 0);

            wrk::blit(q, x, ffn_x);
            wrk::layer_norm(q, L.ln2_w->ptr, L.ln2_b->ptr, ffn_x, LN_EPS);
            wrk::token_shift(q, s.cursors, bvec(L.ffn_mix_k), st_ffn, ffn_x, ffn_kx, 1);
            wrk::token_shift(q, s.cursors, bvec(L.ffn_mix_r), st_ffn, ffn_x, ffn_rx, 1);
            MM(L.ffn_w_k, ffn_kx, ffn_k, WRK_ACT_SQUARED_RELU);
            MM(L.ffn_w_v, ffn_k, ffn_v, WRK_ACT_NONE);
            MM(L.ffn_w_r, ffn_rx, ffn_r, WRK_ACT_NONE);
        }
        wrk::channel_mix_v6(q, s.cursors, st_ffn, ffn_r, ffn_v, ffn_x);
        wrk::binary(q, 0, ffn_x, x, 0, 0, 0);
        if ((li + 1) % d.rescale == 0) wrk::affine(q, x, 0.5f, 0.0f);                                     // v6.rs:953-955
    }
    if (NH > 0) {
        DTensor head_x = make_dense(s.head_x, WRK_F16, D, NH);
        if (identity) wrk::blit(q, make_dense(s.x, WRK_F16, D, NH), head_x);
        else wrk::gather_rows_any(q, x, s.headers, head_x, NH);
        wrk::layer_norm(q, d.ln_out_w->ptr, d.ln_out_b->ptr, head_x, LN_EPS);
        MM(d.head, head_x, make_dense(s.head_o, WRK_F32, V, NH), WRK_ACT_NONE);
    }
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

// ====================================================================== fused RWKV-6 decode (one token per sequence)
// 7 launches per layer instead of ~45 (v6.rs:701-958 restated stage by stage; every f16 store of the reference's
// Runtime<f16> buffers is reproduced as an explicit rounding):
//   K1  LN1 + token_shift(time_mix_x) [prologue] -> time_mix_w1 matvec (tanh)                       -> tmx [5R]
//   K2  v6_mix_kernel: five LoRA up-projections + add(time_mix) + 5-way data-dependent token shift   -> sx5 [5][D]
//   K3  one launch: w_k, w_v, w_r (f32 out), w_g, time_decay_w1 (tanh), each on its own shifted input
//   K4  v6_head_kernel: decay LoRA up + stable_exp, WKV6 with the state in registers, group norm, SiLU gate, state carry
//   K5  w_o + residual
//   K6  LN2 + two token shifts [prologue] -> ffn key (relu^2), ffn receptance
//   K7  ffn value, sigmoid(receptance) gate, residual, ffn shift-state carry  [epilogue]
namespace wrk {

struct V6MixParams {
    uint32_t d, r;                          // D, time_mix rank R
    const f16* w2[5]; uint32_t w2_rb;       // five F16 matrices [D][R] (device rows of w2_rb bytes)
    const f16* tmx;                         // [T][5R] tanh(time_mix_w1 . shifted x)
    const f16* time_mix;                    // [5][D]
    const f16* x_ln;                        // [T][D]  LN1(x)
    const float* state;                     // layer state base; att shift row of batch b at state[b * (S+2) * D]
    const uint32_t* cursors;
    uint32_t batch1;                        // batch + 1 when the host knows it, else 0
    f16* sx5;                               // [5][T][D]
    uint32_t T;
};

// thread (row = tid >> 2, part = tid & 3): 4 lanes share a channel's five R-long dot products
__global__ void __launch_bounds__(256) v6_mix_kernel(const V6MixParams P) {
    const uint32_t tid = threadIdx.x, row = tid >> 2, part = tid & 3u, t = blockIdx.y;
    const uint32_t c = blockIdx.x * 64 + row, D = P.d, R = P.r;
    if (c >= D) return;
    const f16* aux = P.tmx + (size_t)t * 5 * R;
    f16x8 w[5][4], xx[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const uint32_t k = part * 8 + 32 * n;
            if (k < R) { w[i][n] = *(const f16x8*)((const uint8_t*)P.w2[i] + (size_t)c * P.w2_rb + k * 2); xx[i][n] = *(const f16x8*)(aux + i * R + k); }
        }
    const float xl = (float)P.x_ln[(size_t)t * D + c];
    const uint32_t batch = P.batch1 ? P.batch1 - 1 : (P.cursors[t] & 0xffu);
    const float prev = P.state[(size_t)batch * 66 * D + c];
    float tmv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) tmv[i] = (float)P.time_mix[(size_t)i * D + c];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (part * 8 + 32 * n < R) {
                acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[i][n], w[i][n], 0, 1), __builtin_shufflevector(xx[i][n], xx[i][n], 0, 1), acc, false);
                acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[i][n], w[i][n], 2, 3), __builtin_shufflevector(xx[i][n], xx[i][n], 2, 3), acc, false);
                acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[i][n], w[i][n], 4, 5), __builtin_shufflevector(xx[i][n], xx[i][n], 4, 5), acc, false);
                acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[i][n], w[i][n], 6, 7), __builtin_shufflevector(xx[i][n], xx[i][n], 6, 7), acc, false);
            }
        }
        acc += dpp_f32<0xB1>(acc);
        acc += dpp_f32<0x4E>(acc);
        if (part == 0) {
            const float f = r16(r16(acc) + tmv[i]);                                        // matmul (f16 store) + add(time_mix)
            P.sx5[((size_t)i * P.T + t) * D + c] = (f16)wgsl_mix(xl, prev, f);             // token_shift(.., reversed)
        }
    }
}

struct V6HeadParams {
    uint32_t d, w;                          // D, time_decay rank W
    const f16* dw2; uint32_t dw2_rb;        // time_decay_w2 F16 [D][W]
    const f16* att_w;                       // [T][W] tanh(time_decay_w1 . sx_w)
    const f16* time_decay;                  // [D]
    const float* u;                         // time_first f32 [D]
    const float *k, *v, *r;                 // [T][D] f32
    const f16* g;                           // [T][D]
    const f16 *gn_w, *gn_b;
    float* state;                           // layer state base
    const uint32_t* cursors;
    uint32_t batch1;
    const f16* shift_src;                   // LN1(x) [T][D]: becomes the att shift state
    f16* out;                               // [T][D]
    float gn_eps;
};

// One workgroup per (head, token).  Thread (i = tid >> 2, part = tid & 3) owns S[16 part .. +15][i] (quad layout: the
// reduction over j is in-register + two quad shuffles).  The 64 decay values of the head are produced cooperatively
// (thread (row, part) = one quarter of a W-long dot product) and shared through LDS with k, r, u.
__global__ void __launch_bounds__(256) v6_head_kernel(const V6HeadParams P) {
    constexpr int S = 64;
    __shared__ __attribute__((aligned(16))) float sh_w[S], sh_k[S], sh_r[S], sh_u[S], sh_y[S];
    const uint32_t head = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, D = P.d, W = P.w;
    const uint32_t c0 = head * S, row = tid >> 2, part = tid & 3u, i = row, ch = c0 + row;
    // decay LoRA up-projection of channel ch (a key channel j = row of this head)
    f16x8 w[4], xx[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const uint32_t k = part * 8 + 32 * n;
        if (k < W) { w[n] = *(const f16x8*)((const uint8_t*)P.dw2 + (size_t)ch * P.dw2_rb + k * 2); xx[n] = *(const f16x8*)(P.att_w + (size_t)t * W + k); }
    }
    const float td0 = (float)P.time_decay[ch];
    const float kj = P.k[(size_t)t * D + ch], rj = P.r[(size_t)t * D + ch], uj = P.u[ch];
    const float vv = P.v[(size_t)t * D + ch];                   // as value column i = row
    const float gg = (float)P.g[(size_t)t * D + ch];
    const float gnw = (float)P.gn_w[ch], gnb = (float)P.gn_b[ch];
    float shift = 0.0f;
    if (part == 1) shift = (float)P.shift_src[(size_t)t * D + ch];
    const uint32_t batch = P.batch1 ? P.batch1 - 1 : (P.cursors[t] & 0xffu);
    float* st = P.state + ((size_t)batch * (S + 2) + 1) * D + c0 + i;
    float Sreg[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Sreg[jj] = st[(size_t)(part * 16 + jj) * D];
    float acc = 0.0f;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (part * 8 + 32 * n < W) {
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[n], w[n], 0, 1), __builtin_shufflevector(xx[n], xx[n], 0, 1), acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[n], w[n], 2, 3), __builtin_shufflevector(xx[n], xx[n], 2, 3), acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[n], w[n], 4, 5), __builtin_shufflevector(xx[n], xx[n], 4, 5), acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w[n], w[n], 6, 7), __builtin_shufflevector(xx[n], xx[n], 6, 7), acc, false);
        }
    }
    acc += dpp_f32<0xB1>(acc);
    acc += dpp_f32<0x4E>(acc);
    if (part == 0) {
        // time_decay buffer is f32: matmul result unrounded, + time_decay, then activate(StableExp) = exp(-exp(x))
        sh_w[row] = __expf(-__expf(acc + td0));
        sh_k[row] = kj; sh_r[row] = rj; sh_u[row] = uj;
    }
    if (part == 1) P.state[(size_t)batch * (S + 2) * D + ch] = shift;          // att shift-state carry
    __syncthreads();
    float y = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = part * 16 + jj;
        const float kv = sh_k[j] * vv;
        y += sh_r[j] * __builtin_fmaf(sh_u[j], kv, Sreg[jj]);
        const float sn = __builtin_fmaf(sh_w[j], Sreg[jj], kv);
        st[(size_t)j * D] = sn;
    }
    y = y + dpp_f32<0xB1>(y);
    y = y + dpp_f32<0x4E>(y);
    if (part == 0) sh_y[i] = r16(y);                                            // aux_x (f16 store)
    __syncthreads();
    if (tid < S) {      // group norm over the head, SiLU gate: thread = column
        const float yv = sh_y[tid];
        const float mean = wave_sum(yv) * (1.0f / S);
        const float dl = yv - mean;
        const float var = wave_sum(dl * dl) * (1.0f / S) + P.gn_eps;
        // the per-column operands were loaded by thread (row = tid, part 0) = thread 4 * tid: re-load for this mapping
        const uint32_t c = c0 + tid;
        const float o = r16(__builtin_fmaf(dl * (1.0f / sqrtf(var)), (float)P.gn_w[c], (float)P.gn_b[c]));
        const float gv = (float)P.g[(size_t)t * D + c];
        P.out[(size_t)t * D + c] = (f16)((gv / (1.0f + __expf(-gv))) * o);       // mul_activate(att_g Silu, att_x)
    }
    (void)gg; (void)gnw; (void)gnb;
}

}  // namespace wrk

static wrk::MatJob job6(const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    wrk::MatJob j{m->data, m->aux, m->kind, m->flags, m->k, m->m, (uint32_t)m->row_bytes, in, out, act, 0};
    j.scale = m->out_scale;
    return j;
}

// One decode step for T stacked tokens, each its own sequence.  Returns WRK_E_UNSUPPORTED (without launching anything)
// when the model's shapes are outside the fused kernels' range; callers then use enqueue_ops.
int32_t wrk_v6_model::enqueue_fused_decode(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity, uint32_t batch0) {
    hipStream_t q = ctx->op_stream();
    const uint32_t D = d.num_emb, F = d.num_hidden, H = d.num_head, S = 64, V = d.num_vocab, R = d.time_mix, W = d.time_decay;
    if (R > 128 || W > 128 || (R & 7u) || (W & 7u) || (D & 7u) || D > 8192) return WRK_E_UNSUPPORTED;
    for (auto& L : layers) {
        const wrk_matrix* f16m[] = {L.time_mix_w2[0], L.time_mix_w2[1], L.time_mix_w2[2], L.time_mix_w2[3], L.time_mix_w2[4], L.time_decay_w2};
        for (const wrk_matrix* m : f16m) if (m->kind != WRK_MAT_F16) return WRK_E_UNSUPPORTED;
        if (L.time_mix_w2[1]->row_bytes != L.time_mix_w2[0]->row_bytes) return WRK_E_UNSUPPORTED;
    }
    auto vec = [&](void* p, uint32_t c = 0, uint32_t dt = WRK_F16) { return make_dense(p, dt, c ? c : D, T); };
    auto run_jobs = [&](wrk::MatJob* jobs, int n) -> int {
        if (T >= wrk::gemm_min_tokens()) {
            jobs[0].ks_part = s.ks_part; jobs[0].ks_cnt = s.ks_cnt; jobs[0].ks_part_cap = s.ks_part_cap; jobs[0].ks_cnt_cap = s.ks_cnt_cap;
            if (wrk::matmul_mfma_multi(q, jobs, n, ctx->num_cu) == 0) return 0;
            for (int i = 0; i < n; ++i) {
                int rc = wrk::matmul_mfma(q, jobs[i], ctx->num_cu);
                if (rc == -2) rc = wrk::matvec(q, &jobs[i], 1, ctx->num_cu);
                if (rc != 0) return rc;
            }
            return 0;
        }
        return wrk::matvec_grouped(q, jobs, n, ctx->num_cu);
    };
#define LNMIX(P, n) do { if (wrk::ln_mix(q, P, n) != 0) return wrk_fail(ctx, WRK_E_UNSUPPORTED, "ln_mix shape"); } while (0)
    // arrival counters of the K-sliced GEMM: zero at the head of every step (a launch that aborted must not poison the next replay)
    if (T >= 2 && s.ks_cnt && s.ks_cnt_cap) WRK_HIP(ctx, hipMemsetAsync(s.ks_cnt, 0, (size_t)s.ks_cnt_cap * 4, q));
    {   // embedding rows (already gathered into s.input) -> LN0 -> x
        wrk::LnMixParams P{};
        P.src = (const f16*)s.input; P.ln_w = (const f16*)d.ln0_w->ptr; P.ln_b = (const f16*)d.ln0_b->ptr; P.eps = LN_EPS;
        P.d = D; P.nmix = 0; P.ln_out = (f16*)s.x;
        LNMIX(P, T);
    }
    const size_t sxs = (size_t)T * D;                                   // elements between the five shifted inputs
    for (uint32_t li = 0; li < d.num_layer; ++li) {
        const wrk_v6_layer_desc& L = layers[li];
        float* lst = st->layer_ptr(li);
        float* row0 = lst + (size_t)batch0 * (S + 2) * D;               // att shift state (T == 1: of that sequence)
        float* rowf = lst + ((size_t)batch0 * (S + 2) + (S + 1)) * D;   // ffn shift state
        // per layer: can the two LN prologues / the gated epilogue ride the register-input matvec kernels?
        bool single = (T == 1) && D <= 4096;
        if (single) {
            wrk::MatJob a = job6(L.time_mix_w1, vec(s.x), vec(s.tmx, 5 * R), 0); a.pro = 1;
            wrk::MatJob b[2] = {job6(L.ffn_w_k, vec(s.x), vec(s.ffn_k, F), 0), job6(L.ffn_w_r, vec(s.x), vec(s.ffn_r), 0)};
            b[0].pro = b[1].pro = 1;
            wrk::MatJob c = job6(L.ffn_w_v, vec(s.ffn_k, F), vec(s.x), 0); c.gate = s.ffn_r; c.carry_dst = (float*)s.x;
            single = wrk::matvec(q, &a, 1, ctx->num_cu, true) == 0 && wrk::matvec_grouped(q, b, 2, ctx->num_cu, true) == 0 &&
                     wrk::matvec(q, &c, 1, ctx->num_cu, true) == 0;
        }
        {   // K1
            wrk::MatJob j = job6(L.time_mix_w1, vec(s.att_xx), vec(s.tmx, 5 * R), WRK_ACT_TANH);
            if (single) {
                j.in = vec(s.x);
                j.pro = 1; j.pro_eps = LN_EPS; j.ln_w = L.ln1_w->ptr; j.ln_b = L.ln1_b->ptr; j.mixw = L.time_mix_x->ptr; j.prev = row0;
                j.ln_out = s.att_x;
            } else {
                wrk::LnMixParams P{};
                P.src = (const f16*)s.x; P.ln_w = (const f16*)L.ln1_w->ptr; P.ln_b = (const f16*)L.ln1_b->ptr; P.eps = LN_EPS;
                P.d = D; P.nmix = 1; P.mix[0] = (const f16*)L.time_mix_x->ptr; P.out[0] = (f16*)s.att_xx; P.ln_out = (f16*)s.att_x;
                P.state_row = lst; P.state_stride = (size_t)(S + 2) * D; P.cursors = s.cursors; P.no_carry = 1;
                LNMIX(P, T);
            }
            if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 K1 rejected");
        }
        {   // K2
            wrk::V6MixParams P{};
            P.d = D; P.r = R; P.w2_rb = (uint32_t)L.time_mix_w2[0]->row_bytes;
            for (int i = 0; i < 5; ++i) P.w2[i] = (const f16*)L.time_mix_w2[i]->data;
            P.tmx = (const f16*)s.tmx; P.time_mix = (const f16*)L.time_mix->ptr; P.x_ln = (const f16*)s.att_x;
            P.state = lst; P.cursors = s.cursors; P.batch1 = single ? batch0 + 1 : 0; P.sx5 = (f16*)s.att_sx; P.T = T;
            wrk::v6_mix_kernel<<<dim3((D + 63) / 64, T), 256, 0, q>>>(P);
        }
        {   // K3: order of the shifted inputs is w, k, v, r, g (v6.rs:1054-1071)
            f16* sx = (f16*)s.att_sx;
            wrk::MatJob jobs[5] = {job6(L.w_k, vec(sx + 1 * sxs), vec(s.att_k, 0, WRK_F32), WRK_ACT_NONE),
                                   job6(L.w_v, vec(sx + 2 * sxs), vec(s.att_v, 0, WRK_F32), WRK_ACT_NONE),
                                   job6(L.w_r, vec(sx + 3 * sxs), vec(s.att_r, 0, WRK_F32), WRK_ACT_NONE),
                                   job6(L.w_g, vec(sx + 4 * sxs), vec(s.att_g), WRK_ACT_NONE),
                                   job6(L.time_decay_w1, vec(sx + 0 * sxs), vec(s.att_w, W), WRK_ACT_TANH)};
            if (run_jobs(jobs, 5) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 K3 rejected");
        }
        {   // K4
            wrk::V6HeadParams P{};
            P.d = D; P.w = W; P.dw2 = (const f16*)L.time_decay_w2->data; P.dw2_rb = (uint32_t)L.time_decay_w2->row_bytes;
            P.att_w = (const f16*)s.att_w; P.time_decay = (const f16*)L.time_decay->ptr; P.u = (const float*)L.time_first->ptr;
            P.k = s.att_k; P.v = s.att_v; P.r = s.att_r; P.g = (const f16*)s.att_g;
            P.gn_w = (const f16*)L.gn_w->ptr; P.gn_b = (const f16*)L.gn_b->ptr;
            P.state = lst; P.cursors = s.cursors; P.batch1 = single ? batch0 + 1 : 0;
            P.shift_src = (const f16*)s.att_x; P.out = (f16*)s.aux_x; P.gn_eps = GN_EPS;
            wrk::v6_head_kernel<<<dim3(H, T), 256, 0, q>>>(P);
        }
        {   // K5: x += W_o . att
            wrk::MatJob j = job6(L.w_o, vec(s.aux_x), vec(s.x), WRK_ACT_NONE);
            j.has_res = 1; j.res = vec(s.x);
            if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 K5 rejected");
        }
        {   // K6
            wrk::MatJob jobs[2] = {job6(L.ffn_w_k, vec(s.ffn_kx), vec(s.ffn_k, F), WRK_ACT_SQUARED_RELU), job6(L.ffn_w_r, vec(s.ffn_rx), vec(s.ffn_r), WRK_ACT_NONE)};
            if (single) {
                const wrk_buf* mx[2] = {L.ffn_mix_k, L.ffn_mix_r};
                for (int i = 0; i < 2; ++i) {
                    jobs[i].in = vec(s.x);
                    jobs[i].pro = 1; jobs[i].pro_eps = LN_EPS; jobs[i].ln_w = L.ln2_w->ptr; jobs[i].ln_b = L.ln2_b->ptr;
                    jobs[i].mixw = mx[i]->ptr; jobs[i].prev = rowf;
                }
                jobs[0].ln_out = s.ffn_x;
            } else {
                wrk::LnMixParams P{};
                P.src = (const f16*)s.x; P.ln_w = (const f16*)L.ln2_w->ptr; P.ln_b = (const f16*)L.ln2_b->ptr; P.eps = LN_EPS;
                P.d = D; P.nmix = 2; P.mix[0] = (const f16*)L.ffn_mix_k->ptr; P.mix[1] = (const f16*)L.ffn_mix_r->ptr;
                P.out[0] = (f16*)s.ffn_kx; P.out[1] = (f16*)s.ffn_rx; P.ln_out = (f16*)s.ffn_x;
                P.state_row = lst + (size_t)(S + 1) * D; P.state_stride = (size_t)(S + 2) * D; P.cursors = s.cursors;
                LNMIX(P, T);
            }
            if (run_jobs(jobs, 2) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 K6 rejected");
        }
        {   // K7
            wrk::MatJob j = job6(L.ffn_w_v, vec(s.ffn_k, F), vec(s.ffn_v), WRK_ACT_NONE);
            if (single) {
                j.out = vec(s.x); j.has_res = 1; j.res = vec(s.x); j.gate = s.ffn_r; j.carry_src = s.ffn_x; j.carry_dst = rowf;
                if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 K7 rejected");
            } else {
                if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 K7 rejected");
                DTensor st_ffn = make_dense(lst, WRK_F32, D, S + 2, st->num_batch);
                st_ffn.offset[1] = S + 1; st_ffn.shape[1] = 1;
                wrk::channel_mix_v6(q, s.cursors, st_ffn, vec(s.ffn_r), vec(s.ffn_v), vec(s.ffn_x));
                wrk::binary(q, 0, vec(s.ffn_x), vec(s.x), 0, 0, 0);
            }
        }
        if ((li + 1) % d.rescale == 0) wrk::affine(q, vec(s.x), 0.5f, 0.0f);                             // v6.rs:953-955
    }
    if (NH > 0) {
        wrk::LnMixParams P{};
        P.src = (const f16*)s.x; P.ids = identity ? nullptr : s.headers;
        P.ln_w = (const f16*)d.ln_out_w->ptr; P.ln_b = (const f16*)d.ln_out_b->ptr; P.eps = LN_EPS;
        P.d = D; P.nmix = 0; P.ln_out = (f16*)s.head_x;
        LNMIX(P, NH);
        wrk::MatJob j = job6(d.head, make_dense(s.head_x, WRK_F16, D, NH), make_dense(s.head_o, WRK_F32, V, NH), WRK_ACT_NONE);
        if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused V6 head rejected");
    }
#undef LNMIX
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

static void for_each_handle(wrk_v6_model* m, void (*fb)(const wrk_buf*), void (*fm)(const wrk_matrix*)) {
    fb(m->d.ln0_w); fb(m->d.ln0_b); fb(m->d.ln_out_w); fb(m->d.ln_out_b); fb(m->d.emb_f16); fm(m->d.head);
    for (auto& L : m->layers) {
        const wrk_buf* vecs[] = {L.ln1_w, L.ln1_b, L.ln2_w, L.ln2_b, L.time_decay, L.time_first, L.time_mix_x, L.time_mix, L.gn_w, L.gn_b, L.ffn_mix_k, L.ffn_mix_r};
        for (const wrk_buf* b : vecs) fb(b);
        const wrk_matrix* mats[] = {L.time_decay_w1, L.time_decay_w2, L.time_mix_w1, L.time_mix_w2[0], L.time_mix_w2[1], L.time_mix_w2[2], L.time_mix_w2[3],
                                    L.time_mix_w2[4], L.w_k, L.w_v, L.w_r, L.w_g, L.w_o, L.ffn_w_k, L.ffn_w_v, L.ffn_w_r};
        for (const wrk_matrix* x : mats) fm(x);
    }
}

extern "C" {

int32_t wrk_v6_model_create(wrk_ctx* ctx, const wrk_v6_model_desc* desc, wrk_v6_model** out) {
    if (!ctx || !desc || !out) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_ARG(ctx, desc->num_layer >= 1 && desc->num_head >= 1 && desc->num_emb % desc->num_head == 0 && desc->num_emb / desc->num_head == 64, "bad model dims (head size must be 64)");
    WRK_ARG(ctx, desc->layers && desc->head && desc->ln0_w && desc->ln0_b && desc->ln_out_w && desc->ln_out_b, "missing tensors");
    WRK_ARG(ctx, desc->head->k == desc->num_emb && desc->head->m >= desc->num_vocab, "head matrix shape mismatch");
    const uint32_t D = desc->num_emb, F = desc->num_hidden, R = desc->time_mix, W = desc->time_decay;
    for (uint32_t l = 0; l < desc->num_layer; ++l) {
        const wrk_v6_layer_desc& L = desc->layers[l];
        const wrk_buf* vecs[] = {L.ln1_w, L.ln1_b, L.ln2_w, L.ln2_b, L.time_decay, L.time_mix_x, L.gn_w, L.gn_b, L.ffn_mix_k, L.ffn_mix_r};
        for (const wrk_buf* b : vecs) WRK_ARG(ctx, b && b->bytes >= (size_t)D * 2, "layer %u: vector missing or shorter than D f16", l);
        WRK_ARG(ctx, L.time_first && L.time_first->bytes >= (size_t)D * 4, "layer %u: time_first must hold D f32", l);
        WRK_ARG(ctx, L.time_mix && L.time_mix->bytes >= (size_t)D * 5 * 2, "layer %u: time_mix must hold 5*D f16", l);
        struct { const wrk_matrix* m; uint32_t k, mm; } mats[] = {
            {L.time_decay_w1, D, W}, {L.time_decay_w2, W, D}, {L.time_mix_w1, D, 5 * R}, {L.time_mix_w2[0], R, D}, {L.time_mix_w2[1], R, D},
            {L.time_mix_w2[2], R, D}, {L.time_mix_w2[3], R, D}, {L.time_mix_w2[4], R, D}, {L.w_k, D, D}, {L.w_v, D, D}, {L.w_r, D, D}, {L.w_g, D, D},
            {L.w_o, D, D}, {L.ffn_w_k, D, F}, {L.ffn_w_v, F, D}, {L.ffn_w_r, D, D}};
        for (auto& e : mats) WRK_ARG(ctx, e.m && e.m->k == e.k && e.m->m == e.mm, "layer %u: matrix missing or wrong shape (want K=%u M=%u)", l, e.k, e.mm);
    }
    wrk_v6_model* m = new wrk_v6_model();
    m->ctx = ctx;
    m->d = *desc;
    m->d.rescale = desc->rescale ? desc->rescale : 6;
    m->layers.assign(desc->layers, desc->layers + desc->num_layer);
    m->d.layers = m->layers.data();
    for_each_handle(m, [](const wrk_buf* b) { if (b) const_cast<wrk_buf*>(b)->refs.fetch_add(1); },
                    [](const wrk_matrix* x) { if (x) const_cast<wrk_matrix*>(x)->refs.fetch_add(1); });
    *out = m;
    return WRK_OK;
}

int32_t wrk_v6_model_destroy(wrk_v6_model* m) {
    if (!m) return WRK_E_ARG;
    {
        LOCK(m->ctx);
        hipSetDevice(m->ctx->device);
        hipStreamSynchronize(m->ctx->stream);
        m->drop_graphs();
        if (m->scratch) hipFree(m->scratch);
        if (m->history) hipFree(m->history);
    }
    for_each_handle(m, [](const wrk_buf* b) { if (b) wrk_buf_release(const_cast<wrk_buf*>(b)); },
                    [](const wrk_matrix* x) { if (x) wrk_matrix_release(const_cast<wrk_matrix*>(x)); });
    delete m;
    return WRK_OK;
}

size_t wrk_v6_model_token_bytes(const wrk_v6_model* m, uint32_t B) {
    if (!m) return 0;
    const size_t D = m->d.num_emb, S = D / m->d.num_head, V = m->d.num_vocab;
    size_t w = wrk_matrix_stream_bytes(m->d.head) + 4 * D * 2;
    for (auto& L : m->layers) {
        const wrk_matrix* mats[] = {L.time_decay_w1, L.time_decay_w2, L.time_mix_w1, L.time_mix_w2[0], L.time_mix_w2[1], L.time_mix_w2[2], L.time_mix_w2[3],
                                    L.time_mix_w2[4], L.w_k, L.w_v, L.w_r, L.w_g, L.w_o, L.ffn_w_k, L.ffn_w_v, L.ffn_w_r};
        for (const wrk_matrix* x : mats) w += wrk_matrix_stream_bytes(x);
        w += 16 * D * 2 + D * 4;
    }
    return w + 2 * (size_t)m->d.num_layer * D * (S + 2) * 4 * B + (size_t)B * (D * 2 + V * 4);
}

int32_t wrk_v6_state_create(wrk_ctx* ctx, const wrk_v6_model* model, uint32_t num_batch, wrk_v7_state** out) {
    if (!ctx || !model || !out) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_ARG(ctx, num_batch >= 1 && num_batch <= 255, "num_batch must be 1..255");
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    wrk_v7_state* st = new wrk_v7_state();
    st->ctx = ctx; st->num_layer = model->d.num_layer; st->num_emb = model->d.num_emb; st->head_size = 64; st->num_batch = num_batch;
    const size_t bytes = st->layer_elems() * st->num_layer * 4;
    hipError_t e = hipMalloc((void**)&st->data, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(st->data, 0, bytes, ctx->stream);
    if (e != hipSuccess) { delete st; return wrk_fail(ctx, WRK_E_OOM, "state alloc: %s", hipGetErrorString(e)); }
    *out = st;
    return WRK_OK;
}

int32_t wrk_v6_infer(wrk_ctx* ctx, wrk_v6_model* m, wrk_v7_state* st, const uint32_t* tokens, const uint16_t* emb_rows, const uint32_t* cursors,
                     uint32_t T, const uint32_t* headers, uint32_t NH, float* logits, uint32_t* argmax, uint32_t mode) {
    if (!ctx || !m || !st) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    if (T == 0) return WRK_OK;
    WRK_ARG(ctx, cursors && (tokens || emb_rows) && (!tokens || m->d.emb_f16) && (NH == 0 || headers), "missing inputs");
    WRK_ARG(ctx, st->num_emb == m->d.num_emb && st->num_layer == m->d.num_layer, "state does not belong to this model");
    const uint32_t D = m->d.num_emb, V = m->d.num_vocab;
    std::vector<uint8_t> seen(256, 0);
    bool one_token_each = true;
    uint32_t nseq = 0;
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t c = cursors[t], b = c & 0xff, tok = (c >> 8) & 0xffff, len = c >> 24;
        WRK_ARG(ctx, b < st->num_batch, "cursor %u: batch %u >= %u", t, b, st->num_batch);
        WRK_ARG(ctx, len >= 1 && tok <= t && t < tok + len && tok + len <= T, "cursor %u: bad range", t);
        if (tok == t) { WRK_ARG(ctx, !seen[b], "cursor %u: batch %u appears twice", t, b); seen[b] = 1; ++nseq; }
        if (len != 1) one_token_each = false;
        if (tokens) WRK_ARG(ctx, tokens[t] < V, "token %u: id %u >= vocab %u", t, tokens[t], V);
    }
    bool identity = (NH == T);
    for (uint32_t h = 0; h < NH; ++h) { WRK_ARG(ctx, headers[h] < T, "header %u out of range", h); if (headers[h] != h) identity = false; }
    int32_t rc = m->ensure_scratch(T, NH ? NH : 1);
    if (rc != WRK_OK) return rc;
    rc = wrk_buf_write_raw(ctx, m->s.cursors, cursors, (size_t)T * 4);
    if (rc == WRK_OK && NH) rc = wrk_buf_write_raw(ctx, m->s.headers, headers, (size_t)NH * 4);
    if (rc != WRK_OK) return rc;
    if (tokens) {
        rc = wrk_buf_write_raw(ctx, m->s.tokens, tokens, (size_t)T * 4);
        if (rc != WRK_OK) return rc;
        wrk::gather_rows_f16(ctx->op_stream(), m->d.emb_f16->ptr, m->s.tokens, m->s.input, D, T);
    } else {
        rc = wrk_buf_write_raw(ctx, m->s.input, emb_rows, (size_t)T * D * 2);
        if (rc != WRK_OK) return rc;
    }
    m->wkv_nseq = nseq;
    rc = WRK_E_UNSUPPORTED;
    if (mode == 1 && one_token_each) rc = m->enqueue_fused_decode(st, T, NH, identity, cursors[0] & 0xff);
    if (rc == WRK_E_UNSUPPORTED) rc = m->enqueue_ops(st, T, NH, identity, mode == 1 && !one_token_each);
    if (rc != WRK_OK) return rc;
    if (NH && argmax) wrk::argmax_rows(ctx->op_stream(), m->s.head_o, V, V, NH, m->s.argmax);
    WRK_LAUNCH_CHECK(ctx);
    if (ctx->capturing_here()) return WRK_OK;   // recorded into the caller's program: results exist after it has been launched
    if (NH && logits) WRK_HIP(ctx, hipMemcpyAsync(logits, m->s.head_o, (size_t)NH * V * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (NH && argmax) WRK_HIP(ctx, hipMemcpyAsync(argmax, m->s.argmax, (size_t)NH * 4, hipMemcpyDeviceToHost, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WRK_OK;
}

int32_t wrk_v6_generate_greedy(wrk_ctx* ctx, wrk_v6_model* m, wrk_v7_state* st, const uint32_t* first_tokens, uint32_t B, uint32_t steps,
                               uint32_t* out_tokens, float* last_logits, float* elapsed_ms, uint32_t mode) {
    if (!ctx || !m || !st || !first_tokens) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_ARG(ctx, m->d.emb_f16, "generate_greedy needs the device embedding table");
    WRK_ARG(ctx, B >= 1 && B <= st->num_batch, "num_batch %u exceeds the state's %u", B, st->num_batch);
    const uint32_t D = m->d.num_emb, V = m->d.num_vocab;
    for (uint32_t b = 0; b < B; ++b) WRK_ARG(ctx, first_tokens[b] < V, "first token %u out of vocab", first_tokens[b]);
    if (elapsed_ms) *elapsed_ms = 0.0f;
    if (steps == 0) return WRK_OK;
    int32_t rc = m->ensure_scratch(B, B);
    if (rc == WRK_OK) rc = m->ensure_history((size_t)steps * B);
    if (rc != WRK_OK) return rc;
    std::vector<uint32_t> cur(B), hdr(B);
    for (uint32_t b = 0; b < B; ++b) { cur[b] = b | (b << 8) | (1u << 24); hdr[b] = b; }
    rc = wrk_buf_write_raw(ctx, m->s.cursors, cur.data(), (size_t)B * 4);
    if (rc == WRK_OK) rc = wrk_buf_write_raw(ctx, m->s.headers, hdr.data(), (size_t)B * 4);
    if (rc == WRK_OK) rc = wrk_buf_write_raw(ctx, m->s.tokens, first_tokens, (size_t)B * 4);
    if (rc != WRK_OK) return rc;
    WRK_HIP(ctx, hipMemsetAsync(m->s.counter, 0, 4, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const char* ng = getenv("WRK_NO_GRAPH");
    const bool eager = ng && ng[0] == '1';
    auto enqueue_step = [&]() -> int32_t {
        wrk::gather_rows_f16(ctx->op_stream(), m->d.emb_f16->ptr, m->s.tokens, m->s.input, D, B);
        int32_t r = WRK_E_UNSUPPORTED;
        if (mode == 1) r = m->enqueue_fused_decode(st, B, B, true, 0);
        if (r == WRK_E_UNSUPPORTED) r = m->enqueue_ops(st, B, B, true);
        if (r != WRK_OK) return r;
        wrk::argmax_rows(ctx->op_stream(), m->s.head_o, V, V, B, m->s.argmax);
        wrk::advance_tokens(ctx->op_stream(), m->s.argmax, m->s.tokens, m->history, m->s.counter, B);
        return WRK_OK;
    };
    wrk_program* prog = nullptr;
    const auto key = std::make_tuple(st->uid, B, mode);
    if (!eager) {
        auto it = m->graphs.find(key);
        if (it != m->graphs.end()) prog = it->second;
        else {
            rc = wrk_capture_begin(ctx);
            if (rc != WRK_OK) return rc;
            rc = enqueue_step();
            wrk_program* p = nullptr;
            int32_t rc2 = wrk_capture_end(ctx, &p);
            if (rc != WRK_OK) { if (p) wrk_program_destroy(p); return rc; }
            if (rc2 != WRK_OK) return rc2;
            prog = p;
            m->graphs[key] = prog;
        }
    }
    hipEvent_t e0, e1;
    WRK_HIP(ctx, hipEventCreate(&e0));
    WRK_HIP(ctx, hipEventCreate(&e1));
    WRK_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (uint32_t i = 0; i < steps; ++i) {
        if (eager) { rc = enqueue_step(); if (rc != WRK_OK) return rc; }
        else WRK_HIP(ctx, hipGraphLaunch(prog->exec, ctx->stream));
    }
    WRK_HIP(ctx, hipEventRecord(e1, ctx->stream));
    WRK_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.0f;
    WRK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (elapsed_ms) *elapsed_ms = ms;
    if (out_tokens) WRK_HIP(ctx, hipMemcpyAsync(out_tokens, m->history, (size_t)steps * B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (last_logits) WRK_HIP(ctx, hipMemcpyAsync(last_logits, m->s.head_o, (size_t)B * V * 4, hipMemcpyDeviceToHost, ctx->stream));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WRK_OK;
}

}  // extern "C"
