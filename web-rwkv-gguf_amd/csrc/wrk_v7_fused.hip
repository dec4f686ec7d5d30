// Fused RWKV-7 decode step for gfx950: every sequence of the dispatch contributes exactly one token.
//
// The reference encodes ~44 dispatches per layer (v7.rs:716-1007); at batch-1 decode each of them
// moves a few KB, so the step is bound by dispatch-to-dispatch latency.  Here a layer is
// 7 launches, with every elementwise op folded into the kernel that produces or consumes its data:
//
//   K0 ln_mix      LN(ln1) + 6 token shifts + att shift-state carry            (ops 1-3, 12's row-0 write)
//   K1 matvec x7   r,k,v (quantised) + w1(tanh) a1 g1(sigmoid) v1 in ONE launch (ops 4, 5a, 6a, 7a, 10a)
//   K2 head        per (head, sequence): LoRA up-projections w2/a2/g2/v2 + bias/activation, kk = l2norm(k*k_k),
//                  control_k, value residual lerp, WKV7 state update (state in registers), group norm,
//                  time_first bonus, gate                                      (ops 5b-15)
//   K3 matvec      W_o with fused residual add                                  (op 16)
//   K4 ln_mix      LN(ln2) + ffn token shift + ffn shift-state carry            (ops 17-18, 21)
//   K5 matvec      ffn key with squared-ReLU                                    (op 19)
//   K6 matvec      ffn value with fused residual add                            (ops 20, 22)
//
// Every intermediate is rounded to f16 exactly where the op-by-op path (and the reference's
// Runtime<f16> buffers) would store it, so mode 0 and mode 1 agree to f32 summation order.
#include "wrk_device.h"
#include "wrk_v7.h"

namespace wrk {

// ------------------------------------------------------------------ token bookkeeping
__global__ void advance_tokens_kernel(const uint32_t* __restrict__ argmax, uint32_t* __restrict__ tokens,
                                      uint32_t* __restrict__ history, uint32_t* __restrict__ counter, uint32_t b) {
    const uint32_t step = *counter;
    const uint32_t i = threadIdx.x;
    if (i < b) {
        const uint32_t t = argmax[i];
        tokens[i] = t;
        history[(size_t)step * b + i] = t;
    }
    __syncthreads();
    if (i == 0) *counter = step + 1;
}

void advance_tokens(hipStream_t s, const uint32_t* argmax, uint32_t* tokens, uint32_t* history, uint32_t* counter, uint32_t b) {
    advance_tokens_kernel<<<1, 256, 0, s>>>(argmax, tokens, history, counter, b);
}

// ------------------------------------------------------------------ K0 / K4: layer norm + token shifts
struct LnMixParams {
    const f16* src;             // [T][D] rows, or the embedding table when `ids` is set
    const uint32_t* ids;        // optional row index per token (embedding gather)
    const f16 *ln_w, *ln_b;
    float eps;
    uint32_t d, nmix;
    const f16* mix[6];          // token-shift factors
    f16* out[6];                // shifted outputs [T][D]
    f16* ln_out;                // optional: LN output [T][D]
    float* state_row;           // optional: shift state row, element (batch, c) at state_row[batch * state_stride + c]
    size_t state_stride;
    const uint32_t* cursors;    // batch id per token
};

__global__ void __launch_bounds__(256) ln_mix_kernel(const LnMixParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = (float*)smem;           // [D]
    __shared__ float red[4];
    const uint32_t t = blockIdx.x, D = P.d;
    const f16* row = P.src + (size_t)(P.ids ? P.ids[t] : t) * D;
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < D; i += 256) { const float v = (float)row[i]; xs[i] = v; s += v; }
    const float mean = block_sum<4>(s, red) / (float)D;
    float q = 0.0f;
    for (uint32_t i = threadIdx.x; i < D; i += 256) { const float dlt = xs[i] - mean; q += dlt * dlt; }
    const float var = block_sum<4>(q, red) / (float)D + P.eps;
    const float dev = 1.0f / sqrtf(var);
    const uint32_t batch = P.cursors ? (P.cursors[t] & 0xffu) : t;
    float* st = P.state_row ? P.state_row + (size_t)batch * P.state_stride : nullptr;
    for (uint32_t i = threadIdx.x; i < D; i += 256) {
        const float value = (xs[i] - mean) * dev;
        const float y = r16(__builtin_fmaf(value, (float)P.ln_w[i], (float)P.ln_b[i]));   // stored f16 (att_x / ffn_x)
        if (P.ln_out) P.ln_out[(size_t)t * D + i] = (f16)y;
        if (st) {
            const float prev = st[i];
#pragma unroll
            for (uint32_t m = 0; m < 6; ++m)
                if (m < P.nmix) P.out[m][(size_t)t * D + i] = (f16)wgsl_mix(y, prev, (float)P.mix[m][i]);
            st[i] = y;              // shift-state carry (time_mix_v7.wgsl:156-158 / channel_mix.wgsl:99-101)
        }
    }
}

static void ln_mix(hipStream_t s, const LnMixParams& P, uint32_t T) {
    ln_mix_kernel<<<T, 256, (size_t)P.d * 4, s>>>(P);
}

// ------------------------------------------------------------------ K2: the per-head time-mix kernel
struct HeadParams {
    uint32_t d, layer0;
    uint32_t rw, ra, rg, rv;                    // LoRA ranks
    const f16 *w2, *a2, *g2, *v2;               // f16 [D][rank] row-major (device rows may be padded: *_rb bytes per row)
    uint32_t w2_rb, a2_rb, g2_rb, v2_rb;
    const f16 *w0, *a0, *v0;                    // [D]
    const f16 *k_k, *k_a, *r_k, *gn_w, *gn_b;   // [D]
    const f16 *aux_w, *aux_a, *aux_g, *aux_v;   // [T][rank]
    const f16 *r, *k, *v;                       // [T][D]
    f16* v_first;                               // att_v0 [T][D]: written on layer 0, read afterwards
    f16* out;                                   // att_x [T][D]
    float* state;                               // layer state base: element (batch, row, c) at state[(batch*(S+2)+row)*D + c]
    const uint32_t* cursors;
    float gn_eps, l2_eps;
};

// dot of one f16 weight row slice with an f32 vector in LDS; 4 lanes cooperate on a row
__device__ __forceinline__ float lora_row_dot(const f16* __restrict__ wrow, const float* __restrict__ aux, uint32_t rank, uint32_t part) {
    float acc = 0.0f;
    for (uint32_t c = part * 8; c < rank; c += 32) {
        const f16x8 w = *(const f16x8*)(wrow + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_fmaf((float)w[e], aux[c + e], acc);
    }
    acc += __shfl_xor(acc, 1, WAVE);
    acc += __shfl_xor(acc, 2, WAVE);
    return acc;
}

__global__ void __launch_bounds__(256) head_kernel(const HeadParams P) {
    constexpr int S = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* aux = (float*)smem;                      // [rw + ra + rg + rv]
    __shared__ float sh_r[S], sh_w[S], sh_k[S], sh_v[S], sh_a[S], sh_b[S], sh_g[S], sh_kk[S];
    __shared__ float sh_red[4][S];
    const uint32_t head = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const uint32_t D = P.d;
    const uint32_t batch = P.cursors[t] & 0xffu;
    const uint32_t c0 = head * S;

    // 1. LoRA intermediates of this token into LDS
    const uint32_t o_a = P.rw, o_g = o_a + P.ra, o_v = o_g + P.rg, ntot = o_v + (P.layer0 ? 0 : P.rv);
    for (uint32_t i = tid; i < ntot; i += 256) {
        float v;
        if (i < o_a) v = (float)P.aux_w[(size_t)t * P.rw + i];
        else if (i < o_g) v = (float)P.aux_a[(size_t)t * P.ra + (i - o_a)];
        else if (i < o_v) v = (float)P.aux_g[(size_t)t * P.rg + (i - o_g)];
        else v = (float)P.aux_v[(size_t)t * P.rv + (i - o_v)];
        aux[i] = v;
    }
    __syncthreads();

    // 2. up-projections for the 64 channels of this head: thread = (row = tid/4, part = tid%4)
    {
        const uint32_t row = tid >> 2, part = tid & 3u, ch = c0 + row;
        const float dw = lora_row_dot((const f16*)((const uint8_t*)P.w2 + (size_t)ch * P.w2_rb), aux, P.rw, part);
        const float da = lora_row_dot((const f16*)((const uint8_t*)P.a2 + (size_t)ch * P.a2_rb), aux + o_a, P.ra, part);
        const float dg = lora_row_dot((const f16*)((const uint8_t*)P.g2 + (size_t)ch * P.g2_rb), aux + o_g, P.rg, part);
        float dv = 0.0f;
        if (!P.layer0) dv = lora_row_dot((const f16*)((const uint8_t*)P.v2 + (size_t)ch * P.v2_rb), aux + o_v, P.rv, part);
        if (part == 0) {
            const float w = r16((float)P.w0[ch] + r16(dw));                              // add(w0, w)
            const float a = r16(act_sigmoid((float)P.a0[ch] + r16(da)));                 // add_activate(.., Sigmoid)
            const float g = r16(dg);
            const float kraw = (float)P.k[(size_t)t * D + ch];
            float v = (float)P.v[(size_t)t * D + ch];
            if (P.layer0) P.v_first[(size_t)t * D + ch] = (f16)v;                        // blit(att_v, att_v0)
            else {
                const float vv = r16(act_sigmoid((float)P.v0[ch] + r16(dv)));
                v = r16(wgsl_mix(v, (float)P.v_first[(size_t)t * D + ch], vv));          // lerp(att_v0, att_v, att_vv, reversed)
            }
            sh_w[row] = __expf(-0.606531f * act_sigmoid(w));                             // act_w (time_mix_v7.wgsl:68-70)
            sh_a[row] = a;
            sh_g[row] = g;
            sh_v[row] = v;
            sh_r[row] = (float)P.r[(size_t)t * D + ch];
            sh_kk[row] = r16((float)P.k_k[ch] * kraw);                                   // mul(k_k, kk)
            sh_k[row] = r16(kraw * (1.0f + (a - 1.0f) * (float)P.k_a[ch]));             // control_k_v7
        }
    }
    __syncthreads();
    // 3. kk <- l2_norm(kk) over the head; a~ = -kk, b~ = kk * a
    const uint32_t i = tid & 63, g4 = tid >> 6;
    if (g4 == 0) {
        const float kkv = sh_kk[i];
        const float nrm = 1.0f / sqrtf(wave_sum(kkv * kkv) + P.l2_eps);
        const float kkn = r16(kkv * nrm);
        const float a = sh_a[i];
        sh_a[i] = -kkn;
        sh_b[i] = kkn * a;
    }
    __syncthreads();

    // 4. WKV7: thread (i, g4) owns S[16*g4 .. +15][i] in registers
    float* st = P.state + ((size_t)batch * (S + 2) + 1) * D + c0 + i;
    float Sreg[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Sreg[jj] = st[(size_t)(g4 * 16 + jj) * D];
    float sa = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) sa = __builtin_fmaf(Sreg[jj], sh_a[g4 * 16 + jj], sa);
    sh_red[g4][i] = sa;
    __syncthreads();
    sa = (sh_red[0][i] + sh_red[1][i]) + (sh_red[2][i] + sh_red[3][i]);
    const float vv = sh_v[i];
    float y = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = g4 * 16 + jj;
        const float s = Sreg[jj] * sh_w[j] + sh_k[j] * vv + sa * sh_b[j];
        st[(size_t)j * D] = s;
        y = __builtin_fmaf(sh_r[j], s, y);
    }
    __syncthreads();
    sh_red[g4][i] = y;
    __syncthreads();
    if (g4 == 0) {
        y = r16((sh_red[0][i] + sh_red[1][i]) + (sh_red[2][i] + sh_red[3][i]));          // att_x <- y (f16 store)
        // 5. group norm over the head (layer_norm.wgsl GROUP_NORM)
        const float mean = wave_sum(y) * (1.0f / S);
        const float dl = y - mean;
        const float var = wave_sum(dl * dl) * (1.0f / S) + P.gn_eps;
        float o = r16(__builtin_fmaf(dl * (1.0f / sqrtf(var)), (float)P.gn_w[c0 + i], (float)P.gn_b[c0 + i]));
        // 6. time_first: x += (sum_j r_k * k * r) * v
        const float xx = wave_sum((float)P.r_k[c0 + i] * sh_k[i] * sh_r[i]);
        o = r16(o + xx * vv);
        // 7. gate
        o = sh_g[i] * o;
        P.out[(size_t)t * D + c0 + i] = (f16)o;
    }
}

}  // namespace wrk

// ------------------------------------------------------------------ host: enqueue one fused decode step
void wrk_v7_model::drop_graphs() {
    for (auto& kv : graphs) wrk_program_destroy(kv.second);
    graphs.clear();
}

void wrk_v7_model::free_fused() {}

static wrk::MatJob job(const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    return wrk::MatJob{m->data, m->aux, m->kind, m->flags, m->k, m->m, (uint32_t)m->row_bytes, in, out, act, 0};
}

int32_t wrk_v7_model::enqueue_fused_decode(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity_headers) {
    using namespace wrk;
    hipStream_t q = ctx->stream;
    const uint32_t D = d.num_emb, F = d.num_hidden, H = d.num_head, S = D / H, V = d.num_vocab;
    // LoRA matrices must be F16 for the head kernel (the reference keeps them Matrix::Fp16, v7.rs:1108-1113)
    for (auto& L : layers) {
        const wrk_matrix* ms[] = {L.w2, L.a2, L.g2, L.v2};
        for (const wrk_matrix* m : ms)
            if (m && m->kind != WRK_MAT_F16) return enqueue_ops(st, T, NH, identity_headers);
    }
    if (d.lora_w % 8 || d.lora_a % 8 || d.lora_g % 8 || d.lora_v % 8) return enqueue_ops(st, T, NH, identity_headers);
    auto vec = [&](void* p, uint32_t c = 0) { return make_dense(p, WRK_F16, c ? c : D, T); };

    // embed: LN(ln0) on the gathered rows -> x   (v7.rs:649-659)
    {
        LnMixParams P{};
        P.src = (const f16*)s.input; P.ln_w = (const f16*)ln0_w->ptr; P.ln_b = (const f16*)ln0_b->ptr; P.eps = 1.0e-5f;
        P.d = D; P.nmix = 0; P.ln_out = (f16*)s.x;
        ln_mix(q, P, T);
    }
    for (uint32_t li = 0; li < d.num_layer; ++li) {
        const wrk_v7_layer_desc& L = layers[li];
        float* lst = st->layer_ptr(li);
        {   // K0
            LnMixParams P{};
            P.src = (const f16*)s.x; P.ln_w = (const f16*)L.ln1_w->ptr; P.ln_b = (const f16*)L.ln1_b->ptr; P.eps = 1.0e-5f;
            P.d = D; P.nmix = 6;
            const wrk_buf* mx[6] = {L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g};
            void* outs[6] = {s.rx, s.wx, s.kx, s.vx, s.ax, s.gx};
            for (int i = 0; i < 6; ++i) { P.mix[i] = (const f16*)mx[i]->ptr; P.out[i] = (f16*)outs[i]; }
            P.state_row = lst; P.state_stride = (size_t)(S + 2) * D; P.cursors = s.cursors;
            ln_mix(q, P, T);
        }
        {   // K1
            MatJob jobs[7] = {job(L.w_r, vec(s.rx), vec(s.r), WRK_ACT_NONE), job(L.w_k, vec(s.kx), vec(s.k), WRK_ACT_NONE),
                              job(L.w_v, vec(s.vx), vec(s.v), WRK_ACT_NONE),
                              job(L.w1, vec(s.wx), vec(s.aux_w, d.lora_w), WRK_ACT_TANH),
                              job(L.a1, vec(s.ax), vec(s.aux_a, d.lora_a), WRK_ACT_NONE),
                              job(L.g1, vec(s.gx), vec(s.aux_g, d.lora_g), WRK_ACT_SIGMOID),
                              job(li ? L.v1 : L.a1, vec(s.vx), vec(s.aux_v, d.lora_v), WRK_ACT_NONE)};
            if (matvec(q, jobs, li ? 7 : 6, ctx->num_cu) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K1 rejected");
        }
        {   // K2
            HeadParams P{};
            P.d = D; P.layer0 = li == 0;
            P.rw = d.lora_w; P.ra = d.lora_a; P.rg = d.lora_g; P.rv = d.lora_v;
            P.w2 = (const f16*)L.w2->data; P.a2 = (const f16*)L.a2->data; P.g2 = (const f16*)L.g2->data;
            P.w2_rb = (uint32_t)L.w2->row_bytes; P.a2_rb = (uint32_t)L.a2->row_bytes; P.g2_rb = (uint32_t)L.g2->row_bytes;
            if (li) { P.v2 = (const f16*)L.v2->data; P.v2_rb = (uint32_t)L.v2->row_bytes; P.v0 = (const f16*)L.v0->ptr; }
            P.w0 = (const f16*)L.w0->ptr; P.a0 = (const f16*)L.a0->ptr;
            P.k_k = (const f16*)L.k_k->ptr; P.k_a = (const f16*)L.k_a->ptr; P.r_k = (const f16*)L.r_k->ptr;
            P.gn_w = (const f16*)L.gn_w->ptr; P.gn_b = (const f16*)L.gn_b->ptr;
            P.aux_w = (const f16*)s.aux_w; P.aux_a = (const f16*)s.aux_a; P.aux_g = (const f16*)s.aux_g; P.aux_v = (const f16*)s.aux_v;
            P.r = (const f16*)s.r; P.k = (const f16*)s.k; P.v = (const f16*)s.v;
            P.v_first = (f16*)s.att_v0; P.out = (f16*)s.att_x;
            P.state = lst; P.cursors = s.cursors; P.gn_eps = 64.0e-5f; P.l2_eps = 1.0e-12f;
            const size_t smem = (size_t)(d.lora_w + d.lora_a + d.lora_g + d.lora_v) * 4;
            head_kernel<<<dim3(H, T), 256, smem, q>>>(P);
        }
        {   // K3: x += W_o . att_x
            MatJob j = job(L.w_o, vec(s.att_x), vec(s.x), WRK_ACT_NONE);
            j.has_res = 1; j.res = vec(s.x);
            if (matvec(q, &j, 1, ctx->num_cu) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K3 rejected");
        }
        {   // K4
            LnMixParams P{};
            P.src = (const f16*)s.x; P.ln_w = (const f16*)L.ln2_w->ptr; P.ln_b = (const f16*)L.ln2_b->ptr; P.eps = 1.0e-5f;
            P.d = D; P.nmix = 1; P.mix[0] = (const f16*)L.ffn_x_k->ptr; P.out[0] = (f16*)s.ffn_kx;
            P.state_row = lst + (size_t)(S + 1) * D; P.state_stride = (size_t)(S + 2) * D; P.cursors = s.cursors;
            ln_mix(q, P, T);
        }
        {   // K5
            MatJob j = job(L.ffn_w_k, vec(s.ffn_kx), vec(s.ffn_k, F), WRK_ACT_SQUARED_RELU);
            if (matvec(q, &j, 1, ctx->num_cu) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K5 rejected");
        }
        {   // K6: x += W_v . relu(k)^2
            MatJob j = job(L.ffn_w_v, vec(s.ffn_k, F), vec(s.x), WRK_ACT_NONE);
            j.has_res = 1; j.res = vec(s.x);
            if (matvec(q, &j, 1, ctx->num_cu) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K6 rejected");
        }
        if ((li + 1) % d.rescale == 0) wrk::affine(q, vec(s.x), 0.5f, 0.0f);
    }
    if (NH > 0) {
        // header: rows -> LN(ln_out) -> head matmul (f32 logits)
        LnMixParams P{};
        P.src = (const f16*)s.x; P.ids = identity_headers ? nullptr : s.headers;
        P.ln_w = (const f16*)ln_out_w->ptr; P.ln_b = (const f16*)ln_out_b->ptr; P.eps = 1.0e-5f;
        P.d = D; P.nmix = 0; P.ln_out = (f16*)s.head_x;
        ln_mix(q, P, NH);
        MatJob j = job(head, make_dense(s.head_x, WRK_F16, D, NH), make_dense(s.head_o, WRK_F32, V, NH), WRK_ACT_NONE);
        if (matvec(q, &j, 1, ctx->num_cu) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused head rejected");
    }
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}
