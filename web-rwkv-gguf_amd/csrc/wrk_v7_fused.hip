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
#include "wrk_v7_engine.h"
#include "wrk_lora_dev.h"

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
// One workgroup per token.  Every global load (row, LN weights, previous shift state, mix factors)
// is issued up front as 16-byte vectors: the kernel is one memory round trip + two block reductions.
template <int VPT, int NMIX>
__global__ void __launch_bounds__(256) ln_mix_kernel(const LnMixParams P) {
    __shared__ float red[4];
    const uint32_t t = blockIdx.x, D = P.d, nvec = D >> 3, tid = threadIdx.x;
    const size_t rowi = P.ids ? P.ids[t] : t;
    const uint32_t batch = P.batch1 ? P.batch1 - 1 + t : (P.cursors ? (P.cursors[t] & 0xffu) : t);     // host-known: no scalar round trip in front of the state loads
    const f16* row = P.src + rowi * D;
    // every load unconditional (vector index clamped, a launch without shift state reads the LN weights as a mapped dummy): the predicated
    // form made the compiler wait vmcnt(0) in front of the statistics
    const float* st = P.state_row ? P.state_row + (size_t)batch * P.state_stride : (const float*)P.ln_w;
    f16x8 xv[VPT], wv[VPT], bv[VPT], mv[NMIX > 0 ? NMIX : 1][VPT];
    f32x4 pv[VPT][2];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const uint32_t i = min(tid + 256u * v, nvec - 1);
        xv[v] = *(const f16x8*)(row + i * 8);
        wv[v] = *(const f16x8*)(P.ln_w + i * 8);
        bv[v] = *(const f16x8*)(P.ln_b + i * 8);
        if (NMIX > 0) {
            pv[v][0] = *(const f32x4*)(st + i * 8);
            pv[v][1] = *(const f32x4*)(st + i * 8 + 4);
#pragma unroll
            for (int m = 0; m < NMIX; ++m) mv[m][v] = *(const f16x8*)(P.mix[m] + i * 8);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < VPT; ++v)
        if (tid + 256 * v < nvec)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)xv[v][e];
    const float mean = block_sum<4>(s, red) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int v = 0; v < VPT; ++v)
        if (tid + 256 * v < nvec)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = (float)xv[v][e] - mean; q += dl * dl; }
    const float var = block_sum<4>(q, red) / (float)D + P.eps;
    const float dev = 1.0f / sqrtf(var);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const uint32_t i = tid + 256 * v;
        if (i >= nvec) continue;
        float y[8];
        f16x8 yv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float value = ((float)xv[v][e] - mean) * dev;
            yv[e] = (f16)__builtin_fmaf(value, (float)wv[v][e], (float)bv[v][e]);      // stored f16 (att_x / ffn_x)
            y[e] = (float)yv[e];
        }
        if (P.ln_out) *(f16x8*)(P.ln_out + (size_t)t * D + i * 8) = yv;
        if (NMIX > 0) {
#pragma unroll
            for (int m = 0; m < NMIX; ++m) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)wgsl_mix(y[e], pv[v][e >> 2][e & 3], (float)mv[m][v][e]);
                *(f16x8*)(P.out[m] + (size_t)t * D + i * 8) = o;
            }
            // shift-state carry (time_mix_v7.wgsl:156-158 / channel_mix.wgsl:99-101)
            if (!P.no_carry) {
                f32x4 n0 = {y[0], y[1], y[2], y[3]}, n1 = {y[4], y[5], y[6], y[7]};
                float* sw = P.state_row + (size_t)batch * P.state_stride;
                *(f32x4*)(sw + i * 8) = n0;
                *(f32x4*)(sw + i * 8 + 4) = n1;
            }
        }
    }
}

int ln_mix(hipStream_t s, const LnMixParams& P, uint32_t T) {
    const uint32_t nvec = P.d >> 3;
    const int vpt = nvec <= 256 ? 1 : (nvec <= 512 ? 2 : (nvec <= 1024 ? 4 : 0));
    if (vpt == 0 || (P.d & 7u)) return -1;
#define LN_LAUNCH(V, M) ln_mix_kernel<V, M><<<T, 256, 0, s>>>(P)
#define LN_SWITCH(M) do { if (vpt == 1) LN_LAUNCH(1, M); else if (vpt == 2) LN_LAUNCH(2, M); else LN_LAUNCH(4, M); } while (0)
    if (P.nmix == 0) LN_SWITCH(0);
    else if (P.nmix == 1) LN_SWITCH(1);
    else if (P.nmix == 2) LN_SWITCH(2);
    else if (P.nmix == 6) LN_SWITCH(6);
    else return -1;
#undef LN_SWITCH
#undef LN_LAUNCH
    return 0;
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
    const f16* shift_src;                       // optional [T][D]: copied into state row 0 (att shift carry of a fused K0)
    float gn_eps, l2_eps;
    uint32_t batch1;                            // batch id + 1 when the host knows it (single-sequence decode), else 0: read the cursor
    unsigned long long* dbg;
};

template <int GCH>     // 32-column chunks of the gate LoRA held per lane group: rank_g <= 32 * GCH
__global__ void __launch_bounds__(256) head_kernel(const HeadParams P) {
    constexpr int S = 64;
    __shared__ float sh_r[S], sh_w[S], sh_k[S], sh_v[S], sh_a[S], sh_b[S], sh_g[S], sh_kk[S];
    __shared__ float sh_red[4][S];
    const uint32_t head = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const uint32_t D = P.d;
    const uint32_t c0 = head * S;
    const uint32_t i = tid & 63, g4 = tid >> 6;
    WRK_STAMP(P.dbg, 0);

    // (1) everything that does not depend on the cursor goes out first: LoRA up-projection rows for the 64 channels
    //     of this head (thread = (row = tid/4, part = tid%4)) and the per-channel scalars
    const uint32_t row = tid >> 2, part = tid & 3u, ch = c0 + row;
    LoraRegs<4> lw, la, lv;
    LoraRegs<GCH> lg;
    lora_load<4>(lw, (const f16*)((const uint8_t*)P.w2 + (size_t)ch * P.w2_rb), P.aux_w + (size_t)t * P.rw, P.rw, part);
    lora_load<4>(la, (const f16*)((const uint8_t*)P.a2 + (size_t)ch * P.a2_rb), P.aux_a + (size_t)t * P.ra, P.ra, part);
    lora_load<GCH>(lg, (const f16*)((const uint8_t*)P.g2 + (size_t)ch * P.g2_rb), P.aux_g + (size_t)t * P.rg, P.rg, part);
    if (!P.layer0) lora_load<4>(lv, (const f16*)((const uint8_t*)P.v2 + (size_t)ch * P.v2_rb), P.aux_v + (size_t)t * P.rv, P.rv, part);
    const float w0 = (float)P.w0[ch], a0 = (float)P.a0[ch], kkw = (float)P.k_k[ch], kaw = (float)P.k_a[ch];
    const float kraw = (float)P.k[(size_t)t * D + ch], rraw = (float)P.r[(size_t)t * D + ch];
    float v = (float)P.v[(size_t)t * D + ch];
    // unconditional (as in head_split_kernel): layer 0 reads w0 / its own v_first slot as dummies, no shift source reads w0
    const f16 v0h = (P.layer0 ? P.w0 : P.v0)[ch], vfh = P.v_first[(size_t)t * D + ch];
    const float v0w = P.layer0 ? 0.0f : (float)v0h, vfirst = P.layer0 ? 0.0f : (float)vfh;
    const float gnw = (float)P.gn_w[c0 + i], gnb = (float)P.gn_b[c0 + i], rkw = (float)P.r_k[c0 + i];
    const float shift = (float)(P.shift_src ? P.shift_src + (size_t)t * D : P.w0)[c0 + i];

    // (2) state of this thread's column slice S[16*g4 .. +15][i]: needs the batch id; requested last, consumed last
    const uint32_t batch = P.batch1 ? P.batch1 - 1 + t : (P.cursors[t] & 0xffu);
    float* st = P.state + ((size_t)batch * (S + 2) + 1) * D + c0 + i;
    float Sreg[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Sreg[jj] = st[(size_t)(g4 * 16 + jj) * D];

    {
        const float dw = lora_dot<4>(lw, P.rw, part);
        const float da = lora_dot<4>(la, P.ra, part);
        const float dg = lora_dot<GCH>(lg, P.rg, part);
        float dv = 0.0f;
        if (!P.layer0) dv = lora_dot<4>(lv, P.rv, part);
        if (part == 0) {
            const float w = r16(w0 + r16(dw));                                           // add(w0, w)
            const float a = r16(act_sigmoid(a0 + r16(da)));                              // add_activate(.., Sigmoid)
            const float g = r16(dg);
            if (P.layer0) P.v_first[(size_t)t * D + ch] = (f16)v;                        // blit(att_v, att_v0)
            else {
                const float vv = r16(act_sigmoid(v0w + r16(dv)));
                v = r16(wgsl_mix(v, vfirst, vv));                                        // lerp(att_v0, att_v, att_vv, reversed)
            }
            sh_w[row] = __expf(-0.606531f * act_sigmoid(w));                             // act_w (time_mix_v7.wgsl:68-70)
            sh_a[row] = a;
            sh_g[row] = g;
            sh_v[row] = v;
            sh_r[row] = rraw;
            sh_kk[row] = r16(kkw * kraw);                                                // mul(k_k, kk)
            sh_k[row] = r16(kraw * (1.0f + (a - 1.0f) * kaw));                           // control_k_v7
        }
    }
    if (P.shift_src && g4 == 1) P.state[(size_t)batch * (S + 2) * D + c0 + i] = shift;
    WRK_STAMP(P.dbg, 1);     // LoRA rows arrived, per-channel scalars in LDS
    __syncthreads();
    // kk <- l2_norm(kk) over the head; a~ = -kk, b~ = kk * a
    if (g4 == 0) {
        const float kkv = sh_kk[i];
        const float nrm = 1.0f / sqrtf(wave_sum(kkv * kkv) + P.l2_eps);
        const float kkn = r16(kkv * nrm);
        const float a = sh_a[i];
        sh_a[i] = -kkn;
        sh_b[i] = kkn * a;
    }
    __syncthreads();

    WRK_STAMP(P.dbg, 2);
    // WKV7: thread (i, g4) owns S[16*g4 .. +15][i] in registers
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
    WRK_STAMP(P.dbg, 3);     // state updated and stored
    if (g4 == 0) {
        y = r16((sh_red[0][i] + sh_red[1][i]) + (sh_red[2][i] + sh_red[3][i]));          // att_x <- y (f16 store)
        // group norm over the head (layer_norm.wgsl GROUP_NORM)
        const float mean = wave_sum(y) * (1.0f / S);
        const float dl = y - mean;
        const float var = wave_sum(dl * dl) * (1.0f / S) + P.gn_eps;
        float o = r16(__builtin_fmaf(dl * (1.0f / sqrtf(var)), gnw, gnb));
        // time_first: x += (sum_j r_k * k * r) * v
        const float xx = wave_sum(rkw * sh_k[i] * sh_r[i]);
        o = r16(o + xx * vv);
        // gate
        o = sh_g[i] * o;
        P.out[(size_t)t * D + c0 + i] = (f16)o;
    }
    WRK_STAMP(P.dbg, 4);
}

// ------------------------------------------------------------------ K2, split: NS = 4 workgroups per head
// The one-workgroup-per-head kernel pulls 97 KB (LoRA up-projection rows + state) through ONE CU, and a CU sustains ~25-70 GB/s
// (MI355X_MICROARCH.md, "Indexed rows"): 2.4 us of its 3.9 us were that load (WRK_TIMING, round 2).  Here workgroup (head, q)
// owns the value columns i = 16q .. 16q+15 of the head's state S[j][i] (all 64 rows j): sa_i = sum_j S[j][i] a_j and
// y_i = sum_j r_j S'[j][i] are local to a column, so the WKV update needs no exchange between the four workgroups.  Each of
// them recomputes the per-ROW quantities of the head (decay w_j, a_j, k_j, kk_j: the w2 / a2 LoRA rows, 24 KB, L2 hits for three
// of the four: blockIdx = head + H*q puts them on one XCD), and loads only its 16 columns of g2 / v2 / state: 38 KB per CU.
// The group norm needs all 64 outputs of the head, so it moves, with the time_first bonus and the gate, into the prologue of
// the W_o matvec (dmv PRO 3/4); this kernel hands over y (f16, the value the op list stores in att_x), tt = (sum_j r_k k r) v (f32)
// and g (f16).  Thread (wave w, lane j) holds S[j][16q + 4w .. +3]: both reductions over j are DPP wave sums, no LDS round trip.
template <int GC>       // chunks of 128 gate-LoRA columns a 16-lane row group covers: rank_g <= 128 * GC
__global__ void __launch_bounds__(256) head_split_kernel(const HeadParams P, float* __restrict__ tt_out, f16* __restrict__ g_out, uint32_t H) {
    constexpr int S = 64, C = 16;
    __shared__ float sh_r[S], sh_w[S], sh_k[S], sh_a[S], sh_b[S], sh_kk[S], sh_rkr[S];
    __shared__ float sh_v[C], sh_g[C], sh_xx;
    const uint32_t head = blockIdx.x % H, q = blockIdx.x / H, t = blockIdx.y, tid = threadIdx.x;
    const uint32_t D = P.d, c0 = head * S, cq = c0 + C * q;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    WRK_STAMP(P.dbg, 0);

    // (1) cursor-independent loads first: w2 / a2 rows of all 64 channels (thread = (row, part)), g2 / v2 rows of the 16 own
    //     columns (thread = (row16, p16)), per-channel scalars
    const uint32_t row = tid >> 2, part = tid & 3u, ch = c0 + row;
    LoraRegs<4> lw, la;
    lora_load<4>(lw, (const f16*)((const uint8_t*)P.w2 + (size_t)ch * P.w2_rb), P.aux_w + (size_t)t * P.rw, P.rw, part);
    lora_load<4>(la, (const f16*)((const uint8_t*)P.a2 + (size_t)ch * P.a2_rb), P.aux_a + (size_t)t * P.ra, P.ra, part);
    const uint32_t row16 = tid >> 4, p16 = tid & 15u, chq = cq + row16;
    f16x8 gw[GC], gx[GC], vw, vx;
    {
        const f16* grow = (const f16*)((const uint8_t*)P.g2 + (size_t)chq * P.g2_rb);
        const f16* gaux = P.aux_g + (size_t)t * P.rg;
#pragma unroll
        for (int n = 0; n < GC; ++n) {
            const uint32_t c = min(p16 * 8 + 128 * n, P.rg - 8);         // clamped: columns beyond the rank are masked in the dot
            gw[n] = *(const f16x8*)(grow + c);
            gx[n] = *(const f16x8*)(gaux + c);
        }
        const uint32_t cv = min(p16 * 8, P.rv - 8);
        const f16* vrow = P.layer0 ? grow : (const f16*)((const uint8_t*)P.v2 + (size_t)chq * P.v2_rb);
        vw = *(const f16x8*)(vrow + (P.layer0 ? 0 : cv));
        vx = *(const f16x8*)((P.layer0 ? gaux : P.aux_v + (size_t)t * P.rv) + (P.layer0 ? 0 : cv));
    }
    const float w0 = (float)P.w0[ch], a0 = (float)P.a0[ch], kkw = (float)P.k_k[ch], kaw = (float)P.k_a[ch], rkw = (float)P.r_k[ch];
    const float kraw = (float)P.k[(size_t)t * D + ch], rraw = (float)P.r[(size_t)t * D + ch];
    float v = (float)P.v[(size_t)t * D + chq];
    // unconditional loads (round 2: each of these was `if (...) load; s_waitcnt vmcnt(0)` -- two serial memory round trips behind the LoRA
    // burst): layer 0 reads w0 / its own v_first slot as mapped dummies, a launch without shift source reads w0
    const f16 v0h = (P.layer0 ? P.w0 : P.v0)[chq], vfh = P.v_first[(size_t)t * D + chq];
    const f16 shh = (P.shift_src ? P.shift_src + (size_t)t * D + cq : P.w0 + cq)[min(tid, (uint32_t)C - 1)];
    const float v0w = P.layer0 ? 0.0f : (float)v0h, vfirst = P.layer0 ? 0.0f : (float)vfh;
    const float shift = (float)shh;

    // (2) the state slice: thread (wave, lane = j) holds S[j][cq + 4*wave .. +3]; requested last, consumed last
    const uint32_t batch = P.batch1 ? P.batch1 - 1 + t : (P.cursors[t] & 0xffu);
    float* st = P.state + ((size_t)batch * (S + 2) + 1 + lane) * D + cq + 4 * wave;
    f32x4 Sv = *(const f32x4*)st;

    {
        const float dw = lora_dot<4>(lw, P.rw, part);
        const float da = lora_dot<4>(la, P.ra, part);
        if (part == 0) {
            const float w = r16(w0 + r16(dw));                                           // add(w0, w)
            const float a = r16(act_sigmoid(a0 + r16(da)));                              // add_activate(.., Sigmoid)
            const float kc = r16(kraw * (1.0f + (a - 1.0f) * kaw));                      // control_k_v7
            sh_w[row] = __expf(-0.606531f * act_sigmoid(w));
            sh_a[row] = a;
            sh_r[row] = rraw;
            sh_kk[row] = r16(kkw * kraw);                                                // mul(k_k, kk)
            sh_k[row] = kc;
            sh_rkr[row] = rkw * kc * rraw;                                               // time_first's summand
        }
        // gate / value-residual LoRA rows of the 16 own columns: 16 lanes per row
        float dg = 0.0f, dv = 0.0f;
#pragma unroll
        for (int n = 0; n < GC; ++n)
            if (p16 * 8 + 128 * n < P.rg) {
                dg = __builtin_amdgcn_fdot2(__builtin_shufflevector(gw[n], gw[n], 0, 1), __builtin_shufflevector(gx[n], gx[n], 0, 1), dg, false);
                dg = __builtin_amdgcn_fdot2(__builtin_shufflevector(gw[n], gw[n], 2, 3), __builtin_shufflevector(gx[n], gx[n], 2, 3), dg, false);
                dg = __builtin_amdgcn_fdot2(__builtin_shufflevector(gw[n], gw[n], 4, 5), __builtin_shufflevector(gx[n], gx[n], 4, 5), dg, false);
                dg = __builtin_amdgcn_fdot2(__builtin_shufflevector(gw[n], gw[n], 6, 7), __builtin_shufflevector(gx[n], gx[n], 6, 7), dg, false);
            }
        if (!P.layer0 && p16 * 8 < P.rv) {
            dv = __builtin_amdgcn_fdot2(__builtin_shufflevector(vw, vw, 0, 1), __builtin_shufflevector(vx, vx, 0, 1), dv, false);
            dv = __builtin_amdgcn_fdot2(__builtin_shufflevector(vw, vw, 2, 3), __builtin_shufflevector(vx, vx, 2, 3), dv, false);
            dv = __builtin_amdgcn_fdot2(__builtin_shufflevector(vw, vw, 4, 5), __builtin_shufflevector(vx, vx, 4, 5), dv, false);
            dv = __builtin_amdgcn_fdot2(__builtin_shufflevector(vw, vw, 6, 7), __builtin_shufflevector(vx, vx, 6, 7), dv, false);
        }
        // sums over the 16 lanes of a DPP row (all lanes of the row end up with the total)
        dg += dpp_f32<0xB1>(dg); dg += dpp_f32<0x4E>(dg); dg += dpp_f32<0x141>(dg); dg += dpp_f32<0x140>(dg);
        dv += dpp_f32<0xB1>(dv); dv += dpp_f32<0x4E>(dv); dv += dpp_f32<0x141>(dv); dv += dpp_f32<0x140>(dv);
        if (p16 == 0) {
            if (P.layer0) P.v_first[(size_t)t * D + chq] = (f16)v;                       // blit(att_v, att_v0)
            else {
                const float vv = r16(act_sigmoid(v0w + r16(dv)));
                v = r16(wgsl_mix(v, vfirst, vv));                                        // lerp(att_v0, att_v, att_vv, reversed)
            }
            sh_v[row16] = v;
            sh_g[row16] = r16(dg);
        }
    }
    if (P.shift_src && tid < C) P.state[(size_t)batch * (S + 2) * D + cq + tid] = shift;
    WRK_STAMP(P.dbg, 1);
    __syncthreads();
    if (wave == 0) {       // kk <- l2_norm(kk) over the head; a~ = -kk, b~ = kk * a; xx = sum_j r_k k r
        const float kkv = sh_kk[lane];
        const float nrm = 1.0f / sqrtf(wave_sum(kkv * kkv) + P.l2_eps);
        const float kkn = r16(kkv * nrm);
        const float a = sh_a[lane];
        sh_a[lane] = -kkn;
        sh_b[lane] = kkn * a;
        const float xx = wave_sum(sh_rkr[lane]);
        if (lane == 0) sh_xx = xx;
    }
    __syncthreads();
    WRK_STAMP(P.dbg, 2);
    {
        const float aj = sh_a[lane], bj = sh_b[lane], wj = sh_w[lane], kj = sh_k[lane], rj = sh_r[lane];
        float sa[4], y[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) sa[c] = wave_sum(Sv[c] * aj);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s = Sv[c] * wj + kj * sh_v[4 * wave + c] + sa[c] * bj;
            Sv[c] = s;
            y[c] = rj * s;
        }
        *(f32x4*)st = Sv;
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = wave_sum(y[c]);
        WRK_STAMP(P.dbg, 3);
        if (lane < 4) {
            const float yv = lane == 0 ? y[0] : (lane == 1 ? y[1] : (lane == 2 ? y[2] : y[3]));
            const uint32_t i = 4 * wave + lane;
            const size_t o = (size_t)t * D + cq + i;
            P.out[o] = (f16)yv;                                                           // att_x <- y (f16 store)
            tt_out[o] = sh_xx * sh_v[i];                                                  // time_first: + (sum_j r_k k r) * v, added after the group norm
            g_out[o] = (f16)sh_g[i];
        }
    }
    WRK_STAMP(P.dbg, 4);
}

// ------------------------------------------------------------------ greedy sampling, stage 2
// Reduces the per-workgroup (max, first index) partials written by the head matvec, stores the token,
// and (optionally) advances the device-resident generation loop: tokens <- argmax, history, counter.
__global__ void __launch_bounds__(256) argmax_finish_kernel(const float* __restrict__ pv, const uint32_t* __restrict__ pi, uint32_t nwg,
                                                             uint32_t ntok, uint32_t* __restrict__ argmax, uint32_t* __restrict__ tokens,
                                                             uint32_t* __restrict__ history, uint32_t* __restrict__ counter) {
    __shared__ float sv[4];
    __shared__ uint32_t si[4];
    const uint32_t step = counter ? *counter : 0;
    for (uint32_t n = 0; n < ntok; ++n) {
        float bv = -3.0e38f;
        uint32_t bi = 0xffffffffu;
        for (uint32_t w = threadIdx.x; w < nwg; w += 256) {
            const float v = pv[(size_t)w * ntok + n];
            const uint32_t i2 = pi[(size_t)w * ntok + n];
            if (v > bv || (v == bv && i2 < bi)) { bv = v; bi = i2; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, WAVE);
            const uint32_t oi = __shfl_xor(bi, o, WAVE);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k)
                if (sv[k] > bv || (sv[k] == bv && si[k] < bi)) { bv = sv[k]; bi = si[k]; }
            if (bi == 0xffffffffu) bi = 0;
            argmax[n] = bi;
            if (tokens) { tokens[n] = bi; history[(size_t)step * ntok + n] = bi; }
        }
    }
    if (counter && threadIdx.x == 0) *counter = step + 1;
}

void argmax_finish(hipStream_t s, const float* pv, const uint32_t* pi, uint32_t nwg, uint32_t ntok, uint32_t* argmax, uint32_t* tokens,
                   uint32_t* history, uint32_t* counter) {
    argmax_finish_kernel<<<1, 256, 0, s>>>(pv, pi, nwg, ntok, argmax, tokens, history, counter);
}

}  // namespace wrk

// ------------------------------------------------------------------ debug: in-kernel timeline of one layer (WRK_TIMING=1)
static constexpr uint32_t TIMED_LAYER = 5;
namespace {
struct TimingState {
    unsigned long long* dev = nullptr;
    std::vector<std::string> labels;
    bool enabled = false, init = false;
} g_timing;
}
unsigned long long* wrk::timing_slot(wrk_ctx* ctx, const char* label) {
    if (!g_timing.init) { const char* e = getenv("WRK_TIMING"); g_timing.enabled = e && e[0] == '1'; g_timing.init = true; }
    if (!g_timing.enabled) return nullptr;
    if (!g_timing.dev) {        // first call comes from wrk_v7_generate_greedy BEFORE any stream capture (label == nullptr)
        if (hipMalloc((void**)&g_timing.dev, 64 * 16 * 8) != hipSuccess) return nullptr;
        hipMemset(g_timing.dev, 0, 64 * 16 * 8);
    }
    if (!label) return nullptr;
    for (size_t i = 0; i < g_timing.labels.size(); ++i)
        if (g_timing.labels[i] == label) return g_timing.dev + i * 16;
    if (g_timing.labels.size() >= 64) return nullptr;
    g_timing.labels.push_back(label);
    return g_timing.dev + (g_timing.labels.size() - 1) * 16;
}
void wrk::timing_report(wrk_ctx* ctx) {
    if (!g_timing.enabled || !g_timing.dev || g_timing.labels.empty()) return;
    std::vector<unsigned long long> h(64 * 16);
    hipMemcpy(h.data(), g_timing.dev, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (size_t i = 0; i < g_timing.labels.size() * 16; ++i) if (h[i] && h[i] < t0) t0 = h[i];
    fprintf(stderr, "[WRK_TIMING] layer %u of the last decode step; ns since the first stamp (100 MHz clock); first WG | last WG\n", TIMED_LAYER);
    for (size_t i = 0; i < g_timing.labels.size(); ++i) {
        fprintf(stderr, "  %-40s", g_timing.labels[i].c_str());
        for (int w = 0; w < 2; ++w) {
            for (int k = 0; k < 5; ++k) {
                const unsigned long long v = h[i * 16 + w * 8 + k];
                if (v) fprintf(stderr, " %6llu", (v - t0) * 10); else fprintf(stderr, "      -");
            }
            if (w == 0) fprintf(stderr, "  |");
        }
        fprintf(stderr, "\n");
    }
}

// ------------------------------------------------------------------ host: enqueue one fused decode step
void wrk_v7_model::drop_graphs() {
    for (auto& kv : graphs) wrk_program_destroy(kv.second);
    graphs.clear();
}

void wrk_v7_model::free_fused() {
    if (amax_val) hipFree(amax_val);
    if (amax_idx) hipFree(amax_idx);
    amax_val = nullptr; amax_idx = nullptr; amax_cap = 0;
}

static wrk::MatJob job(const wrk_matrix* m, DTensor in, DTensor out, uint32_t act) {
    wrk::MatJob j{m->data, m->aux, m->kind, m->flags, m->k, m->m, (uint32_t)m->row_bytes, in, out, act, 0};
    j.scale = m->out_scale;
    return j;
}

int32_t wrk_v7_model::enqueue_fused_decode(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity_headers, bool from_tokens,
                                            bool want_argmax, bool advance, uint32_t cursor0_batch, bool contiguous) {
    using namespace wrk;
    hipStream_t q = ctx->op_stream();
    const uint32_t D = d.num_emb, F = d.num_hidden, H = d.num_head, S = D / H, V = d.num_vocab;
    // what the fused kernels assume; anything else takes the op-by-op path
    bool ok = (D % 8 == 0) && D <= 8192 && d.lora_w % 8 == 0 && d.lora_a % 8 == 0 && d.lora_g % 8 == 0 && d.lora_v % 8 == 0 &&
              d.lora_w <= 128 && d.lora_a <= 128 && d.lora_v <= 128 && d.lora_g <= 512;
    for (auto& L : layers) {
        const wrk_matrix* ms[] = {L.w2, L.a2, L.g2, L.v2};     // Matrix::Fp16 in the reference (v7.rs:1108-1113)
        for (const wrk_matrix* m : ms)
            if (m && m->kind != WRK_MAT_F16) ok = false;
    }
    if (!ok) {
        if (from_tokens) gather_rows_f16(q, emb->ptr, s.tokens, s.input, D, T);
        int32_t rc = enqueue_ops(st, T, NH, identity_headers);
        if (rc != WRK_OK) return rc;
        if (want_argmax && NH) {
            argmax_rows(q, s.head_o, V, V, NH, s.argmax);
            if (advance) advance_tokens(q, s.argmax, s.tokens, history, s.counter, NH);
        }
        return WRK_OK;
    }
    auto vec = [&](void* p, uint32_t c = 0) { return make_dense(p, WRK_F16, c ? c : D, T); };
    // batched decode (many sequences): each matrix goes to the MFMA GEMM; few sequences: one multi-matrix matvec launch
    bool single = false;        // this layer runs the 5-launch structure (prologue-fused dmv kernels)
    auto run_jobs = [&](MatJob* jobs, int n) -> int {
        if (!single && T >= gemm_min_tokens()) {
            jobs[0].ks_part = s.ks_part; jobs[0].ks_cnt = s.ks_cnt; jobs[0].ks_part_cap = s.ks_part_cap; jobs[0].ks_cnt_cap = s.ks_cnt_cap;
            jobs[0].xsum = ctx->gemm_scratch; jobs[0].xsum_cap = ctx->gemm_scratch_cap;
            if (matmul_mfma_multi(q, jobs, n, ctx->num_cu) == 0) return 0;      // all matrices of the stage in one launch
            for (int i = 0; i < n; ++i) {
                int rc = matmul_mfma(q, jobs[i], ctx->num_cu);
                if (rc == -2) rc = matvec(q, &jobs[i], 1, ctx->num_cu);
                if (rc != 0) return rc;
            }
            return 0;
        }
        return matvec_grouped(q, jobs, n, ctx->num_cu);
    };
#define LN(P, n)                                                                             \
    do {                                                                                     \
        if (ln_mix(q, P, n) != 0) return wrk_fail(ctx, WRK_E_UNSUPPORTED, "ln_mix shape");   \
    } while (0)

    // the K-sliced GEMM's arrival counters must be zero when a step starts; its kernels leave them zero, but a launch that faulted or was
    // aborted would poison every later replay of this frame -- so the step program zeroes them first (a memset node, ~1 KB)
    if (T >= 2 && s.ks_cnt && s.ks_cnt_cap) WRK_HIP(ctx, hipMemsetAsync(s.ks_cnt, 0, (size_t)s.ks_cnt_cap * 4, q));
    // embed: gather (device table) + LN(ln0) -> x   (v7.rs:438-474, 649-659).  One sequence's token on the engine: the engine takes the row from
    // its table of normalised embedding rows itself (wrk_v7_engine.hip) -- no launch here
    const bool eng_embeds = T == 1 && from_tokens && !skip_embed && layer_begin == 0 && engine_on() && !engine_skip_once && wrk_v7_engine_has_table(engine) &&
                            d.num_layer > 0;
    if (!skip_embed && !eng_embeds) {
        LnMixParams P{};
        if (from_tokens) { P.src = (const f16*)emb->ptr; P.ids = s.tokens; }
        else P.src = (const f16*)s.input;
        P.ln_w = (const f16*)ln0_w->ptr; P.ln_b = (const f16*)ln0_b->ptr; P.eps = 1.0e-5f;
        P.d = D; P.nmix = 0; P.ln_out = (f16*)s.x;
        LN(P, T);
    }
    // one sequence, one token: the persistent engine takes every layer of the step in ONE launch (wrk_v7_engine.hip)
    uint32_t first_launch_layer = layer_begin;
    if (T == 1 && engine_on() && !engine_skip_once) {
        const uint32_t l1 = std::min<uint32_t>(d.num_layer, layer_end);
        if (layer_begin < l1) {
            const int32_t rc = wrk_v7_engine_enqueue(engine, q, st, cursor0_batch, layer_begin, l1, s.x, s.x, s.att_v0, eng_embeds ? s.tokens : nullptr);
            if (rc != WRK_OK) return rc;
        }
        first_launch_layer = l1;
    }
    for (uint32_t li = first_launch_layer; li < d.num_layer && li < layer_end; ++li) {
        const wrk_v7_layer_desc& L = layers[li];
        float* lst = st->layer_ptr(li);
        // batch-1 decode folds LN + token shift into the prologue of the matvec that consumes them (5 launches per layer
        // instead of 7; 0.865 vs 0.963 ms/token on MI355X, round 1).  The prologue lives in the register-input kernels
        // only: a dry run of the three launches decides per layer; WRK_FUSE_LN=0 forces the 7-launch path.
        static const bool fuse_ln = [] { const char* e = getenv("WRK_FUSE_LN"); return !(e && e[0] == '0'); }();
        // 2 .. 4 sequences (round 2): the same structure on the dmv kernels' multi-token instantiations when the sequences' state
        // rows are a constant stride apart (token t = batch cursor0_batch + t, host-checked); WRK_DMV_TOKENS=1 restores the MFMA path
        static const uint32_t few_max = [] { const char* e = getenv("WRK_DMV_TOKENS"); const int v = e ? atoi(e) : 4; return (uint32_t)(v < 1 ? 1 : (v > 4 ? 4 : v)); }();
        single = fuse_ln && (T == 1 || (T <= few_max && contiguous)) && D <= 4096;
        if (single) {
            MatJob k1[7] = {job(L.w_r, vec(s.x), vec(s.r), 0), job(L.w_k, vec(s.x), vec(s.k), 0), job(L.w_v, vec(s.x), vec(s.v), 0),
                            job(L.w1, vec(s.x), vec(s.aux_w, d.lora_w), 0), job(L.a1, vec(s.x), vec(s.aux_a, d.lora_a), 0),
                            job(L.g1, vec(s.x), vec(s.aux_g, d.lora_g), 0), job(li ? L.v1 : L.a1, vec(s.x), vec(s.aux_v, d.lora_v), 0)};
            for (MatJob& j : k1) j.pro = 1;
            MatJob k5 = job(L.ffn_w_k, vec(s.x), vec(s.ffn_k, F), 0);
            k5.pro = 1;
            MatJob k6 = job(L.ffn_w_v, vec(s.ffn_k, F), vec(s.x), 0);
            k6.has_res = 1; k6.res = vec(s.x);
            k6.carry_dst = (float*)s.x;     // any non-null pointer: classification only
            single = matvec_grouped(q, k1, li ? 7 : 6, ctx->num_cu, true) == 0 && matvec(q, &k5, 1, ctx->num_cu, true) == 0 &&
                     matvec(q, &k6, 1, ctx->num_cu, true) == 0;
        }
        // split head (4 workgroups per head, group norm in W_o's prologue): batch-1 decode whose W_o launch the dmv kernels take
        // (a model that has the persistent engine computes one-token jobs with ONE workgroup per head there; wherever such a job falls back
        // to the launches -- concurrent pipelines -- it keeps that arithmetic, so that the results do not depend on which of the two ran;
        // the layer inspection entry point keeps the split head, whose hand-over buffers its callers read)
        const bool want_split = split_head_env_on() && !(T == 1 && engine != nullptr && engine_env_on_public() && !engine_skip_once);
        const uint32_t state_stride = (S + 2) * D;       // floats between the state rows of consecutive sequences
        bool split_head = want_split && single && d.lora_w >= 8 && d.lora_a >= 8 && d.lora_g >= 8 && d.lora_v >= 8 && d.lora_g <= 512;
        if (split_head) {
            MatJob k3 = job(L.w_o, vec(s.att_x), vec(s.x), 0);
            k3.has_res = 1; k3.res = vec(s.x);
            k3.pro = 2; k3.ln_w = L.gn_w->ptr; k3.ln_b = L.gn_b->ptr; k3.mixw = s.g; k3.prev = (const float*)s.n;
            split_head = matvec(q, &k3, 1, ctx->num_cu, true) == 0;
        }
        // (next-launch weight prefetch -- every launch touching the lines its successor will stream -- was built and measured in
        // round 2: every edge made the token slower, profiles/r02_prefetch_ab.txt; HBM streaming is not what bounds batch 1)
        uint32_t batch0 = 0;
        if (single) batch0 = cursor0_batch;
        float* row0 = lst + (size_t)batch0 * (S + 2) * D;                    // att shift state of the sequence
        if (!single) {   // K0
            LnMixParams P{};
            P.src = (const f16*)s.x; P.ln_w = (const f16*)L.ln1_w->ptr; P.ln_b = (const f16*)L.ln1_b->ptr; P.eps = 1.0e-5f;
            P.d = D; P.nmix = 6;
            const wrk_buf* mx[6] = {L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g};
            void* outs[6] = {s.rx, s.wx, s.kx, s.vx, s.ax, s.gx};
            for (int i = 0; i < 6; ++i) { P.mix[i] = (const f16*)mx[i]->ptr; P.out[i] = (f16*)outs[i]; }
            P.state_row = lst; P.state_stride = (size_t)(S + 2) * D; P.cursors = s.cursors; P.batch1 = contiguous ? cursor0_batch + 1 : 0;
            LN(P, T);
        }
        {   // K1
            MatJob jobs[7] = {job(L.w_r, vec(s.rx), vec(s.r), WRK_ACT_NONE), job(L.w_k, vec(s.kx), vec(s.k), WRK_ACT_NONE),
                              job(L.w_v, vec(s.vx), vec(s.v), WRK_ACT_NONE),
                              job(L.w1, vec(s.wx), vec(s.aux_w, d.lora_w), WRK_ACT_TANH),
                              job(L.a1, vec(s.ax), vec(s.aux_a, d.lora_a), WRK_ACT_NONE),
                              job(L.g1, vec(s.gx), vec(s.aux_g, d.lora_g), WRK_ACT_SIGMOID),
                              job(li ? L.v1 : L.a1, vec(s.vx), vec(s.aux_v, d.lora_v), WRK_ACT_NONE)};
            if (single) {
                const wrk_buf* mx[7] = {L.x_r, L.x_k, L.x_v, L.x_w, L.x_a, L.x_g, L.x_v};
                for (int i = 0; i < 7; ++i) {
                    jobs[i].in = vec(s.x);
                    jobs[i].pro = 1; jobs[i].pro_eps = 1.0e-5f; jobs[i].ln_w = L.ln1_w->ptr; jobs[i].ln_b = L.ln1_b->ptr;
                    jobs[i].mixw = mx[i]->ptr; jobs[i].prev = row0; jobs[i].tok_prev_stride = state_stride;
                }
                jobs[0].ln_out = s.ln_tmp;      // LN(x): becomes the att shift state in K2
            }
            if (li == TIMED_LAYER) jobs[0].dbg = jobs[(li ? 7 : 6) - 1].dbg = wrk::timing_slot(ctx, "K1 r,k,v + LoRA-1 (LN1 prologue)");
            const int rc = run_jobs(jobs, li ? 7 : 6);
            if (rc != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K1 rejected (%d)", rc);
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
            P.shift_src = single ? (const f16*)s.ln_tmp : nullptr;     // fused K0: the state carry happens here
            P.batch1 = (single || contiguous) ? cursor0_batch + 1 : 0;  // host-known batches: token t is batch cursor0_batch + t (no cursor load in front of the state loads)
            P.dbg = li == TIMED_LAYER ? wrk::timing_slot(ctx, "K2 head: LoRA-2 + WKV7 + group norm") : nullptr;
            if (split_head) {
                if (d.lora_g <= 256) head_split_kernel<2><<<dim3(H * 4, T), 256, 0, q>>>(P, (float*)s.n, (f16*)s.g, H);
                else head_split_kernel<4><<<dim3(H * 4, T), 256, 0, q>>>(P, (float*)s.n, (f16*)s.g, H);
            }
            else if (d.lora_g <= 256) head_kernel<8><<<dim3(H, T), 256, 0, q>>>(P);
            else head_kernel<16><<<dim3(H, T), 256, 0, q>>>(P);
        }
        {   // K3: x += W_o . att_x
            MatJob j = job(L.w_o, vec(s.att_x), vec(s.x), WRK_ACT_NONE);
            j.has_res = 1; j.res = vec(s.x);
            if (split_head) {       // group norm + time_first + gate of the split head kernel's hand-over, in the prologue
                j.pro = 2; j.pro_eps = 64.0e-5f; j.ln_w = L.gn_w->ptr; j.ln_b = L.gn_b->ptr; j.mixw = s.g; j.prev = (const float*)s.n;
                j.tok_mix_stride = D; j.tok_prev_stride = D;
            }
            if (li == TIMED_LAYER) j.dbg = wrk::timing_slot(ctx, "K3 w_o + residual");
            if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K3 rejected");
        }
        float* rowf = lst + ((size_t)batch0 * (S + 2) + (S + 1)) * D;          // ffn shift state of the sequence
        if (!single) {   // K4
            LnMixParams P{};
            P.src = (const f16*)s.x; P.ln_w = (const f16*)L.ln2_w->ptr; P.ln_b = (const f16*)L.ln2_b->ptr; P.eps = 1.0e-5f;
            P.d = D; P.nmix = 1; P.mix[0] = (const f16*)L.ffn_x_k->ptr; P.out[0] = (f16*)s.ffn_kx;
            P.state_row = lst + (size_t)(S + 1) * D; P.state_stride = (size_t)(S + 2) * D; P.cursors = s.cursors; P.batch1 = contiguous ? cursor0_batch + 1 : 0;
            LN(P, T);
        }
        {   // K5
            MatJob j = job(L.ffn_w_k, vec(s.ffn_kx), vec(s.ffn_k, F), WRK_ACT_SQUARED_RELU);
            if (single) {
                j.in = vec(s.x);
                j.pro = 1; j.pro_eps = 1.0e-5f; j.ln_w = L.ln2_w->ptr; j.ln_b = L.ln2_b->ptr; j.mixw = L.ffn_x_k->ptr; j.prev = rowf;
                j.tok_prev_stride = state_stride;
                j.ln_out = s.ffn_x;             // LN(x): becomes the ffn shift state in K6's epilogue
            }
            if (li == TIMED_LAYER) j.dbg = wrk::timing_slot(ctx, "K5 ffn key (LN2 prologue)");
            if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K5 rejected");
        }
        {   // K6: x += W_v . relu(k)^2
            MatJob j = job(L.ffn_w_v, vec(s.ffn_k, F), vec(s.x), WRK_ACT_NONE);
            j.has_res = 1; j.res = vec(s.x);
            if (single) { j.carry_src = s.ffn_x; j.carry_dst = rowf; j.tok_carry_src_stride = D; j.tok_carry_dst_stride = state_stride; }
            if (li == TIMED_LAYER) j.dbg = wrk::timing_slot(ctx, "K6 ffn value + residual");
            if (run_jobs(&j, 1) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused K6 rejected");
        }
        if ((li + 1) % d.rescale == 0) wrk::affine(q, vec(s.x), 0.5f, 0.0f);
    }
    if (NH > 0) {
        // header: rows -> LN(ln_out) -> head matmul (f32 logits) [+ arg-max partials in the same launch]
        LnMixParams P{};
        P.src = (const f16*)s.x; P.ids = identity_headers ? nullptr : s.headers;
        P.ln_w = (const f16*)ln_out_w->ptr; P.ln_b = (const f16*)ln_out_b->ptr; P.eps = 1.0e-5f;
        P.d = D; P.nmix = 0; P.ln_out = (f16*)s.head_x;
        LN(P, NH);
        MatJob j = job(head, make_dense(s.head_x, WRK_F16, D, NH), make_dense(s.head_o, WRK_F32, V, NH), WRK_ACT_NONE);
        uint32_t nwg = 0;
        // a few header rows: the multi-token dmv kernel streams the head once for all of them, arg-max partials fused
        bool head_mv = NH < gemm_min_tokens();
        if (!head_mv && NH <= 4) {
            MatJob probe = j;
            if (want_argmax) { probe.amax_val = amax_val; probe.amax_idx = amax_idx; }
            head_mv = (size_t)matvec_num_wg(&j, 1, ctx->num_cu, nullptr) * NH <= amax_cap && matvec(q, &probe, 1, ctx->num_cu, true, false, true) == 0;
        }
        // several header rows: the head goes to the matrix cores too and the arg-max becomes its own (tiny) kernel
        if (!head_mv && matmul_mfma(q, j, ctx->num_cu) == 0) {
            if (want_argmax) {
                wrk::argmax_rows(q, (const float*)s.head_o, V, V, NH, s.argmax);
                if (advance) wrk::advance_tokens(q, s.argmax, s.tokens, history, s.counter, NH);
            }
        } else {
            if (want_argmax) {
                nwg = matvec_num_wg(&j, 1, ctx->num_cu, nullptr);
                if ((size_t)nwg * NH > amax_cap) return wrk_fail(ctx, WRK_E_ARG, "arg-max scratch too small");
                j.amax_val = amax_val; j.amax_idx = amax_idx;
            }
            if (matvec(q, &j, 1, ctx->num_cu) != 0) return wrk_fail(ctx, WRK_E_ARG, "fused head rejected");
            if (want_argmax)
                argmax_finish(q, amax_val, amax_idx, nwg, NH, s.argmax, advance ? s.tokens : nullptr, history, advance ? s.counter : nullptr);
        }
    }
#undef LN
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}
