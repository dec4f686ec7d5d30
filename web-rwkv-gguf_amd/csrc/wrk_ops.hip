// Op-by-op kernels: one kernel per reference TensorOp on the V7/V6 path.
// These are the "mode 0" path of wrk_v7_infer, the prefill path's elementwise layer, and the
// kernels behind the wrk_op_* entry points.  The decode fast path fuses them (wrk_v7_fused.hip).
//
// All arithmetic is f32; loads/stores convert to the buffer dtype (f16 for Runtime<f16> buffers,
// f32 for state / logits), which reproduces the reference's rounding points.
#include "wrk_device.h"

namespace wrk {

// ------------------------------------------------------------------ layer_norm / group_norm
// shaders/layer_norm.wgsl:63-121.  One workgroup per row; statistics two-pass in f32
// (mean, then centred second moment: same quantities as the shader's Welford merge).
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) layer_norm_kernel(const f16* __restrict__ w, const f16* __restrict__ b, DTensor src, DTensor x,
                                                            float eps, int group) {
    __shared__ float red[BLOCK / 64];
    const uint32_t C = x.shape[0];
    const uint32_t token = blockIdx.x, batch = blockIdx.y;
    const size_t base = dt_index(x, 0, token, batch), sbase = dt_index(src, 0, token, batch);      // src == x: in place
    const uint32_t wofs = group ? token * C : 0;    // GROUP_NORM: h = token * stride
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < C; i += BLOCK) s += dt_load(src, sbase + i);
    const float mean = block_sum<BLOCK / 64>(s, red) / (float)C;
    float q = 0.0f;
    for (uint32_t i = threadIdx.x; i < C; i += BLOCK) { float d = dt_load(src, sbase + i) - mean; q += d * d; }
    const float var = block_sum<BLOCK / 64>(q, red) / (float)C + eps;
    const float dev = 1.0f / sqrtf(var);
    for (uint32_t i = threadIdx.x; i < C; i += BLOCK) {
        float value = (dt_load(src, sbase + i) - mean) * dev;
        dt_store(x, base + i, __builtin_fmaf(value, (float)w[wofs + i], (float)b[wofs + i]));
    }
}

// The same arithmetic for f16 rows of up to 256 NV channels (round 3): thread i still owns channels i, i + 256, ... and adds them in that order --
// the sums are bit-identical to the kernel above -- but every operand is requested ONCE, up front and unconditionally, and kept in registers
// for the three passes (the generic kernel re-loads the row per pass through dt_load, whose branch on the element type makes each load a
// memory round trip of its own: 8.8 us per 128-token launch in pp512, a quarter of it after this).
template <int NV>
__global__ void __launch_bounds__(256) layer_norm_f16_kernel(const f16* __restrict__ w, const f16* __restrict__ b, DTensor src, DTensor x, float eps) {
    __shared__ float red[4];
    const uint32_t C = x.shape[0];
    const uint32_t token = blockIdx.x, batch = blockIdx.y, tid = threadIdx.x;
    const f16* sp = (const f16*)src.p + dt_index(src, 0, token, batch);
    f16* xp = (f16*)x.p + dt_index(x, 0, token, batch);
    f16 v[NV], wv[NV], bv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { const uint32_t i = min(tid + 256u * k, C - 1); v[k] = sp[i]; wv[k] = w[i]; bv[k] = b[i]; }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) if (tid + 256u * k < C) s += (float)v[k];
    const float mean = block_sum<4>(s, red) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) if (tid + 256u * k < C) { const float d = (float)v[k] - mean; q += d * d; }
    const float var = block_sum<4>(q, red) / (float)C + eps;
    const float dev = 1.0f / sqrtf(var);
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (tid + 256u * k < C) xp[tid + 256u * k] = (f16)__builtin_fmaf(((float)v[k] - mean) * dev, (float)wv[k], (float)bv[k]);
}
static bool ln_dense_f16(const DTensor& d) { return d.dtype == WRK_F16 && d.stride[0] == d.shape[0] && d.offset[0] == 0; }
static void layer_norm_launch(hipStream_t s, const void* w, const void* b, DTensor src, DTensor x, float eps) {
    dim3 grid(x.shape[1], x.shape[2]);
    if (grid.x == 0 || grid.y == 0) return;
    const uint32_t C = x.shape[0];
    static const bool fast = [] { const char* e = getenv("WRK_LN_FAST"); return !(e && e[0] == '0'); }();
    if (fast && ln_dense_f16(src) && ln_dense_f16(x) && src.shape[0] == C && C <= 4096) {
        if (C <= 2048) layer_norm_f16_kernel<8><<<grid, 256, 0, s>>>((const f16*)w, (const f16*)b, src, x, eps);
        else layer_norm_f16_kernel<16><<<grid, 256, 0, s>>>((const f16*)w, (const f16*)b, src, x, eps);
        return;
    }
    layer_norm_kernel<256><<<grid, 256, 0, s>>>((const f16*)w, (const f16*)b, src, x, eps, 0);
}

void layer_norm(hipStream_t s, const void* w, const void* b, DTensor x, float eps) { layer_norm_launch(s, w, b, x, x, eps); }

// blit(src, x) + layer_norm(x) in one pass (the copy is exact, so the result is the same)
void layer_norm_from(hipStream_t s, const void* w, const void* b, DTensor src, DTensor x, float eps) { layer_norm_launch(s, w, b, src, x, eps); }

void group_norm(hipStream_t s, const void* w, const void* b, DTensor x, float eps) {
    // x [S, H, T]: "token" = head, "batch" = token (ops.rs:460-508)
    dim3 grid(x.shape[1], x.shape[2]);
    if (grid.x == 0 || grid.y == 0) return;
    layer_norm_kernel<64><<<grid, 64, 0, s>>>((const f16*)w, (const f16*)b, x, x, eps, 1);
}

// ------------------------------------------------------------------ l2_norm (normalize.wgsl:117-152)
__global__ void __launch_bounds__(64) l2_norm_kernel(DTensor x, float eps) {
    const uint32_t C = x.shape[0];
    const size_t base = dt_index(x, 0, blockIdx.x, blockIdx.y);
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < C; i += 64) { float v = dt_load(x, base + i); s += v * v; }
    s = wave_sum(s);
    const float norm = 1.0f / sqrtf(s + eps);
    for (uint32_t i = threadIdx.x; i < C; i += 64) dt_store(x, base + i, dt_load(x, base + i) * norm);
}

void l2_norm(hipStream_t s, DTensor x, float eps) {
    dim3 grid(x.shape[1], x.shape[2]);
    if (grid.x == 0 || grid.y == 0) return;
    l2_norm_kernel<<<grid, 64, 0, s>>>(x, eps);
}

// ------------------------------------------------------------------ 8-wide fast paths
// The element-per-thread kernels below follow the reference shaders one to one and take any view.  The model's own
// buffers are dense f16 rows: for those a thread owns 8 consecutive channels (one 16-byte access per operand), which
// is what the HBM-bound multi-token passes need (rocprof, 4096 stacked tokens: 17-25 us per pass vs ~7 us of bytes).
static bool vec8_ok(const DTensor& d) {
    return d.dtype == WRK_F16 && (d.shape[0] & 7u) == 0 && (d.offset[0] & 7u) == 0 && (d.stride[0] & 7u) == 0 && (((uintptr_t)d.p) & 15u) == 0;
}
__device__ __forceinline__ f16x8 ld8(const DTensor& d, size_t idx) { return *(const f16x8*)((const f16*)d.p + idx); }
__device__ __forceinline__ void st8(const DTensor& d, size_t idx, f16x8 v) { *(f16x8*)((f16*)d.p + idx) = v; }

// ------------------------------------------------------------------ token_shift (token_shift.wgsl:85-117)
// mix is [C, A?, I]: one factor vector (A? == 1) or one per stacked token (V6's data-dependent shift), and I
// outputs per call (count axis = blockIdx.z), written to out [C, T, I].
__global__ void __launch_bounds__(256) token_shift_kernel(const uint32_t* __restrict__ cursors, DTensor mixw, DTensor st, DTensor in,
                                                           DTensor out, int reversed) {
    const uint32_t C = in.shape[0];
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stack = blockIdx.y, count = blockIdx.z;
    if (c >= C) return;
    const Cursor cur = unpack_cursor(cursors[stack]);
    const float f = dt_load(mixw, dt_index(mixw, c, mixw.shape[1] == 1 ? 0 : stack, count));
    const float xt = dt_load(in, dt_index(in, c, stack, 0));
    const float prev = (stack == cur.token) ? dt_load(st, dt_index(st, c, 0, cur.batch))
                                            : dt_load(in, dt_index(in, c, stack - 1, 0));
    const float v = reversed ? wgsl_mix(xt, prev, f) : wgsl_mix(prev, xt, f);
    dt_store(out, dt_index(out, c, stack, count), v);
}

__global__ void __launch_bounds__(256) token_shift_v8_kernel(const uint32_t* __restrict__ cursors, DTensor mixw, DTensor st, DTensor in,
                                                              DTensor out, int reversed) {
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) * 8;
    const uint32_t stack = blockIdx.y, count = blockIdx.z;
    if (c >= in.shape[0]) return;
    const Cursor cur = unpack_cursor(cursors[stack]);
    const f16x8 f = ld8(mixw, dt_index(mixw, c, mixw.shape[1] == 1 ? 0 : stack, count));
    const f16x8 xt = ld8(in, dt_index(in, c, stack, 0));
    float prev[8];
    if (stack == cur.token) {
        const float* sp = (const float*)st.p + dt_index(st, c, 0, cur.batch);
        const f32x4 p0 = *(const f32x4*)sp, p1 = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { prev[e] = p0[e]; prev[4 + e] = p1[e]; }
    } else {
        const f16x8 pv = ld8(in, dt_index(in, c, stack - 1, 0));
#pragma unroll
        for (int e = 0; e < 8; ++e) prev[e] = (float)pv[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)(reversed ? wgsl_mix((float)xt[e], prev[e], (float)f[e]) : wgsl_mix(prev[e], (float)xt[e], (float)f[e]));
    st8(out, dt_index(out, c, stack, count), o);
}

void token_shift(hipStream_t s, const uint32_t* cursors, DTensor mixw, DTensor st, DTensor in, DTensor out, int reversed) {
    if (in.shape[1] == 0) return;
    const uint32_t nz = out.shape[2] ? out.shape[2] : 1;
    if (vec8_ok(mixw) && vec8_ok(in) && vec8_ok(out) && st.dtype == WRK_F32 && (st.offset[0] & 3u) == 0 && (st.stride[0] & 3u) == 0 &&
        (((uintptr_t)st.p) & 15u) == 0) {
        token_shift_v8_kernel<<<dim3((in.shape[0] / 8 + 255) / 256, in.shape[1], nz), 256, 0, s>>>(cursors, mixw, st, in, out, reversed);
        return;
    }
    dim3 grid((in.shape[0] + 255) / 256, in.shape[1], nz);
    token_shift_kernel<<<grid, 256, 0, s>>>(cursors, mixw, st, in, out, reversed);
}

// Several token_shift ops over the SAME input and state row in one pass (the six att shifts of an RWKV-7 layer,
// v7.rs:760-790): x and its predecessor are read once, every output is the same wgsl_mix expression as above.
struct ShiftSet { DTensor mixw[6]; DTensor out[6]; int n; };
__global__ void __launch_bounds__(256) token_shift_multi_v8_kernel(const uint32_t* __restrict__ cursors, ShiftSet S, DTensor st, DTensor in, int reversed) {
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) * 8;
    const uint32_t stack = blockIdx.y;
    if (c >= in.shape[0]) return;
    const Cursor cur = unpack_cursor(cursors[stack]);
    const f16x8 xt = ld8(in, dt_index(in, c, stack, 0));
    float prev[8];
    if (stack == cur.token) {
        const float* sp = (const float*)st.p + dt_index(st, c, 0, cur.batch);
        const f32x4 p0 = *(const f32x4*)sp, p1 = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { prev[e] = p0[e]; prev[4 + e] = p1[e]; }
    } else {
        const f16x8 pv = ld8(in, dt_index(in, c, stack - 1, 0));
#pragma unroll
        for (int e = 0; e < 8; ++e) prev[e] = (float)pv[e];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (i >= S.n) break;
        const f16x8 f = ld8(S.mixw[i], dt_index(S.mixw[i], c, 0, 0));
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(reversed ? wgsl_mix((float)xt[e], prev[e], (float)f[e]) : wgsl_mix(prev[e], (float)xt[e], (float)f[e]));
        st8(S.out[i], dt_index(S.out[i], c, stack, 0), o);
    }
}

void token_shift_multi(hipStream_t s, const uint32_t* cursors, const DTensor* mixw, const DTensor* out, int n, DTensor st, DTensor in, int reversed) {
    if (in.shape[1] == 0 || n <= 0) return;
    bool fast = n <= 6 && vec8_ok(in) && st.dtype == WRK_F32 && (st.offset[0] & 3u) == 0 && (st.stride[0] & 3u) == 0 && (((uintptr_t)st.p) & 15u) == 0;
    for (int i = 0; i < n && fast; ++i) fast = vec8_ok(mixw[i]) && vec8_ok(out[i]) && mixw[i].shape[1] == 1 && out[i].shape[2] <= 1;
    if (!fast) {
        for (int i = 0; i < n; ++i) token_shift(s, cursors, mixw[i], st, in, out[i], reversed);
        return;
    }
    ShiftSet S;
    S.n = n;
    for (int i = 0; i < 6; ++i) { S.mixw[i] = mixw[i < n ? i : 0]; S.out[i] = out[i < n ? i : 0]; }
    token_shift_multi_v8_kernel<<<dim3((in.shape[0] / 8 + 255) / 256, in.shape[1]), 256, 0, s>>>(cursors, S, st, in, reversed);
}

// ------------------------------------------------------------------ add / mul (binary.wgsl:38-78)
__global__ void __launch_bounds__(256) binary_kernel(int is_mul, DTensor in, DTensor out, uint32_t ax, uint32_t ay, uint32_t ao) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t t = blockIdx.y, b = blockIdx.z;
    if (c >= out.shape[0]) return;
    const float x = dt_load(in, dt_index(in, c, in.shape[1] == 1 ? 0 : t, in.shape[2] == 1 ? 0 : b));
    const size_t o = dt_index(out, c, t, b);
    const float y = dt_load(out, o);
    const float xv = act_apply(ax, x), yv = act_apply(ay, y);
    dt_store(out, o, act_apply(ao, is_mul ? xv * yv : xv + yv));
}

__global__ void __launch_bounds__(256) binary_v8_kernel(int is_mul, DTensor in, DTensor out, uint32_t ax, uint32_t ay, uint32_t ao) {
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) * 8;
    const uint32_t t = blockIdx.y, b = blockIdx.z;
    if (c >= out.shape[0]) return;
    const f16x8 x = ld8(in, dt_index(in, c, in.shape[1] == 1 ? 0 : t, in.shape[2] == 1 ? 0 : b));
    const size_t o = dt_index(out, c, t, b);
    const f16x8 y = ld8(out, o);
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xv = act_apply(ax, (float)x[e]), yv = act_apply(ay, (float)y[e]);
        r[e] = (f16)act_apply(ao, is_mul ? xv * yv : xv + yv);
    }
    st8(out, o, r);
}

void binary(hipStream_t s, int is_mul, DTensor in, DTensor out, uint32_t ax, uint32_t ay, uint32_t ao) {
    if (out.shape[1] == 0 || out.shape[2] == 0) return;
    if (vec8_ok(in) && vec8_ok(out)) {
        binary_v8_kernel<<<dim3((out.shape[0] / 8 + 255) / 256, out.shape[1], out.shape[2]), 256, 0, s>>>(is_mul, in, out, ax, ay, ao);
        return;
    }
    dim3 grid((out.shape[0] + 255) / 256, out.shape[1], out.shape[2]);
    binary_kernel<<<grid, 256, 0, s>>>(is_mul, in, out, ax, ay, ao);
}

// ------------------------------------------------------------------ lerp (lerp.wgsl:74-92)
__global__ void __launch_bounds__(256) lerp_kernel(DTensor x, DTensor y, DTensor f, int reversed) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t t = blockIdx.y, b = blockIdx.z;
    if (c >= y.shape[0]) return;
    const float fv = dt_load(f, dt_index(f, c, f.shape[1] == 1 ? 0 : t, f.shape[2] == 1 ? 0 : b));
    const float xv = dt_load(x, dt_index(x, c, t, b));
    const size_t o = dt_index(y, c, t, b);
    const float yv = dt_load(y, o);
    dt_store(y, o, reversed ? wgsl_mix(yv, xv, fv) : wgsl_mix(xv, yv, fv));
}

__global__ void __launch_bounds__(256) lerp_v8_kernel(DTensor x, DTensor y, DTensor f, int reversed) {
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) * 8;
    const uint32_t t = blockIdx.y, b = blockIdx.z;
    if (c >= y.shape[0]) return;
    const f16x8 fv = ld8(f, dt_index(f, c, f.shape[1] == 1 ? 0 : t, f.shape[2] == 1 ? 0 : b));
    const f16x8 xv = ld8(x, dt_index(x, c, t, b));
    const size_t o = dt_index(y, c, t, b);
    const f16x8 yv = ld8(y, o);
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (f16)(reversed ? wgsl_mix((float)yv[e], (float)xv[e], (float)fv[e]) : wgsl_mix((float)xv[e], (float)yv[e], (float)fv[e]));
    st8(y, o, r);
}

void lerp(hipStream_t s, DTensor x, DTensor y, DTensor f, int reversed) {
    if (y.shape[1] == 0 || y.shape[2] == 0) return;
    if (vec8_ok(x) && vec8_ok(y) && vec8_ok(f)) {
        lerp_v8_kernel<<<dim3((y.shape[0] / 8 + 255) / 256, y.shape[1], y.shape[2]), 256, 0, s>>>(x, y, f, reversed);
        return;
    }
    dim3 grid((y.shape[0] + 255) / 256, y.shape[1], y.shape[2]);
    lerp_kernel<<<grid, 256, 0, s>>>(x, y, f, reversed);
}

// ------------------------------------------------------------------ blit / affine / activate
__global__ void __launch_bounds__(256) blit_kernel(DTensor in, DTensor out) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t t = blockIdx.y, b = blockIdx.z;
    if (c >= out.shape[0]) return;
    dt_store(out, dt_index(out, c, t, b), dt_load(in, dt_index(in, c, t, b)));
}

__global__ void __launch_bounds__(256) blit_v8_kernel(DTensor in, DTensor out) {
    const uint32_t c = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c >= out.shape[0]) return;
    st8(out, dt_index(out, c, blockIdx.y, blockIdx.z), ld8(in, dt_index(in, c, blockIdx.y, blockIdx.z)));
}

void blit(hipStream_t s, DTensor in, DTensor out) {
    if (out.shape[1] == 0 || out.shape[2] == 0) return;
    if (vec8_ok(in) && vec8_ok(out)) {
        blit_v8_kernel<<<dim3((out.shape[0] / 8 + 255) / 256, out.shape[1], out.shape[2]), 256, 0, s>>>(in, out);
        return;
    }
    dim3 grid((out.shape[0] + 255) / 256, out.shape[1], out.shape[2]);
    blit_kernel<<<grid, 256, 0, s>>>(in, out);
}

__global__ void __launch_bounds__(256) affine_kernel(DTensor x, float scale, float bias, uint32_t act, int do_affine) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= x.shape[0]) return;
    const size_t o = dt_index(x, c, blockIdx.y, blockIdx.z);
    float v = dt_load(x, o);
    v = do_affine ? __builtin_fmaf(scale, v, bias) : act_apply(act, v);
    dt_store(x, o, v);
}

void affine(hipStream_t s, DTensor x, float scale, float bias) {
    if (x.shape[1] == 0 || x.shape[2] == 0) return;
    dim3 grid((x.shape[0] + 255) / 256, x.shape[1], x.shape[2]);
    affine_kernel<<<grid, 256, 0, s>>>(x, scale, bias, 0, 1);
}

void activate(hipStream_t s, DTensor x, uint32_t act) {
    if (x.shape[1] == 0 || x.shape[2] == 0) return;
    dim3 grid((x.shape[0] + 255) / 256, x.shape[1], x.shape[2]);
    affine_kernel<<<grid, 256, 0, s>>>(x, 0.f, 0.f, act, 0);
}

// ------------------------------------------------------------------ control_k_v7 (control_k_v7.wgsl:60-75)
__global__ void __launch_bounds__(256) control_k_kernel(const f16* __restrict__ p, DTensor a, DTensor k) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k.shape[0]) return;
    const float pv = (float)p[c];
    const float av = dt_load(a, dt_index(a, c, blockIdx.y, blockIdx.z));
    const size_t o = dt_index(k, c, blockIdx.y, blockIdx.z);
    const float kv = dt_load(k, o);
    dt_store(k, o, kv * (1.0f + (av - 1.0f) * pv));
}

void control_k_v7(hipStream_t s, const void* p, DTensor a, DTensor k) {
    if (k.shape[1] == 0 || k.shape[2] == 0) return;
    dim3 grid((k.shape[0] + 255) / 256, k.shape[1], k.shape[2]);
    control_k_kernel<<<grid, 256, 0, s>>>((const f16*)p, a, k);
}

// ------------------------------------------------------------------ time_mix_v7 (time_mix_v7.wgsl:143-221)
// One workgroup per (head, sequence-chunk).  The 64x64 f32 state of the head lives in registers for
// the whole chunk: thread (i = tid & 63, g = tid >> 6) owns S[16g .. 16g+15][i].  grid = (H, T):
// the workgroup whose stacked token index is the first of a sequence runs that sequence, the others
// exit, so sequences of one dispatch run concurrently (the reference serialises them).
//   w~ = exp(-0.606531 * sigmoid(w));  a~ = -kk;  b~ = kk * a
//   sa[i]  = sum_j S[j,i] * a~[j]
//   S[j,i] = S[j,i] * w~[j] + k[j] * v[i] + sa[i] * b~[j]
//   y[i]   = sum_j r[j] * S[j,i]
// r, w, x are [S, H, T]; n is [S, H, T, 4] = (k, v, a, kk); state view [C, S+1, B] rows: 0 = shift.
__global__ void __launch_bounds__(256) time_mix_v7_kernel(const uint32_t* __restrict__ cursors, DTensor st, DTensor r, DTensor w,
                                                           DTensor n, DTensor x) {
    constexpr int S = 64;
    __shared__ float sh_r[S], sh_w[S], sh_k[S], sh_a[S], sh_b[S];
    __shared__ float sh_red[4][S];
    const uint32_t head = blockIdx.x, t0 = blockIdx.y;
    const Cursor cur = unpack_cursor(cursors[t0]);
    if (cur.token != t0) return;                     // not the first token of a sequence chunk
    const uint32_t tid = threadIdx.x, i = tid & 63, g = tid >> 6;
    const uint32_t ch = head * S + i;                // channel of this thread's value column

    // token-shift carry: state row 0 <- att_x of the sequence's last token (read before it is overwritten)
    if (g == 0) {
        const uint32_t last = cur.token + cur.len - 1;
        dt_store(st, dt_index(st, ch, 0, cur.batch), dt_load(x, dt_index(x, i, head, last)));
    }
    float Sreg[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Sreg[jj] = dt_load(st, dt_index(st, ch, 1 + g * 16 + jj, cur.batch));

    for (uint32_t t = cur.token; t < cur.token + cur.len; ++t) {
        __syncthreads();
        if (g == 0) {
            sh_r[i] = dt_load(r, dt_index(r, i, head, t));
            sh_w[i] = __expf(-0.606531f * act_sigmoid(dt_load(w, dt_index(w, i, head, t))));
            sh_k[i] = dt_load(n, dt_index4(n, i, head, t, 0));
            const float a = dt_load(n, dt_index4(n, i, head, t, 2));
            const float kk = dt_load(n, dt_index4(n, i, head, t, 3));
            sh_a[i] = -kk;
            sh_b[i] = kk * a;
        }
        __syncthreads();
        float sa = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) sa = __builtin_fmaf(Sreg[jj], sh_a[g * 16 + jj], sa);
        sh_red[g][i] = sa;
        __syncthreads();
        sa = (sh_red[0][i] + sh_red[1][i]) + (sh_red[2][i] + sh_red[3][i]);
        const float vv = dt_load(n, dt_index4(n, i, head, t, 1));
        float y = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = g * 16 + jj;
            const float s = Sreg[jj] * sh_w[j] + sh_k[j] * vv + sa * sh_b[j];
            Sreg[jj] = s;
            y = __builtin_fmaf(sh_r[j], s, y);
        }
        __syncthreads();
        sh_red[g][i] = y;
        __syncthreads();
        if (g == 0) {
            y = (sh_red[0][i] + sh_red[1][i]) + (sh_red[2][i] + sh_red[3][i]);
            dt_store(x, dt_index(x, i, head, t), y);
        }
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) dt_store(st, dt_index(st, ch, 1 + g * 16 + jj, cur.batch), Sreg[jj]);
}


// The s-th sequence of a dispatch (the s-th token t whose cursor says "first token of its sequence": cursor.token == t), found by every wave
// for itself: 16 cursors per lane in flight, a ballot per 64 tokens.  Round 3: the chunk kernels are launched over (head, sequence slot)
// instead of (head, token) -- a 32 x 128-token chunk was 131 072 workgroups of which 1 024 stayed, placed wherever the dispatcher happened
// to have a free slot (some SIMDs three live waves, some none).
__device__ __forceinline__ bool find_sequence(const uint32_t* __restrict__ cursors, uint32_t T, uint32_t s, Cursor& out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t seen = 0;
    for (uint32_t base = 0; base < T; base += 1024) {
        uint32_t c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = cursors[min(base + 64 * u + lane, T - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t idx = base + 64 * u + lane;
            const bool first = idx < T && ((c[u] >> 8) & 0xffffu) == idx;
            const unsigned long long mask = __ballot(first);
            const uint32_t cnt = (uint32_t)__popcll(mask);
            if (s < seen + cnt) {
                const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                const unsigned long long hit = __ballot(first && rank == s - seen);
                const int src = __ffsll((long long)hit) - 1;
                out = unpack_cursor((uint32_t)__builtin_amdgcn_readlane((int)c[u], src));
                return true;
            }
            seen += cnt;
        }
    }
    return false;
}

// Fast path for the layout the runtime uses (dense f16 r, w, n, x; f32 state): same arithmetic, different mapping.
// Thread (i = tid >> 2, part = tid & 3) owns S[16 part .. 16 part + 15][i], so the two reductions over j are 16
// in-register FMAs plus two quad shuffles -- no LDS traffic and no workgroup barrier for them.  Only w~ (two
// transcendentals per key channel) is prepared cooperatively: thread j < 64 computes w~[j] of the NEXT token into a
// double-buffered LDS row, one barrier per token.  The f16 inputs of the next token are prefetched while the current
// one is multiplied.  Measured (1.5B, 128-token chunk): 292 us -> see DESIGN.md.
// WD: the decays w~ come precomputed (f32 [T][D], written by pre_wkv_v7 from the same f16 w with the same expression): no LDS row, NO BARRIER
// per token step -- the waves of a head run free (round 3; what a lone sequence's step costs is the chain, not the instruction count)
template <int NP, bool WD>      // NP = threads per state column: 4 (16 rows each; round 2) or 8 (8 rows each: few sequences, see time_mix_v7)
__global__ void __launch_bounds__(64 * NP) time_mix_v7_fast_kernel(const uint32_t* __restrict__ cursors, DTensor st, DTensor r, DTensor w,
                                                                DTensor n, DTensor x, uint32_t ntok, const float* __restrict__ wdec) {
    constexpr int S = 64, JJ = S / NP, NV = JJ / 8;
    __shared__ __attribute__((aligned(16))) float sh_w[2][S];
    const uint32_t head = blockIdx.x;
    Cursor cur;
    if (!find_sequence(cursors, ntok, blockIdx.y, cur)) return;       // fewer sequences in this dispatch than slots (uniform over the workgroup)
    const uint32_t tid = threadIdx.x, i = blockIdx.z * (blockDim.x / NP) + tid / NP, part = tid % NP;      // blockIdx.z: this workgroup's share of the head's 64 columns (WD only)
    const uint32_t ch = head * S + i;
    const uint32_t tend = cur.token + cur.len;

    // token-shift carry: state row 0 <- att_x of the sequence's last token (read before it is overwritten)
    if (part == 0) dt_store(st, dt_index(st, ch, 0, cur.batch), dt_load(x, dt_index(x, i, head, tend - 1)));
    // the state is f32 here (host-checked): plain loads, all sixteen in flight at once (dt_load branches on the element type, and the
    // compiler then waited vmcnt(0) after every one of them: sixteen serial round trips at the head of every chunk)
    float Sreg[JJ];
    float* sbase = (float*)st.p;
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) Sreg[jj] = sbase[dt_index(st, ch, 1 + part * JJ + jj, cur.batch)];

    struct Tok { f16x8 r[NV], k[NV], a[NV], kk[NV]; f16 v; f16 wraw; f32x4 wd[JJ / 4]; };
    // per-token pointers advance by constant strides (dense f16 views): no index arithmetic inside the loop
    const size_t rstep = (size_t)r.stride[1] * r.stride[0], nstep = (size_t)n.stride[1] * n.stride[0];
    const size_t wstep = (size_t)w.stride[1] * w.stride[0], xstep = (size_t)x.stride[1] * x.stride[0];
    const f16* rp = (const f16*)r.p + dt_index(r, part * JJ, head, cur.token);
    const f16* kp = (const f16*)n.p + dt_index4(n, part * JJ, head, cur.token, 0);
    const f16* ap = (const f16*)n.p + dt_index4(n, part * JJ, head, cur.token, 2);
    const f16* qp = (const f16*)n.p + dt_index4(n, part * JJ, head, cur.token, 3);
    const f16* vp = (const f16*)n.p + dt_index4(n, i, head, cur.token, 1);
    const f16* wp = (const f16*)w.p + dt_index(w, tid & 63u, head, cur.token);
    const float* dp = (WD ? wdec : (const float*)st.p) + dt_index(w, part * JJ, head, cur.token);      // same [D, T] indexing as w (dense)
    f16* xp = (f16*)x.p + dt_index(x, i, head, cur.token);
    // Every load is unconditional (round 2): the predicated form (`if (more) load`, `if (tid < 64) wraw = *wp`) made the compiler wait
    // vmcnt(0) right behind the prefetch, so each token paid a full memory round trip (2.8 us per token in the 32 x 128 prefill).
    // The pointers stop advancing at the last token instead (`adv` = 0): its prefetch re-reads that token and is discarded.
    auto load_tok = [&](Tok& T, bool adv) {       // advances the pointers (unless at the end), then loads the token they stand on
        const size_t rs = adv ? rstep : 0, ns = adv ? nstep : 0, ws = adv ? wstep : 0;
        rp += rs; kp += ns; ap += ns; qp += ns; vp += ns; wp += ws; dp += ws;
        if (WD) {
#pragma unroll
            for (int q = 0; q < JJ / 4; ++q) T.wd[q] = *(const f32x4*)(dp + 4 * q);
        } else T.wraw = *wp;                               // first of its group: the decay of token t + 1 is needed one step before the rest
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            T.r[q] = *(const f16x8*)(rp + 8 * q); T.k[q] = *(const f16x8*)(kp + 8 * q);
            T.a[q] = *(const f16x8*)(ap + 8 * q); T.kk[q] = *(const f16x8*)(qp + 8 * q);
        }
        T.v = *vp;
    };
    // Tokens are requested NPF - 1 steps ahead: a ring of NPF register sets, the loop unrolled by NPF so every index is static, and NO
    // control flow around the steps (steps beyond the chunk run on the clamped last token and are masked: the first attempt had a `break`
    // in the unrolled body and the compiler fell back to vmcnt(0)).  With ONE token of lookahead every step ended in `s_waitcnt vmcnt(0)`
    // for the decay of the next token -- the youngest load -- i.e. a full memory round trip per token: 2.4 us x 128 tokens per layer in
    // the 32 x 128-token prefill (found in the ISA, round 2).
    // (measured, 32 x 128-token prefill, tokens/s: ring of 3 101.6 k | ring of 2 99.6 k | ring of 4 100.7 k | register caps that restore four
    // workgroups per CU spill: 100.3 k / 87.1 k; one token of lookahead, before: 97.5 k)
    constexpr int NPF = 3;
    Tok T[NPF];
    uint32_t lpos = cur.token;                      // token the pointers stand on
    load_tok(T[0], false);
#pragma unroll
    for (int u = 1; u < NPF; ++u) { const bool adv = lpos + 1 < tend; load_tok(T[u], adv); lpos += adv ? 1u : 0u; }
    if (!WD && tid < S) sh_w[cur.token & 1u][tid] = __expf(-0.606531f * act_sigmoid((float)T[0].wraw));
    for (uint32_t tb = cur.token; tb < tend; tb += NPF) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const uint32_t t = tb + u;
            const bool valid = t < tend;                            // uniform; a masked step changes nothing
            const Tok& curT = T[u];
            if (!WD) __syncthreads();                               // w~ of token t is in sh_w[t & 1]
            // w~ of the NEXT token goes into the other buffer right away (its decay arrived a step ago; the last token's again at the end
            // of the chunk): the exponentials of the one wave that computes it are off the path to the next barrier
            if (!WD && tid < S) sh_w[(t + 1) & 1u][tid] = __expf(-0.606531f * act_sigmoid((float)T[(u + 1) % NPF].wraw));
            const float* wt = sh_w[t & 1u] + part * JJ;
            float wv[JJ];
#pragma unroll
            for (int q = 0; q < JJ / 4; ++q) { const f32x4 v4 = WD ? curT.wd[q] : *(const f32x4*)(wt + 4 * q); wv[4 * q] = v4[0]; wv[4 * q + 1] = v4[1]; wv[4 * q + 2] = v4[2]; wv[4 * q + 3] = v4[3]; }
            // four independent FMA chains per reduction (a wave runs alone on its SIMD: latency, not issue, is the cost)
            float kkf[JJ], s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) { kkf[jj] = (float)curT.kk[jj >> 3][jj & 7]; s4[jj & 3] = __builtin_fmaf(Sreg[jj], -kkf[jj], s4[jj & 3]); }   // a~ = -kk
            float sa = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            sa = sa + dpp_f32<0xB1>(sa);          // lane ^ 1, lane ^ 2 inside the quad: DPP, not ds_bpermute (an LDS round trip each)
            sa = sa + dpp_f32<0x4E>(sa);
            if (NP == 8) sa = sa + dpp_f32<0x141>(sa);      // the other quad of the column's eight lanes (row_half_mirror: lane i <-> 7 - i)
            const float vv = (float)curT.v;
            float y4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const float kj = (float)curT.k[jj >> 3][jj & 7], aj = (float)curT.a[jj >> 3][jj & 7];
                const float sn = Sreg[jj] * wv[jj] + kj * vv + sa * (kkf[jj] * aj);                                  // b~ = kk * a
                Sreg[jj] = valid ? sn : Sreg[jj];
                y4[jj & 3] = __builtin_fmaf((float)curT.r[jj >> 3][jj & 7], sn, y4[jj & 3]);
            }
            float y = (y4[0] + y4[1]) + (y4[2] + y4[3]);
            y = y + dpp_f32<0xB1>(y);
            y = y + dpp_f32<0x4E>(y);
            if (NP == 8) y = y + dpp_f32<0x141>(y);
            if (valid && part == 0) *xp = (f16)y;
            xp += valid ? xstep : 0;
            // this register set is free: request token t + NPF (the last token again once the chunk ends; discarded)
            { const bool adv = lpos + 1 < tend; load_tok(T[u], adv); lpos += adv ? 1u : 0u; }
        }
    }
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) sbase[dt_index(st, ch, 1 + part * JJ + jj, cur.batch)] = Sreg[jj];
}

// v_fma_mix_f32 with an f16 operand taken from half HI of a dword, unconverted (plain asm: the compiler may move and drop them)
template <int HI> __device__ __forceinline__ float mix_fma(uint32_t h2, float f, float acc) {           // (float)h * f + acc, one rounding
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h2), "v"(f));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h2), "v"(f));
    return acc;
}
template <int HI> __device__ __forceinline__ float mix_fma_neg(float f, uint32_t h2, float acc) {       // f * -(float)h + acc
    if (HI) asm("v_fma_mix_f32 %0, %1, -%2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(f), "v"(h2));
    else asm("v_fma_mix_f32 %0, %1, -%2, %0 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(f), "v"(h2));
    return acc;
}
template <int HI> __device__ __forceinline__ float mix_fma0(uint32_t h2, float f) {                     // (float)h * f + 0: head of a chain
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f));
    else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f));
    return d;
}
template <int HI> __device__ __forceinline__ float mix_fma_neg0(float f, uint32_t h2) {                 // f * -(float)h + 0
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, -%2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(f), "v"(h2));
    else asm("v_fma_mix_f32 %0, %1, -%2, 0 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(f), "v"(h2));
    return d;
}
template <int HI> __device__ __forceinline__ float mix_mul(uint32_t h2, float f, float negzero) {       // (float)h * f rounded to f32 (x + -0 = x)
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f), "v"(negzero));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(f), "v"(negzero));
    return d;
}

// Round 3: ONE WAVE per (head, sequence).  Lane i owns the whole state column S[0..63][i] in 64 registers, so both reductions over j are
// private FMA chains (no shuffles, no barrier, no workgroup) and every per-j quantity -- w~, b~ = kk * a, r, k, a~ = -kk -- is the SAME for all
// lanes: lane j prepares channel j of the NEXT token once (two transcendentals, one product; the quad kernel above converted each of them in 64
// threads: 49 conversions + 29 register moves of its ~250 instructions per wave and step, times four waves per head) and leaves it in a
// double-buffered LDS row that the wave reads back at a uniform address (broadcast).  ~330 vector instructions per head and step instead of
// ~1000; 32 sequences x 32 heads = one wave per SIMD of the chip instead of two rounds of two 4-wave workgroups per CU.
// The summation orders are those of the quad kernel (16 chains of 4, the same trees), so the two are bit-identical.
__global__ void __launch_bounds__(64) time_mix_v7_wave_kernel(const uint32_t* __restrict__ cursors, DTensor st, DTensor r, DTensor w, DTensor n,
                                                               DTensor x, uint32_t ntok) {
    constexpr int S = 64;
    struct Slot { float wt[S], bt[S]; f16 r[S], k[S], q[S]; };
    __shared__ __attribute__((aligned(16))) Slot sh[2];
    const uint32_t head = blockIdx.x;
    Cursor cur;
    if (!find_sequence(cursors, ntok, blockIdx.y, cur)) return;       // fewer sequences in this dispatch than slots
    const uint32_t lane = threadIdx.x;
    const uint32_t ch = head * S + lane;
    const uint32_t tend = cur.token + cur.len;

    // token-shift carry: state row 0 <- att_x of the sequence's last token (read before it is overwritten)
    dt_store(st, dt_index(st, ch, 0, cur.batch), dt_load(x, dt_index(x, lane, head, tend - 1)));
    float Sreg[S];
    float* sbase = (float*)st.p;
#pragma unroll
    for (int j = 0; j < S; ++j) Sreg[j] = sbase[dt_index(st, ch, 1 + j, cur.batch)];

    struct Tok { f16 w, a, q, v, r, k; };
    const size_t rstep = (size_t)r.stride[1] * r.stride[0], nstep = (size_t)n.stride[1] * n.stride[0];
    const size_t wstep = (size_t)w.stride[1] * w.stride[0], xstep = (size_t)x.stride[1] * x.stride[0];
    const f16* rp = (const f16*)r.p + dt_index(r, lane, head, cur.token);
    const f16* kp = (const f16*)n.p + dt_index4(n, lane, head, cur.token, 0);
    const f16* vp = (const f16*)n.p + dt_index4(n, lane, head, cur.token, 1);
    const f16* ap = (const f16*)n.p + dt_index4(n, lane, head, cur.token, 2);
    const f16* qp = (const f16*)n.p + dt_index4(n, lane, head, cur.token, 3);
    const f16* wp = (const f16*)w.p + dt_index(w, lane, head, cur.token);
    f16* xp = (f16*)x.p + dt_index(x, lane, head, cur.token);
    uint32_t lpos = cur.token;
    float negzero = -0.0f;
    asm volatile("" : "+v"(negzero));
    auto load_tok = [&](Tok& T, bool adv) {       // unconditional loads: the pointers stop at the last token
        const size_t rs = adv ? rstep : 0, ns = adv ? nstep : 0, ws = adv ? wstep : 0;
        rp += rs; kp += ns; vp += ns; ap += ns; qp += ns; wp += ws;
        lpos += adv ? 1u : 0u;
        T.w = *wp; T.a = *ap; T.q = *qp; T.v = *vp; T.r = *rp; T.k = *kp;
    };
    auto prepare = [&](const Tok& T, Slot& L) {   // lane j: channel j of a token
        L.wt[lane] = __expf(-0.606531f * act_sigmoid((float)T.w));
        L.bt[lane] = (float)T.q * (float)T.a;     // b~ = kk * a (exact in f32)
        L.r[lane] = T.r; L.k[lane] = T.k; L.q[lane] = T.q;
    };
    // Tokens are requested NPF - 1 steps ahead (a memory round trip under load is about three steps): a ring of register sets with STATIC
    // indices -- rotating the sets (T0 = T1 ...) moves registers whose loads are still in flight, which is a wait for all of them.  The main
    // loop takes whole groups of NPF tokens without any control flow; the last len % NPF tokens run behind uniform branches.
    constexpr int NPF = 4;
    Tok T[NPF];
    load_tok(T[0], false);
#pragma unroll
    for (int u = 1; u < NPF; ++u) load_tok(T[u], lpos + 1 < tend);
    prepare(T[0], sh[cur.token & 1u]);
    auto step = [&](uint32_t t, Tok& Tc, const Tok& Tn) {
        const Slot& L = sh[t & 1u];
        prepare(Tn, sh[(t + 1) & 1u]);              // the next token's row: off the critical path of this step
        const float vv = (float)Tc.v;
        load_tok(Tc, lpos + 1 < tend);              // this register set is free: token t + NPF (the last one again once the chunk ends)
        // sa[i] = sum_j S[j][i] * a~[j]: 16 chains (part p = j / 16, c = j % 4) of 4, as in the quad kernel.  The f16 operands go into
        // v_fma_mix_f32 as they are (the compiler converts each of them first and then packs pairs with register moves: 196 + 136 instructions)
        // LDS reads are requested a group ahead of their use (left alone the compiler reads each operand right in front of its first use and
        // waits: 56 exposed LDS round trips per step)
        u32x4 q8[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) q8[g] = *(const u32x4*)(L.q + 8 * g);
        struct Grp { u32x4 k, r; f32x4 w0, w1, b0, b1; };
        auto load_grp = [&](Grp& G, int g) {
            G.k = *(const u32x4*)(L.k + 8 * g); G.r = *(const u32x4*)(L.r + 8 * g);
            G.w0 = *(const f32x4*)(L.wt + 8 * g); G.w1 = *(const f32x4*)(L.wt + 8 * g + 4);
            G.b0 = *(const f32x4*)(L.bt + 8 * g); G.b1 = *(const f32x4*)(L.bt + 8 * g + 4);
        };
        Grp G0, G1;
        load_grp(G0, 0);
        float ch16[16];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const int j = 8 * g + e, c = (j >> 4) * 4 + (j & 3);
                if ((j & 15) < 4) {         // head of its chain
                    ch16[c] = mix_fma_neg0<0>(Sreg[j], q8[g][e >> 1]);
                    ch16[c + 1] = mix_fma_neg0<1>(Sreg[j + 1], q8[g][e >> 1]);
                } else {
                    ch16[c] = mix_fma_neg<0>(Sreg[j], q8[g][e >> 1], ch16[c]);
                    ch16[c + 1] = mix_fma_neg<1>(Sreg[j + 1], q8[g][e >> 1], ch16[c + 1]);
                }
            }
        }
        float P[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) P[p] = (ch16[4 * p] + ch16[4 * p + 1]) + (ch16[4 * p + 2] + ch16[4 * p + 3]);
        const float sa = (P[0] + P[1]) + (P[2] + P[3]);
        const f32x2 sa2 = {sa, sa};
        auto update = [&](const Grp& G, int g) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const int j = 8 * g + e, c = (j >> 4) * 4 + (j & 3);
                const f32x2 wj = e < 4 ? (f32x2){G.w0[e & 3], G.w0[(e & 3) + 1]} : (f32x2){G.w1[e & 3], G.w1[(e & 3) + 1]};
                const f32x2 bj = e < 4 ? (f32x2){G.b0[e & 3], G.b0[(e & 3) + 1]} : (f32x2){G.b1[e & 3], G.b1[(e & 3) + 1]};
                const f32x2 s2 = {Sreg[j], Sreg[j + 1]};
                const f32x2 kv = {mix_mul<0>(G.k[e >> 1], vv, negzero), mix_mul<1>(G.k[e >> 1], vv, negzero)};      // k * v, rounded once
                const f32x2 sn = (s2 * wj + kv) + sa2 * bj;
                Sreg[j] = sn[0];
                Sreg[j + 1] = sn[1];
                if ((j & 15) < 4) {
                    ch16[c] = mix_fma0<0>(G.r[e >> 1], sn[0]);
                    ch16[c + 1] = mix_fma0<1>(G.r[e >> 1], sn[1]);
                } else {
                    ch16[c] = mix_fma<0>(G.r[e >> 1], sn[0], ch16[c]);
                    ch16[c + 1] = mix_fma<1>(G.r[e >> 1], sn[1], ch16[c + 1]);
                }
            }
        };
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
            load_grp(G1, g + 1);
            update(G0, g);
            if (g + 2 < 8) load_grp(G0, g + 2);
            update(G1, g + 1);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) P[p] = (ch16[4 * p] + ch16[4 * p + 1]) + (ch16[4 * p + 2] + ch16[4 * p + 3]);
        const float y = (P[0] + P[1]) + (P[2] + P[3]);
        *xp = (f16)y;
        xp += xstep;
    };
    uint32_t tb = cur.token;
    for (; tb + NPF <= tend; tb += NPF) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) step(tb + u, T[u], T[(u + 1) % NPF]);
    }
#pragma unroll
    for (int u = 0; u < NPF - 1; ++u)
        if (tb + u < tend) step(tb + u, T[u], T[(u + 1) % NPF]);
#pragma unroll
    for (int j = 0; j < S; ++j) sbase[dt_index(st, ch, 1 + j, cur.batch)] = Sreg[j];
}

static bool dense_f16_heads(const DTensor& d) {
    return d.dtype == WRK_F16 && d.shape[0] == 64 && d.stride[0] == 64 && d.offset[0] == 0 && (((uintptr_t)d.p) & 15u) == 0;
}

void time_mix_v7(hipStream_t s, const uint32_t* cursors, DTensor st, DTensor r, DTensor w, DTensor n, DTensor x, uint32_t nseq_hint, const float* wdec) {
    if (r.shape[2] == 0) return;
    const uint32_t H = r.shape[1], T = r.shape[2];
    if (dense_f16_heads(r) && dense_f16_heads(w) && dense_f16_heads(n) && dense_f16_heads(x) && st.dtype == WRK_F32) {
        // sequence slots: at most one sequence per batch of the state and per token.  One wave per head once the sequences of the dispatch
        // (nseq_hint; unknown = as many as the state has batches) put a wave on three of four SIMDs of the chip, four waves per head below that:
        // a lone wave issues one vector instruction per ~4 cycles whatever it does (one sequence: 1.19 us per step against 0.84)
        const uint32_t slots = std::min(T, st.shape[2]);
        const uint32_t nseq = nseq_hint ? std::min(nseq_hint, slots) : slots;
        const char* fe = getenv("WRK_WKV_WAVE");                // 0 / 1: A/B and tests (read per call; captured programs keep their kernel)
        const int force = fe ? atoi(fe) : -1;
        const bool wave = force >= 0 ? force != 0 : nseq * H >= 768;     // measured (1.5B, 32 heads): 16 sequences 110 k vs 107 k tok/s for four waves, 32 sequences 117 k vs 124 k
        dim3 grid(H, slots);
        // few sequences (CUs idle): eight threads per state column instead of four -- half the instructions per wave and step, which is what a
        // lone sequence's step costs (pp512: 0.69 -> see DESIGN 4.5); from 256 heads on every SIMD has a wave either way
        const char* oe = getenv("WRK_WKV_OCT");
        const bool oct = oe ? atoi(oe) != 0 : true;      // measured, x 32 heads: 1 / 4 / 8 / 16 sequences +3 / +2.4 / +2.3 / +2 % end to end over four threads per column
        if (wave) time_mix_v7_wave_kernel<<<grid, 64, 0, s>>>(cursors, st, r, w, n, x, T);
        else if (oct && wdec) {
            // The columns of a state are independent (sa_i, y_i never leave column i): with few sequences a head is dealt over 2 or 4 workgroups of
            // 32 / 16 columns, so that its eight waves run on eight SIMDs of two or four CUs instead of two per SIMD of one (a lone sequence's
            // step is issue-bound at two waves per SIMD: 0.47 us).  Every workgroup re-reads the per-row operands (r, k, a, kk, w~: L2 hits).
            const char* ce = getenv("WRK_WKV_CSPLIT");
            uint32_t cs = ce ? (uint32_t)atoi(ce) : (nseq * H <= 32 ? 4u : (nseq * H <= 128 ? 2u : 1u));      // measured (x 32 heads, tokens/s, 1 | 2 | 4 workgroups per head): 1 sequence 23.7 | 25.6 | 25.7 k, 2: 38.9 | 41.2 | 41.0 k, 4: 48.1 | 50.8 | 49.1 k, 8: 82.0 | 82.1 k
            if (cs != 2 && cs != 4) cs = 1;
            time_mix_v7_fast_kernel<8, true><<<dim3(H, slots, cs), 512 / cs, 0, s>>>(cursors, st, r, w, n, x, T, wdec);
        }
        else if (oct) time_mix_v7_fast_kernel<8, false><<<grid, 512, 0, s>>>(cursors, st, r, w, n, x, T, nullptr);
        else { if (wdec) time_mix_v7_fast_kernel<4, true><<<grid, 256, 0, s>>>(cursors, st, r, w, n, x, T, wdec); else time_mix_v7_fast_kernel<4, false><<<grid, 256, 0, s>>>(cursors, st, r, w, n, x, T, nullptr); }
    } else {
        dim3 grid(H, T);
        time_mix_v7_kernel<<<grid, 256, 0, s>>>(cursors, st, r, w, n, x);
    }
}

// ------------------------------------------------------------------ merged element-wise stages of an RWKV-7 layer (mode 1, multi-token)
// The op list between the projections and the WKV kernel (v7.rs:826-905) is a dozen launches of one f16 row each:
//   w += w0 | a = sigmoid(a + a0) | kk = l2_norm(k * k_k) per head | k *= 1 + (a - 1) * k_a | v = mix(v, v0, sigmoid(vv + v0p))
//   (layer 0: v0 = v) | n[0..3] = k, v, a, kk
// One wave per (head, token) does all of it with every intermediate rounded to f16 exactly where the separate ops
// store it, and the same wave_sum for the per-head norm, so the results are bit-identical to the op chain.
// f32 -> f16 of a value that must first exist as an f32: without the barrier the compiler folds `(f16)(h * f)` into
// v_fma_mixlo_f16 (ONE rounding of the exact product), while the separate ops round the product to f32 and then to f16 --
// rare last-bit differences (seen as 2e-3 on a logit after 33 tokens).
__device__ __forceinline__ f16 to_h(float v) { asm volatile("" : "+v"(v)); return (f16)v; }
struct PreWkvParams {
    f16 *w, *a, *k, *v, *vv, *v0, *n;         // dense f16 [D, T] rows (n: [S, H, T, 4])
    const f16 *w0, *a0, *k_k, *k_a, *v0p;
    uint32_t D, T, first_layer;
    float l2_eps;
    float* wdec;                              // optional f32 [D, T]: w~ = exp(-0.606531 sigmoid(w)) of the stored (f16) w, for the chunk kernel
};
__global__ void __launch_bounds__(64) pre_wkv_v7_kernel(const PreWkvParams P) {
    const uint32_t head = blockIdx.x, t = blockIdx.y, c = head * 64 + threadIdx.x;
    const size_t i = (size_t)t * P.D + c, plane = (size_t)P.T * P.D;
    // every operand requested up front (round 2: the loads were interleaved with the rounding barriers of to_h and each group ended in
    // vmcnt(0): seven serial round trips per wave); layer 0 reads its own v0 slot / vv as dummies
    const f16 w0c = P.w0[c], wi = P.w[i], a0c = P.a0[c], ai = P.a[i], k0 = P.k[i], kkc = P.k_k[c], kac = P.k_a[c], vi = P.v[i];
    const f16 v0pc = (P.first_layer ? P.k_a : P.v0p)[c], vvi = (P.first_layer ? P.v : P.vv)[i], v0i = P.v0[i];
    const f16 wn = to_h((float)w0c + (float)wi);
    const f16 an = to_h(act_sigmoid((float)a0c + (float)ai));
    const f16 kk0 = to_h((float)kkc * (float)k0);
    const float ss = wave_sum((float)kk0 * (float)kk0);
    const f16 kk1 = to_h((float)kk0 * (1.0f / sqrtf(ss + P.l2_eps)));
    const f16 kn = to_h((float)k0 * (1.0f + ((float)an - 1.0f) * (float)kac));
    f16 vn = vi;
    if (P.first_layer) P.v0[i] = vn;
    else {
        const f16 f = to_h(act_sigmoid((float)v0pc + (float)vvi));
        vn = to_h(wgsl_mix((float)vn, (float)v0i, (float)f));
    }
    P.w[i] = wn;
    if (P.wdec) P.wdec[i] = __expf(-0.606531f * act_sigmoid((float)wn));
    P.n[i] = kn;
    P.n[plane + i] = vn;
    P.n[2 * plane + i] = an;
    P.n[3 * plane + i] = kk1;
}
void pre_wkv_v7(hipStream_t s, void* w, void* a, void* k, void* v, void* vv, void* v0, void* n, const void* w0, const void* a0, const void* k_k,
                const void* k_a, const void* v0p, uint32_t D, uint32_t T, bool first_layer, float l2_eps, float* wdec) {
    if (T == 0) return;
    PreWkvParams P{(f16*)w, (f16*)a, (f16*)k, (f16*)v, (f16*)vv, (f16*)v0, (f16*)n, (const f16*)w0, (const f16*)a0, (const f16*)k_k, (const f16*)k_a,
                   (const f16*)v0p, D, T, first_layer ? 1u : 0u, l2_eps, wdec};
    pre_wkv_v7_kernel<<<dim3(D / 64, T), 64, 0, s>>>(P);
}

// group_norm -> time_first -> * g after the WKV kernel (v7.rs:918-946), same contract as above
struct PostWkvParams {
    f16* x;                                    // WKV output, [D, T]
    const f16 *r, *g, *n, *gn_w, *gn_b, *r_k;
    uint32_t D, T;
    float gn_eps;
};
__global__ void __launch_bounds__(64) post_wkv_v7_kernel(const PostWkvParams P) {
    const uint32_t head = blockIdx.x, t = blockIdx.y, c = head * 64 + threadIdx.x;
    const size_t i = (size_t)t * P.D + c, plane = (size_t)P.T * P.D;
    const f16 xi = P.x[i], gw = P.gn_w[c], gb = P.gn_b[c], rk = P.r_k[c], n0 = P.n[i], ri = P.r[i], n1 = P.n[plane + i], gi = P.g[i];    // all up front
    const float x0 = (float)xi;
    const float mean = wave_sum(x0) / 64.0f;
    const float dlt = x0 - mean;
    const float var = wave_sum(dlt * dlt) / 64.0f + P.gn_eps;
    const float dev = 1.0f / sqrtf(var);
    const f16 y = to_h(__builtin_fmaf((x0 - mean) * dev, (float)gw, (float)gb));
    const float xx = wave_sum((float)rk * (float)n0 * (float)ri);
    const f16 y2 = to_h((float)y + xx * (float)n1);
    P.x[i] = to_h((float)gi * (float)y2);
}
void post_wkv_v7(hipStream_t s, void* x, const void* r, const void* g, const void* n, const void* gn_w, const void* gn_b, const void* r_k,
                 uint32_t D, uint32_t T, float gn_eps) {
    if (T == 0) return;
    PostWkvParams P{(f16*)x, (const f16*)r, (const f16*)g, (const f16*)n, (const f16*)gn_w, (const f16*)gn_b, (const f16*)r_k, D, T, gn_eps};
    post_wkv_v7_kernel<<<dim3(D / 64, T), 64, 0, s>>>(P);
}

// ------------------------------------------------------------------ time_first_v7 (time_mix_v7.wgsl:223-262)
// x[i] += (sum_j u[j] * k[j] * r[j]) * v[i] per head; one wave per (head, token)
__global__ void __launch_bounds__(64) time_first_v7_kernel(const f16* __restrict__ u, DTensor r, DTensor n, DTensor x) {
    const uint32_t head = blockIdx.x, t = blockIdx.y, i = threadIdx.x;
    const float uu = (float)u[head * 64 + i];
    const float kk = dt_load(n, dt_index4(n, i, head, t, 0));
    const float rr = dt_load(r, dt_index(r, i, head, t));
    const float xx = wave_sum(uu * kk * rr);
    const float vv = dt_load(n, dt_index4(n, i, head, t, 1));
    const size_t o = dt_index(x, i, head, t);
    dt_store(x, o, dt_load(x, o) + xx * vv);
}

void time_first_v7(hipStream_t s, const void* u, DTensor r, DTensor n, DTensor x) {
    if (r.shape[2] == 0) return;
    dim3 grid(r.shape[1], r.shape[2]);
    time_first_v7_kernel<<<grid, 64, 0, s>>>((const f16*)u, r, n, x);
}

// ------------------------------------------------------------------ channel_mix (V7) (channel_mix.wgsl:83-107)
__global__ void __launch_bounds__(256) channel_mix_v7_kernel(const uint32_t* __restrict__ cursors, DTensor st, DTensor v, DTensor x) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stack = blockIdx.y;
    if (c >= x.shape[0]) return;
    const Cursor cur = unpack_cursor(cursors[stack]);
    const size_t o = dt_index(x, c, stack, 0);
    if (stack - cur.token + 1 == cur.len) dt_store(st, dt_index(st, c, 0, cur.batch), dt_load(x, o));
    dt_store(x, o, dt_load(v, dt_index(v, c, stack, 0)));
}

void channel_mix_v7(hipStream_t s, const uint32_t* cursors, DTensor st, DTensor v, DTensor x) {
    if (x.shape[1] == 0) return;
    dim3 grid((x.shape[0] + 255) / 256, x.shape[1]);
    channel_mix_v7_kernel<<<grid, 256, 0, s>>>(cursors, st, v, x);
}

// ------------------------------------------------------------------ transpose (reshape.wgsl:59-78): out[c, b, t] = in[c, t, b]
__global__ void __launch_bounds__(256) transpose_kernel(DTensor in, DTensor out) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= out.shape[0]) return;
    dt_store(out, dt_index(out, c, blockIdx.z, blockIdx.y), dt_load(in, dt_index(in, c, blockIdx.y, blockIdx.z)));
}

void transpose(hipStream_t s, DTensor in, DTensor out) {
    if (in.shape[1] == 0 || in.shape[2] == 0) return;
    dim3 grid((in.shape[0] + 255) / 256, in.shape[1], in.shape[2]);
    transpose_kernel<<<grid, 256, 0, s>>>(in, out);
}

// ------------------------------------------------------------------ time_mix_v6 (time_mix_v6.wgsl:83-155)
//   y[i]   = sum_j r[j] * (u[j] * k[j] * v[i] + S[j,i]);   S[j,i] <- w[j] * S[j,i] + k[j] * v[i]
// Same decomposition as time_mix_v7_kernel: one workgroup per (head, sequence chunk), state in registers.
__global__ void __launch_bounds__(256) time_mix_v6_kernel(const uint32_t* __restrict__ cursors, DTensor decay, const float* __restrict__ u,
                                                           DTensor st, DTensor k, DTensor v, DTensor r, DTensor x) {
    constexpr int S = 64;
    __shared__ float sh_r[S], sh_w[S], sh_k[S], sh_u[S];
    __shared__ float sh_red[4][S];
    const uint32_t head = blockIdx.x, t0 = blockIdx.y;
    const Cursor cur = unpack_cursor(cursors[t0]);
    if (cur.token != t0) return;
    const uint32_t tid = threadIdx.x, i = tid & 63, g = tid >> 6;
    const uint32_t ch = head * S + i;
    if (g == 0) {
        const uint32_t last = cur.token + cur.len - 1;
        dt_store(st, dt_index(st, ch, 0, cur.batch), dt_load(x, dt_index(x, i, head, last)));
        sh_u[i] = u[ch];
    }
    float Sreg[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Sreg[jj] = dt_load(st, dt_index(st, ch, 1 + g * 16 + jj, cur.batch));
    for (uint32_t t = cur.token; t < cur.token + cur.len; ++t) {
        __syncthreads();
        if (g == 0) {
            sh_r[i] = dt_load(r, dt_index(r, i, head, t));
            sh_k[i] = dt_load(k, dt_index(k, i, head, t));
            sh_w[i] = dt_load(decay, dt_index(decay, i, head, t));
        }
        __syncthreads();
        const float vv = dt_load(v, dt_index(v, i, head, t));
        float y = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = g * 16 + jj;
            const float kv = sh_k[j] * vv;
            y += sh_r[j] * __builtin_fmaf(sh_u[j], kv, Sreg[jj]);
            Sreg[jj] = __builtin_fmaf(sh_w[j], Sreg[jj], kv);
        }
        sh_red[g][i] = y;
        __syncthreads();
        if (g == 0) dt_store(x, dt_index(x, i, head, t), (sh_red[0][i] + sh_red[1][i]) + (sh_red[2][i] + sh_red[3][i]));
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) dt_store(st, dt_index(st, ch, 1 + g * 16 + jj, cur.batch), Sreg[jj]);
}

// Fast path for the runtime's layout (dense f32 decay / k / v / r, f16 x, f32 state): thread (i = tid >> 2, part = tid & 3)
// owns S[16 part .. +15][i]; the reduction over j is 16 in-register FMAs + two quad shuffles, so a token needs no LDS and
// no barrier at all (the generic kernel above spends four barriers per token).  Next token's operands are prefetched.
template <int NP>      // threads per state column: 4 (16 rows each) or 8 (8 rows each; few sequences, with the head's columns over blockIdx.z workgroups: as RWKV-7)
__global__ void __launch_bounds__(64 * NP) time_mix_v6_fast_kernel(const uint32_t* __restrict__ cursors, DTensor decay, const float* __restrict__ u,
                                                                DTensor st, DTensor k, DTensor v, DTensor r, DTensor x, uint32_t ntok) {
    constexpr int S = 64, JJ = S / NP;
    const uint32_t head = blockIdx.x;
    Cursor cur;
    if (!find_sequence(cursors, ntok, blockIdx.y, cur)) return;       // launched over (head, sequence slot), as the RWKV-7 chunk kernels (round 3)
    const uint32_t tid = threadIdx.x, i = blockIdx.z * (blockDim.x / NP) + tid / NP, part = tid % NP;
    const uint32_t ch = head * S + i;
    const uint32_t tend = cur.token + cur.len;
    if (part == 0) dt_store(st, dt_index(st, ch, 0, cur.batch), dt_load(x, dt_index(x, i, head, tend - 1)));
    float Sreg[JJ], uu[JJ];
    float* sbase = (float*)st.p;         // f32 state (host-checked): plain loads, all in flight at once
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) { Sreg[jj] = sbase[dt_index(st, ch, 1 + part * JJ + jj, cur.batch)]; uu[jj] = u[head * S + part * JJ + jj]; }
    struct Tok { f32x4 r[JJ / 4], k[JJ / 4], w[JJ / 4]; float v; };
    const size_t rstep = (size_t)r.stride[1] * r.stride[0], kstep = (size_t)k.stride[1] * k.stride[0];
    const size_t wstep = (size_t)decay.stride[1] * decay.stride[0], vstep = (size_t)v.stride[1] * v.stride[0], xstep = (size_t)x.stride[1] * x.stride[0];
    const float* rp = (const float*)r.p + dt_index(r, part * JJ, head, cur.token);
    const float* kp = (const float*)k.p + dt_index(k, part * JJ, head, cur.token);
    const float* wp = (const float*)decay.p + dt_index(decay, part * JJ, head, cur.token);
    const float* vp = (const float*)v.p + dt_index(v, i, head, cur.token);
    f16* xp = (f16*)x.p + dt_index(x, i, head, cur.token);
    // As the RWKV-7 kernel (round 3; this one still had the round-1 form): every load unconditional -- the pointers stop at the last token --,
    // a ring of NPF register sets with static indices (`if (more) load`, `cur = next` each cost a wait for everything in flight per token),
    // steps beyond the chunk masked.
    uint32_t lpos = cur.token;
    auto load_tok = [&](Tok& T, bool adv) {
        rp += adv ? rstep : 0; kp += adv ? kstep : 0; wp += adv ? wstep : 0; vp += adv ? vstep : 0;
        lpos += adv ? 1u : 0u;
#pragma unroll
        for (int q = 0; q < JJ / 4; ++q) { T.r[q] = *(const f32x4*)(rp + 4 * q); T.k[q] = *(const f32x4*)(kp + 4 * q); T.w[q] = *(const f32x4*)(wp + 4 * q); }
        T.v = *vp;
    };
    constexpr int NPF = 3;
    Tok T[NPF];
    load_tok(T[0], false);
#pragma unroll
    for (int q = 1; q < NPF; ++q) load_tok(T[q], lpos + 1 < tend);
    for (uint32_t tb = cur.token; tb < tend; tb += NPF) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const bool valid = tb + q < tend;           // uniform; a masked step changes nothing
            const Tok& curT = T[q];
            const float vv = curT.v;
            float y = 0.0f;
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const float kv = curT.k[jj >> 2][jj & 3] * vv;
                y += curT.r[jj >> 2][jj & 3] * __builtin_fmaf(uu[jj], kv, Sreg[jj]);
                const float sn = __builtin_fmaf(curT.w[jj >> 2][jj & 3], Sreg[jj], kv);
                Sreg[jj] = valid ? sn : Sreg[jj];
            }
            y = y + dpp_f32<0xB1>(y);
            y = y + dpp_f32<0x4E>(y);
            if (NP == 8) y = y + dpp_f32<0x141>(y);         // the other quad of the column's eight lanes
            if (valid && part == 0) *xp = (f16)y;
            xp += valid ? xstep : 0;
            load_tok(T[q], lpos + 1 < tend);            // this register set is free: token t + NPF (the last one again beyond the chunk; discarded)
        }
    }
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) sbase[dt_index(st, ch, 1 + part * JJ + jj, cur.batch)] = Sreg[jj];
}

// One wave per (head, sequence), as time_mix_v7_wave_kernel (round 3): lane i owns the state column S[0..63][i]; the per-row operands (decay w, k, r: f32 here)
// are loaded lane-parallel -- lane j takes channel j of the next tokens -- and read back from a double-buffered LDS row at uniform addresses; the bonus u is
// constant per head (64 registers).  The sum over j runs as four chains of sixteen in the quad kernel's order, (P0 + P1) + (P2 + P3): bit-identical to it.
// 16 sequences x 64 heads of the 7B model: one wave per SIMD of the chip instead of two rounds of 4-wave workgroups.
__global__ void __launch_bounds__(64) time_mix_v6_wave_kernel(const uint32_t* __restrict__ cursors, DTensor decay, const float* __restrict__ u, DTensor st, DTensor k,
                                                               DTensor v, DTensor r, DTensor x, uint32_t ntok) {
    constexpr int S = 64;
    struct Slot { float w[S], k[S], r[S]; };
    __shared__ __attribute__((aligned(16))) Slot sh[2];
    const uint32_t head = blockIdx.x;
    Cursor cur;
    if (!find_sequence(cursors, ntok, blockIdx.y, cur)) return;
    const uint32_t lane = threadIdx.x, ch = head * S + lane;
    const uint32_t tend = cur.token + cur.len;
    dt_store(st, dt_index(st, ch, 0, cur.batch), dt_load(x, dt_index(x, lane, head, tend - 1)));       // token-shift carry (read before x is overwritten)
    float Sreg[S], uu[S];
    float* sbase = (float*)st.p;
#pragma unroll
    for (int j = 0; j < S; ++j) { Sreg[j] = sbase[dt_index(st, ch, 1 + j, cur.batch)]; uu[j] = u[head * S + j]; }
    struct Tok { float w, k, r, v; };
    const size_t rstep = (size_t)r.stride[1] * r.stride[0], kstep = (size_t)k.stride[1] * k.stride[0];
    const size_t wstep = (size_t)decay.stride[1] * decay.stride[0], vstep = (size_t)v.stride[1] * v.stride[0], xstep = (size_t)x.stride[1] * x.stride[0];
    const float* rp = (const float*)r.p + dt_index(r, lane, head, cur.token);
    const float* kp = (const float*)k.p + dt_index(k, lane, head, cur.token);
    const float* wp = (const float*)decay.p + dt_index(decay, lane, head, cur.token);
    const float* vp = (const float*)v.p + dt_index(v, lane, head, cur.token);
    f16* xp = (f16*)x.p + dt_index(x, lane, head, cur.token);
    uint32_t lpos = cur.token;
    auto load_tok = [&](Tok& T, bool adv) {       // unconditional loads: the pointers stop at the last token
        rp += adv ? rstep : 0; kp += adv ? kstep : 0; wp += adv ? wstep : 0; vp += adv ? vstep : 0;
        lpos += adv ? 1u : 0u;
        T.w = *wp; T.k = *kp; T.r = *rp; T.v = *vp;
    };
    auto prepare = [&](const Tok& T, Slot& L) { L.w[lane] = T.w; L.k[lane] = T.k; L.r[lane] = T.r; };
    constexpr int NPF = 4;
    Tok T[NPF];
    load_tok(T[0], false);
#pragma unroll
    for (int q = 1; q < NPF; ++q) load_tok(T[q], lpos + 1 < tend);
    prepare(T[0], sh[cur.token & 1u]);
    auto step = [&](uint32_t t, Tok& Tc, const Tok& Tn) {
        const Slot& L = sh[t & 1u];
        prepare(Tn, sh[(t + 1) & 1u]);
        const float vv = Tc.v;
        load_tok(Tc, lpos + 1 < tend);
        float P[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float y = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *(const f32x4*)(L.w + 16 * p + 4 * q), k4 = *(const f32x4*)(L.k + 16 * p + 4 * q), r4 = *(const f32x4*)(L.r + 16 * p + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 16 * p + 4 * q + e;
                    const float kv = k4[e] * vv;
                    y += r4[e] * __builtin_fmaf(uu[j], kv, Sreg[j]);
                    Sreg[j] = __builtin_fmaf(w4[e], Sreg[j], kv);
                }
            }
            P[p] = y;
        }
        const float y = (P[0] + P[1]) + (P[2] + P[3]);
        *xp = (f16)y;
        xp += xstep;
    };
    uint32_t tb = cur.token;
    for (; tb + NPF <= tend; tb += NPF) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) step(tb + q, T[q], T[(q + 1) % NPF]);
    }
#pragma unroll
    for (int q = 0; q < NPF - 1; ++q)
        if (tb + q < tend) step(tb + q, T[q], T[(q + 1) % NPF]);
#pragma unroll
    for (int j = 0; j < S; ++j) sbase[dt_index(st, ch, 1 + j, cur.batch)] = Sreg[j];
}

static bool dense_f32_heads(const DTensor& d) {
    return d.dtype == WRK_F32 && d.shape[0] == 64 && d.stride[0] == 64 && d.offset[0] == 0 && (((uintptr_t)d.p) & 15u) == 0;
}

void time_mix_v6(hipStream_t s, const uint32_t* cursors, DTensor decay, const void* u, DTensor st, DTensor k, DTensor v, DTensor r, DTensor x, uint32_t nseq_hint) {
    if (r.shape[2] == 0) return;
    dim3 grid(r.shape[1], r.shape[2]);
    if (dense_f32_heads(decay) && dense_f32_heads(k) && dense_f32_heads(v) && dense_f32_heads(r) && dense_f16_heads(x) && st.dtype == WRK_F32) {
        const uint32_t T = r.shape[2], slots = std::min(T, st.shape[2]);       // at most one sequence per batch of the state and per token
        // one wave per head once the sequences of the dispatch put a wave on three of four SIMDs (no hint: as many as the state has batches)
        const char* fe = getenv("WRK_WKV_WAVE");
        const uint32_t nseq = nseq_hint ? std::min(nseq_hint, slots) : slots;
        const bool wave = fe ? atoi(fe) != 0 : (size_t)nseq * r.shape[1] >= 768;
        if (wave) time_mix_v6_wave_kernel<<<dim3(r.shape[1], slots), 64, 0, s>>>(cursors, decay, (const float*)u, st, k, v, r, x, T);
        else {
            // few sequences: eight threads per column and the head's columns over 2 or 4 workgroups (the columns never exchange anything), as RWKV-7;
            // WRK_WKV_OCT=0: four threads per column, one workgroup per head
            const char* oe = getenv("WRK_WKV_OCT");
            const uint32_t H = r.shape[1];
            if (oe && atoi(oe) == 0) time_mix_v6_fast_kernel<4><<<dim3(H, slots), 256, 0, s>>>(cursors, decay, (const float*)u, st, k, v, r, x, T);
            else {
                const char* ce = getenv("WRK_WKV_CSPLIT");
                uint32_t cs = ce ? (uint32_t)atoi(ce) : (nseq * H <= 32 ? 4u : (nseq * H <= 128 ? 2u : 1u));
                if (cs != 2 && cs != 4) cs = 1;
                time_mix_v6_fast_kernel<8><<<dim3(H, slots, cs), 512 / cs, 0, s>>>(cursors, decay, (const float*)u, st, k, v, r, x, T);
            }
        }
        return;
    }
    time_mix_v6_kernel<<<grid, 256, 0, s>>>(cursors, decay, (const float*)u, st, k, v, r, x);
}

// ------------------------------------------------------------------ channel_mix (V6) (channel_mix.wgsl:83-107): x <- sigmoid(r) * v
__global__ void __launch_bounds__(256) channel_mix_v6_kernel(const uint32_t* __restrict__ cursors, DTensor st, DTensor r, DTensor v, DTensor x) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stack = blockIdx.y;
    if (c >= x.shape[0]) return;
    const Cursor cur = unpack_cursor(cursors[stack]);
    const size_t o = dt_index(x, c, stack, 0);
    if (stack - cur.token + 1 == cur.len) dt_store(st, dt_index(st, c, 0, cur.batch), dt_load(x, o));
    const float rr = 1.0f / (1.0f + __expf(-dt_load(r, dt_index(r, c, stack, 0))));
    dt_store(x, o, rr * dt_load(v, dt_index(v, c, stack, 0)));
}

void channel_mix_v6(hipStream_t s, const uint32_t* cursors, DTensor st, DTensor r, DTensor v, DTensor x) {
    if (x.shape[1] == 0) return;
    dim3 grid((x.shape[0] + 255) / 256, x.shape[1]);
    channel_mix_v6_kernel<<<grid, 256, 0, s>>>(cursors, st, r, v, x);
}

// ------------------------------------------------------------------ softmax (softmax.wgsl) -- next (f)1
__global__ void __launch_bounds__(256) softmax_kernel(DTensor x) {
    __shared__ float red[4];
    const uint32_t C = x.shape[0];
    const size_t base = dt_index(x, 0, blockIdx.x, blockIdx.y);
    float m = -3.0e38f;
    for (uint32_t i = threadIdx.x; i < C; i += 256) m = fmaxf(m, dt_load(x, base + i));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < C; i += 256) s += __expf(dt_load(x, base + i) - m);
    s = block_sum<4>(s, red);
    for (uint32_t i = threadIdx.x; i < C; i += 256) dt_store(x, base + i, __expf(dt_load(x, base + i) - m) / s);
}

void softmax(hipStream_t s, DTensor x) {
    if (x.shape[1] == 0 || x.shape[2] == 0) return;
    dim3 grid(x.shape[1], x.shape[2]);
    softmax_kernel<<<grid, 256, 0, s>>>(x);
}

// ------------------------------------------------------------------ embedding gather / header gather / argmax
// RnnJob::load gathers embedding rows on the CPU (v7.rs:438-474); with the f16 table resident on the
// device the same rows are gathered here (values identical, one 2*D-byte row per token).
__global__ void __launch_bounds__(256) gather_rows_f16_kernel(const f16* __restrict__ table, const uint32_t* __restrict__ ids,
                                                               f16* __restrict__ out, uint32_t d) {
    const uint32_t t = blockIdx.y;
    const size_t row = ids[t];
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < d; c += gridDim.x * 256) out[(size_t)t * d + c] = table[row * d + c];
}

void gather_rows_f16(hipStream_t s, const void* table, const uint32_t* ids, void* out, uint32_t d, uint32_t n) {
    if (n == 0) return;
    dim3 grid((d + 255) / 256, n);
    gather_rows_f16_kernel<<<grid, 256, 0, s>>>((const f16*)table, ids, (f16*)out, d);
}

// RnnRedirect::op (rnn.rs:101-134): copy the header rows of x into head_x
__global__ void __launch_bounds__(256) gather_rows_any_kernel(DTensor in, const uint32_t* __restrict__ rows, DTensor out) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= out.shape[0]) return;
    dt_store(out, dt_index(out, c, blockIdx.y, 0), dt_load(in, dt_index(in, c, rows[blockIdx.y], 0)));
}

void gather_rows_any(hipStream_t s, DTensor in, const uint32_t* rows, DTensor out, uint32_t n) {
    if (n == 0) return;
    dim3 grid((out.shape[0] + 255) / 256, n);
    gather_rows_any_kernel<<<grid, 256, 0, s>>>(in, rows, out);
}

// greedy sampling on device: first index of the maximum (the reference's `sample(&output, 0.0)` after
// softmax picks the arg-max probability; softmax is monotone so the logits' argmax is the same token)
__global__ void __launch_bounds__(1024) argmax_rows_kernel(const float* __restrict__ logits, uint32_t v, uint32_t v_stride,
                                                            uint32_t* __restrict__ out) {
    __shared__ float smax[16];
    __shared__ uint32_t sidx[16];
    const float* row = logits + (size_t)blockIdx.x * v_stride;
    float best = -3.0e38f;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < v; i += 1024) {
        const float x = row[i];
        if (x > best) { best = x; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, WAVE);
        const uint32_t oi = __shfl_xor(bi, o, WAVE);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = best; sidx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (smax[k] > best || (smax[k] == best && sidx[k] < bi)) { best = smax[k]; bi = sidx[k]; }
        out[blockIdx.x] = bi == 0xffffffffu ? 0u : bi;
    }
}

void argmax_rows(hipStream_t s, const float* logits, uint32_t v, uint32_t v_stride, uint32_t n, uint32_t* out) {
    if (n == 0) return;
    argmax_rows_kernel<<<n, 1024, 0, s>>>(logits, v, v_stride, out);
}

}  // namespace wrk
