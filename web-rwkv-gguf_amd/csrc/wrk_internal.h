// Internal host-side structures behind the opaque handles of include/wrk_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "wrk_hip.h"

struct wrk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // submission stream (ops, programs)
    hipStream_t read_stream = nullptr;  // read-back stream (context.rs:148-162 readback thread)
    hipEvent_t read_event = nullptr;
    std::recursive_mutex mu;
    std::string err;
    // Context::encode (ops.rs:79-143) runs on tokio `spawn_blocking` workers while the runtime task submits cached jobs
    // (runtime/mod.rs:139-167).  So an open capture belongs to the ENCODING THREAD and records on a private stream of
    // its own: the submission stream and the read-back stream are never in capture mode, any number of threads may
    // encode at once, and submissions / reads / uploads from other threads proceed meanwhile.  (Guarded by `mu`.)
    std::map<std::thread::id, hipStream_t> sessions;    // open captures
    std::vector<hipStream_t> capture_pool;              // idle capture streams
    // the stream a wrk_op_* call of THIS thread records on: its open capture, else the submission stream
    hipStream_t op_stream() {
        auto it = sessions.find(std::this_thread::get_id());
        return it == sessions.end() ? stream : it->second;
    }
    bool capturing_here() { return sessions.count(std::this_thread::get_id()) != 0; }
    int num_cu = 256;
    void* staging = nullptr;            // pinned host staging for wrk_buf_write
    size_t staging_bytes = 0;
    // scratch of the prefill GEMM (sub-block input sums of the stacked tokens); grown outside captures, never shrunk, one per context:
    // the programs of a context run on one submission stream, in order
    void* gemm_scratch = nullptr;
    size_t gemm_scratch_cap = 0;
};
int32_t wrk_ctx_reserve_gemm_scratch(wrk_ctx* ctx, size_t bytes);      // WRK_OK also when it cannot grow now (inside a capture): the GEMM then keeps its older kernels

struct wrk_buf {
    wrk_ctx* ctx;
    void* ptr;
    size_t bytes;
    std::atomic<int> refs;
};

struct wrk_matrix {
    wrk_ctx* ctx;
    uint32_t kind, k, m, flags;
    uint8_t* data;          // re-laid-out weight stream (device)
    size_t row_bytes;       // device bytes per row (16-byte aligned)
    size_t stored_bytes;    // algorithmic bytes = GGUF/f16 stored size of the tensor
    uint8_t* aux = nullptr;     // Matrix::Fp4 { q }: the 16 f32 levels (device); side tables live in the row planes
    size_t aux_bytes = 0;
    std::atomic<int> refs;
    float out_scale = 1.0f;     // y = out_scale * (W . x): the 2^-k layer discount of load_matrix_discount kept OUT of the blocks
};

struct wrk_program {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

inline int32_t wrk_fail(wrk_ctx* ctx, int32_t code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
inline int32_t wrk_fail(wrk_ctx* ctx, int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define WRK_HIP(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return wrk_fail((ctx), _e == hipErrorOutOfMemory ? WRK_E_OOM : WRK_E_HIP, "%s: %s (%s:%d)", \
                            #expr, hipGetErrorString(_e), __FILE__, __LINE__);                          \
    } while (0)

#define WRK_ARG(ctx, cond, ...)                                     \
    do {                                                            \
        if (!(cond)) return wrk_fail((ctx), WRK_E_ARG, __VA_ARGS__); \
    } while (0)

#define WRK_LAUNCH_CHECK(ctx) WRK_HIP(ctx, hipGetLastError())

// ------------------------------------------------------------------ device-side tensor descriptor
// Mirrors `View` addressing (see wrk_hip.h).  Passed by value to kernels.
struct DTensor {
    void* p;
    uint32_t dtype;     // WRK_F16 / WRK_F32
    uint32_t shape[4];
    uint32_t stride[4];
    uint32_t offset[4];
};

inline DTensor make_dtensor(const wrk_tensor* t) {
    DTensor d;
    d.p = t->buf ? t->buf->ptr : nullptr;
    d.dtype = t->dtype;
    for (int i = 0; i < 4; ++i) {
        d.shape[i] = t->view.shape[i];
        d.stride[i] = t->view.stride[i];
        d.offset[i] = t->view.offset[i];
    }
    return d;
}

inline DTensor make_dense(void* p, uint32_t dtype, uint32_t c, uint32_t t = 1, uint32_t b = 1, uint32_t w = 1) {
    DTensor d;
    d.p = p;
    d.dtype = dtype;
    d.shape[0] = c; d.shape[1] = t; d.shape[2] = b; d.shape[3] = w;
    d.stride[0] = c; d.stride[1] = t; d.stride[2] = b; d.stride[3] = w;
    d.offset[0] = d.offset[1] = d.offset[2] = d.offset[3] = 0;
    return d;
}

inline size_t dtype_size(uint32_t dt) { return dt == WRK_F16 ? 2 : (dt == WRK_U8 ? 1 : 4); }

// number of elements the view's parent tensor must hold for the view to be in bounds
inline size_t dtensor_extent(const DTensor& d) {
    size_t s0 = d.stride[0], s1 = d.stride[1], s2 = d.stride[2] ? d.stride[2] : 1;
    size_t last_w = (size_t)d.offset[3] + (d.shape[3] ? d.shape[3] - 1 : 0);
    size_t last_b = (size_t)d.offset[2] + (d.shape[2] ? d.shape[2] - 1 : 0);
    size_t last_t = (size_t)d.offset[1] + (d.shape[1] ? d.shape[1] - 1 : 0);
    size_t last_c = (size_t)d.offset[0] + (d.shape[0] ? d.shape[0] - 1 : 0);
    return ((last_w * s2 + last_b) * s1 + last_t) * s0 + last_c + 1;
}

// ------------------------------------------------------------------ internal launchers (stream ordered)
namespace wrk {

// wrk_ops.hip
void layer_norm(hipStream_t s, const void* w, const void* b, DTensor x, float eps);
void layer_norm_from(hipStream_t s, const void* w, const void* b, DTensor src, DTensor x, float eps);     // x = LN(src), same shapes
void group_norm(hipStream_t s, const void* w, const void* b, DTensor x, float eps);
void l2_norm(hipStream_t s, DTensor x, float eps);
void token_shift(hipStream_t s, const uint32_t* cursors, DTensor mix, DTensor state, DTensor in, DTensor out, int reversed);
// merged element-wise stages of an RWKV-7 layer over dense f16 [D, T] rows with 64-wide heads (bit-identical to the op chain)
void pre_wkv_v7(hipStream_t s, void* w, void* a, void* k, void* v, void* vv, void* v0, void* n, const void* w0, const void* a0, const void* k_k,
                const void* k_a, const void* v0p, uint32_t D, uint32_t T, bool first_layer, float l2_eps, float* wdec = nullptr);
void post_wkv_v7(hipStream_t s, void* x, const void* r, const void* g, const void* n, const void* gn_w, const void* gn_b, const void* r_k,
                 uint32_t D, uint32_t T, float gn_eps);
// n <= 6 token_shift ops over the same input / state row in one pass (falls back to n launches for views it cannot vectorise)
void token_shift_multi(hipStream_t s, const uint32_t* cursors, const DTensor* mix, const DTensor* out, int n, DTensor state, DTensor in, int reversed);
void transpose(hipStream_t s, DTensor in, DTensor out);
void time_mix_v6(hipStream_t s, const uint32_t* cursors, DTensor decay, const void* u_f32, DTensor state, DTensor k, DTensor v, DTensor r, DTensor x,
                 uint32_t nseq_hint = 0);      // 0: unknown (as many sequences as the state has batches)
void channel_mix_v6(hipStream_t s, const uint32_t* cursors, DTensor state, DTensor r, DTensor v, DTensor x);
void binary(hipStream_t s, int is_mul, DTensor in, DTensor out, uint32_t ax, uint32_t ay, uint32_t ao);
void lerp(hipStream_t s, DTensor x, DTensor y, DTensor f, int reversed);
void blit(hipStream_t s, DTensor in, DTensor out);
void affine(hipStream_t s, DTensor x, float scale, float bias);
void activate(hipStream_t s, DTensor x, uint32_t act);
void control_k_v7(hipStream_t s, const void* p, DTensor a, DTensor k);
void time_mix_v7(hipStream_t s, const uint32_t* cursors, DTensor state, DTensor r, DTensor w, DTensor n, DTensor x, uint32_t nseq_hint = 0, const float* wdec = nullptr);   // nseq_hint 0: unknown; wdec: precomputed decays f32 [D, T] (pre_wkv_v7) or nullptr
void time_first_v7(hipStream_t s, const void* u, DTensor r, DTensor n, DTensor x);
void channel_mix_v7(hipStream_t s, const uint32_t* cursors, DTensor state, DTensor v, DTensor x);
void softmax(hipStream_t s, DTensor x);
void gather_rows_f16(hipStream_t s, const void* table, const uint32_t* ids, void* out, uint32_t d, uint32_t n);
void gather_rows_any(hipStream_t s, DTensor in, const uint32_t* rows, DTensor out, uint32_t n);
void argmax_rows(hipStream_t s, const float* logits, uint32_t v, uint32_t v_stride, uint32_t n, uint32_t* out);

// WRK_TIMING=1 (debug): in-kernel wall-clock stamps of one decode layer, printed after wrk_v7_generate_greedy
unsigned long long* timing_slot(wrk_ctx* ctx, const char* label);   // nullptr unless enabled
void timing_report(wrk_ctx* ctx);

// wrk_matvec.hip
struct MatJob {
    const uint8_t* w;       // matrix data (device layout)
    const uint8_t* aux;
    uint32_t kind, flags;
    uint32_t k, m;
    uint32_t row_bytes;
    DTensor in;             // [K, T, B]
    DTensor out;            // [M, T, B]
    uint32_t act;
    uint32_t sparse;
    uint32_t has_res = 0;   // fused `add`: out = round_to_out_dtype(act(W.x)) + res
    DTensor res{};
    float* amax_val = nullptr;      // optional fused arg-max partials [num_wg][ntok]
    uint32_t* amax_idx = nullptr;
    // fused decode prologue / epilogue (single input vector only; matvec() returns -3 if it cannot honour them)
    uint32_t pro = 0;               // 1: input = mix(LN(in; ln_w, ln_b, pro_eps), prev, mixw)
                                    // 2: input = mixw * r16(r16(GN64(in; ln_w, ln_b, pro_eps)) + prev): the post-WKV stage of a split head
                                    //    (in = WKV output, prev = f32 time_first term, mixw = gate), dmv kernels only
    float pro_eps = 0.0f;
    const void *ln_w = nullptr, *ln_b = nullptr, *mixw = nullptr;
    const float* prev = nullptr;    // f32 shift-state row of this sequence
    void* ln_out = nullptr;         // f16 [K]: LN(in), published by the first workgroup of the job
    const void* carry_src = nullptr;    // epilogue: carry_dst[row] = carry_src[row]
    float* carry_dst = nullptr;
    float scale = 1.0f;                 // wrk_matrix::out_scale
    const void* gate = nullptr;         // f16 [M]: out = round(sigmoid(gate[row]) * round(act(W.x)))  (channel_mix.wgsl:104-106, RWKV-6), before the residual
    unsigned long long* dbg = nullptr;  // WRK_TIMING=1: 16 device timestamps of this launch (first and last workgroup)
    // several input vectors (2 .. 4 sequences decoding together, dmv kernels): element strides from one token's operand to the next
    uint32_t tok_prev_stride = 0, tok_mix_stride = 0, tok_carry_src_stride = 0, tok_carry_dst_stride = 0, tok_gate_stride = 0;
    // decode batches on the matrix cores: scratch of the K-sliced GEMM (f32 partial tiles, per-row-group arrival counters that are
    // zero between launches); taken from the first job of a launch, nullptr = the kernel is not used
    float* ks_part = nullptr;
    uint32_t* ks_cnt = nullptr;
    size_t ks_part_cap = 0;         // floats
    uint32_t ks_cnt_cap = 0;
    // third-generation prefill tile (>= 512 stacked tokens, Q4_K): scratch for the sub-block input sums, taken from the first job of a
    // launch (wrk_ctx::gemm_scratch); nullptr = the kernel is not used
    void* xsum = nullptr;
    size_t xsum_cap = 0;            // bytes
};
uint32_t matvec_num_wg(const MatJob* jobs, int njobs, int num_cu, uint32_t* rows_per_wg);
// dry_run: classify only (0 = a launch would honour every job's prologue / carry request, -3 = it cannot)
// dmv_only: with 2 .. 4 input vectors, -5 unless the multi-token dmv kernels take the launch (the caller then prefers the matrix cores)
int matvec(hipStream_t s, const MatJob* jobs, int njobs, int num_cu, bool dry_run = false, bool no_catchall = false, bool dmv_only = false);
// as matvec, but jobs of several quantised kinds are split into one launch per kind (F16 jobs ride with the first)
int matvec_grouped(hipStream_t s, const MatJob* jobs, int njobs, int num_cu, bool dry_run = false);
// MFMA dequant-GEMM (wrk_gemm.hip); -2 = not applicable (caller uses the matvec kernels)
int matmul_mfma(hipStream_t s, const MatJob& job, int num_cu);
// several matrices x the same tokens in ONE launch (-2 if any job is not for the MFMA path)
int matmul_mfma_multi(hipStream_t s, const MatJob* jobs, int njobs, int num_cu);
uint32_t gemm_min_tokens();
// wrk_quant.hip
void quantize_int8(hipStream_t s, const void* src_f16, uint8_t* dst, uint32_t k, uint32_t m, uint32_t row_bytes);
void quantize_nf4(hipStream_t s, const void* src_f16, const float* levels, uint8_t* dst, uint32_t k, uint32_t m, uint32_t row_bytes);
size_t repack_row_bytes(uint32_t kind, uint32_t k);
uint32_t int8_row_blocks(uint32_t k);   // (min, max) entries stored per Int8 row
int repack_rows(uint32_t kind, uint32_t k, uint32_t m, const uint8_t* src, uint8_t* dst);   // host side
size_t stored_bytes(uint32_t kind, uint32_t k, uint32_t m);

}  // namespace wrk
