/*
 * wrk_hip.h -- C ABI of the MI355X (gfx950) backend for the RWKV hot path of
 * JoelTankard/web-rwkv-gguf.
 *
 * This is the drop-in boundary described in SURVEY.md section 8(b): everything the reference's
 * `Context` (src/context.rs) does through wgpu for the model path -- allocate/upload/read back
 * buffers, build a TensorOp, encode a list of ops, submit -- has one entry point here.  The
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an int32_t status (WRK_OK == 0); out-params come last;
 *     nothing unwinds across the boundary; `wrk_last_error(ctx)` returns the message.
 *   - plain pointers and sizes only; host pointers are borrowed for the duration of the call.
 *   - handles are reference counted where the reference uses Arc<Buffer>.
 *   - tensors are `[x = fastest, y, z, w]` exactly like src/tensor/shape.rs:95-99 and are
 *     addressed through `wrk_view` == `View { shape, stride, offset }` (src/tensor/mod.rs:27-44):
 *     element(b, t, c) = ((b + offset[2]) * stride[1] + (t + offset[1])) * stride[0] + c + offset[0]
 *     (`stride` holds the parent tensor's dims, as in the WGSL `compute_index` helpers).
 *   - `wrk_op_*` functions ENQUEUE work on the context's submission stream (they are the HIP analogue of
 *     building a TensorOp and encoding it); between `wrk_capture_begin/end` ON THE SAME THREAD they are
 *     recorded into a `wrk_program` (a hipGraph) instead, which is the analogue of the CommandBuffer the
 *     reference keeps in an `RnnJob` (src/runtime/v7.rs:423-432) and replays with `queue.submit`.
 *   - threading = the reference's: jobs are ENCODED concurrently on tokio `spawn_blocking` workers
 *     while the runtime task SUBMITS cached ones (src/runtime/mod.rs:139-167) and a dedicated thread blocks in
 *     read-backs (src/context.rs:148-162).  A capture belongs to the thread that began it and records on a
 *     private stream: the submission stream and the read-back stream are never in capture mode, any number of
 *     threads may have a capture open, and launches / uploads / allocations / reads from other threads proceed
 *     meanwhile.  Individual calls on one context are serialised by an internal mutex (a capture is not).
 *     Nothing recorded executes before its program is launched; `wrk_buf_write` is never recorded (it is
 *     `queue.write_buffer`, not an encoder command); `wrk_buf_copy` and the state snapshot copies are.
 */
#ifndef WRK_HIP_H
#define WRK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WRK_ABI_VERSION 1

typedef struct wrk_ctx wrk_ctx;         /* Context              src/context.rs:51-64        */
typedef struct wrk_buf wrk_buf;         /* Arc<Buffer>          src/context.rs:368-394      */
typedef struct wrk_matrix wrk_matrix;   /* enum Matrix          src/tensor/matrix.rs:82-131 */
typedef struct wrk_program wrk_program; /* Vec<CommandBuffer>   src/tensor/ops.rs:79-143    */
typedef struct wrk_v7_model wrk_v7_model; /* v7::Model          src/runtime/v7.rs:35-142    */
typedef struct wrk_v7_state wrk_v7_state; /* v7::State          src/runtime/v7.rs:146-277   */

enum {
    WRK_OK = 0,
    WRK_E_ARG = 1,          /* TensorError::{Shape,Size,Type,...}                      */
    WRK_E_OOM = 2,
    WRK_E_HIP = 3,          /* ContextError / device lost                              */
    WRK_E_UNSUPPORTED = 4
};

enum { WRK_F16 = 0, WRK_F32 = 1, WRK_U8 = 2, WRK_U32 = 3 };

/* Activation (src/tensor/ops.rs:146-160, definitions :205-235) */
enum {
    WRK_ACT_NONE = 0,
    WRK_ACT_SQUARED_RELU = 1,
    WRK_ACT_TANH = 2,
    WRK_ACT_STABLE_EXP = 3,
    WRK_ACT_OPPOSITE_EXP = 4,
    WRK_ACT_SOFTPLUS = 5,
    WRK_ACT_SIGMOID = 6,
    WRK_ACT_SILU = 7
};

/* Matrix variants (src/tensor/matrix.rs:82-131).  Values of the GGUF kinds equal the ggml type
 * ids (src/runtime/gguf.rs:888-923) so a loader can pass `info.tensor_type` through. */
enum {
    WRK_MAT_F32 = 0,        /* converted to f16 on upload (loader.rs:117-121)           */
    WRK_MAT_F16 = 1,
    WRK_MAT_Q8_0 = 8,
    WRK_MAT_Q4_K = 12,
    WRK_MAT_Q5_K = 13,
    WRK_MAT_Q6_K = 14,
    WRK_MAT_INT8 = 100,     /* web-rwkv Int8: u8 + per-128 (min,max) f16                 */
    WRK_MAT_NF4 = 101       /* web-rwkv NF4 : u4 + per-64 absmax f16 + 16-entry table    */
};

/* wrk_matrix_create flags */
enum {
    WRK_MATRIX_EXACT = 0,        /* ggml-canonical inline dequantisation in f32 (north_star)          */
    WRK_MATRIX_ROUND_F16 = 1     /* reproduce the reference at HEAD: every dequantised weight is
                                    rounded to f16 first (gguf.rs:129,135)               */
};

typedef struct wrk_view {
    uint32_t shape[4];
    uint32_t stride[4];
    uint32_t offset[4];
} wrk_view;

typedef struct wrk_tensor {     /* TensorGpuView: buffer + dtype + view */
    wrk_buf* buf;
    uint32_t dtype;             /* WRK_F16 / WRK_F32 */
    wrk_view view;
} wrk_tensor;

/* ---------------------------------------------------------------- context (src/context.rs) */
int32_t wrk_abi_version(void);
/* ContextBuilder::build (context.rs:113-165) */
int32_t wrk_ctx_create(int32_t device, wrk_ctx** out);
/* Drop for Context (context.rs:66-78) */
int32_t wrk_ctx_destroy(wrk_ctx* ctx);
const char* wrk_last_error(wrk_ctx* ctx);
/* device.poll(Wait) (context.rs:439-470, v7.rs:1076-1080) */
int32_t wrk_ctx_sync(wrk_ctx* ctx);
/* the HIP stream ops are enqueued on (for callers that time with events or interoperate) */
void* wrk_ctx_stream(wrk_ctx* ctx);

/* ---------------------------------------------------------------- buffers */
/* checkout_buffer(_init) (context.rs:368-394), TensorGpu::from_data_u8 (tensor/mod.rs:603-626) */
int32_t wrk_buf_create(wrk_ctx* ctx, size_t bytes, const void* init_or_null, wrk_buf** out);
int32_t wrk_buf_retain(wrk_buf* buf);
int32_t wrk_buf_release(wrk_buf* buf);                     /* TensorGpu::destroy (tensor/mod.rs:797) */
size_t wrk_buf_size(const wrk_buf* buf);
void* wrk_buf_device_ptr(const wrk_buf* buf);
/* TensorGpu::load / load_batch -> queue.write_buffer (tensor/mod.rs:774-795); stream ordered,
 * the source is copied before the call returns */
int32_t wrk_buf_write(wrk_ctx* ctx, wrk_buf* buf, size_t offset, const void* src, size_t bytes);
/* TensorGpu::back / read_back_buffer (tensor/mod.rs:671-714, context.rs:439-470); blocking */
int32_t wrk_buf_read(wrk_ctx* ctx, const wrk_buf* buf, size_t offset, void* dst, size_t bytes);
/* copy_tensor(_batch) (tensor/ops.rs:35-75) */
int32_t wrk_buf_copy(wrk_ctx* ctx, const wrk_buf* src, size_t src_off, wrk_buf* dst, size_t dst_off, size_t bytes);

/* ---------------------------------------------------------------- programs (encode / submit) */
/* Context::encode (ops.rs:79-143): ops enqueued between begin/end are recorded, not run */
int32_t wrk_capture_begin(wrk_ctx* ctx);
int32_t wrk_capture_end(wrk_ctx* ctx, wrk_program** out);
/* queue.submit(commands) (v7.rs:476-479) */
int32_t wrk_program_launch(wrk_ctx* ctx, wrk_program* prog);
int32_t wrk_program_destroy(wrk_program* prog);

/* ---------------------------------------------------------------- matrices */
/* Loader::load_matrix / try_load_matrix_direct / load_matrix_f16 (loader.rs:617-641,756-921).
 * `data` is the raw GGUF block stream (row-major, rows of K elements, M rows) for the GGUF
 * kinds, or row-major f16/f32 values.  Blocks are re-laid-out on upload;
 * no bytes are added to the weight stream beyond 16-byte row alignment. */
int32_t wrk_matrix_create(wrk_ctx* ctx, uint32_t kind, uint32_t k, uint32_t m,
                          const void* data, size_t bytes, uint32_t flags, wrk_matrix** out);
/* web-rwkv's own formats through wrk_matrix_create (Matrix::Int8 { w, m } / Matrix::Fp4 { w, q, m },
 * matrix.rs:79-130; the direct-load arms loader.rs:808-820, 901-918):
 *   WRK_MAT_INT8: data = u8 codes [K*M] ++ (min, max) f16 pairs, one per 128 flattened elements; K % 16 == 0, K*M % 128 == 0
 *   WRK_MAT_NF4 : data = nibbles [K*M/2] (element 2i in the low nibble) ++ absmax f16, one per 64 flattened
 *                 elements, optionally ++ the 16 f32 levels of `q` (default: the NF4 levels matrix.rs:50-67;
 *                 pass Float4Quant::new_student's for SF4); K % 64 == 0
 *
 * Matrix::quant_u8 / quant_nf4 / quant_sf4 (matrix.rs:211-271; quant_mat_int8.wgsl, quant_mat_nf4.wgsl):
 * on-device quantisation of an f16 [K, M] tensor (`f16_data`: M rows of K values).  `levels`: NULL for the
 * NF4 levels, else 16 f32 (SF4); ignored for WRK_MAT_INT8. */
int32_t wrk_matrix_quantize(wrk_ctx* ctx, uint32_t kind, uint32_t k, uint32_t m,
                            const wrk_buf* f16_data, const float* levels, wrk_matrix** out);
/* read the quantised planes back in wrk_matrix_create's layout (tests; Matrix serialisation): INT8 / NF4 only */
int32_t wrk_matrix_export(wrk_matrix* mat, void* dst, size_t capacity, size_t* bytes);
/* Loader::load_matrix_discount (loader.rs:923-951): the reference multiplies every weight by 2^-(layer / rescale) at load,
 * which forces the F16 path.  A power-of-two factor commutes with the contraction, so the blocks stay quantised and the
 * factor is applied to the f32 dot product instead: y = act(scale * (W . x)). */
int32_t wrk_matrix_set_scale(wrk_matrix* mat, float scale);
int32_t wrk_matrix_release(wrk_matrix* mat);
/* stored bytes read per full pass over the matrix (the roofline's algorithmic bytes) */
size_t wrk_matrix_stream_bytes(const wrk_matrix* mat);

/* ---------------------------------------------------------------- TensorOp constructors
 * One function per reference op on the V7/V6 path.  Shapes are validated as the
 * reference does (TensorError -> WRK_E_ARG). */

/* Matrix::matmul_op (matrix.rs:185-209): output[M, T, B] = act(W[K, M] . input[K, T, B]).
 * turbo != 0 selects the MFMA GEMM kernels for >= 2 stacked tokens (the reference takes T % 32 == 0 there; here token
 * tiles are padded, so any count works); sparse = matmul_op_sparse */
int32_t wrk_op_matmul(wrk_ctx* ctx, const wrk_matrix* mat, const wrk_tensor* input, const wrk_tensor* output,
                      uint32_t act, int32_t turbo, int32_t sparse);
/* TensorOp::layer_norm (ops.rs:407-454): x[C, T, B] in place, w/b f16 [C] */
int32_t wrk_op_layer_norm(wrk_ctx* ctx, const wrk_buf* w, const wrk_buf* b, const wrk_tensor* x, float eps);
/* TensorOp::group_norm (ops.rs:460-508): x[S, H, T], w/b f16 [S*H] */
int32_t wrk_op_group_norm(wrk_ctx* ctx, const wrk_buf* w, const wrk_buf* b, const wrk_tensor* x, float eps);
/* TensorOp::l2_norm (ops.rs:642-691): x[S, H, T] */
int32_t wrk_op_l2_norm(wrk_ctx* ctx, const wrk_tensor* x, float eps);
/* TensorOp::token_shift (ops.rs:2119-2187): cursors u32 [T]; time_mix [C, 1 or T, I] (one factor vector, or V6's
 * per-token factors, I shifts per call); state f32 view [C, 1, B]; input [C, T, 1]; output [C, T, I] */
int32_t wrk_op_token_shift(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* time_mix, const wrk_tensor* state,
                           const wrk_tensor* input, const wrk_tensor* output, int32_t reversed);
/* TensorOp::transpose (ops.rs:2847-2905): output[C, B, T] = input[C, T, B] */
int32_t wrk_op_transpose(wrk_ctx* ctx, const wrk_tensor* input, const wrk_tensor* output);
/* TensorOp::time_mix_v6 (ops.rs:2327-2394): time_decay/k/v/r/x [S, H, T]; time_first f32 [S*H]; state f32 view [C, S+1, B] */
int32_t wrk_op_time_mix_v6(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* time_decay, const wrk_buf* time_first,
                           const wrk_tensor* state, const wrk_tensor* k, const wrk_tensor* v, const wrk_tensor* r, const wrk_tensor* x);
/* TensorOp::channel_mix (ops.rs:2586-2641, V6): x <- sigmoid(r) * v, plus the ffn shift-state save */
int32_t wrk_op_channel_mix(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* state, const wrk_tensor* r, const wrk_tensor* v, const wrk_tensor* x);
/* TensorOp::add_activate / mul_activate (ops.rs:1953-2117): output = act_o(act_x(input) (+|*) act_y(output)),
 * input broadcast over T/B when its extent is 1 */
int32_t wrk_op_add(wrk_ctx* ctx, const wrk_tensor* input, const wrk_tensor* output, uint32_t act_x, uint32_t act_y, uint32_t act_o);
int32_t wrk_op_mul(wrk_ctx* ctx, const wrk_tensor* input, const wrk_tensor* output, uint32_t act_x, uint32_t act_y, uint32_t act_o);
/* TensorOp::lerp (ops.rs:3010-3076): y <- reversed ? mix(y, x, f) : mix(x, y, f) */
int32_t wrk_op_lerp(wrk_ctx* ctx, const wrk_tensor* x, const wrk_tensor* y, const wrk_tensor* f, int32_t reversed);
/* TensorOp::blit (ops.rs:2741-2793): strided copy with dtype conversion */
int32_t wrk_op_blit(wrk_ctx* ctx, const wrk_tensor* input, const wrk_tensor* output);
/* TensorOp::affine (ops.rs:3078-3117): x <- scale * x + bias */
int32_t wrk_op_affine(wrk_ctx* ctx, const wrk_tensor* x, float scale, float bias);
/* TensorOp::activate (ops.rs:2699-2739) */
int32_t wrk_op_activate(wrk_ctx* ctx, const wrk_tensor* x, uint32_t act);
/* TensorOp::control_k_v7 (ops.rs:2526-2584): k <- k * (1 + (a - 1) * p), p f16 [C] */
int32_t wrk_op_control_k_v7(wrk_ctx* ctx, const wrk_buf* p, const wrk_tensor* a, const wrk_tensor* k);
/* TensorOp::time_mix_v7 (ops.rs:2405-2467): state f32 view [C, S+1, B]; r, w, x [S, H, T];
 * n [S, H, T, 4] = (k, v, a, kk) */
int32_t wrk_op_time_mix_v7(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* state, const wrk_tensor* r,
                           const wrk_tensor* w, const wrk_tensor* n, const wrk_tensor* x);
/* TensorOp::time_first_v7 (ops.rs:2469-2524): u f16 [S, H] */
int32_t wrk_op_time_first_v7(wrk_ctx* ctx, const wrk_buf* u, const wrk_tensor* r, const wrk_tensor* n, const wrk_tensor* x);
/* TensorOp::channel_mix_v7 (ops.rs:2643-2697): state f32 view [C, 1, B]; v, x [C, T] */
int32_t wrk_op_channel_mix_v7(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* state, const wrk_tensor* v, const wrk_tensor* x);
/* TensorOp::softmax (ops.rs:300-344) -- "next" row (f)1 */
int32_t wrk_op_softmax(wrk_ctx* ctx, const wrk_tensor* x);

/* ---------------------------------------------------------------- fused RWKV-7 fast path
 * New entry points (no reference counterpart at this granularity): they run what
 * v7::Bundle::dispatch (v7.rs:598-713) encodes for one chunk, with the elementwise ops fused
 * into the producing kernels.  Results follow the same rounding points as the op-by-op path. */

typedef struct wrk_v7_layer_desc {
    /* f16 vectors [D] unless noted; v7.rs:77-128 */
    const wrk_buf *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const wrk_buf *x_r, *x_w, *x_k, *x_v, *x_a, *x_g;
    const wrk_buf *w0, *a0, *v0;
    const wrk_matrix *w1, *w2, *a1, *a2, *g1, *g2, *v1, *v2;     /* v1/v2/v0 unused on layer 0 */
    const wrk_buf *r_k, *k_k, *k_a;
    const wrk_matrix *w_k, *w_v, *w_r, *w_o;
    const wrk_buf *gn_w, *gn_b;
    const wrk_buf* ffn_x_k;
    const wrk_matrix *ffn_w_k, *ffn_w_v;
} wrk_v7_layer_desc;

typedef struct wrk_v7_model_desc {
    uint32_t num_layer, num_emb, num_hidden, num_vocab, num_head;
    uint32_t lora_w, lora_a, lora_g, lora_v;          /* v7::CustomInfo (v7.rs:54-60) */
    uint32_t rescale;                                 /* Model::rescale, default 1024 (v7.rs:50) */
    const wrk_buf *ln0_w, *ln0_b, *ln_out_w, *ln_out_b;
    const wrk_buf* emb_f16;                           /* [D, V] f16 table on device, or NULL when the
                                                         caller gathers rows itself (v7.rs:438-474) */
    const wrk_matrix* head;
    const wrk_v7_layer_desc* layers;
} wrk_v7_model_desc;

/* ModelBuilder::build_v7 after the tensors are uploaded (v7.rs:1038-1227); retains every handle */
int32_t wrk_v7_model_create(wrk_ctx* ctx, const wrk_v7_model_desc* desc, wrk_v7_model** out);
int32_t wrk_v7_model_destroy(wrk_v7_model* model);
/* algorithmic bytes read+written per decoded token for `num_batch` sequences */
size_t wrk_v7_model_token_bytes(const wrk_v7_model* model, uint32_t num_batch);

/* v7::Bundle::new state allocation (v7.rs:514-536): L tensors f32 [D, S+2, B], zeroed */
int32_t wrk_v7_state_create(wrk_ctx* ctx, const wrk_v7_model* model, uint32_t num_batch, wrk_v7_state** out);
int32_t wrk_v7_state_destroy(wrk_v7_state* state);
/* State::load / State::back (v7.rs:152-170, 210-217): host f32 [D, S+2, L] for one batch */
int32_t wrk_v7_state_load(wrk_ctx* ctx, wrk_v7_state* state, uint32_t batch, const float* src);
int32_t wrk_v7_state_back(wrk_ctx* ctx, const wrk_v7_state* state, uint32_t batch, float* dst);
/* State::read / State::write (v7.rs:229-262): device-resident snapshot of one batch, f32 [D, S+2, L] in `buf`
 * (stream-ordered device-to-device copies, no host round trip: multi-session serving keeps snapshots in HBM) */
int32_t wrk_v7_state_read(wrk_ctx* ctx, const wrk_v7_state* state, uint32_t batch, wrk_buf* buf);
int32_t wrk_v7_state_write(wrk_ctx* ctx, wrk_v7_state* state, uint32_t batch, const wrk_buf* buf);

/* One RnnJob (load + submit + back, v7.rs:434-492) for a chunk of `num_token` stacked tokens.
 *   tokens     u32 [num_token] ids (device gather from emb_f16), or NULL with
 *   emb_rows   f16 [D, num_token] rows gathered by the caller (Token::Embed / CPU gather)
 *   cursors    packed Cursor per token (tensor/mod.rs:53-60)
 *   headers    stacked row indices fed to the head (RnnRedirect::headers, rnn.rs:41-81)
 *   logits     host f32 [num_vocab, num_header] or NULL
 *   argmax     host u32 [num_header] or NULL (greedy token per header row, computed on device)
 *   mode       0 = op-by-op (one kernel per reference TensorOp, the launch list of v7.rs:716-1007);
 *              1 = fast paths: one token per sequence -> the fused 5-launch decode layer; multi-token chunks -> the
 *                  same op list with merged launches (six shifts in one pass, projections grouped per stage, the
 *                  element-wise chains around the WKV kernel as one kernel each, residual adds in the GEMM epilogue),
 *                  whose logits and state are bit-identical to mode 0 above 64 stacked tokens
 */
int32_t wrk_v7_infer(wrk_ctx* ctx, wrk_v7_model* model, wrk_v7_state* state,
                     const uint32_t* tokens, const uint16_t* emb_rows, const uint32_t* cursors, uint32_t num_token,
                     const uint32_t* headers, uint32_t num_header, float* logits, uint32_t* argmax, uint32_t mode);

/* Bundle::<F>::new (v7.rs:514-536; Runtime<F> v7.rs:281-364 is generic over the activation type): WRK_F16 = Bundle::<f16>, the
 * reference's default build; WRK_F32 = Bundle::<f32>: every frame buffer but `input` holds f32, matmuls read f32 inputs
 * (IN_FP32 shader variants), the op-by-op launch list runs whatever `mode` says.  Call between jobs; drops cached programs. */
int32_t wrk_v7_model_set_frame_dtype(wrk_ctx* ctx, wrk_v7_model* model, uint32_t dtype);

/* The persistent batch-1 decode engine (one launch for all layers of a token; the device-side analogue of the reference's
 * speculative job queue, runtime/mod.rs:110-209, which keeps the next job ready while the current one runs).  It is built on the
 * first one-token job of a model and used by mode 1 whenever it exists and WRK_ENGINE != 0.  Returns 1 when the engine exists,
 * 0 when it does not (reason copied to `why`, NUL-terminated, when why != NULL), < 0 on error.  Builds it if no job has run yet. */
int32_t wrk_v7_model_engine_status(wrk_ctx* ctx, wrk_v7_model* model, char* why, size_t capacity);

/* Parity instrumentation at the reference's own seam: v7::Hook / HookMap closures receive the `Frame` (state + Runtime<F>
 * buffers) at every stage of a layer (v7.rs:386-421, 497-502) and examples/inspect.rs:100-248 reads each buffer back per layer.
 *   wrk_v7_infer_layer: run ONLY `layer` of a job (mode as wrk_v7_infer) on a caller-supplied layer input x [D, num_token] and
 *                       layer-0 value v_first [D, num_token] (NULL for layer 0), both in the frame dtype; the state slice of the layer advances
 *   wrk_v7_frame_read : TensorGpu::back of one frame buffer, by inspect.rs's name (x, att_x, att_r, att_w, att_k, att_v, att_a, att_g,
 *                       att_o, att_kk, att_vv, att_n, att_rx ... att_gx, aux_w/a/g/v, ffn_x, ffn_kx, ffn_k, ffn_v; att_x_ln = LN1(x) of the
 *                       fused decode path); dst NULL returns the size.  In mode 1 only the buffers the fused kernels materialise are meaningful. */
int32_t wrk_v7_infer_layer(wrk_ctx* ctx, wrk_v7_model* model, wrk_v7_state* state, uint32_t layer, const void* x, const void* v_first,
                           const uint32_t* cursors, uint32_t num_token, uint32_t mode);
int32_t wrk_v7_frame_read(wrk_ctx* ctx, wrk_v7_model* model, const char* name, uint32_t num_token, void* dst, size_t capacity, size_t* bytes);

/* Greedy decode loop kept on the device (the reference's bench loop, examples/bench.rs:224-236,
 * with softmax+argmax moved on device): every sequence b feeds `first_tokens[b]`, then its own
 * argmax, for `steps` steps.  out_tokens: host u32 [steps, num_batch] or NULL.  One hipGraph per
 * step shape is built on first use and replayed.  elapsed_ms_or_null receives the HIP-event time
 * of the `steps` replays on the context's stream.
 * mode: bits 0-7 as wrk_v7_infer (0 op-by-op, 1 fused); bits 8-15 = G > 1: the num_batch INDEPENDENT sequences (separate state
 * slices, v7.rs:519-521) are dealt in contiguous blocks over G concurrent pipelines -- each with its own Runtime<F> frame, cached
 * step program and HIP stream, all sharing the one set of weights -- so that several latency-bound decode pipelines overlap on the
 * GPU (the in-GPU analogue of sharding streams over GPUs; results equal running each block on its own). */
int32_t wrk_v7_generate_greedy(wrk_ctx* ctx, wrk_v7_model* model, wrk_v7_state* state,
                               const uint32_t* first_tokens, uint32_t num_batch, uint32_t steps,
                               uint32_t* out_tokens, float* last_logits_or_null, float* elapsed_ms_or_null, uint32_t mode);

/* ---------------------------------------------------------------- RWKV-6 (v6::Model, src/runtime/v6.rs)
 * Same chunk semantics, state layout ([D, S+2, B] per layer: v6.rs:150-214 == v7) and entry points as the V7
 * runner; one kernel per reference TensorOp (v6.rs:701-958), decode steps replayed from a hipGraph. */
typedef struct wrk_v6_model wrk_v6_model;

typedef struct wrk_v6_layer_desc {
    const wrk_buf *ln1_w, *ln1_b, *ln2_w, *ln2_b;               /* f16 [D] */
    const wrk_buf *time_decay;                                  /* f16 [D]                        (v6.rs:1051) */
    const wrk_buf *time_first;                                  /* f32 [D] = [S, H]               (v6.rs:1052, load_vector_f32) */
    const wrk_buf *time_mix_x;                                  /* f16 [D] */
    const wrk_buf *time_mix;                                    /* f16 [D, 1, 5] = w, k, v, r, g  (v6.rs:1054-1071) */
    const wrk_matrix *time_decay_w1, *time_decay_w2, *time_mix_w1;
    const wrk_matrix *time_mix_w2[5];                           /* the batched [R, D, 5] matrix as five [R -> D] matrices */
    const wrk_matrix *w_k, *w_v, *w_r, *w_g, *w_o;
    const wrk_buf *gn_w, *gn_b;
    const wrk_buf *ffn_mix_k, *ffn_mix_r;
    const wrk_matrix *ffn_w_k, *ffn_w_v, *ffn_w_r;
} wrk_v6_layer_desc;

typedef struct wrk_v6_model_desc {
    uint32_t num_layer, num_emb, num_hidden, num_vocab, num_head;
    uint32_t time_mix, time_decay;                              /* v6::CustomInfo (v6.rs:53-59) */
    uint32_t rescale;                                           /* default 6 (v6.rs:49) */
    const wrk_buf *ln0_w, *ln0_b, *ln_out_w, *ln_out_b, *emb_f16;
    const wrk_matrix* head;
    const wrk_v6_layer_desc* layers;
} wrk_v6_model_desc;

int32_t wrk_v6_model_create(wrk_ctx* ctx, const wrk_v6_model_desc* desc, wrk_v6_model** out);
int32_t wrk_v6_model_destroy(wrk_v6_model* model);
size_t wrk_v6_model_token_bytes(const wrk_v6_model* model, uint32_t num_batch);
/* v6::Bundle::new state allocation; the handle type is shared with V7 (identical layout) */
int32_t wrk_v6_state_create(wrk_ctx* ctx, const wrk_v6_model* model, uint32_t num_batch, wrk_v7_state** out);
/* as wrk_v7_infer / wrk_v7_generate_greedy.  mode 1: jobs whose tokens are one per sequence (decode) run the fused
 * 7-launch layer (LN prologues in the consumer matvec, v6_mix / v6_head kernels, gated ffn epilogue); everything else,
 * and mode 0, runs one kernel per reference op (v6.rs:701-958) */
int32_t wrk_v6_infer(wrk_ctx* ctx, wrk_v6_model* model, wrk_v7_state* state,
                     const uint32_t* tokens, const uint16_t* emb_rows, const uint32_t* cursors, uint32_t num_token,
                     const uint32_t* headers, uint32_t num_header, float* logits, uint32_t* argmax, uint32_t mode);
int32_t wrk_v6_generate_greedy(wrk_ctx* ctx, wrk_v6_model* model, wrk_v7_state* state,
                               const uint32_t* first_tokens, uint32_t num_batch, uint32_t steps,
                               uint32_t* out_tokens, float* last_logits_or_null, float* elapsed_ms_or_null, uint32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* WRK_HIP_H */
