/*
 * wrk_runtime.h -- C surface of the native host layer that sits ABOVE the backend boundary
 * (wrk_hip.h).  In the reference this layer is Rust and stays Rust (north_star); no Rust toolchain
 * exists in this image, so it is restated in C++ (web-rwkv-gguf_amd/host/) with the reference's
 * names and semantics, and exported here so tests and bench.py can drive it through ctypes:
 *
 *   GgufReader            src/runtime/gguf.rs:1150-1158, 1331-1413, 1540-1795
 *   Loader::info          src/runtime/loader.rs:238-371
 *   ModelBuilder::build_v7 src/runtime/v7.rs:1038-1227  (+ loader.rs:563-951)
 *   v7::Bundle::new       src/runtime/v7.rs:514-536
 *   RnnInput / RnnIter / RnnInfo::redirect   src/runtime/infer/rnn.rs:41-81, 204-335
 *   SimpleRuntime::infer  src/runtime/mod.rs:238-263
 *   State::{load, back}   src/runtime/v7.rs:152-170, 210-217
 */
#ifndef WRK_RUNTIME_H
#define WRK_RUNTIME_H

#include "wrk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wrk_gguf wrk_gguf;
typedef struct wrk_rnn_input wrk_rnn_input;
typedef struct wrk_rnn_iter wrk_rnn_iter;
typedef struct wrk_runtime wrk_runtime;

/* thread-local message of the last failing wrk_gguf_ / wrk_rnn_ / wrk_runtime_ call */
const char* wrk_host_last_error(void);

/* ------------------------------------------------------------ GgufReader */
/* GgufReader::new over a memory-mapped file (examples/chat.rs:208-214 mmaps the model) */
int32_t wrk_gguf_open(const char* path, wrk_gguf** out);
/* GgufReader::new(&data): `data` is borrowed and must outlive the reader */
int32_t wrk_gguf_from_memory(const void* data, size_t bytes, wrk_gguf** out);
int32_t wrk_gguf_close(wrk_gguf* g);
uint32_t wrk_gguf_version(const wrk_gguf* g);
uint64_t wrk_gguf_tensor_data_offset(const wrk_gguf* g);
/* Reader::contains */
int32_t wrk_gguf_contains(const wrk_gguf* g, const char* name);
/* Reader::shape: safetensors-order dims; returns WRK_E_ARG when absent */
int32_t wrk_gguf_shape(const wrk_gguf* g, const char* name, uint32_t dims[4], uint32_t* ndim);
/* Reader::tensor converted by tensor_f16_from_reader (loader.rs:104-132): f16 bits, K-quants
 * dequantised on the CPU exactly as gguf.rs:11-274 */
int32_t wrk_gguf_tensor_f16(const wrk_gguf* g, const char* name, uint16_t* out, size_t capacity, size_t* count);
/* Reader::quantized_tensor with the K-quant gate lifted: ggml type id + raw block pointer */
int32_t wrk_gguf_raw(const wrk_gguf* g, const char* name, uint32_t* ggml_type, const void** data, size_t* bytes);
/* metadata lookup (u32/u64 values only) */
int32_t wrk_gguf_meta_u64(const wrk_gguf* g, const char* key, uint64_t* out);

typedef struct wrk_model_info {         /* ModelInfo (model.rs:30-63) + v7::CustomInfo */
    uint32_t version;                   /* 6 or 7 */
    uint32_t num_layer, num_emb, num_hidden, num_vocab, num_head;
    uint32_t lora_w, lora_a, lora_g, lora_v;   /* V7: w/a/g/v ranks; V6: lora_w = time_mix, lora_a = time_decay (v6::CustomInfo) */
} wrk_model_info;
/* Loader::info */
int32_t wrk_gguf_info(const wrk_gguf* g, wrk_model_info* out);

/* read_state (v7.rs:1229-1262, v6.rs:1176-1208): the pre-trained initial state of a state-tuned model file,
 * `blocks.N.att.time_state` [H, S, S] placed in rows 1..S of a zeroed f32 [D, S+2, L] state (count = L*(S+2)*D);
 * feed it to wrk_v7_state_load.  WRK_E_ARG when the file has no time_state tensors. */
int32_t wrk_gguf_read_state(const wrk_gguf* g, float* out, size_t capacity, size_t* count);

/* quantile_student (src/tensor/matrix.rs:29-44) = the 16 levels of Float4Quant::new_student(nu) ("SF4"; nu = 5 for most cases):
 * Student-t quantiles at p = delta + i (0.5 - delta) / 7 (i = 0..6) and 0.5 + i (0.5 - delta) / 8 (i = 0..8) with
 * delta = (1/32 + 1/30) / 2, divided by the largest.  The reference takes the inverse CDF from the un-vendored statrs 0.18.0
 * (Cargo.toml:51); here it is the root of the t CDF (regularised incomplete beta, continued fraction) refined by Newton steps in
 * f64 -- PARITY UNPINNED against statrs, pinned against scipy.stats.t.ppf to 1e-6 in tests/test_abi_host.py.
 * Pass the result to wrk_matrix_quantize(WRK_MAT_NF4, ..., levels) for Matrix::quant_sf4 (matrix.rs:251-271). */
int32_t wrk_quantile_student(double nu, float* out16);

/* ------------------------------------------------------------ RnnInput / RnnIter */
enum { WRK_RNN_NONE = -1, WRK_RNN_LAST = 0, WRK_RNN_FULL = 1 };

/* RnnInput::new(batches, token_chunk_size) with empty batches, option Last */
int32_t wrk_rnn_input_create(uint32_t num_batch, uint32_t token_chunk_size, wrk_rnn_input** out);
int32_t wrk_rnn_input_destroy(wrk_rnn_input* in);
uint32_t wrk_rnn_input_token_chunk_size(const wrk_rnn_input* in);
/* RnnInputBatch::append / option */
int32_t wrk_rnn_input_append(wrk_rnn_input* in, uint32_t batch, const uint32_t* tokens, uint32_t n);
int32_t wrk_rnn_input_set_option(wrk_rnn_input* in, uint32_t batch, int32_t option);
uint32_t wrk_rnn_input_remaining(const wrk_rnn_input* in, uint32_t batch);
/* JobInput::step */
int32_t wrk_rnn_input_step(wrk_rnn_input* in);
/* (&input).into_iter() */
int32_t wrk_rnn_iter_create(const wrk_rnn_input* in, wrk_rnn_iter** out);
int32_t wrk_rnn_iter_destroy(wrk_rnn_iter* it);
/* RnnIter::next -> RnnInfo: lens[num_batch], options[num_batch] (WRK_RNN_*) */
int32_t wrk_rnn_iter_next(wrk_rnn_iter* it, uint32_t* lens, int32_t* options);
/* RnnInfo::redirect: headers (capacity = sum lens), inputs/outputs as (start,end) pairs [num_batch][2] */
int32_t wrk_rnn_redirect(const uint32_t* lens, const int32_t* options, uint32_t num_batch,
                         uint32_t* headers, uint32_t* num_header, uint32_t* inputs, uint32_t* outputs);

/* ------------------------------------------------------------ ModelBuilder + Bundle + Runtime */
enum {
    WRK_WEIGHTS_INLINE = 0,       /* north_star: raw blocks on device, inline dequant in f32              */
    WRK_WEIGHTS_INLINE_F16 = 1,   /* same kernels, each dequantised weight rounded to f16 (reference's
                                     effective arithmetic)                                      */
    WRK_WEIGHTS_REFERENCE = 2     /* the reference at HEAD literally: CPU dequant to f16, F16 matrices     */
};

enum { WRK_QUANT_NONE = 0, WRK_QUANT_INT8 = 1, WRK_QUANT_NF4 = 2 };   /* Quant (runtime/model.rs); SF4 needs the caller's
                                                                          Student-t levels: wrk_matrix_quantize only */

typedef struct wrk_build_options {
    uint32_t rescale;             /* ModelBuilder::rescale, 0 = default 1024 (v7.rs:50)  */
    uint32_t weights;             /* WRK_WEIGHTS_* */
    /* ModelBuilder::quant (HashMap<usize, Quant>): quant[l] = WRK_QUANT_* for layer l < num_quant, NONE beyond.
     * Applies to att.{key,value,receptance,gate,output} and ffn.{key,value,receptance} (v7.rs:1168-1186,
     * v6.rs:1110-1132) exactly as Loader::load_matrix / load_matrix_discount do (loader.rs:756-951):
     * Q8_0 + Int8 and Q4_0 + NF4 tensors loaded without discount are repacked on the host (gguf.rs:429-627),
     * everything else goes f16 -> (discount) -> on-device quantisation. */
    const uint8_t* quant;
    uint32_t num_quant;
} wrk_build_options;

/* ModelBuilder::new(&context, reader).build_v7() + v7::Bundle::new(model, num_batch) + SimpleRuntime::new */
int32_t wrk_runtime_create(wrk_ctx* ctx, const wrk_gguf* g, const wrk_build_options* opt, uint32_t num_batch, wrk_runtime** out);
int32_t wrk_runtime_destroy(wrk_runtime* rt);
int32_t wrk_runtime_info(const wrk_runtime* rt, wrk_model_info* out);
wrk_v7_model* wrk_runtime_model(wrk_runtime* rt);
wrk_v7_state* wrk_runtime_state(wrk_runtime* rt);
/* non-NULL when the file is an RWKV-6 model (ModelVersion::V6); wrk_runtime_model is NULL then */
wrk_v6_model* wrk_runtime_model_v6(wrk_runtime* rt);

/* runtime.infer(input) -> (input, output) (mod.rs:238-263): dispatch the next chunk, load, submit,
 * read back, then input.step().  `logits` receives the rows of every batch back to back
 * ([num_vocab] each, in batch order); rows[b] = number of rows for batch b (RnnOutputBatch sizes).
 * Returns WRK_E_ARG with "input iterator exhausted" when nothing is left (RuntimeError::InputExhausted). */
int32_t wrk_runtime_infer(wrk_runtime* rt, wrk_rnn_input* in, float* logits, size_t capacity_rows, uint32_t* rows, uint32_t mode);

#ifdef __cplusplus
}
#endif
#endif
