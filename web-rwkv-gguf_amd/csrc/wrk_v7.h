// Internal definitions of the RWKV-7 model / state handles.
#pragma once
#include <atomic>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "wrk_internal.h"
#include "wrk_device.h"

struct wrk_v7_state {
    // captured graphs bake the state's addresses and strides in: they are keyed by this id, never reused, rather than by
    // the handle's address (a destroyed state's address can come back with another num_batch)
    const void* uid = next_uid();
    static const void* next_uid() { static std::atomic<uintptr_t> n{1}; return (const void*)(n.fetch_add(1) << 4); }
    wrk_ctx* ctx = nullptr;
    uint32_t num_layer = 0, num_emb = 0, head_size = 0, num_batch = 0;
    float* data = nullptr;      // [L][B][S+2][D] f32 == L tensors [D, S+2, B] (v7.rs:514-527)
    size_t layer_elems() const { return (size_t)num_batch * (head_size + 2) * num_emb; }
    float* layer_ptr(uint32_t l) const { return data + layer_elems() * l; }
};

struct V7Scratch {      // Runtime<f16> + Header<f16> (v7.rs:281-383); f16 unless noted
    void *input, *x, *att_x, *att_v0, *rx, *wx, *kx, *vx, *ax, *gx, *r, *w, *k, *v, *a, *g, *o, *kk, *vv, *n;
    void *aux_w, *aux_a, *aux_g, *aux_v, *ffn_x, *ffn_kx, *ffn_k, *ffn_v, *ln_tmp, *head_x;
    float* head_o;      // f32 [V, num_header]
    float* ks_part; uint32_t* ks_cnt; size_t ks_part_cap; uint32_t ks_cnt_cap;     // K-sliced GEMM scratch (2 .. 64 tokens), see MatJob
    uint32_t *cursors, *tokens, *headers, *argmax, *counter;
};

struct wrk_v7_model {
    wrk_ctx* ctx = nullptr;
    wrk_v7_model_desc d{};
    std::vector<wrk_v7_layer_desc> layers;
    const wrk_buf *ln0_w = nullptr, *ln0_b = nullptr, *ln_out_w = nullptr, *ln_out_b = nullptr, *emb = nullptr;
    const wrk_matrix* head = nullptr;

    void* scratch = nullptr;
    uint32_t scratch_tokens = 0, scratch_headers = 0;
    V7Scratch s{};
    uint32_t* history = nullptr;    // generated tokens [steps][B] (device)
    size_t history_cap = 0;

    // b: tokens (generate_greedy: sequences); mode: 0/1 for generate_greedy, or 16 + flag bits for wrk_v7_infer jobs
    // (the analogue of the reference's cached RnnJob per RnnInfo, runtime/mod.rs:110-209); nh: header rows
    struct GraphKey {
        const void* state; uint32_t b, mode, nh = 0;
        bool operator<(const GraphKey& o) const { return std::tie(state, b, mode, nh) < std::tie(o.state, o.b, o.mode, o.nh); }
    };
    std::map<GraphKey, wrk_program*> graphs;

    // fused decode path (wrk_v7_fused.hip): arg-max partials of the head matvec [num_wg][num_header]
    float* amax_val = nullptr;
    uint32_t* amax_idx = nullptr;
    size_t amax_cap = 0;

    // teacher-forced single-layer runs (wrk_v7_infer_layer): the layer range the op list covers and whether the embedding
    // stage (LN(ln0) + blit) is part of it; the defaults are the whole model
    uint32_t layer_begin = 0, layer_end = 0xffffffffu;
    bool skip_embed = false;
    uint32_t wkv_nseq = 0;          // sequences of the job being enqueued (0: unknown): picks the WKV chunk kernel, wrk::time_mix_v7
    // activation dtype of the frame: WRK_F16 = Bundle::<f16> (the reference's default), WRK_F32 = Bundle::<f32> (v7.rs:281-320
    // is generic over F).  F32 frames always take the op-by-op path with the f32-input matvec.
    uint32_t act_dtype = WRK_F16;

    // Concurrent pipelines (generate_greedy with groups > 1): lane g is a clone of this model -- same weight handles, its own frame,
    // history and cached programs -- whose decode graphs are replayed on a stream of its own, so that several latency-bound
    // pipelines overlap on the GPU (independent sequences: separate state slices, no synchronisation between lanes)
    std::vector<wrk_v7_model*> lanes;
    std::vector<hipStream_t> lane_streams;
    std::vector<hipEvent_t> lane_events;

    // persistent batch-1 decode engine (wrk_v7_engine.hip): built on first use, nullptr when the model / device does not fit it
    struct wrk_v7_engine* engine = nullptr;
    bool engine_tried = false;
    bool engine_skip_once = false;      // wrk_v7_infer_layer: the frame buffers must be materialised -> launches (unless WRK_ENGINE_INSPECT=1)
    bool engine_blocked = false;        // set while several pipelines share the GPU (generate_greedy with groups > 1)
    std::string engine_why;             // why the engine is not available (diagnostics)
    int32_t ensure_engine();            // outside captures; WRK_OK also when the engine is unavailable
    bool engine_on() const;             // WRK_ENGINE != 0 and the engine exists
    int32_t ensure_scratch(uint32_t T, uint32_t NH);
    int32_t ensure_history(size_t n);
    void drop_graphs();
    int32_t enqueue_ops(wrk_v7_state* st, uint32_t T, uint32_t NH, bool identity_headers, bool merged = false);
    // from_tokens: gather embedding rows of s.tokens on the device; want_argmax: greedy token per header row into
    // s.argmax; advance: also feed it back as the next token (device-resident generation loop)
    int32_t enqueue_fused_decode(wrk_v7_state* st, uint32_t B, uint32_t NH, bool identity_headers, bool from_tokens,
                                 bool want_argmax, bool advance, uint32_t cursor0_batch, bool contiguous);   // batch id of token 0; contiguous: token t is batch cursor0_batch + t
    void free_fused();
};

int32_t wrk_buf_write_raw(wrk_ctx* ctx, void* dst, const void* src, size_t bytes);
bool split_head_env_on();
bool engine_env_on_public();  // WRK_ENGINE != 0, read per call     // WRK_SPLIT_HEAD != 0, read per call (part of the graph keys)

namespace wrk {
// tokens <- argmax; history[counter][b] = argmax[b]; counter += 1   (one tiny kernel)
void advance_tokens(hipStream_t s, const uint32_t* argmax, uint32_t* tokens, uint32_t* history, uint32_t* counter, uint32_t b);
void argmax_finish(hipStream_t s, const float* pv, const uint32_t* pi, uint32_t nwg, uint32_t ntok, uint32_t* argmax, uint32_t* tokens,
                   uint32_t* history, uint32_t* counter);
}

// ------------------------------------------------------------------ shared by the fused V7 / V6 decode paths (wrk_v7_fused.hip)
namespace wrk {
// K0 / K4: layer norm + token shifts of stacked tokens, one workgroup per token
struct LnMixParams {
    const f16* src;             // [T][D] rows, or the embedding table when `ids` is set
    const uint32_t* ids;        // optional row index per token (embedding gather / header rows)
    const f16 *ln_w, *ln_b;
    float eps;
    uint32_t d, nmix;
    const f16* mix[6];          // token-shift factors
    f16* out[6];                // shifted outputs [T][D]
    f16* ln_out;                // optional: LN output [T][D]
    float* state_row;           // optional: shift state row, element (batch, c) at state_row[batch * state_stride + c]
    size_t state_stride;
    const uint32_t* cursors;    // batch id per token
    uint32_t batch1;            // host-known batches: token t is batch batch1 - 1 + t (0: read the cursor)
    uint32_t no_carry;          // 1: leave the shift state alone (a later kernel of the layer still reads it: RWKV-6)
};
int ln_mix(hipStream_t s, const LnMixParams& P, uint32_t T);      // -1: unsupported shape (D % 8, D > 8192, nmix not in {0, 1, 2, 6})
void argmax_finish(hipStream_t s, const float* pv, const uint32_t* pi, uint32_t nwg, uint32_t ntok, uint32_t* argmax, uint32_t* tokens,
                   uint32_t* history, uint32_t* counter);
}  // namespace wrk
