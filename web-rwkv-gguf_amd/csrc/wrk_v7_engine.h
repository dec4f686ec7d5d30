// Persistent batch-1 decode engine for RWKV-7 (wrk_v7_engine.hip): shared host / device definitions.  DESIGN.md section 4.8.
#pragma once
#include <stdint.h>

#include "wrk_internal.h"

namespace wrk {

constexpr int ENG_K1_JOBS = 7;                  // r, k, v, w1, a1, g1, v1
constexpr uint32_t ENG_NO_HEAD = 0xffffffffu;

// one matrix of stage K1; the same for every layer (pointers live in EngLayer)
struct EngJob {
    uint32_t rows, row_bytes, f16;      // f16: 1 = F16 rows (LoRA down-projection), 0 = the model's quantised kind
    uint32_t act, mix;                  // activation; index of the token-shift factor (0 r, 1 w, 2 k, 3 v, 4 a, 5 g)
    uint32_t wg0, nwg, rows_per_wg;     // workgroups [wg0, wg0 + nwg) take rows_per_wg rows each
    uint32_t gbase;                     // LoRA: first output granule; r / k / v: slot (0, 1, 2) inside a head's block
    uint32_t headed;                    // 1: r / k / v (outputs laid out per head)
};

struct EngShape {
    uint32_t D, F, H, nwg;
    uint32_t rw, ra, rg, rv;                        // LoRA ranks
    uint32_t rb_d, rb_f;                            // device row bytes of the D-wide / F-wide quantised matrices
    uint32_t rb_w2, rb_a2, rb_g2, rb_v2;            // row bytes of the F16 up-projections
    EngJob k1[ENG_K1_JOBS];
    uint32_t k3_rpw, k5_rpw, k6_rpw;                // rows per workgroup of W_o, ffn key, ffn value
    // LDS layout, byte offsets
    uint32_t lds_slot1, lds_slot5, lds_slot6, lds_slot3, lds_xraw0, lds_xraw1, lds_xs, lds_ln, lds_misc, lds_total;
    // granule buffer, offsets in 8-byte granules
    uint32_t g_x, g_k1, g_o, g_x1, g_k, g_total, g_aux;   // g_aux: granules of the LoRA intermediates in front of the per-head r / k / v blocks
    float ln_eps, gn_eps, l2_eps;
    uint32_t layer_begin, layer_end, rescale;       // layers [begin, end) run in this launch
    uint32_t batch, state_rows;                     // sequence; S + 2
    uint32_t x_from_granules;                       // unused (layer_begin's input is always the plain x vector)
};

// matrix pointers of a layer (read by the loader wave a layer ahead, and by the head workgroups once per layer)
struct EngLayer {
    const uint8_t *w_r, *w_k, *w_v, *w1, *a1, *g1, *v1;      // K1 (v1 unused on layer 0)
    const uint8_t *w2, *a2, *g2, *v2;                        // K2, F16 rows
    const uint8_t *w_o, *ffn_k, *ffn_v;
};
// The f16 vectors of every layer are PACKED into one buffer [L][ENG_NV][D] when the engine is built, so that a stage computes their
// addresses instead of loading them: a pointer fetched through a table is a dependent scalar round trip at the head of a stage, and the
// compiler must repeat it after every barrier (the barrier's asm clobbers memory) -- the first build spent ~1 us per stage there.
enum { ENG_V_LN1W, ENG_V_LN1B, ENG_V_LN2W, ENG_V_LN2B, ENG_V_MIX0, ENG_V_W0 = ENG_V_MIX0 + 6, ENG_V_A0, ENG_V_V0, ENG_V_RK, ENG_V_KK, ENG_V_KA,
       ENG_V_GNW, ENG_V_GNB, ENG_V_FFNXK, ENG_NV };
// wrk_matrix::out_scale of a layer's matrices: [L][ENG_NS] floats: K1 jobs 0..6, then W_o, ffn key, ffn value
enum { ENG_S_O = ENG_K1_JOBS, ENG_S_FK, ENG_S_FV, ENG_NS = 16 };

struct EngArgs {
    EngShape S;
    const EngLayer* layers;         // [num_layer]
    const void* vecs;               // f16 [num_layer][ENG_NV][D]
    const float* scal;              // [num_layer][ENG_NS]
    unsigned long long* gran;       // granule buffer, zeroed before every launch
    const void* x_in;               // f16 [D]: input of layer_begin (embedding LN output); with x_row: base of a table of such rows
    const uint32_t* x_row;          // optional: row of x_in to take (the token id: the table is LN(ln0) of every embedding row, made at build time)
    void* x_out;                    // f16 [D]: output of layer_end - 1
    void* v_first;                  // f16 [D]: att_v0 (written by layer 0, read when the launch starts above it)
    float* state;                   // state base; layer l of sequence b at state + (l * num_batch + b) * state_rows * D
    uint32_t num_batch;
    uint32_t* fail;                 // set when a bounded spin gave up
    unsigned long long* stamps;     // optional timeline [nwg][ENG_STAMPS] of layer `stamp_layer`
    uint32_t stamp_layer;
};
constexpr int ENG_STAMPS = 32;

}  // namespace wrk

// host side (wrk_v7_engine.hip)
struct wrk_v7_model;
struct wrk_v7_state;
struct wrk_v7_engine;
int32_t wrk_v7_engine_create(wrk_v7_model* m, wrk_v7_engine** out);    // WRK_E_UNSUPPORTED: the model / device does not fit the engine
void wrk_v7_engine_destroy(wrk_v7_engine* e);
// layers [l0, l1) of one decode token of sequence `batch`: x_in -> x_out (both f16 [D], may alias)
// token: device pointer to the token id, or nullptr.  With a token (and layer 0 first) the engine reads its input from its table of
// normalised embedding rows -- the embedding launch of the step is not needed -- and x_in is ignored.
int32_t wrk_v7_engine_enqueue(wrk_v7_engine* e, hipStream_t q, wrk_v7_state* st, uint32_t batch, uint32_t l0, uint32_t l1, const void* x_in,
                              void* x_out, void* v_first, const uint32_t* token = nullptr);
bool wrk_v7_engine_has_table(const wrk_v7_engine* e);
int32_t wrk_v7_engine_check(wrk_v7_engine* e);       // after a synchronisation: WRK_E_HIP if a launch gave up
void wrk_v7_engine_report(wrk_v7_engine* e);         // WRK_TIMING=1: print the in-kernel timeline of the stamped layer
