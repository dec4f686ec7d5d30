// Native host layer above the backend boundary (include/wrk_runtime.h): GGUF reader, loader,
// V7 model builder, chunk scheduler and the infer loop -- C++ restatement of the reference's Rust
// host code for this path, with the reference's names.  File:line citations are to /root/reference.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "wrk_runtime.h"

namespace {

thread_local std::string g_err;
int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// ------------------------------------------------------------------ f16 helpers (half::f16 semantics)
inline float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
inline uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t b; memcpy(&b, &h, 2); return b; }   // RNE
inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
inline float ld_f16(const uint8_t* p) { uint16_t b; memcpy(&b, p, 2); return h2f(b); }

// ------------------------------------------------------------------ GGML types (gguf.rs:888-1075)
enum : uint32_t { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q8_0 = 8, T_Q2K = 10, T_Q3K = 11, T_Q4K = 12, T_Q5K = 13, T_Q6K = 14, T_BF16 = 30 };
size_t type_size(uint32_t t) {
    switch (t) {
        case T_F32: return 4; case T_F16: return 2; case T_BF16: return 2;
        case T_Q4_0: return 18; case 3: return 20; case 6: return 22; case 7: return 24;
        case T_Q8_0: return 34; case 9: return 36;
        case 10: return 84; case 11: return 110; case T_Q4K: return 144; case T_Q5K: return 176; case T_Q6K: return 210; case 15: return 292;
        case 24: return 1; case 25: return 2; case 26: return 4; case 27: return 8; case 28: return 8;
        default: return 0;
    }
}
size_t block_size(uint32_t t) {
    switch (t) {
        case T_Q4_0: case 3: case 6: case 7: case T_Q8_0: case 9: return 32;
        case 10: case 11: case T_Q4K: case T_Q5K: case T_Q6K: case 15: return 256;
        default: return 1;
    }
}
bool is_quantized(uint32_t t) { return block_size(t) > 1; }

// ------------------------------------------------------------------ CPU dequantisers -> f16 (gguf.rs:11-274)
// Load-time only: embedding table, WRK_WEIGHTS_REFERENCE mode, and Reader::tensor().
void get_scale_min_k4(int j, const uint8_t* s, uint8_t& sc, uint8_t& m) {      // gguf.rs:81-89
    if (j < 4) { sc = s[j] & 63; m = s[j + 4] & 63; }
    else { sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4); m = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
}

void dequant_q8_0(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:11-37
    const long long nb = (long long)(n / 32);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 34;
        const float scale = ld_f16(blk);
        for (int i = 0; i < 32; ++i) out[b * 32 + i] = f2h((float)(int8_t)blk[2 + i] * scale);
    }
}

void dequant_q4_0(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:42-75 (interleaved order)
    const long long nb = (long long)(n / 32);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 18;
        const float scale = ld_f16(blk);
        for (int i = 0; i < 16; ++i) {
            const int lo = (int)(blk[2 + i] & 0x0F) - 8, hi = (int)(blk[2 + i] >> 4) - 8;
            out[b * 32 + 2 * i] = f2h((float)lo * scale);
            out[b * 32 + 2 * i + 1] = f2h((float)hi * scale);
        }
    }
}

// repack_q8_0_to_int8 (gguf.rs:429-520): four Q8_0 blocks -> one 128-element Int8 block.  Output in
// wrk_matrix_create's WRK_MAT_INT8 layout: codes [n] ++ (min, max) f16 per block.  n % 128 == 0 here.
void repack_q8_0_to_int8(const uint8_t* d, size_t n, std::vector<uint8_t>& blob) {
    const long long nblk = (long long)(n / 128);
    blob.assign(n + (size_t)nblk * 4, 0);
    uint8_t* codes = blob.data();
    uint16_t* minmax = (uint16_t*)(blob.data() + n);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nblk; ++b) {
        float val[128];
        float mn = FLT_MAX, mx = -FLT_MAX;
        for (int i = 0; i < 4; ++i) {
            const uint8_t* blk = d + (b * 4 + i) * 34;
            const float scale = ld_f16(blk);
            for (int j = 0; j < 32; ++j) {
                const float v = (float)(int8_t)blk[2 + j] * scale;
                val[i * 32 + j] = v;
                mn = std::min(mn, v);
                mx = std::max(mx, v);
            }
        }
        minmax[2 * b] = f2h(mn);
        minmax[2 * b + 1] = f2h(mx);
        const float range = mx - mn, inv = range > 0.0f ? 255.0f / range : 0.0f;
        for (int e = 0; e < 128; ++e) {
            const float t = roundf((val[e] - mn) * inv);                 // f32::round: half away from zero
            codes[b * 128 + e] = (uint8_t)std::min(std::max(t, 0.0f), 255.0f);   // `as u8` saturates
        }
    }
}

// repack_q4_0_to_nf4 (gguf.rs:528-627): two Q4_0 blocks -> one 64-element NF4 block; nibbles [n/2] ++ absmax f16
void repack_q4_0_to_nf4(const uint8_t* d, size_t n, std::vector<uint8_t>& blob) {
    static const float LEVELS[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                     -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                     0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                     0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
    const long long nblk = (long long)(n / 64);
    blob.assign(n / 2 + (size_t)nblk * 2, 0);
    uint8_t* packed = blob.data();
    uint16_t* absmax = (uint16_t*)(blob.data() + n / 2);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nblk; ++b) {
        float val[64];
        float amax = 0.0f;
        for (int i = 0; i < 2; ++i) {
            const uint8_t* blk = d + (b * 2 + i) * 18;
            const float scale = ld_f16(blk);
            for (int j = 0; j < 16; ++j) {
                const float lo = (float)((int)(blk[2 + j] & 0x0F) - 8) * scale, hi = (float)((int)(blk[2 + j] >> 4) - 8) * scale;
                val[i * 32 + 2 * j] = lo;
                val[i * 32 + 2 * j + 1] = hi;
                amax = std::max(std::max(amax, std::fabs(lo)), std::fabs(hi));
            }
        }
        absmax[b] = f2h(amax);
        const float inv = amax > 0.0f ? 1.0f / amax : 0.0f;
        auto nearest = [&](float x) {       // Iterator::min_by keeps the FIRST minimum
            int best = 0;
            float err = std::fabs(LEVELS[0] - x);
            for (int q = 1; q < 16; ++q) {
                const float e = std::fabs(LEVELS[q] - x);
                if (e < err) { err = e; best = q; }
            }
            return (uint8_t)best;
        };
        for (int e = 0; e < 32; ++e) packed[b * 32 + e] = nearest(val[2 * e] * inv) | (nearest(val[2 * e + 1] * inv) << 4);
    }
}

void dequant_q2_k(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:372-423
    const long long nb = (long long)(n / 256);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 84;
        const uint8_t* scales = blk;
        const uint8_t* qs = blk + 16;
        const float dd = ld_f16(blk + 80), dmin = ld_f16(blk + 82);
        uint16_t* o = out + b * 256;
        int is = 0;
        for (int n128 = 0; n128 < 2; ++n128) {
            for (int j = 0; j < 4; ++j) {
                for (int h = 0; h < 2; ++h) {
                    const uint8_t sc = scales[is++];
                    const float dl = dd * (float)(sc & 0xF), ml = dmin * (float)(sc >> 4);
                    for (int l = 0; l < 16; ++l) {
                        const int q = (qs[n128 * 32 + h * 16 + l] >> (2 * j)) & 3;
                        *o++ = f2h(dl * (float)q - ml);
                    }
                }
            }
        }
    }
}

void dequant_q3_k(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:280-366
    const long long nb = (long long)(n / 256);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 110;
        const uint8_t* hmask = blk;
        const uint8_t* qs = blk + 32;
        const float d_all = ld_f16(blk + 108);
        uint32_t aux[4], tmp;
        memcpy(&aux[0], blk + 96, 4); memcpy(&aux[1], blk + 100, 4); memcpy(&tmp, blk + 104, 4);
        const uint32_t K1 = 0x03030303u, K2 = 0x0f0f0f0fu;
        aux[2] = ((aux[0] >> 4) & K2) | (((tmp >> 4) & K1) << 4);
        aux[3] = ((aux[1] >> 4) & K2) | (((tmp >> 6) & K1) << 4);
        aux[0] = (aux[0] & K2) | (((tmp >> 0) & K1) << 4);
        aux[1] = (aux[1] & K2) | (((tmp >> 2) & K1) << 4);
        int8_t scales[16];
        memcpy(scales, aux, 16);
        uint16_t* o = out + b * 256;
        int is = 0;
        uint8_t m = 1;
        for (int n128 = 0; n128 < 2; ++n128) {
            for (int j = 0; j < 4; ++j) {
                for (int h = 0; h < 2; ++h) {
                    const float dl = d_all * (float)((int)scales[is++] - 32);
                    for (int l = 0; l < 16; ++l) {
                        const int q = (qs[n128 * 32 + h * 16 + l] >> (2 * j)) & 3;
                        const int hv = (hmask[h * 16 + l] & m) ? 0 : -4;
                        *o++ = f2h(dl * (float)(q + hv));
                    }
                }
                m = (uint8_t)(m << 1);
            }
        }
    }
}

void dequant_q4_k(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:95-143
    const long long nb = (long long)(n / 256);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 144;
        const float dd = ld_f16(blk), dmin = ld_f16(blk + 2);
        const uint8_t* scales = blk + 4;
        const uint8_t* qs = blk + 16;
        uint16_t* o = out + b * 256;
        for (int g = 0; g < 4; ++g) {
            uint8_t sc0, m0, sc1, m1;
            get_scale_min_k4(2 * g, scales, sc0, m0);
            get_scale_min_k4(2 * g + 1, scales, sc1, m1);
            const volatile float d1 = dd * (float)sc0, mv1 = dmin * (float)m0, d2 = dd * (float)sc1, mv2 = dmin * (float)m1;
            for (int l = 0; l < 32; ++l) { volatile float p = d1 * (float)(qs[32 * g + l] & 0xF); *o++ = f2h(p - mv1); }
            for (int l = 0; l < 32; ++l) { volatile float p = d2 * (float)(qs[32 * g + l] >> 4); *o++ = f2h(p - mv2); }
        }
    }
}

void dequant_q5_k(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:149-204
    const long long nb = (long long)(n / 256);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 176;
        const float dd = ld_f16(blk), dmin = ld_f16(blk + 2);
        const uint8_t *scales = blk + 4, *qh = blk + 16, *ql = blk + 48;
        uint16_t* o = out + b * 256;
        uint8_t u1 = 1, u2 = 2;
        for (int g = 0; g < 4; ++g) {
            uint8_t sc0, m0, sc1, m1;
            get_scale_min_k4(2 * g, scales, sc0, m0);
            get_scale_min_k4(2 * g + 1, scales, sc1, m1);
            const volatile float d1 = dd * (float)sc0, mv1 = dmin * (float)m0, d2 = dd * (float)sc1, mv2 = dmin * (float)m1;
            for (int l = 0; l < 32; ++l) {
                const int q = (ql[32 * g + l] & 0xF) + ((qh[l] & u1) ? 16 : 0);
                volatile float p = d1 * (float)q; *o++ = f2h(p - mv1);
            }
            for (int l = 0; l < 32; ++l) {
                const int q = (ql[32 * g + l] >> 4) + ((qh[l] & u2) ? 16 : 0);
                volatile float p = d2 * (float)q; *o++ = f2h(p - mv2);
            }
            u1 <<= 2; u2 <<= 2;
        }
    }
}

void dequant_q6_k(const uint8_t* d, size_t n, uint16_t* out) {                  // gguf.rs:210-274
    const long long nb = (long long)(n / 256);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nb; ++b) {
        const uint8_t* blk = d + b * 210;
        const uint8_t *ql = blk, *qh = blk + 128;
        const int8_t* sc = (const int8_t*)(blk + 192);
        const float dd = ld_f16(blk + 208);
        uint16_t* o = out + b * 256;
        for (int nn = 0; nn < 2; ++nn) {
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                const int a = ql[64 * nn + l], c = ql[64 * nn + l + 32], h = qh[32 * nn + l];
                const int q1 = (int)(int8_t)((a & 0xF) | ((h & 3) << 4)) - 32;
                const int q2 = (int)(int8_t)((c & 0xF) | (((h >> 2) & 3) << 4)) - 32;
                const int q3 = (int)(int8_t)((a >> 4) | (((h >> 4) & 3) << 4)) - 32;
                const int q4 = (int)(int8_t)((c >> 4) | (((h >> 6) & 3) << 4)) - 32;
                volatile float s0 = dd * (float)sc[8 * nn + is], s2 = dd * (float)sc[8 * nn + is + 2];
                volatile float s4 = dd * (float)sc[8 * nn + is + 4], s6 = dd * (float)sc[8 * nn + is + 6];
                o[128 * nn + l] = f2h(s0 * (float)q1);
                o[128 * nn + 32 + l] = f2h(s2 * (float)q2);
                o[128 * nn + 64 + l] = f2h(s4 * (float)q3);
                o[128 * nn + 96 + l] = f2h(s6 * (float)q4);
            }
        }
    }
}

// ------------------------------------------------------------------ name map (gguf.rs:1173-1329)
struct Pair { const char* gguf; const char* st; };
const Pair TOP[] = {
    {"token_embd.weight", "emb.weight"}, {"output_norm.weight", "ln_out.weight"}, {"output_norm.bias", "ln_out.bias"},
    {"output.weight", "head.weight"}, {"token_embd_norm.weight", "blocks.0.ln0.weight"}, {"token_embd_norm.bias", "blocks.0.ln0.bias"}};
const Pair BLK[] = {
    {"attn_norm.weight", "ln1.weight"}, {"attn_norm.bias", "ln1.bias"}, {"attn_norm_2.weight", "ln2.weight"}, {"attn_norm_2.bias", "ln2.bias"},
    {"ffn_norm.weight", "ln2.weight"}, {"ffn_norm.bias", "ln2.bias"},
    {"attn_k.weight", "att.key.weight"}, {"attn_v.weight", "att.value.weight"}, {"attn_r.weight", "att.receptance.weight"},
    {"attn_g.weight", "att.gate.weight"}, {"attn_output.weight", "att.output.weight"},
    {"attn_time_decay", "att.time_decay"}, {"attn_time_first", "att.time_first"},
    {"attn_time_mix_k", "att.time_mix_k"}, {"attn_time_mix_v", "att.time_mix_v"}, {"attn_time_mix_r", "att.time_mix_r"},
    {"attn_time_mix_g", "att.time_mix_g"}, {"attn_time_mix_x", "att.time_mix_x"}, {"attn_time_mix_w", "att.time_mix_w"},
    {"attn_time_mix_w1", "att.time_mix_w1"}, {"attn_time_mix_w2", "att.time_mix_w2"},
    {"attn_time_decay_w1", "att.time_decay_w1"}, {"attn_time_decay_w2", "att.time_decay_w2"},
    {"time_maa_w1", "att.time_mix_w1"}, {"time_maa_w2", "att.time_mix_w2"}, {"time_decay_w1", "att.time_decay_w1"}, {"time_decay_w2", "att.time_decay_w2"},
    {"attn_ln_x.weight", "att.ln_x.weight"}, {"attn_ln_x.bias", "att.ln_x.bias"}, {"attn_time_state", "att.time_state"},
    {"ffn_k.weight", "ffn.key.weight"}, {"ffn_v.weight", "ffn.value.weight"}, {"ffn_r.weight", "ffn.receptance.weight"},
    {"ffn_time_mix_k", "ffn.time_mix_k"}, {"ffn_time_mix_r", "ffn.time_mix_r"},
    {"ffn.key.weight", "ffn.key.weight"}, {"ffn.value.weight", "ffn.value.weight"}, {"ffn.receptance.weight", "ffn.receptance.weight"},
    {"channel_mix_key.weight", "ffn.key.weight"}, {"channel_mix_value.weight", "ffn.value.weight"},
    {"channel_mix_receptance.weight", "ffn.receptance.weight"}, {"channel_mix_lerp_k.weight", "ffn.x_k"},
    {"time_mix_key.weight", "att.key.weight"}, {"time_mix_value.weight", "att.value.weight"},
    {"time_mix_receptance.weight", "att.receptance.weight"}, {"time_mix_gate.weight", "att.gate.weight"},
    {"time_mix_output.weight", "att.output.weight"}, {"time_mix_lerp_fused.weight", "att.time_maa"},
    {"time_mix_ln.weight", "att.ln_x.weight"}, {"time_mix_ln.bias", "att.ln_x.bias"}, {"ffn_x_k", "ffn.x_k"}};
// llama.cpp's RWKV-6 tensor names (the ones the reference's own assets/scripts/convert_hf_to_gguf.py:455-525 emits through
// gguf-py's MODEL_ARCH.RWKV6 table).  The reference's map (gguf.rs:1212-1229) only knows the attn_time_* / time_maa_* spellings and
// sends `time_mix_w1.weight` to the RWKV-7 tensor att.w1 (gguf.rs:1261) -- SURVEY H6 -- so a real llama.cpp RWKV-6 file does not
// load there.  This table applies when general.architecture == "rwkv6" and takes precedence over the V7 rules for the names both
// families use (time_mix_w1 / w2, channel_mix_lerp_k).  Layouts need no change: the converter's transposes give exactly the dims
// the attn_* spellings carry ([D, 5R], [R, D, 5], [D, W], [W, D], [S, H]).  PARITY UNPINNED: gguf-py is not in this image; the
// names are pinned only by that converter script.
const Pair BLK_V6[] = {
    {"time_mix_lerp_x.weight", "att.time_mix_x"}, {"time_mix_lerp_w.weight", "att.time_mix_w"}, {"time_mix_lerp_k.weight", "att.time_mix_k"},
    {"time_mix_lerp_v.weight", "att.time_mix_v"}, {"time_mix_lerp_r.weight", "att.time_mix_r"}, {"time_mix_lerp_g.weight", "att.time_mix_g"},
    {"time_mix_w1.weight", "att.time_mix_w1"}, {"time_mix_w2.weight", "att.time_mix_w2"},
    {"time_mix_decay.weight", "att.time_decay"}, {"time_mix_decay_w1.weight", "att.time_decay_w1"}, {"time_mix_decay_w2.weight", "att.time_decay_w2"},
    {"time_mix_first.weight", "att.time_first"},
    {"channel_mix_lerp_k.weight", "ffn.time_mix_k"}, {"channel_mix_lerp_r.weight", "ffn.time_mix_r"}};
const char* V7_SHORT[] = {"w0", "w1", "w2", "a0", "a1", "a2", "g1", "g2", "v0", "v1", "v2", "r_k", "k_k", "k_a"};
const char* V7_LERP[] = {"x_r", "x_w", "x_k", "x_v", "x_a", "x_g"};

bool gguf_to_safetensors_name(const std::string& g, std::string& out, bool arch_v6 = false) {
    for (const Pair& p : TOP) if (g == p.gguf) { out = p.st; return true; }
    if (g.rfind("blk.", 0) != 0) return false;
    const std::string rest = g.substr(4);
    const size_t dot = rest.find('.');
    if (dot == std::string::npos) return false;
    const std::string blk = rest.substr(0, dot), rem = rest.substr(dot + 1);
    if (arch_v6)
        for (const Pair& p : BLK_V6) if (rem == p.gguf) { out = "blocks." + blk + "." + p.st; return true; }
    for (const Pair& p : BLK) if (rem == p.gguf) { out = "blocks." + blk + "." + p.st; return true; }
    for (const char* s : V7_SHORT) {
        const std::string v(s);
        if (rem == "time_mix_" + v + ".weight" || rem == "attn_" + v || rem == "att_" + v) { out = "blocks." + blk + ".att." + v; return true; }
    }
    for (const char* s : V7_LERP) {
        const std::string v(s);
        if (rem == "attn_" + v || rem == "att_" + v) { out = "blocks." + blk + ".att." + v; return true; }
    }
    return false;
}

struct TensorInfo {
    std::string name;
    std::vector<uint64_t> dims;
    uint32_t type = 0;
    uint64_t offset = 0;
    size_t num_elements() const { size_t n = 1; for (auto d : dims) n *= d; return n; }
    size_t data_size() const {                                                   // gguf.rs:1137-1147
        const size_t bs = block_size(type), ts = type_size(type);
        return bs == 1 ? num_elements() * ts : (num_elements() / bs) * ts;
    }
    // dims and offsets come from the file: every product / sum is overflow-checked before a pointer is formed from it
    // (a crafted offset near 2^64 would otherwise wrap past the range check and yield a pointer before the mapping)
    bool checked_sizes(uint64_t& elems, uint64_t& bytes) const {
        elems = 1;
        for (uint64_t d : dims)
            if (__builtin_mul_overflow(elems, d, &elems)) return false;
        const uint64_t bs = block_size(type), ts = type_size(type);
        if (bs == 0 || ts == 0) { bytes = 0; return true; }      // unknown type: rejected when the tensor is read
        return !__builtin_mul_overflow(elems / bs, ts, &bytes);
    }
};

struct MetaValue { int kind = 0; uint64_t u = 0; double f = 0; std::string s; };    // kind: 1 uint, 2 int, 3 float, 4 bool, 5 string, 6 array

}  // namespace

struct wrk_gguf {
    const uint8_t* data = nullptr;
    size_t size = 0;
    bool mapped = false;
    uint32_t version = 0;
    uint64_t tensor_count = 0, tensor_data_offset = 0;
    std::unordered_map<std::string, MetaValue> metadata;
    std::unordered_map<std::string, TensorInfo> tensors;
    std::unordered_map<std::string, std::string> name_map;
    std::vector<std::string> order;     // tensor names in file order

    // ---- cursor (gguf.rs:1419-1538)
    size_t pos = 0;
    bool eof = false;
    const uint8_t* take(size_t n) {
        if (size - pos < n) { eof = true; return nullptr; }
        const uint8_t* p = data + pos; pos += n; return p;
    }
    template <class T> T rd() { const uint8_t* p = take(sizeof(T)); T v{}; if (p) memcpy(&v, p, sizeof(T)); return v; }
    bool rd_string(std::string& s) {
        const uint64_t n = rd<uint64_t>();
        if (eof || size - pos < n) { eof = true; return false; }
        s.assign((const char*)data + pos, n); pos += n;
        return true;
    }
    bool rd_value(uint32_t t, MetaValue& v, int depth = 0) {
        switch (t) {
            case 0: v.kind = 1; v.u = rd<uint8_t>(); break;
            case 1: v.kind = 2; v.u = (uint64_t)(int64_t)rd<int8_t>(); break;
            case 2: v.kind = 1; v.u = rd<uint16_t>(); break;
            case 3: v.kind = 2; v.u = (uint64_t)(int64_t)rd<int16_t>(); break;
            case 4: v.kind = 1; v.u = rd<uint32_t>(); break;
            case 5: v.kind = 2; v.u = (uint64_t)(int64_t)rd<int32_t>(); break;
            case 6: v.kind = 3; v.f = rd<float>(); break;
            case 7: v.kind = 4; v.u = rd<uint8_t>() != 0; break;
            case 8: v.kind = 5; if (!rd_string(v.s)) return false; break;
            case 9: {
                const uint32_t at = rd<uint32_t>();
                const uint64_t n = rd<uint64_t>();
                if (eof || depth > 4) return false;
                v.kind = 6; v.u = n;
                for (uint64_t i = 0; i < n; ++i) { MetaValue e; if (!rd_value(at, e, depth + 1) || eof) return false; }
                break;
            }
            case 10: v.kind = 1; v.u = rd<uint64_t>(); break;
            case 11: v.kind = 2; v.u = (uint64_t)rd<int64_t>(); break;
            case 12: v.kind = 3; v.f = rd<double>(); break;
            default: return false;
        }
        return !eof;
    }

    int32_t parse() {                                                            // gguf.rs:1331-1402
        const uint32_t magic = rd<uint32_t>();
        if (eof) return fail(WRK_E_ARG, "unexpected end of file");
        if (magic != 0x46554747u) return fail(WRK_E_ARG, "invalid magic number: expected 0x46554747, got 0x%08X", magic);
        version = rd<uint32_t>();
        if (eof) return fail(WRK_E_ARG, "unexpected end of file");
        if (version < 2 || version > 3) return fail(WRK_E_UNSUPPORTED, "unsupported version: %u (supported: 3)", version);
        tensor_count = rd<uint64_t>();
        const uint64_t nkv = rd<uint64_t>();
        if (eof) return fail(WRK_E_ARG, "unexpected end of file");
        for (uint64_t i = 0; i < nkv; ++i) {
            std::string key;
            if (!rd_string(key)) return fail(WRK_E_ARG, "unexpected end of file");
            const uint32_t t = rd<uint32_t>();
            MetaValue v;
            if (!rd_value(t, v)) return fail(WRK_E_ARG, eof ? "unexpected end of file" : "invalid metadata value type: %u", t);
            metadata[key] = v;
        }
        uint64_t alignment = 32;
        auto al = metadata.find("general.alignment");
        if (al != metadata.end() && (al->second.kind == 1 || al->second.kind == 2) && al->second.u > 0) alignment = al->second.u;
        for (uint64_t i = 0; i < tensor_count; ++i) {
            TensorInfo ti;
            if (!rd_string(ti.name)) return fail(WRK_E_ARG, "unexpected end of file");
            const uint32_t nd = rd<uint32_t>();
            if (eof || nd > 8) return fail(WRK_E_ARG, "unexpected end of file");
            for (uint32_t k = 0; k < nd; ++k) ti.dims.push_back(rd<uint64_t>());
            ti.type = rd<uint32_t>();
            ti.offset = rd<uint64_t>();
            if (eof) return fail(WRK_E_ARG, "unexpected end of file");
            order.push_back(ti.name);
            tensors[ti.name] = ti;
        }
        tensor_data_offset = pos + (alignment - (pos % alignment)) % alignment;      // align_offset, gguf.rs:1415-1417
        bool arch_v6 = false;
        {
            auto ar = metadata.find("general.architecture");
            arch_v6 = ar != metadata.end() && ar->second.kind == 5 && ar->second.s == "rwkv6";
        }
        for (const std::string& g : order) {                                         // build_rwkv_name_map
            std::string st;
            if (gguf_to_safetensors_name(g, st, arch_v6)) name_map[st] = g;
            name_map[g] = g;
        }
        // every tensor must lie inside the file: the loader hands these pointers to the device
        if (tensor_data_offset > size) return fail(WRK_E_ARG, "tensor data offset %llu exceeds the file", (unsigned long long)tensor_data_offset);
        const uint64_t room = size - tensor_data_offset;
        for (auto& kv : tensors) {
            const TensorInfo& ti = kv.second;
            uint64_t elems, bytes;
            if (!ti.checked_sizes(elems, bytes)) return fail(WRK_E_ARG, "tensor %s: dimensions overflow", ti.name.c_str());
            if (ti.offset > room || bytes > room - ti.offset) return fail(WRK_E_ARG, "tensor %s exceeds the file", ti.name.c_str());
        }
        // fused time_maa tensors are read as six [D] slices (try_get_fused_slice, gguf.rs:1545-1571): they must hold them
        for (auto& kv : name_map) {
            const std::string sfx = ".att.time_maa";
            if (kv.first.size() < sfx.size() || kv.first.compare(kv.first.size() - sfx.size(), sfx.size(), sfx) != 0) continue;
            const TensorInfo& ti = tensors[kv.second];
            uint64_t elems, bytes;
            ti.checked_sizes(elems, bytes);
            if (ti.dims.empty() || ti.dims[0] == 0 || elems / ti.dims[0] < 6)
                return fail(WRK_E_ARG, "tensor %s: a fused lerp tensor must hold 6 slices of dims[0] elements", ti.name.c_str());
        }
        return WRK_OK;
    }

    const TensorInfo* info(const std::string& name) const {
        auto it = name_map.find(name);
        if (it == name_map.end()) return nullptr;
        auto jt = tensors.find(it->second);
        return jt == tensors.end() ? nullptr : &jt->second;
    }
    const uint8_t* tensor_data(const TensorInfo& ti) const { return data + tensor_data_offset + ti.offset; }
    uint64_t head_size() const {
        for (const char* k : {"rwkv7.wkv.head_size", "rwkv6.wkv.head_size"}) {
            auto it = metadata.find(k);
            if (it != metadata.end() && it->second.kind == 1) return it->second.u;
        }
        return 0;
    }
    // try_get_fused_slice (gguf.rs:1545-1571)
    bool fused_slice(const std::string& name, std::string& fused, int& idx) const {
        if (name.rfind("blocks.", 0) != 0 || name.find(".att.x_") == std::string::npos) return false;
        for (int i = 0; i < 6; ++i) {
            const std::string sfx = std::string(".att.") + V7_LERP[i];
            if (name.size() > sfx.size() && name.compare(name.size() - sfx.size(), sfx.size(), sfx) == 0) {
                fused = name.substr(0, name.size() - sfx.size()) + ".att.time_maa";
                if (name_map.count(fused)) { idx = i; return true; }
            }
        }
        return false;
    }
    bool contains(const std::string& name) const {
        std::string f; int i;
        return name_map.count(name) || fused_slice(name, f, i);
    }
    // Reader::shape (gguf.rs:1600-1648)
    bool shape(const std::string& name, std::vector<size_t>& out) const {
        std::string f; int idx;
        if (fused_slice(name, f, idx)) {
            const TensorInfo* ti = info(f);
            if (!ti || ti->dims.empty()) return false;
            out = {(size_t)ti->dims[0]};
            return true;
        }
        const TensorInfo* ti = info(name);
        if (!ti) return false;
        out.assign(ti->dims.begin(), ti->dims.end());
        const std::string sfx = ".att.r_k";
        if (out.size() == 1 && name.size() >= sfx.size() && name.compare(name.size() - sfx.size(), sfx.size(), sfx) == 0) {
            const uint64_t hs = head_size();
            if (hs) { out = {out[0] / (size_t)hs, (size_t)hs}; return true; }
        }
        if (out.size() > 1) std::reverse(out.begin(), out.end());
        return true;
    }
    // Reader::tensor + tensor_f16_from_reader (gguf.rs:1650-1773, loader.rs:104-132) -> f16 bits
    int32_t tensor_f16(const std::string& name, std::vector<uint16_t>& out) const {
        std::string f; int idx;
        const TensorInfo* ti;
        size_t n, skip = 0;
        if (fused_slice(name, f, idx)) {
            ti = info(f);
            if (!ti) return fail(WRK_E_ARG, "tensor not found: %s", f.c_str());
            if (is_quantized(ti->type) || (ti->type != T_F32 && ti->type != T_F16 && ti->type != T_BF16))
                return fail(WRK_E_UNSUPPORTED, "unsupported tensor type: %u", ti->type);
            n = (size_t)ti->dims[0];
            skip = (size_t)idx * n;
        } else {
            ti = info(name);
            if (!ti) return fail(WRK_E_ARG, "tensor not found: %s", name.c_str());
            n = ti->num_elements();
        }
        const uint8_t* d = tensor_data(*ti);
        out.assign(n, 0);
        switch (ti->type) {
            case T_F16: memcpy(out.data(), d + skip * 2, n * 2); break;
            case T_F32: { const float* p = (const float*)d + skip; for (size_t i = 0; i < n; ++i) out[i] = f2h(p[i]); break; }
            case T_BF16: { const uint16_t* p = (const uint16_t*)d + skip; for (size_t i = 0; i < n; ++i) out[i] = f2h(bf2f(p[i])); break; }
            case T_Q8_0: dequant_q8_0(d, n, out.data()); break;
            case T_Q4_0: dequant_q4_0(d, n, out.data()); break;
            case T_Q2K: dequant_q2_k(d, n, out.data()); break;
            case T_Q3K: dequant_q3_k(d, n, out.data()); break;
            case T_Q4K: dequant_q4_k(d, n, out.data()); break;
            case T_Q5K: dequant_q5_k(d, n, out.data()); break;
            case T_Q6K: dequant_q6_k(d, n, out.data()); break;
            default: return fail(WRK_E_UNSUPPORTED, "unsupported tensor type: %u", ti->type);
        }
        return WRK_OK;
    }
};

// ------------------------------------------------------------------ RnnInput / RnnIter (rnn.rs)
static constexpr uint32_t MIN_TOKEN_CHUNK_SIZE = 32;

struct wrk_rnn_input {
    std::vector<std::vector<uint32_t>> tokens;
    std::vector<int32_t> option;
    uint32_t token_chunk_size = 128;
};

struct wrk_rnn_iter {
    // BatchState::Gen == -1, Read(n) == n  (rnn.rs:344-348)
    std::vector<int64_t> state;
    std::vector<int32_t> option;
    uint32_t token_chunk_size = 128;

    void next(uint32_t* lens, int32_t* options) {                                // rnn.rs:280-335
        const size_t nb = state.size();
        std::vector<uint64_t> remains(nb);
        uint64_t total = 0;
        for (size_t i = 0; i < nb; ++i) { remains[i] = state[i] < 0 ? 1 : (uint64_t)state[i]; total += remains[i]; }
        uint64_t num_token = std::min<uint64_t>(total, token_chunk_size);
        if (num_token > MIN_TOKEN_CHUNK_SIZE) num_token -= num_token % MIN_TOKEN_CHUNK_SIZE;
        for (size_t i = 0; i < nb; ++i) lens[i] = 0;
        while (num_token > 0) {
            uint64_t mid0 = 0;
            for (uint64_t r : remains) if (r > 0 && (mid0 == 0 || r < mid0)) mid0 = r;
            for (size_t i = 0; i < nb; ++i) {
                if (remains[i] == 0) continue;
                const uint64_t mid = std::min(mid0, num_token);
                num_token -= mid;
                lens[i] += (uint32_t)mid;
                remains[i] -= mid;
            }
        }
        for (size_t i = 0; i < nb; ++i) {
            if (lens[i] > 0) state[i] = remains[i] == 0 ? -1 : (int64_t)remains[i];
            if (option[i] == WRK_RNN_LAST) options[i] = remains[i] == 0 ? WRK_RNN_LAST : WRK_RNN_NONE;
            else options[i] = WRK_RNN_FULL;
        }
    }
};

static void make_iter(const wrk_rnn_input& in, wrk_rnn_iter& it) {
    it.state.clear();
    for (auto& t : in.tokens) it.state.push_back((int64_t)t.size());
    it.option = in.option;
    it.token_chunk_size = in.token_chunk_size;
}

// RnnInfo::redirect (rnn.rs:41-81)
static void redirect(const uint32_t* lens, const int32_t* options, uint32_t nb, std::vector<uint32_t>& headers,
                     std::vector<std::pair<uint32_t, uint32_t>>& inputs, std::vector<std::pair<uint32_t, uint32_t>>& outputs) {
    headers.clear();
    inputs.assign(nb, {0, 0});
    outputs.assign(nb, {0, 0});
    uint32_t p_in = 0, p_out = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t len = lens[b];
        inputs[b] = {p_in, p_in + len};
        if (options[b] == WRK_RNN_NONE) {
            outputs[b] = {p_out, p_out};
        } else if (options[b] == WRK_RNN_LAST) {
            if (len == 0) outputs[b] = {p_out, p_out};
            else { outputs[b] = {p_out, p_out + 1}; headers.push_back(p_in + len - 1); p_out += 1; }
        } else {
            outputs[b] = {p_out, p_out + len};
            for (uint32_t i = 0; i < len; ++i) headers.push_back(p_in + i);
            p_out += len;
        }
        p_in += len;
    }
}

// ------------------------------------------------------------------ runtime
struct wrk_runtime {
    wrk_ctx* ctx = nullptr;
    wrk_model_info info{};
    std::vector<wrk_buf*> bufs;
    std::vector<wrk_matrix*> mats;
    std::vector<wrk_v7_layer_desc> layers;
    std::vector<wrk_v6_layer_desc> layers6;
    wrk_v7_model* model = nullptr;
    wrk_v6_model* model6 = nullptr;
    wrk_v7_state* state = nullptr;
    uint32_t num_batch = 0;
    ~wrk_runtime() {
        if (state) wrk_v7_state_destroy(state);
        if (model) wrk_v7_model_destroy(model);
        if (model6) wrk_v6_model_destroy(model6);
        for (auto* m : mats) wrk_matrix_release(m);
        for (auto* b : bufs) wrk_buf_release(b);
    }
};

static int32_t loader_info(const wrk_gguf& g, wrk_model_info& out) {               // loader.rs:238-371
    uint32_t num_layer = 0;
    for (auto& kv : g.name_map) {
        const std::string& n = kv.first;
        if (n.rfind("blocks.", 0) == 0) {
            const std::string rest = n.substr(7);
            const size_t dot = rest.find('.');
            if (dot == std::string::npos || dot == 0) continue;
            char* end = nullptr;
            const unsigned long v = strtoul(rest.substr(0, dot).c_str(), &end, 10);
            if (end && *end == 0) num_layer = std::max<uint32_t>(num_layer, (uint32_t)v);
        }
    }
    num_layer += 1;
    std::vector<size_t> embed, ffn, rk;
    if (!g.shape("emb.weight", embed) || embed.size() != 2) return fail(WRK_E_ARG, "tensor not found: emb.weight");
    if (!g.shape("blocks.0.ffn.key.weight", ffn) || ffn.size() != 2) return fail(WRK_E_ARG, "tensor not found: blocks.0.ffn.key.weight");
    const char* sep[] = {"x_r", "x_w", "x_k", "x_v", "x_a", "x_g", "w0", "w1", "w2", "a0", "a1", "a2", "g1", "g2", "r_k", "k_k", "k_a"};
    const char* fused[] = {"time_maa", "w0", "w1", "w2", "a0", "a1", "a2", "g1", "g2", "r_k", "k_k", "k_a"};
    const char* v6n[] = {"time_mix_x", "time_mix_w", "time_mix_k", "time_mix_v", "time_mix_r", "time_mix_g", "time_mix_w1", "time_mix_w2",
                         "time_decay_w1", "time_decay_w2"};
    bool v7s = true, v7f = true, v6 = g.contains("blocks.0.ffn.time_mix_k") && g.contains("blocks.0.ffn.time_mix_r");
    for (const char* s : sep) v7s = v7s && g.contains(std::string("blocks.0.att.") + s);
    for (const char* s : fused) v7f = v7f && g.contains(std::string("blocks.0.att.") + s);
    for (const char* s : v6n) v6 = v6 && g.contains(std::string("blocks.0.att.") + s);
    auto rank = [&](const char* n, uint32_t& o) {
        std::vector<size_t> s;
        if (!g.shape(n, s) || s.empty()) return false;
        o = (uint32_t)s[0];
        return true;
    };
    out.num_layer = num_layer;
    out.num_emb = (uint32_t)embed[1];
    out.num_vocab = (uint32_t)embed[0];
    out.num_hidden = (uint32_t)ffn[0];
    if (v6 && !(v7s || v7f)) {                                                 // (_, _, true, false) => V6, loader.rs:325-331
        std::vector<size_t> tf;
        if (!g.shape("blocks.0.att.time_first", tf) || tf.empty()) return fail(WRK_E_ARG, "tensor not found: blocks.0.att.time_first");
        out.version = 6;
        out.num_head = (uint32_t)tf[0];
        uint32_t w1 = 0;
        if (!rank("blocks.0.att.time_mix_w1", w1) || !rank("blocks.0.att.time_decay_w1", out.lora_a)) return fail(WRK_E_ARG, "V6 LoRA tensors missing");
        out.lora_w = w1 / 5;                                                   // CustomInfo::time_mix (loader.rs:345)
        out.lora_g = out.lora_v = 0;
        return WRK_OK;
    }
    if (!(v7s || v7f)) return fail(WRK_E_UNSUPPORTED, "invalid model version (RWKV-6 and RWKV-7 are built)");
    if (!g.shape("blocks.0.att.r_k", rk) || rk.size() != 2) return fail(WRK_E_ARG, "blocks.0.att.r_k: cannot derive num_head (rwkv7.wkv.head_size missing?)");
    out.version = 7;
    out.num_head = (uint32_t)rk[0];
    if (!rank("blocks.0.att.w1", out.lora_w) || !rank("blocks.0.att.a1", out.lora_a) || !rank("blocks.0.att.g1", out.lora_g))
        return fail(WRK_E_ARG, "LoRA tensors missing");
    if (!rank("blocks.1.att.v1", out.lora_v)) return fail(WRK_E_ARG, "tensor not found: blocks.1.att.v1");
    return WRK_OK;
}

extern "C" {

const char* wrk_host_last_error(void) { return g_err.c_str(); }

int32_t wrk_gguf_from_memory(const void* data, size_t bytes, wrk_gguf** out) {
    if (!data || !out) return fail(WRK_E_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<wrk_gguf> g(new wrk_gguf());
    g->data = (const uint8_t*)data;
    g->size = bytes;
    const int32_t rc = g->parse();
    if (rc != WRK_OK) return rc;
    *out = g.release();
    return WRK_OK;
}

int32_t wrk_gguf_open(const char* path, wrk_gguf** out) {
    if (!path || !out) return fail(WRK_E_ARG, "null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(WRK_E_ARG, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { close(fd); return fail(WRK_E_ARG, "cannot stat %s", path); }
    void* p = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(WRK_E_OOM, "mmap failed for %s", path);
    std::unique_ptr<wrk_gguf> g(new wrk_gguf());
    g->data = (const uint8_t*)p;
    g->size = (size_t)sb.st_size;
    g->mapped = true;
    const int32_t rc = g->parse();
    if (rc != WRK_OK) { munmap(p, g->size); return rc; }
    *out = g.release();
    return WRK_OK;
}

int32_t wrk_gguf_close(wrk_gguf* g) {
    if (!g) return WRK_E_ARG;
    if (g->mapped) munmap((void*)g->data, g->size);
    delete g;
    return WRK_OK;
}

uint32_t wrk_gguf_version(const wrk_gguf* g) { return g ? g->version : 0; }
uint64_t wrk_gguf_tensor_data_offset(const wrk_gguf* g) { return g ? g->tensor_data_offset : 0; }
int32_t wrk_gguf_contains(const wrk_gguf* g, const char* name) { return g && name && g->contains(name) ? 1 : 0; }

int32_t wrk_gguf_shape(const wrk_gguf* g, const char* name, uint32_t dims[4], uint32_t* ndim) {
    if (!g || !name || !dims || !ndim) return fail(WRK_E_ARG, "null argument");
    std::vector<size_t> s;
    if (!g->shape(name, s)) return fail(WRK_E_ARG, "tensor not found: %s", name);
    if (s.size() > 4) return fail(WRK_E_UNSUPPORTED, "rank %zu > 4", s.size());
    *ndim = (uint32_t)s.size();
    for (size_t i = 0; i < 4; ++i) dims[i] = i < s.size() ? (uint32_t)s[i] : 1;
    return WRK_OK;
}

int32_t wrk_gguf_tensor_f16(const wrk_gguf* g, const char* name, uint16_t* out, size_t capacity, size_t* count) {
    if (!g || !name || !count) return fail(WRK_E_ARG, "null argument");
    std::vector<uint16_t> v;
    const int32_t rc = g->tensor_f16(name, v);
    if (rc != WRK_OK) return rc;
    *count = v.size();
    if (out) {
        if (capacity < v.size()) return fail(WRK_E_ARG, "buffer too small: %zu < %zu", capacity, v.size());
        memcpy(out, v.data(), v.size() * 2);
    }
    return WRK_OK;
}

int32_t wrk_gguf_raw(const wrk_gguf* g, const char* name, uint32_t* ggml_type, const void** data, size_t* bytes) {
    if (!g || !name || !ggml_type || !data || !bytes) return fail(WRK_E_ARG, "null argument");
    std::string f; int idx;
    if (g->fused_slice(name, f, idx)) return fail(WRK_E_UNSUPPORTED, "virtual slice %s has no raw form", name);
    const TensorInfo* ti = g->info(name);
    if (!ti) return fail(WRK_E_ARG, "tensor not found: %s", name);
    *ggml_type = ti->type;
    *data = g->tensor_data(*ti);
    *bytes = ti->data_size();
    return WRK_OK;
}

int32_t wrk_gguf_meta_u64(const wrk_gguf* g, const char* key, uint64_t* out) {
    if (!g || !key || !out) return fail(WRK_E_ARG, "null argument");
    auto it = g->metadata.find(key);
    if (it == g->metadata.end() || (it->second.kind != 1 && it->second.kind != 2)) return fail(WRK_E_ARG, "metadata key not found: %s", key);
    *out = it->second.u;
    return WRK_OK;
}

int32_t wrk_gguf_info(const wrk_gguf* g, wrk_model_info* out) {
    if (!g || !out) return fail(WRK_E_ARG, "null argument");
    return loader_info(*g, *out);
}

// ---------------------------------------------------------------- RnnInput
int32_t wrk_rnn_input_create(uint32_t num_batch, uint32_t token_chunk_size, wrk_rnn_input** out) {
    if (!out || num_batch == 0) return fail(WRK_E_ARG, "bad argument");
    wrk_rnn_input* in = new wrk_rnn_input();
    in->tokens.resize(num_batch);
    in->option.assign(num_batch, WRK_RNN_LAST);
    uint32_t t = std::max(token_chunk_size, MIN_TOKEN_CHUNK_SIZE);              // rnn.rs:204-212
    t = (t + MIN_TOKEN_CHUNK_SIZE - 1) / MIN_TOKEN_CHUNK_SIZE * MIN_TOKEN_CHUNK_SIZE;
    in->token_chunk_size = t;
    *out = in;
    return WRK_OK;
}
int32_t wrk_rnn_input_destroy(wrk_rnn_input* in) { delete in; return WRK_OK; }
uint32_t wrk_rnn_input_token_chunk_size(const wrk_rnn_input* in) { return in ? in->token_chunk_size : 0; }
int32_t wrk_rnn_input_append(wrk_rnn_input* in, uint32_t batch, const uint32_t* tokens, uint32_t n) {
    if (!in || batch >= in->tokens.size() || (!tokens && n)) return fail(WRK_E_ARG, "bad argument");
    in->tokens[batch].insert(in->tokens[batch].end(), tokens, tokens + n);
    return WRK_OK;
}
int32_t wrk_rnn_input_set_option(wrk_rnn_input* in, uint32_t batch, int32_t option) {
    if (!in || batch >= in->tokens.size() || (option != WRK_RNN_LAST && option != WRK_RNN_FULL)) return fail(WRK_E_ARG, "bad argument");
    in->option[batch] = option;
    return WRK_OK;
}
uint32_t wrk_rnn_input_remaining(const wrk_rnn_input* in, uint32_t batch) {
    return in && batch < in->tokens.size() ? (uint32_t)in->tokens[batch].size() : 0;
}
int32_t wrk_rnn_input_step(wrk_rnn_input* in) {                                  // rnn.rs:233-240
    if (!in) return fail(WRK_E_ARG, "null argument");
    wrk_rnn_iter it;
    make_iter(*in, it);
    std::vector<uint32_t> lens(in->tokens.size());
    std::vector<int32_t> opts(in->tokens.size());
    it.next(lens.data(), opts.data());
    for (size_t b = 0; b < in->tokens.size(); ++b) in->tokens[b].erase(in->tokens[b].begin(), in->tokens[b].begin() + lens[b]);
    return WRK_OK;
}
int32_t wrk_rnn_iter_create(const wrk_rnn_input* in, wrk_rnn_iter** out) {
    if (!in || !out) return fail(WRK_E_ARG, "null argument");
    wrk_rnn_iter* it = new wrk_rnn_iter();
    make_iter(*in, *it);
    *out = it;
    return WRK_OK;
}
int32_t wrk_rnn_iter_destroy(wrk_rnn_iter* it) { delete it; return WRK_OK; }
int32_t wrk_rnn_iter_next(wrk_rnn_iter* it, uint32_t* lens, int32_t* options) {
    if (!it || !lens || !options) return fail(WRK_E_ARG, "null argument");
    it->next(lens, options);
    return WRK_OK;
}
int32_t wrk_rnn_redirect(const uint32_t* lens, const int32_t* options, uint32_t nb, uint32_t* headers, uint32_t* num_header,
                         uint32_t* inputs, uint32_t* outputs) {
    if (!lens || !options || !num_header) return fail(WRK_E_ARG, "null argument");
    std::vector<uint32_t> h;
    std::vector<std::pair<uint32_t, uint32_t>> i, o;
    redirect(lens, options, nb, h, i, o);
    *num_header = (uint32_t)h.size();
    if (headers) memcpy(headers, h.data(), h.size() * 4);
    for (uint32_t b = 0; b < nb; ++b) {
        if (inputs) { inputs[2 * b] = i[b].first; inputs[2 * b + 1] = i[b].second; }
        if (outputs) { outputs[2 * b] = o[b].first; outputs[2 * b + 1] = o[b].second; }
    }
    return WRK_OK;
}

// ---------------------------------------------------------------- quantile_student (matrix.rs:29-44)
namespace {
// regularised incomplete beta I_x(a, b) by Lentz's continued fraction (f64)
double betacf(double a, double b, double x) {
    const double tiny = 1e-300;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, d = 1.0 - qab * x / qap;
    if (std::fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}
double betai(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double bt = std::exp(std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log1p(-x));
    return x < (a + 1.0) / (a + b + 2.0) ? bt * betacf(a, b, x) / a : 1.0 - bt * betacf(b, a, 1.0 - x) / b;
}
double student_cdf(double t, double nu) {
    const double tail = 0.5 * betai(0.5 * nu, 0.5, nu / (nu + t * t));
    return t >= 0.0 ? 1.0 - tail : tail;
}
double student_pdf(double t, double nu) {
    return std::exp(std::lgamma(0.5 * (nu + 1.0)) - std::lgamma(0.5 * nu) - 0.5 * std::log(nu * M_PI) - 0.5 * (nu + 1.0) * std::log1p(t * t / nu));
}
double student_inv_cdf(double p, double nu) {
    if (p == 0.5) return 0.0;
    double lo = -1.0, hi = 1.0;
    while (student_cdf(lo, nu) > p) lo *= 2.0;
    while (student_cdf(hi, nu) < p) hi *= 2.0;
    for (int i = 0; i < 200 && hi - lo > 1e-15 * std::max(1.0, std::fabs(lo)); ++i) {      // bisection to the last bits ...
        const double mid = 0.5 * (lo + hi);
        if (student_cdf(mid, nu) < p) lo = mid; else hi = mid;
    }
    double t = 0.5 * (lo + hi);
    for (int i = 0; i < 3; ++i) {                                                           // ... polished by Newton steps
        const double f = student_cdf(t, nu) - p, g = student_pdf(t, nu);
        if (g > 0.0 && std::isfinite(f / g)) t -= f / g;
    }
    return t;
}
}  // namespace

int32_t wrk_quantile_student(double nu, float* out16) {
    if (!out16 || !(nu > 0.0) || !std::isfinite(nu)) return fail(WRK_E_ARG, "quantile_student: nu must be positive and finite");
    const double delta = (1.0 / 32.0 + 1.0 / 30.0) / 2.0;
    double p[16], q[16];
    double step = (0.5 - delta) / 7.0;
    for (int i = 0; i < 7; ++i) p[i] = delta + step * i;
    step = (1.0 - delta - 0.5) / 8.0;
    for (int i = 0; i < 9; ++i) p[7 + i] = 0.5 + step * i;
    double mx = -1e300;
    for (int i = 0; i < 16; ++i) { q[i] = student_inv_cdf(p[i], nu); mx = std::max(mx, q[i]); }
    for (int i = 0; i < 16; ++i) out16[i] = (float)(q[i] / mx);
    return WRK_OK;
}

// ---------------------------------------------------------------- read_state (v7.rs:1229-1262)
int32_t wrk_gguf_read_state(const wrk_gguf* g, float* out, size_t capacity, size_t* count) {
    if (!g || !count) return fail(WRK_E_ARG, "null argument");
    wrk_model_info I{};
    int32_t rc = loader_info(*g, I);
    if (rc != WRK_OK) return rc;
    const uint32_t D = I.num_emb, H = I.num_head, S = D / H, L = I.num_layer;
    const size_t per = (size_t)(S + 2) * D;
    *count = per * L;
    if (!out) return WRK_OK;
    if (capacity < *count) return fail(WRK_E_ARG, "state needs %zu floats, capacity %zu", *count, capacity);
    std::fill(out, out + *count, 0.0f);
    for (uint32_t l = 0; l < L; ++l) {
        const std::string name = "blocks." + std::to_string(l) + ".att.time_state";
        std::vector<uint16_t> h;
        rc = g->tensor_f16(name, h);
        if (rc != WRK_OK) return rc;
        if (h.size() != (size_t)H * S * S) return fail(WRK_E_ARG, "%s: %zu elements, expected %u x %u x %u", name.c_str(), h.size(), H, S, S);
        // file order [h][j][c]; after transpose + blit: row 1 + j, channel h*S + c
        for (uint32_t hh = 0; hh < H; ++hh)
            for (uint32_t j = 0; j < S; ++j)
                for (uint32_t c = 0; c < S; ++c) out[l * per + (size_t)(1 + j) * D + hh * S + c] = h2f(h[((size_t)hh * S + j) * S + c]);
    }
    return WRK_OK;
}

// ---------------------------------------------------------------- ModelBuilder::build_v7
int32_t wrk_runtime_create(wrk_ctx* ctx, const wrk_gguf* g, const wrk_build_options* opt, uint32_t num_batch, wrk_runtime** out) {
    if (!ctx || !g || !out) return fail(WRK_E_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<wrk_runtime> rt(new wrk_runtime());
    rt->ctx = ctx;
    rt->num_batch = num_batch;
    int32_t rc = loader_info(*g, rt->info);
    if (rc != WRK_OK) return rc;
    const wrk_model_info& I = rt->info;
    const uint32_t weights = opt ? opt->weights : WRK_WEIGHTS_INLINE;
    const uint32_t rescale = opt && opt->rescale ? opt->rescale : 1024;
    const uint32_t D = I.num_emb;
    if (I.num_head == 0 || D % I.num_head != 0 || D / I.num_head != 64) return fail(WRK_E_UNSUPPORTED, "head size must be 64");

#define HOSTCHK(expr)                                                                        \
    do {                                                                                     \
        int32_t _r = (expr);                                                                 \
        if (_r != WRK_OK) return fail(_r, "%s failed: %s", #expr, wrk_last_error(ctx));      \
    } while (0)

    // load_vector_f16 (loader.rs:563-615): any float dtype -> f16 on device
    auto vec = [&](const std::string& name, size_t expect, wrk_buf** o) -> int32_t {
        std::vector<uint16_t> v;
        int32_t r = g->tensor_f16(name, v);
        if (r != WRK_OK) return r;
        if (v.size() != expect) return fail(WRK_E_ARG, "%s: %zu elements, expected %zu", name.c_str(), v.size(), expect);
        wrk_buf* b = nullptr;
        r = wrk_buf_create(ctx, v.size() * 2, v.data(), &b);
        if (r != WRK_OK) return fail(r, "upload %s: %s", name.c_str(), wrk_last_error(ctx));
        rt->bufs.push_back(b);
        *o = b;
        return WRK_OK;
    };
    // load_matrix / load_matrix_f16 / load_matrix_discount (loader.rs:617-670, 756-951)
    uint32_t layer_quant = WRK_QUANT_NONE;     // Quant of the layer being built (ModelBuilder::quant)
    auto mat = [&](const std::string& name, uint32_t k, uint32_t m, bool big, float discount, wrk_matrix** o) -> int32_t {
        std::vector<size_t> shp;
        if (!g->shape(name, shp) || shp.size() != 2) return fail(WRK_E_ARG, "tensor not found: %s", name.c_str());
        if (shp[0] != m || shp[1] != k) return fail(WRK_E_ARG, "%s: shape [%zu, %zu], expected [%u, %u]", name.c_str(), shp[0], shp[1], m, k);
        const TensorInfo* ti = g->info(name);
        wrk_matrix* mt = nullptr;
        int32_t r;
        const uint32_t quant = big ? layer_quant : (uint32_t)WRK_QUANT_NONE;
        if (quant != WRK_QUANT_NONE) {
            const size_t n = (size_t)k * m;
            const uint32_t kind = quant == WRK_QUANT_INT8 ? WRK_MAT_INT8 : WRK_MAT_NF4;
            // try_load_matrix_direct's live arms (loader.rs:808-820, 901-918); load_matrix_discount never takes them
            if (discount == 1.0f && quant == WRK_QUANT_INT8 && ti->type == T_Q8_0 && k % 128 == 0) {
                std::vector<uint8_t> blob;
                repack_q8_0_to_int8(g->tensor_data(*ti), n, blob);
                r = wrk_matrix_create(ctx, kind, k, m, blob.data(), blob.size(), WRK_MATRIX_EXACT, &mt);
            } else if (discount == 1.0f && quant == WRK_QUANT_NF4 && ti->type == T_Q4_0 && k % 64 == 0) {
                std::vector<uint8_t> blob;
                repack_q4_0_to_nf4(g->tensor_data(*ti), n, blob);
                r = wrk_matrix_create(ctx, kind, k, m, blob.data(), blob.size(), WRK_MATRIX_EXACT, &mt);
            } else {
                // load_in_place_matrix_f16[_discount] + Matrix::quant_u8 / quant_nf4 (loader.rs:770-781, 932-943)
                std::vector<uint16_t> v;
                r = g->tensor_f16(name, v);
                if (r != WRK_OK) return r;
                if (discount != 1.0f)
                    for (auto& h : v) h = f2h(discount * h2f(h));
                wrk_buf* src = nullptr;
                r = wrk_buf_create(ctx, v.size() * 2, v.data(), &src);
                if (r != WRK_OK) return fail(r, "upload %s: %s", name.c_str(), wrk_last_error(ctx));
                r = wrk_matrix_quantize(ctx, kind, k, m, src, nullptr, &mt);
                wrk_ctx_sync(ctx);
                wrk_buf_release(src);
            }
            if (r != WRK_OK) return fail(r, "quantise %s: %s", name.c_str(), wrk_last_error(ctx));
            rt->mats.push_back(mt);
            *o = mt;
            return WRK_OK;
        }
        // the layer discount 2^-(layer / rescale) is a power of two: it is applied to the dot product (wrk_matrix_set_scale)
        // and the blocks stay quantised, instead of forcing the F16 path as load_matrix_discount does
        const bool direct = big && weights != WRK_WEIGHTS_REFERENCE &&
                            (ti->type == T_Q4K || ti->type == T_Q5K || ti->type == T_Q6K || ti->type == T_Q8_0) &&
                            (ti->type == T_Q8_0 ? k % 32 == 0 : k % 256 == 0);
        if (direct) {
            r = wrk_matrix_create(ctx, ti->type, k, m, g->tensor_data(*ti), ti->data_size(),
                                  weights == WRK_WEIGHTS_INLINE_F16 ? WRK_MATRIX_ROUND_F16 : WRK_MATRIX_EXACT, &mt);
            if (r == WRK_OK && discount != 1.0f) r = wrk_matrix_set_scale(mt, discount);
        } else {
            std::vector<uint16_t> v;
            r = g->tensor_f16(name, v);
            if (r != WRK_OK) return r;
            if (discount != 1.0f)                                                    // loader.rs:650-652
                for (auto& h : v) h = f2h(discount * h2f(h));
            r = wrk_matrix_create(ctx, WRK_MAT_F16, k, m, v.data(), v.size() * 2, WRK_MATRIX_EXACT, &mt);
        }
        if (r != WRK_OK) return fail(r, "upload %s: %s", name.c_str(), wrk_last_error(ctx));
        rt->mats.push_back(mt);
        *o = mt;
        return WRK_OK;
    };
#define VEC(name, n, dst) do { int32_t _r = vec(name, n, &dst); if (_r != WRK_OK) return _r; } while (0)
#define MAT(name, k, m, big, disc, dst) do { int32_t _r = mat(name, k, m, big, disc, &dst); if (_r != WRK_OK) return _r; } while (0)

    if (I.version == 6) {
        // ModelBuilder::build_v6 (v6.rs:995-1170)
        const uint32_t rescale6 = opt && opt->rescale ? opt->rescale : 6;
        const uint32_t R = I.lora_w, W = I.lora_a;
        wrk_v6_model_desc d6{};
        d6.num_layer = I.num_layer; d6.num_emb = D; d6.num_hidden = I.num_hidden; d6.num_vocab = I.num_vocab; d6.num_head = I.num_head;
        d6.time_mix = R; d6.time_decay = W; d6.rescale = rescale6;
        wrk_buf *a, *b6, *c, *e, *emb6;
        VEC("blocks.0.ln0.weight", D, a); VEC("blocks.0.ln0.bias", D, b6); VEC("ln_out.weight", D, c); VEC("ln_out.bias", D, e);
        VEC("emb.weight", (size_t)I.num_vocab * D, emb6);
        wrk_matrix* head6;
        MAT("head.weight", D, I.num_vocab, true, 1.0f, head6);
        d6.ln0_w = a; d6.ln0_b = b6; d6.ln_out_w = c; d6.ln_out_b = e; d6.emb_f16 = emb6; d6.head = head6;
        rt->layers6.resize(I.num_layer);
        for (uint32_t l = 0; l < I.num_layer; ++l) {
            wrk_v6_layer_desc& L = rt->layers6[l];
            const float discount = 1.0f / (float)(1u << std::min<uint32_t>(l / rescale6, 30));
            layer_quant = opt && opt->quant && l < opt->num_quant ? opt->quant[l] : (uint32_t)WRK_QUANT_NONE;
            if (layer_quant > WRK_QUANT_NF4) return fail(WRK_E_ARG, "quant[%u] = %u is not a WRK_QUANT_* value", l, layer_quant);
            const std::string blk = "blocks." + std::to_string(l), att = blk + ".att", ffn = blk + ".ffn";
            wrk_buf* vb;
            wrk_matrix* m;
#define V6(field, name) VEC(name, D, vb); L.field = vb
#define M6(field, name, k, mm, big, disc) MAT(name, k, mm, big, disc, m); L.field = m
            V6(ln1_w, blk + ".ln1.weight"); V6(ln1_b, blk + ".ln1.bias"); V6(ln2_w, blk + ".ln2.weight"); V6(ln2_b, blk + ".ln2.bias");
            V6(time_decay, att + ".time_decay"); V6(time_mix_x, att + ".time_mix_x");
            {   // load_vector_f32 (loader.rs:443-478): f16-rounded values widened to f32
                std::vector<uint16_t> h;
                int32_t r = g->tensor_f16(att + ".time_first", h);
                if (r != WRK_OK) return r;
                if (h.size() != D) return fail(WRK_E_ARG, "%s.time_first: %zu elements, expected %u", att.c_str(), h.size(), D);
                std::vector<float> f(h.size());
                for (size_t i = 0; i < h.size(); ++i) f[i] = h2f(h[i]);
                wrk_buf* fb = nullptr;
                r = wrk_buf_create(ctx, f.size() * 4, f.data(), &fb);
                if (r != WRK_OK) return fail(r, "upload time_first: %s", wrk_last_error(ctx));
                rt->bufs.push_back(fb);
                L.time_first = fb;
            }
            {   // time_mix stack [D, 1, 5] = w, k, v, r, g (v6.rs:1054-1071)
                std::vector<uint16_t> all;
                for (const char* n : {"w", "k", "v", "r", "g"}) {
                    std::vector<uint16_t> h;
                    int32_t r = g->tensor_f16(att + ".time_mix_" + n, h);
                    if (r != WRK_OK) return r;
                    if (h.size() != D) return fail(WRK_E_ARG, "%s.time_mix_%s: bad size", att.c_str(), n);
                    all.insert(all.end(), h.begin(), h.end());
                }
                wrk_buf* tb = nullptr;
                int32_t r = wrk_buf_create(ctx, all.size() * 2, all.data(), &tb);
                if (r != WRK_OK) return fail(r, "upload time_mix: %s", wrk_last_error(ctx));
                rt->bufs.push_back(tb);
                L.time_mix = tb;
            }
            M6(time_decay_w1, att + ".time_decay_w1", D, W, false, 1.0f); M6(time_decay_w2, att + ".time_decay_w2", W, D, false, 1.0f);
            M6(time_mix_w1, att + ".time_mix_w1", D, 5 * R, false, 1.0f);
            {   // batched time_mix_w2 [R, D, 5] -> five F16 matrices [R -> D]
                std::vector<size_t> shp;
                if (!g->shape(att + ".time_mix_w2", shp) || shp.size() != 3 || shp[0] != 5 || shp[1] != D || shp[2] != R)
                    return fail(WRK_E_ARG, "%s.time_mix_w2: expected shape [5, %u, %u]", att.c_str(), D, R);
                std::vector<uint16_t> h;
                int32_t r = g->tensor_f16(att + ".time_mix_w2", h);
                if (r != WRK_OK) return r;
                for (int i = 0; i < 5; ++i) {
                    wrk_matrix* mt = nullptr;
                    r = wrk_matrix_create(ctx, WRK_MAT_F16, R, D, h.data() + (size_t)i * D * R, (size_t)D * R * 2, WRK_MATRIX_EXACT, &mt);
                    if (r != WRK_OK) return fail(r, "upload time_mix_w2[%d]: %s", i, wrk_last_error(ctx));
                    rt->mats.push_back(mt);
                    L.time_mix_w2[i] = mt;
                }
            }
            V6(gn_w, att + ".ln_x.weight"); V6(gn_b, att + ".ln_x.bias");
            M6(w_k, att + ".key.weight", D, D, true, 1.0f); M6(w_v, att + ".value.weight", D, D, true, 1.0f);
            M6(w_r, att + ".receptance.weight", D, D, true, 1.0f); M6(w_g, att + ".gate.weight", D, D, true, 1.0f);
            M6(w_o, att + ".output.weight", D, D, true, discount);
            V6(ffn_mix_k, ffn + ".time_mix_k"); V6(ffn_mix_r, ffn + ".time_mix_r");
            M6(ffn_w_k, ffn + ".key.weight", D, I.num_hidden, true, 1.0f); M6(ffn_w_v, ffn + ".value.weight", I.num_hidden, D, true, discount);
            M6(ffn_w_r, ffn + ".receptance.weight", D, D, true, 1.0f);
#undef V6
#undef M6
        }
        d6.layers = rt->layers6.data();
        HOSTCHK(wrk_v6_model_create(ctx, &d6, &rt->model6));
        HOSTCHK(wrk_v6_state_create(ctx, rt->model6, num_batch, &rt->state));
        *out = rt.release();
        return WRK_OK;
    }

    wrk_v7_model_desc desc{};
    desc.num_layer = I.num_layer; desc.num_emb = D; desc.num_hidden = I.num_hidden; desc.num_vocab = I.num_vocab; desc.num_head = I.num_head;
    desc.lora_w = I.lora_w; desc.lora_a = I.lora_a; desc.lora_g = I.lora_g; desc.lora_v = I.lora_v;
    desc.rescale = rescale;
    wrk_buf *ln0_w, *ln0_b, *lno_w, *lno_b, *emb;
    VEC("blocks.0.ln0.weight", D, ln0_w);
    VEC("blocks.0.ln0.bias", D, ln0_b);
    VEC("ln_out.weight", D, lno_w);
    VEC("ln_out.bias", D, lno_b);
    VEC("emb.weight", (size_t)I.num_vocab * D, emb);          // f16 table (v7.rs:1065), kept on the device
    wrk_matrix* head;
    MAT("head.weight", D, I.num_vocab, true, 1.0f, head);
    desc.ln0_w = ln0_w; desc.ln0_b = ln0_b; desc.ln_out_w = lno_w; desc.ln_out_b = lno_b; desc.emb_f16 = emb; desc.head = head;

    rt->layers.resize(I.num_layer);
    for (uint32_t l = 0; l < I.num_layer; ++l) {
        wrk_v7_layer_desc& L = rt->layers[l];
        const float discount = 1.0f / (float)(1u << std::min<uint32_t>(l / rescale, 30));      // 2^-(layer / rescale), v7.rs:1090
        layer_quant = opt && opt->quant && l < opt->num_quant ? opt->quant[l] : (uint32_t)WRK_QUANT_NONE;   // v7.rs:1089
        if (layer_quant > WRK_QUANT_NF4) return fail(WRK_E_ARG, "quant[%u] = %u is not a WRK_QUANT_* value", l, layer_quant);
        const std::string blk = "blocks." + std::to_string(l), att = blk + ".att", ffn = blk + ".ffn";
        wrk_buf* b;
        wrk_matrix* m;
#define V(field, name) VEC(name, D, b); L.field = b
#define M(field, name, k, mm, big, disc) MAT(name, k, mm, big, disc, m); L.field = m
        V(ln1_w, blk + ".ln1.weight"); V(ln1_b, blk + ".ln1.bias"); V(ln2_w, blk + ".ln2.weight"); V(ln2_b, blk + ".ln2.bias");
        V(x_r, att + ".x_r"); V(x_w, att + ".x_w"); V(x_k, att + ".x_k"); V(x_v, att + ".x_v"); V(x_a, att + ".x_a"); V(x_g, att + ".x_g");
        V(w0, att + ".w0"); V(a0, att + ".a0");
        M(w1, att + ".w1", D, I.lora_w, false, 1.0f); M(w2, att + ".w2", I.lora_w, D, false, 1.0f);
        M(a1, att + ".a1", D, I.lora_a, false, 1.0f); M(a2, att + ".a2", I.lora_a, D, false, 1.0f);
        M(g1, att + ".g1", D, I.lora_g, false, 1.0f); M(g2, att + ".g2", I.lora_g, D, false, 1.0f);
        if (l == 0) { L.v0 = nullptr; L.v1 = nullptr; L.v2 = nullptr; }              // v7.rs:1115-1116: unused placeholders
        else {
            V(v0, att + ".v0");
            M(v1, att + ".v1", D, I.lora_v, false, 1.0f); M(v2, att + ".v2", I.lora_v, D, false, 1.0f);
        }
        V(r_k, att + ".r_k"); V(k_k, att + ".k_k"); V(k_a, att + ".k_a");
        V(gn_w, att + ".ln_x.weight"); V(gn_b, att + ".ln_x.bias");
        M(w_k, att + ".key.weight", D, D, true, 1.0f); M(w_v, att + ".value.weight", D, D, true, 1.0f);
        M(w_r, att + ".receptance.weight", D, D, true, 1.0f); M(w_o, att + ".output.weight", D, D, true, discount);
        V(ffn_x_k, ffn + ".x_k");
        M(ffn_w_k, ffn + ".key.weight", D, I.num_hidden, true, 1.0f); M(ffn_w_v, ffn + ".value.weight", I.num_hidden, D, true, discount);
#undef V
#undef M
    }
    desc.layers = rt->layers.data();
    HOSTCHK(wrk_v7_model_create(ctx, &desc, &rt->model));
    HOSTCHK(wrk_v7_state_create(ctx, rt->model, num_batch, &rt->state));
    *out = rt.release();
    return WRK_OK;
}

int32_t wrk_runtime_destroy(wrk_runtime* rt) { delete rt; return WRK_OK; }
int32_t wrk_runtime_info(const wrk_runtime* rt, wrk_model_info* out) { if (!rt || !out) return WRK_E_ARG; *out = rt->info; return WRK_OK; }
wrk_v7_model* wrk_runtime_model(wrk_runtime* rt) { return rt ? rt->model : nullptr; }
wrk_v7_state* wrk_runtime_state(wrk_runtime* rt) { return rt ? rt->state : nullptr; }
wrk_v6_model* wrk_runtime_model_v6(wrk_runtime* rt) { return rt ? rt->model6 : nullptr; }

// SimpleRuntime::infer (mod.rs:238-263) with RnnJob::{load, submit, back} (v7.rs:434-492)
int32_t wrk_runtime_infer(wrk_runtime* rt, wrk_rnn_input* in, float* logits, size_t capacity_rows, uint32_t* rows, uint32_t mode) {
    if (!rt || !in || !rows) return fail(WRK_E_ARG, "null argument");
    const uint32_t nb = (uint32_t)in->tokens.size();
    if (nb != rt->num_batch) return fail(WRK_E_ARG, "input has %u batches, bundle was built for %u", nb, rt->num_batch);
    wrk_rnn_iter it;
    make_iter(*in, it);
    std::vector<uint32_t> lens(nb);
    std::vector<int32_t> opts(nb);
    it.next(lens.data(), opts.data());
    uint32_t T = 0;
    for (uint32_t b = 0; b < nb; ++b) T += lens[b];
    for (uint32_t b = 0; b < nb; ++b) rows[b] = 0;
    if (T == 0) return fail(WRK_E_ARG, "input iterator exhausted");
    std::vector<uint32_t> headers;
    std::vector<std::pair<uint32_t, uint32_t>> inputs, outputs;
    redirect(lens.data(), opts.data(), nb, headers, inputs, outputs);
    if (headers.size() > capacity_rows) return fail(WRK_E_ARG, "logits buffer holds %zu rows, chunk produces %zu", capacity_rows, headers.size());
    // chunk + TensorStack cursors (tensor/mod.rs:1185-1233, into_cursors :70-84)
    std::vector<uint32_t> toks, cursors;
    uint32_t token = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        if (lens[b] > 255) return fail(WRK_E_UNSUPPORTED, "batch %u: %u tokens in one chunk exceed the cursor's u8 length (tensor/mod.rs:53-60)", b, lens[b]);
        for (uint32_t i = 0; i < lens[b]; ++i) {
            toks.push_back(in->tokens[b][i]);
            cursors.push_back((b & 0xff) | ((token & 0xffff) << 8) | ((lens[b] & 0xff) << 24));
        }
        token += lens[b];
    }
    const int32_t rc = rt->model6
        ? wrk_v6_infer(rt->ctx, rt->model6, rt->state, toks.data(), nullptr, cursors.data(), T, headers.data(), (uint32_t)headers.size(), logits, nullptr, mode)
        : wrk_v7_infer(rt->ctx, rt->model, rt->state, toks.data(), nullptr, cursors.data(), T, headers.data(), (uint32_t)headers.size(), logits, nullptr, mode);
    if (rc != WRK_OK) return fail(rc, "infer: %s", wrk_last_error(rt->ctx));
    for (uint32_t b = 0; b < nb; ++b) rows[b] = outputs[b].second - outputs[b].first;
    for (uint32_t b = 0; b < nb; ++b) in->tokens[b].erase(in->tokens[b].begin(), in->tokens[b].begin() + lens[b]);   // input.step()
    return WRK_OK;
}

}  // extern "C"
