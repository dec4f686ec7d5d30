/*
 * wrk_oracle.c -- plain-C restatement of the reference's RWKV-7 decode path on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests as a second, independent checker
 * of oracle/rwkv7.py and by bench.py's `cpu_baseline` leg ("reference path restated on CPU",
 * BASELINE.md section 3).  Never linked into or called from the product libraries.
 *
 * It is the reference's EFFECTIVE arithmetic at HEAD: every matrix is dequantised
 * to f16 on the CPU at load (src/runtime/gguf.rs:11-274), activations are stored as f16 between
 * ops (Bundle::<f16>), accumulation is f32, state and logits are f32.  One sequence, one token per
 * call (decode); op order = src/runtime/v7.rs:649-659, 716-1036.
 *
 * Build: oracle/c/Makefile (gcc -O3 -fopenmp -mavx2 -mf16c, -ffp-contract=off so a*b-c keeps the
 * two roundings the reference's Rust code has).
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint16_t h16;

static inline float h2f(h16 h) { return _cvtsh_ss(h); }
static inline h16 f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }   /* RNE == half::f16::from_f32 */
static inline float r16(float f) { return h2f(f2h(f)); }

/* ------------------------------------------------------------------ dequantisers (gguf.rs) */
static void scale_min_k4(int j, const uint8_t* s, uint8_t* sc, uint8_t* m) {          /* gguf.rs:81-89 */
    if (j < 4) { *sc = s[j] & 63; *m = s[j + 4] & 63; }
    else { *sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4); *m = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
}

static float ldh(const uint8_t* p) { h16 b; memcpy(&b, p, 2); return h2f(b); }

/* type ids = ggml (gguf.rs:888-923): 0 F32, 1 F16, 8 Q8_0, 12 Q4_K, 13 Q5_K, 14 Q6_K */
int orc_dequant_f16(uint32_t type, const uint8_t* d, size_t n, h16* out) {
    if (type == 1) { memcpy(out, d, n * 2); return 0; }
    if (type == 0) {                                                                  /* loader.rs:117-121 */
        const float* f = (const float*)d;
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)n; ++i) out[i] = f2h(f[i]);
        return 0;
    }
    if (type == 8) {                                                                  /* gguf.rs:11-37 */
#pragma omp parallel for schedule(static)
        for (long long b = 0; b < (long long)(n / 32); ++b) {
            const uint8_t* blk = d + b * 34;
            const float sc = ldh(blk);
            for (int i = 0; i < 32; ++i) out[b * 32 + i] = f2h((float)(int8_t)blk[2 + i] * sc);
        }
        return 0;
    }
    if (type == 12 || type == 13) {                                                   /* gguf.rs:95-204 */
        const size_t bb = type == 12 ? 144 : 176;
#pragma omp parallel for schedule(static)
        for (long long b = 0; b < (long long)(n / 256); ++b) {
            const uint8_t* blk = d + b * bb;
            const float dd = ldh(blk), dmin = ldh(blk + 2);
            const uint8_t* scales = blk + 4;
            const uint8_t* qh = blk + 16;
            const uint8_t* ql = type == 12 ? blk + 16 : blk + 48;
            h16* o = out + b * 256;
            for (int g = 0; g < 4; ++g) {
                uint8_t s0, m0, s1, m1;
                scale_min_k4(2 * g, scales, &s0, &m0);
                scale_min_k4(2 * g + 1, scales, &s1, &m1);
                const float d1 = dd * (float)s0, mv1 = dmin * (float)m0, d2 = dd * (float)s1, mv2 = dmin * (float)m1;
                for (int l = 0; l < 32; ++l) {
                    int q = ql[32 * g + l] & 0xF;
                    if (type == 13 && (qh[l] & (1 << (2 * g)))) q += 16;
                    const float p = d1 * (float)q;
                    *o++ = f2h(p - mv1);
                }
                for (int l = 0; l < 32; ++l) {
                    int q = ql[32 * g + l] >> 4;
                    if (type == 13 && (qh[l] & (2 << (2 * g)))) q += 16;
                    const float p = d2 * (float)q;
                    *o++ = f2h(p - mv2);
                }
            }
        }
        return 0;
    }
    if (type == 14) {                                                                 /* gguf.rs:210-274 */
#pragma omp parallel for schedule(static)
        for (long long b = 0; b < (long long)(n / 256); ++b) {
            const uint8_t* blk = d + b * 210;
            const uint8_t *ql = blk, *qh = blk + 128;
            const int8_t* sc = (const int8_t*)(blk + 192);
            const float dd = ldh(blk + 208);
            h16* o = out + b * 256;
            for (int nn = 0; nn < 2; ++nn)
                for (int l = 0; l < 32; ++l) {
                    const int is = l / 16, a = ql[64 * nn + l], c = ql[64 * nn + l + 32], h = qh[32 * nn + l];
                    const int q1 = ((a & 0xF) | ((h & 3) << 4)) - 32, q2 = ((c & 0xF) | (((h >> 2) & 3) << 4)) - 32;
                    const int q3 = ((a >> 4) | (((h >> 4) & 3) << 4)) - 32, q4 = ((c >> 4) | (((h >> 6) & 3) << 4)) - 32;
                    const float s0 = dd * (float)sc[8 * nn + is], s2 = dd * (float)sc[8 * nn + is + 2];
                    const float s4 = dd * (float)sc[8 * nn + is + 4], s6 = dd * (float)sc[8 * nn + is + 6];
                    o[128 * nn + l] = f2h(s0 * (float)q1);
                    o[128 * nn + 32 + l] = f2h(s2 * (float)q2);
                    o[128 * nn + 64 + l] = f2h(s4 * (float)q3);
                    o[128 * nn + 96 + l] = f2h(s6 * (float)q4);
                }
        }
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------ model */
typedef struct {
    const h16 *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const h16 *x_r, *x_w, *x_k, *x_v, *x_a, *x_g;
    const h16 *w0, *a0, *v0;
    const h16 *w1, *w2, *a1, *a2, *g1, *g2, *v1, *v2;
    const h16 *r_k, *k_k, *k_a, *gn_w, *gn_b;
    const h16 *w_k, *w_v, *w_r, *w_o;
    const h16 *ffn_x_k, *ffn_w_k, *ffn_w_v;
} orc_layer;

typedef struct {
    uint32_t num_layer, num_emb, num_hidden, num_vocab, num_head;
    uint32_t lora_w, lora_a, lora_g, lora_v;
    const h16 *emb, *ln0_w, *ln0_b, *ln_out_w, *ln_out_b, *head;
    const orc_layer* layers;
} orc_model;

/* y[m] = act(W[m][k] . x[k]) with f16 weights/inputs, f32 accumulate (matmul_vec_fp16.wgsl:48-110) */
static void matvec(const h16* w, const float* x, float* y, uint32_t k, uint32_t m) {
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < (long long)m; ++r) {
        const h16* row = w + (size_t)r * k;
        __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
        uint32_t i = 0;
        for (; i + 16 <= k; i += 16) {
            const __m256 a0 = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(row + i)));
            const __m256 a1 = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(row + i + 8)));
            acc0 = _mm256_add_ps(acc0, _mm256_mul_ps(a0, _mm256_loadu_ps(x + i)));
            acc1 = _mm256_add_ps(acc1, _mm256_mul_ps(a1, _mm256_loadu_ps(x + i + 8)));
        }
        float tmp[8];
        _mm256_storeu_ps(tmp, _mm256_add_ps(acc0, acc1));
        float s = ((tmp[0] + tmp[1]) + (tmp[2] + tmp[3])) + ((tmp[4] + tmp[5]) + (tmp[6] + tmp[7]));
        for (; i < k; ++i) s += h2f(row[i]) * x[i];
        y[r] = s;
    }
}

static float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

/* layer_norm.wgsl:63-121 (statistics in double, see oracle/rwkv7.py) ; output rounded to f16 */
static void layer_norm16(const float* x, const h16* w, const h16* b, float eps, float* y, uint32_t n) {
    double s = 0, q = 0;
    for (uint32_t i = 0; i < n; ++i) s += x[i];
    const double mean = s / n;
    for (uint32_t i = 0; i < n; ++i) q += (x[i] - mean) * (x[i] - mean);
    const float meanf = (float)mean, dev = 1.0f / sqrtf((float)(q / n) + eps);
    for (uint32_t i = 0; i < n; ++i) y[i] = r16((x[i] - meanf) * dev * h2f(w[i]) + h2f(b[i]));
}

static float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }

/* One decode step of one sequence.  state: f32 [L][S+2][D] (v7.rs:146-208).  scratch: >= 24*D + F + 600 floats. */
void orc_v7_decode(const orc_model* m, float* state, uint32_t token, float* logits, float* scratch) {
    const uint32_t D = m->num_emb, F = m->num_hidden, H = m->num_head, S = D / H;
    float* x = scratch;          float* ax = x + D;      float* rx = ax + D;   float* wx = rx + D;   float* kx = wx + D;
    float* vx = kx + D;          float* aax = vx + D;    float* gx = aax + D;  float* r = gx + D;    float* w = r + D;
    float* k = w + D;            float* v = k + D;       float* a = v + D;     float* g = a + D;     float* kk = g + D;
    float* v0 = kk + D;          float* y = v0 + D;      float* o = y + D;     float* fx = o + D;    float* fkx = fx + D;
    float* fv = fkx + D;         float* tmp = fv + D;    float* fk = tmp + D;  float* aux = fk + F;
    /* embed: f16 row, LN(ln0) (v7.rs:438-474, 649-659) */
    for (uint32_t i = 0; i < D; ++i) tmp[i] = h2f(m->emb[(size_t)token * D + i]);
    layer_norm16(tmp, m->ln0_w, m->ln0_b, 1e-5f, x, D);
    for (uint32_t li = 0; li < m->num_layer; ++li) {
        const orc_layer* L = &m->layers[li];
        float* st = state + (size_t)li * (S + 2) * D;
        layer_norm16(x, L->ln1_w, L->ln1_b, 1e-5f, ax, D);                                  /* 1-2 */
        for (uint32_t i = 0; i < D; ++i) {                                                  /* 3 (token_shift REVERSED) */
            const float p = st[i], c = ax[i];
            rx[i] = r16(mixf(c, p, h2f(L->x_r[i]))); wx[i] = r16(mixf(c, p, h2f(L->x_w[i])));
            kx[i] = r16(mixf(c, p, h2f(L->x_k[i]))); vx[i] = r16(mixf(c, p, h2f(L->x_v[i])));
            aax[i] = r16(mixf(c, p, h2f(L->x_a[i]))); gx[i] = r16(mixf(c, p, h2f(L->x_g[i])));
        }
        matvec(L->w_r, rx, r, D, D); matvec(L->w_k, kx, k, D, D); matvec(L->w_v, vx, v, D, D);          /* 4 */
        for (uint32_t i = 0; i < D; ++i) { r[i] = r16(r[i]); k[i] = r16(k[i]); v[i] = r16(v[i]); }
        matvec(L->w1, wx, aux, D, m->lora_w);                                                             /* 5 */
        for (uint32_t i = 0; i < m->lora_w; ++i) aux[i] = r16(aux[i] > 42.0f ? 1.0f : tanhf(aux[i]));
        matvec(L->w2, aux, w, m->lora_w, D);
        for (uint32_t i = 0; i < D; ++i) w[i] = r16(h2f(L->w0[i]) + r16(w[i]));
        matvec(L->a1, aax, aux, D, m->lora_a);                                                            /* 6 */
        for (uint32_t i = 0; i < m->lora_a; ++i) aux[i] = r16(aux[i]);
        matvec(L->a2, aux, a, m->lora_a, D);
        for (uint32_t i = 0; i < D; ++i) a[i] = r16(sigmoidf(h2f(L->a0[i]) + r16(a[i])));
        matvec(L->g1, gx, aux, D, m->lora_g);                                                             /* 7 */
        for (uint32_t i = 0; i < m->lora_g; ++i) aux[i] = r16(sigmoidf(aux[i]));
        matvec(L->g2, aux, g, m->lora_g, D);
        for (uint32_t i = 0; i < D; ++i) g[i] = r16(g[i]);
        for (uint32_t i = 0; i < D; ++i) kk[i] = r16(h2f(L->k_k[i]) * k[i]);                              /* 8 */
        for (uint32_t h = 0; h < H; ++h) {
            double s2 = 0;
            for (uint32_t i = 0; i < S; ++i) s2 += (double)kk[h * S + i] * kk[h * S + i];
            const float nrm = 1.0f / sqrtf((float)s2 + 1e-12f);
            for (uint32_t i = 0; i < S; ++i) kk[h * S + i] = r16(kk[h * S + i] * nrm);
        }
        for (uint32_t i = 0; i < D; ++i) k[i] = r16(k[i] * (1.0f + (a[i] - 1.0f) * h2f(L->k_a[i])));     /* 9 */
        if (li == 0) memcpy(v0, v, D * 4);                                                                /* 10 */
        else {
            matvec(L->v1, vx, aux, D, m->lora_v);
            for (uint32_t i = 0; i < m->lora_v; ++i) aux[i] = r16(aux[i]);
            matvec(L->v2, aux, tmp, m->lora_v, D);
            for (uint32_t i = 0; i < D; ++i) {
                const float vv = r16(sigmoidf(h2f(L->v0[i]) + r16(tmp[i])));
                v[i] = r16(mixf(v[i], v0[i], vv));
            }
        }
        memcpy(st, ax, D * 4);                                                         /* 12: shift-state carry */
#pragma omp parallel for schedule(static)
        for (long long h = 0; h < (long long)H; ++h) {                                 /* 12: WKV7 (time_mix_v7.wgsl:143-221) */
            float sa[64], yy[64];
            for (uint32_t i = 0; i < S; ++i) { sa[i] = 0; yy[i] = 0; }
            for (uint32_t j = 0; j < S; ++j) {
                const float aj = -kk[h * S + j];
                const float* row = st + (size_t)(1 + j) * D + h * S;
                for (uint32_t i = 0; i < S; ++i) sa[i] += row[i] * aj;
            }
            for (uint32_t j = 0; j < S; ++j) {
                const float wj = expf(-0.606531f * sigmoidf(w[h * S + j])), kj = k[h * S + j], bj = kk[h * S + j] * a[h * S + j], rj = r[h * S + j];
                float* row = st + (size_t)(1 + j) * D + h * S;
                for (uint32_t i = 0; i < S; ++i) {
                    const float s = row[i] * wj + kj * v[h * S + i] + sa[i] * bj;
                    row[i] = s;
                    yy[i] += rj * s;
                }
            }
            /* 13 group norm, 14 time_first, 15 gate */
            double s1 = 0, q1 = 0, xx = 0;
            for (uint32_t i = 0; i < S; ++i) { yy[i] = r16(yy[i]); s1 += yy[i]; }
            const double mean = s1 / S;
            for (uint32_t i = 0; i < S; ++i) q1 += (yy[i] - mean) * (yy[i] - mean);
            const float meanf = (float)mean, dev = 1.0f / sqrtf((float)(q1 / S) + 64e-5f);
            for (uint32_t i = 0; i < S; ++i) xx += (double)(h2f(L->r_k[h * S + i]) * k[h * S + i] * r[h * S + i]);
            for (uint32_t i = 0; i < S; ++i) {
                float t = r16((yy[i] - meanf) * dev * h2f(L->gn_w[h * S + i]) + h2f(L->gn_b[h * S + i]));
                t = r16(t + (float)xx * v[h * S + i]);
                y[h * S + i] = r16(g[h * S + i] * t);
            }
        }
        matvec(L->w_o, y, o, D, D);                                                                        /* 16 */
        for (uint32_t i = 0; i < D; ++i) x[i] = r16(r16(o[i]) + x[i]);
        layer_norm16(x, L->ln2_w, L->ln2_b, 1e-5f, fx, D);                                                  /* 17 */
        float* sf = st + (size_t)(S + 1) * D;
        for (uint32_t i = 0; i < D; ++i) fkx[i] = r16(mixf(fx[i], sf[i], h2f(L->ffn_x_k[i])));             /* 18 */
        matvec(L->ffn_w_k, fkx, fk, D, F);                                                                  /* 19 */
        for (uint32_t i = 0; i < F; ++i) { const float p = fk[i] > 0 ? fk[i] : 0; fk[i] = r16(p * p); }
        matvec(L->ffn_w_v, fk, fv, F, D);                                                                   /* 20 */
        memcpy(sf, fx, D * 4);                                                                              /* 21 */
        for (uint32_t i = 0; i < D; ++i) x[i] = r16(r16(fv[i]) + x[i]);                                     /* 22 */
    }
    layer_norm16(x, m->ln_out_w, m->ln_out_b, 1e-5f, tmp, D);                          /* head (v7.rs:1009-1036) */
    matvec(m->head, tmp, logits, D, m->num_vocab);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
