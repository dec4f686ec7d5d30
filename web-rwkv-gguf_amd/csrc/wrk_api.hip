// C ABI: context, buffers, programs, matrices and the wrk_op_* TensorOp entry points
// (include/wrk_hip.h).  Each function validates shapes the way the reference's TensorOp
// constructors do (TensorError -> WRK_E_ARG) and enqueues the kernels of wrk_ops.hip /
// wrk_matvec.hip on the context's stream.
#include "wrk_internal.h"

#include <cmath>
#include <memory>

#define LOCK(ctx) std::lock_guard<std::recursive_mutex> _lk((ctx)->mu)

extern "C" {

int32_t wrk_abi_version(void) { return WRK_ABI_VERSION; }

// ---------------------------------------------------------------- context
int32_t wrk_ctx_create(int32_t device, wrk_ctx** out) {
    if (!out) return WRK_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return WRK_E_HIP;   // no HIP device: fail loudly
    if (device < 0 || device >= count) return WRK_E_ARG;
    std::unique_ptr<wrk_ctx> ctx(new wrk_ctx());
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) return WRK_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return WRK_E_HIP;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return WRK_E_HIP;
    if (hipStreamCreateWithFlags(&ctx->read_stream, hipStreamNonBlocking) != hipSuccess) return WRK_E_HIP;
    if (hipEventCreateWithFlags(&ctx->read_event, hipEventDisableTiming) != hipSuccess) return WRK_E_HIP;
    *out = ctx.release();
    return WRK_OK;
}

int32_t wrk_ctx_destroy(wrk_ctx* ctx) {
    if (!ctx) return WRK_E_ARG;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->read_stream);
    if (ctx->staging) hipHostFree(ctx->staging);
    if (ctx->gemm_scratch) hipFree(ctx->gemm_scratch);
    for (auto& kv : ctx->sessions) { hipGraph_t g = nullptr; hipStreamEndCapture(kv.second, &g); if (g) hipGraphDestroy(g); hipStreamDestroy(kv.second); }
    for (hipStream_t s : ctx->capture_pool) hipStreamDestroy(s);
    hipEventDestroy(ctx->read_event);
    hipStreamDestroy(ctx->read_stream);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return WRK_OK;
}

}  // extern "C"

int32_t wrk_ctx_reserve_gemm_scratch(wrk_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->gemm_scratch_cap) return WRK_OK;
    if (ctx->capturing_here()) return WRK_OK;
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));        // programs in flight may still read the old buffer
    void* p = nullptr;
    WRK_HIP(ctx, hipMalloc(&p, bytes));
    if (ctx->gemm_scratch) hipFree(ctx->gemm_scratch);
    ctx->gemm_scratch = p;
    ctx->gemm_scratch_cap = bytes;
    return WRK_OK;
}

extern "C" {

const char* wrk_last_error(wrk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t wrk_ctx_sync(wrk_ctx* ctx) {
    if (!ctx) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WRK_OK;
}

void* wrk_ctx_stream(wrk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---------------------------------------------------------------- buffers
int32_t wrk_buf_create(wrk_ctx* ctx, size_t bytes, const void* init, wrk_buf** out) {
    if (!ctx || !out) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    const size_t alloc = bytes ? ((bytes + 255) & ~(size_t)255) : 256;
    WRK_HIP(ctx, hipMalloc(&p, alloc));
    if (init) {
        hipError_t e = hipMemcpyAsync(p, init, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // source borrowed only for this call
        if (e != hipSuccess) { hipFree(p); return wrk_fail(ctx, WRK_E_HIP, "upload: %s", hipGetErrorString(e)); }
    } else {
        hipError_t e = hipMemsetAsync(p, 0, alloc, ctx->stream);
        if (e != hipSuccess) { hipFree(p); return wrk_fail(ctx, WRK_E_HIP, "memset: %s", hipGetErrorString(e)); }
    }
    wrk_buf* b = new wrk_buf{ctx, p, bytes, {1}};
    *out = b;
    return WRK_OK;
}

int32_t wrk_buf_retain(wrk_buf* buf) {
    if (!buf) return WRK_E_ARG;
    buf->refs.fetch_add(1);
    return WRK_OK;
}

int32_t wrk_buf_release(wrk_buf* buf) {
    if (!buf) return WRK_E_ARG;
    if (buf->refs.fetch_sub(1) == 1) {
        wrk_ctx* ctx = buf->ctx;
        LOCK(ctx);
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);      // no kernel may still reference it
        hipFree(buf->ptr);
        delete buf;
    }
    return WRK_OK;
}

size_t wrk_buf_size(const wrk_buf* buf) { return buf ? buf->bytes : 0; }
void* wrk_buf_device_ptr(const wrk_buf* buf) { return buf ? buf->ptr : nullptr; }

int32_t wrk_buf_write(wrk_ctx* ctx, wrk_buf* buf, size_t offset, const void* src, size_t bytes) {
    if (!ctx || !buf || (!src && bytes)) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, offset + bytes <= buf->bytes, "wrk_buf_write: range %zu+%zu exceeds buffer of %zu bytes", offset, bytes, buf->bytes);
    if (bytes == 0) return WRK_OK;
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    // stream ordered; the source must be consumed before returning -> stage through pinned memory
    if (ctx->staging_bytes < bytes) {
        WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->staging) hipHostFree(ctx->staging);
        ctx->staging = nullptr;
        ctx->staging_bytes = 0;
        size_t nb = bytes < (1u << 20) ? (1u << 20) : bytes;
        WRK_HIP(ctx, hipHostMalloc(&ctx->staging, nb, hipHostMallocDefault));
        ctx->staging_bytes = nb;
    } else {
        // the previous write may still be reading the staging area
        WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    memcpy(ctx->staging, src, bytes);
    WRK_HIP(ctx, hipMemcpyAsync((char*)buf->ptr + offset, ctx->staging, bytes, hipMemcpyHostToDevice, ctx->stream));
    return WRK_OK;
}

int32_t wrk_buf_read(wrk_ctx* ctx, const wrk_buf* buf, size_t offset, void* dst, size_t bytes) {
    if (!ctx || !buf || (!dst && bytes)) return WRK_E_ARG;
    {
        LOCK(ctx);
        WRK_ARG(ctx, offset + bytes <= buf->bytes, "wrk_buf_read: range %zu+%zu exceeds buffer of %zu bytes", offset, bytes, buf->bytes);
        if (bytes == 0) return WRK_OK;
        WRK_HIP(ctx, hipSetDevice(ctx->device));
        // order the read after everything submitted so far, on the read-back stream
        WRK_HIP(ctx, hipEventRecord(ctx->read_event, ctx->stream));
        WRK_HIP(ctx, hipStreamWaitEvent(ctx->read_stream, ctx->read_event, 0));
        WRK_HIP(ctx, hipMemcpyAsync(dst, (const char*)buf->ptr + offset, bytes, hipMemcpyDeviceToHost, ctx->read_stream));
    }
    WRK_HIP(ctx, hipStreamSynchronize(ctx->read_stream));       // blocking, outside the lock
    return WRK_OK;
}

int32_t wrk_buf_copy(wrk_ctx* ctx, const wrk_buf* src, size_t so, wrk_buf* dst, size_t dof, size_t bytes) {
    if (!ctx || !src || !dst) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, so + bytes <= src->bytes && dof + bytes <= dst->bytes, "wrk_buf_copy: out of range");
    if (bytes == 0) return WRK_OK;
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_HIP(ctx, hipMemcpyAsync((char*)dst->ptr + dof, (const char*)src->ptr + so, bytes, hipMemcpyDeviceToDevice, ctx->op_stream()));     // an encoder command
    return WRK_OK;
}

// ---------------------------------------------------------------- programs
int32_t wrk_capture_begin(wrk_ctx* ctx) {
    if (!ctx) return WRK_E_ARG;
    LOCK(ctx);
    WRK_ARG(ctx, !ctx->capturing_here(), "this thread already has a capture in progress on the context");
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    if (!ctx->capture_pool.empty()) { s = ctx->capture_pool.back(); ctx->capture_pool.pop_back(); }
    else WRK_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // thread-local mode: only THIS thread is restricted while its capture is open; other threads allocate, upload,
    // launch and read as usual
    const hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { ctx->capture_pool.push_back(s); return wrk_fail(ctx, WRK_E_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
    ctx->sessions[std::this_thread::get_id()] = s;
    return WRK_OK;
}

int32_t wrk_capture_end(wrk_ctx* ctx, wrk_program** out) {
    if (!ctx || !out) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    auto it = ctx->sessions.find(std::this_thread::get_id());
    WRK_ARG(ctx, it != ctx->sessions.end(), "no capture in progress on this thread");
    hipStream_t s = it->second;
    ctx->sessions.erase(it);
    ctx->capture_pool.push_back(s);
    hipGraph_t g = nullptr;
    WRK_HIP(ctx, hipStreamEndCapture(s, &g));
    hipGraphExec_t e = nullptr;
    hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    if (err != hipSuccess) { hipGraphDestroy(g); return wrk_fail(ctx, WRK_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(err)); }
    wrk_program* p = new wrk_program();
    p->graph = g;
    p->exec = e;
    *out = p;
    return WRK_OK;
}

int32_t wrk_program_launch(wrk_ctx* ctx, wrk_program* prog) {
    if (!ctx || !prog) return WRK_E_ARG;
    LOCK(ctx);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    WRK_HIP(ctx, hipGraphLaunch(prog->exec, ctx->stream));
    return WRK_OK;
}

int32_t wrk_program_destroy(wrk_program* prog) {
    if (!prog) return WRK_E_ARG;
    if (prog->exec) hipGraphExecDestroy(prog->exec);
    if (prog->graph) hipGraphDestroy(prog->graph);
    delete prog;
    return WRK_OK;
}

// ---------------------------------------------------------------- matrices
static inline uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;       // round-to-nearest-even, like half::f16::from_f32
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}

// Float4Quant::default (matrix.rs:50-67)
static const float NF4_LEVELS[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                     -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                     0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                     0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

static int32_t upload_levels(wrk_ctx* ctx, wrk_matrix* mt, const float* levels) {
    void* q = nullptr;
    WRK_HIP(ctx, hipMalloc(&q, 64));
    hipError_t e = hipMemcpyAsync(q, levels ? levels : NF4_LEVELS, 64, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(q); return wrk_fail(ctx, WRK_E_HIP, "fp4 level upload: %s", hipGetErrorString(e)); }
    mt->aux = (uint8_t*)q;
    mt->aux_bytes = 64;
    return WRK_OK;
}

int32_t wrk_matrix_create(wrk_ctx* ctx, uint32_t kind, uint32_t k, uint32_t m, const void* data, size_t bytes, uint32_t flags,
                          wrk_matrix** out) {
    if (!ctx || !out || !data) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_ARG(ctx, k > 0 && m > 0, "matrix dims must be positive");
    uint32_t dev_kind = kind;
    size_t expect = 0;
    const float* levels = nullptr;
    switch (kind) {
        case WRK_MAT_F32: expect = (size_t)k * m * 4; dev_kind = WRK_MAT_F16; break;
        case WRK_MAT_F16: expect = (size_t)k * m * 2; break;
        case WRK_MAT_Q8_0:
            WRK_ARG(ctx, k % 32 == 0, "Q8_0 needs K %% 32 == 0 (loader.rs:884-887)");
            expect = (size_t)k / 32 * 34 * m;
            break;
        case WRK_MAT_Q4_K:
        case WRK_MAT_Q5_K:
        case WRK_MAT_Q6_K:
            WRK_ARG(ctx, k % 256 == 0, "K-quants need K %% 256 == 0 (loader.rs:824-827)");
            expect = wrk::stored_bytes(kind, k, m);
            break;
        case WRK_MAT_INT8:
            WRK_ARG(ctx, k >= 128 && k % 16 == 0 && ((size_t)k * m) % 128 == 0, "Int8 matrices need K %% 16 == 0 and K*M %% 128 == 0 (INT8_BLOCK_SIZE, ops.rs:36)");
            expect = wrk::stored_bytes(kind, k, m);
            break;
        case WRK_MAT_NF4:
            WRK_ARG(ctx, k % 64 == 0, "NF4 matrices need K %% 64 == 0 (NF4_BLOCK_SIZE, ops.rs)");
            expect = wrk::stored_bytes(kind, k, m);
            if (bytes == expect + 64) { levels = (const float*)((const uint8_t*)data + expect); bytes = expect; }
            break;
        default: return wrk_fail(ctx, WRK_E_UNSUPPORTED, "matrix kind %u not supported by wrk_matrix_create", kind);
    }
    WRK_ARG(ctx, bytes == expect, "matrix data is %zu bytes, expected %zu for kind %u [%u x %u]", bytes, expect, kind, k, m);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = wrk::repack_row_bytes(dev_kind, k);
    const size_t total = rb * m;
    std::vector<uint8_t> host(total, 0);
    if (kind == WRK_MAT_F32) {
        std::vector<uint16_t> tmp((size_t)k * m);
        const float* f = (const float*)data;
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)tmp.size(); ++i) tmp[i] = f32_to_f16_bits(f[i]);
        wrk::repack_rows(WRK_MAT_F16, k, m, (const uint8_t*)tmp.data(), host.data());
    } else {
        wrk::repack_rows(dev_kind, k, m, (const uint8_t*)data, host.data());
    }
    void* p = nullptr;
    WRK_HIP(ctx, hipMalloc(&p, total + 256));
    hipError_t e = hipMemcpyAsync(p, host.data(), total, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(p); return wrk_fail(ctx, WRK_E_HIP, "matrix upload: %s", hipGetErrorString(e)); }
    wrk_matrix* mt = new wrk_matrix{ctx, dev_kind, k, m, flags, (uint8_t*)p, rb, wrk::stored_bytes(kind, k, m), nullptr, 0, {1}};
    if (kind == WRK_MAT_NF4) {
        const int32_t rc = upload_levels(ctx, mt, levels);
        if (rc != WRK_OK) { hipFree(p); delete mt; return rc; }
    }
    *out = mt;
    return WRK_OK;
}

int32_t wrk_matrix_quantize(wrk_ctx* ctx, uint32_t kind, uint32_t k, uint32_t m, const wrk_buf* f16_data, const float* levels,
                            wrk_matrix** out) {
    if (!ctx || !out || !f16_data) return WRK_E_ARG;
    LOCK(ctx);
    *out = nullptr;
    WRK_ARG(ctx, k > 0 && m > 0, "matrix dims must be positive");
    if (kind != WRK_MAT_INT8 && kind != WRK_MAT_NF4)
        return wrk_fail(ctx, WRK_E_UNSUPPORTED, "wrk_matrix_quantize: kind %u is not Int8 / NF4", kind);
    WRK_ARG(ctx, kind != WRK_MAT_INT8 || (k >= 128 && k % 16 == 0 && ((size_t)k * m) % 128 == 0), "Int8 matrices need K %% 16 == 0 and K*M %% 128 == 0");
    WRK_ARG(ctx, kind != WRK_MAT_NF4 || k % 64 == 0, "NF4 matrices need K %% 64 == 0");
    WRK_ARG(ctx, f16_data->bytes >= (size_t)k * m * 2, "source holds %zu bytes, [%u x %u] f16 needs %zu", f16_data->bytes, k, m, (size_t)k * m * 2);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rb = wrk::repack_row_bytes(kind, k);
    void* p = nullptr;
    WRK_HIP(ctx, hipMalloc(&p, rb * m + 256));
    wrk_matrix* mt = new wrk_matrix{ctx, kind, k, m, 0, (uint8_t*)p, rb, wrk::stored_bytes(kind, k, m), nullptr, 0, {1}};
    hipError_t e = hipMemsetAsync(p, 0, rb * m + 256, ctx->stream);
    if (e == hipSuccess && kind == WRK_MAT_INT8) wrk::quantize_int8(ctx->stream, f16_data->ptr, mt->data, k, m, (uint32_t)rb);
    if (e == hipSuccess && kind == WRK_MAT_NF4) {
        const int32_t rc = upload_levels(ctx, mt, levels);
        if (rc != WRK_OK) { hipFree(p); delete mt; return rc; }
        wrk::quantize_nf4(ctx->stream, f16_data->ptr, (const float*)mt->aux, mt->data, k, m, (uint32_t)rb);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { hipFree(p); if (mt->aux) hipFree(mt->aux); delete mt; return wrk_fail(ctx, WRK_E_HIP, "matrix quantize: %s", hipGetErrorString(e)); }
    *out = mt;
    return WRK_OK;
}

int32_t wrk_matrix_export(wrk_matrix* mat, void* dst, size_t capacity, size_t* bytes) {
    if (!mat || !bytes) return WRK_E_ARG;
    wrk_ctx* ctx = mat->ctx;
    LOCK(ctx);
    if (mat->kind != WRK_MAT_INT8 && mat->kind != WRK_MAT_NF4)
        return wrk_fail(ctx, WRK_E_UNSUPPORTED, "wrk_matrix_export: only Int8 / NF4 matrices keep their source layout");
    const uint32_t k = mat->k, m = mat->m;
    const size_t total = wrk::stored_bytes(mat->kind, k, m) + (mat->kind == WRK_MAT_NF4 ? 64 : 0);
    *bytes = total;
    if (!dst) return WRK_OK;
    WRK_ARG(ctx, capacity >= total, "export needs %zu bytes, capacity %zu", total, capacity);
    WRK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint8_t> host(mat->row_bytes * m);
    WRK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WRK_HIP(ctx, hipMemcpy(host.data(), mat->data, host.size(), hipMemcpyDeviceToHost));
    uint8_t* o = (uint8_t*)dst;
    const size_t code_row = mat->kind == WRK_MAT_INT8 ? k : k / 2;
    for (uint32_t r = 0; r < m; ++r) memcpy(o + (size_t)r * code_row, host.data() + (size_t)r * mat->row_bytes, code_row);
    if (mat->kind == WRK_MAT_INT8) {       // block b lives (at least) in the row holding its first element
        for (size_t b = 0; b < (size_t)k * m / 128; ++b) {
            const size_t r = b * 128 / k;
            memcpy(o + (size_t)m * k + b * 4, host.data() + r * mat->row_bytes + k + (b - r * k / 128) * 4, 4);
        }
    } else {
        const size_t side_row = (size_t)(k / 64) * 2;
        for (uint32_t r = 0; r < m; ++r)
            memcpy(o + (size_t)m * code_row + (size_t)r * side_row, host.data() + (size_t)r * mat->row_bytes + code_row, side_row);
    }
    if (mat->kind == WRK_MAT_NF4) WRK_HIP(ctx, hipMemcpy(o + total - 64, mat->aux, 64, hipMemcpyDeviceToHost));
    return WRK_OK;
}

int32_t wrk_matrix_set_scale(wrk_matrix* mat, float scale) {
    if (!mat || !(scale > 0.0f) || !std::isfinite(scale)) return WRK_E_ARG;
    mat->out_scale = scale;
    return WRK_OK;
}

int32_t wrk_matrix_release(wrk_matrix* mat) {
    if (!mat) return WRK_E_ARG;
    if (mat->refs.fetch_sub(1) == 1) {
        wrk_ctx* ctx = mat->ctx;
        LOCK(ctx);
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
        hipFree(mat->data);
        if (mat->aux) hipFree(mat->aux);
        delete mat;
    }
    return WRK_OK;
}

size_t wrk_matrix_stream_bytes(const wrk_matrix* mat) { return mat ? mat->stored_bytes : 0; }

// ---------------------------------------------------------------- op helpers
static int32_t check_tensor(wrk_ctx* ctx, const wrk_tensor* t, const char* name) {
    WRK_ARG(ctx, t && t->buf, "%s: null tensor", name);
    WRK_ARG(ctx, t->dtype == WRK_F16 || t->dtype == WRK_F32, "%s: dtype must be f16 or f32", name);
    const DTensor d = make_dtensor(t);
    for (int i = 0; i < 3; ++i)
        WRK_ARG(ctx, (size_t)d.offset[i] + d.shape[i] <= (size_t)(d.stride[i] ? d.stride[i] : 1) || d.shape[i] == 0,
                "%s: view [%u..+%u) exceeds parent extent %u on axis %d", name, d.offset[i], d.shape[i], d.stride[i], i);
    if (d.shape[0] && d.shape[1] && d.shape[2] && d.shape[3])
        WRK_ARG(ctx, dtensor_extent(d) * dtype_size(t->dtype) <= t->buf->bytes, "%s: view exceeds its %zu-byte buffer", name, t->buf->bytes);
    return WRK_OK;
}
#define CHECK_T(t, name)                                       \
    do {                                                       \
        int32_t _r = check_tensor(ctx, (t), (name));           \
        if (_r != WRK_OK) return _r;                           \
    } while (0)
#define SAME_SHAPE(a, b) ((a)->view.shape[0] == (b)->view.shape[0] && (a)->view.shape[1] == (b)->view.shape[1] && (a)->view.shape[2] == (b)->view.shape[2])
#define ENTER(ctx)                              \
    if (!(ctx)) return WRK_E_ARG;               \
    LOCK(ctx);                                  \
    WRK_HIP(ctx, hipSetDevice((ctx)->device))

int32_t wrk_op_matmul(wrk_ctx* ctx, const wrk_matrix* mat, const wrk_tensor* input, const wrk_tensor* output, uint32_t act,
                      int32_t turbo, int32_t sparse) {
    ENTER(ctx);
    WRK_ARG(ctx, mat, "matmul: null matrix");
    CHECK_T(input, "matmul input");
    CHECK_T(output, "matmul output");
    WRK_ARG(ctx, input->view.shape[0] == mat->k, "matmul: input has %u channels, matrix K = %u", input->view.shape[0], mat->k);
    WRK_ARG(ctx, output->view.shape[0] == mat->m, "matmul: output has %u channels, matrix M = %u", output->view.shape[0], mat->m);
    WRK_ARG(ctx, input->view.shape[1] == output->view.shape[1] && input->view.shape[2] == output->view.shape[2], "matmul: token/batch mismatch");
    wrk::MatJob j{mat->data, mat->aux, mat->kind, mat->flags, mat->k, mat->m, (uint32_t)mat->row_bytes,
                  make_dtensor(input), make_dtensor(output), act, (uint32_t)sparse};
    j.scale = mat->out_scale;
    int rc = -2;
    const size_t ntok = (size_t)input->view.shape[1] * input->view.shape[2];
    if (turbo && ntok >= 128 && (mat->kind == WRK_MAT_Q4_K || mat->kind == WRK_MAT_Q5_K)) {                             // the third-generation prefill tile wants its sum scratch
        // (+ the f32 partial tiles of a K-split launch: chunks of up to 256 tokens, at most one slice per 256-block)
        const size_t part = ntok <= 256 ? ntok * (size_t)mat->m * (mat->k >> 8) * 4 : 0;
        const int32_t rs = wrk_ctx_reserve_gemm_scratch(ctx, ntok * (mat->k >> 5) * 4 + 1024 + part);
        if (rs != WRK_OK) return rs;
    }
    j.xsum = ctx->gemm_scratch; j.xsum_cap = ctx->gemm_scratch_cap;
    if (turbo && ntok >= 2) rc = wrk::matmul_mfma(ctx->op_stream(), j, ctx->num_cu);     // tiles are padded to 16 tokens
    if (rc == -2) rc = wrk::matvec(ctx->op_stream(), &j, 1, ctx->num_cu);
    WRK_ARG(ctx, rc == 0, "matmul: launch configuration rejected");
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_layer_norm(wrk_ctx* ctx, const wrk_buf* w, const wrk_buf* b, const wrk_tensor* x, float eps) {
    ENTER(ctx);
    CHECK_T(x, "layer_norm x");
    WRK_ARG(ctx, w && b && w->bytes >= (size_t)x->view.shape[0] * 2 && b->bytes >= (size_t)x->view.shape[0] * 2, "layer_norm: w/b must hold C f16");
    wrk::layer_norm(ctx->op_stream(), w->ptr, b->ptr, make_dtensor(x), eps);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_group_norm(wrk_ctx* ctx, const wrk_buf* w, const wrk_buf* b, const wrk_tensor* x, float eps) {
    ENTER(ctx);
    CHECK_T(x, "group_norm x");
    const size_t need = (size_t)x->view.shape[0] * x->view.shape[1] * 2;
    WRK_ARG(ctx, w && b && w->bytes >= need && b->bytes >= need, "group_norm: w/b must hold S*H f16");
    WRK_ARG(ctx, x->view.shape[0] <= 64 * 1024, "group_norm: head size too large");
    wrk::group_norm(ctx->op_stream(), w->ptr, b->ptr, make_dtensor(x), eps);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_l2_norm(wrk_ctx* ctx, const wrk_tensor* x, float eps) {
    ENTER(ctx);
    CHECK_T(x, "l2_norm x");
    wrk::l2_norm(ctx->op_stream(), make_dtensor(x), eps);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_token_shift(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* time_mix, const wrk_tensor* state,
                           const wrk_tensor* input, const wrk_tensor* output, int32_t reversed) {
    ENTER(ctx);
    CHECK_T(time_mix, "token_shift time_mix");
    CHECK_T(state, "token_shift state");
    CHECK_T(input, "token_shift input");
    CHECK_T(output, "token_shift output");
    WRK_ARG(ctx, input->view.shape[0] == output->view.shape[0] && input->view.shape[1] == output->view.shape[1], "token_shift: input/output shape mismatch");
    WRK_ARG(ctx, input->view.shape[2] == 1, "token_shift: input must be [C, T, 1]");
    WRK_ARG(ctx, time_mix->view.shape[0] == input->view.shape[0] && (time_mix->view.shape[1] == 1 || time_mix->view.shape[1] == input->view.shape[1]) &&
                     time_mix->view.shape[2] == output->view.shape[2], "token_shift: time_mix must be [C, 1 or T, I] with I = output.shape[2]");
    WRK_ARG(ctx, state->view.shape[0] == input->view.shape[0] && state->view.shape[1] == 1, "token_shift: state must be [C, 1, B]");
    WRK_ARG(ctx, cursors && cursors->bytes >= (size_t)input->view.shape[1] * 4, "token_shift: cursors must hold T u32");
    wrk::token_shift(ctx->op_stream(), (const uint32_t*)cursors->ptr, make_dtensor(time_mix), make_dtensor(state), make_dtensor(input), make_dtensor(output), reversed);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_transpose(wrk_ctx* ctx, const wrk_tensor* input, const wrk_tensor* output) {
    ENTER(ctx);
    CHECK_T(input, "transpose input");
    CHECK_T(output, "transpose output");
    WRK_ARG(ctx, input->view.shape[0] == output->view.shape[0] && input->view.shape[1] == output->view.shape[2] && input->view.shape[2] == output->view.shape[1],
            "transpose: output must be [C, B, T] for input [C, T, B]");
    wrk::transpose(ctx->op_stream(), make_dtensor(input), make_dtensor(output));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_time_mix_v6(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* time_decay, const wrk_buf* time_first, const wrk_tensor* state,
                           const wrk_tensor* k, const wrk_tensor* v, const wrk_tensor* r, const wrk_tensor* x) {
    ENTER(ctx);
    CHECK_T(time_decay, "time_mix_v6 time_decay");
    CHECK_T(state, "time_mix_v6 state");
    CHECK_T(k, "time_mix_v6 k");
    CHECK_T(v, "time_mix_v6 v");
    CHECK_T(r, "time_mix_v6 r");
    CHECK_T(x, "time_mix_v6 x");
    const uint32_t S = r->view.shape[0], H = r->view.shape[1], T = r->view.shape[2];
    WRK_ARG(ctx, S == 64, "time_mix_v6: head size %u unsupported (64 only)", S);
    WRK_ARG(ctx, SAME_SHAPE(r, k) && SAME_SHAPE(r, v) && SAME_SHAPE(r, x) && SAME_SHAPE(r, time_decay), "time_mix_v6: shape mismatch");
    WRK_ARG(ctx, state->view.shape[0] == S * H && state->view.shape[1] == S + 1 && state->dtype == WRK_F32, "time_mix_v6: state must be f32 [C, S+1, B]");
    WRK_ARG(ctx, time_first && time_first->bytes >= (size_t)S * H * 4, "time_mix_v6: time_first must hold S*H f32");
    WRK_ARG(ctx, cursors && cursors->bytes >= (size_t)T * 4, "time_mix_v6: cursors must hold T u32");
    wrk::time_mix_v6(ctx->op_stream(), (const uint32_t*)cursors->ptr, make_dtensor(time_decay), time_first->ptr, make_dtensor(state), make_dtensor(k),
                     make_dtensor(v), make_dtensor(r), make_dtensor(x));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_channel_mix(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* state, const wrk_tensor* r, const wrk_tensor* v, const wrk_tensor* x) {
    ENTER(ctx);
    CHECK_T(state, "channel_mix state");
    CHECK_T(r, "channel_mix r");
    CHECK_T(v, "channel_mix v");
    CHECK_T(x, "channel_mix x");
    WRK_ARG(ctx, SAME_SHAPE(v, x) && SAME_SHAPE(r, x), "channel_mix: r/v/x shape mismatch");
    WRK_ARG(ctx, state->view.shape[0] == x->view.shape[0] && state->view.shape[1] == 1, "channel_mix: state must be [C, 1, B]");
    WRK_ARG(ctx, cursors && cursors->bytes >= (size_t)x->view.shape[1] * 4, "channel_mix: cursors must hold T u32");
    wrk::channel_mix_v6(ctx->op_stream(), (const uint32_t*)cursors->ptr, make_dtensor(state), make_dtensor(r), make_dtensor(v), make_dtensor(x));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

static int32_t binary_op(wrk_ctx* ctx, int is_mul, const wrk_tensor* input, const wrk_tensor* output, uint32_t ax, uint32_t ay, uint32_t ao) {
    ENTER(ctx);
    CHECK_T(input, "binary input");
    CHECK_T(output, "binary output");
    WRK_ARG(ctx, input->view.shape[0] == output->view.shape[0], "binary: channel mismatch");
    WRK_ARG(ctx, input->view.shape[1] == 1 || input->view.shape[1] == output->view.shape[1], "binary: token extent must be 1 or equal");
    WRK_ARG(ctx, input->view.shape[2] == 1 || input->view.shape[2] == output->view.shape[2], "binary: batch extent must be 1 or equal");
    wrk::binary(ctx->op_stream(), is_mul, make_dtensor(input), make_dtensor(output), ax, ay, ao);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}
int32_t wrk_op_add(wrk_ctx* ctx, const wrk_tensor* i, const wrk_tensor* o, uint32_t ax, uint32_t ay, uint32_t ao) { return binary_op(ctx, 0, i, o, ax, ay, ao); }
int32_t wrk_op_mul(wrk_ctx* ctx, const wrk_tensor* i, const wrk_tensor* o, uint32_t ax, uint32_t ay, uint32_t ao) { return binary_op(ctx, 1, i, o, ax, ay, ao); }

int32_t wrk_op_lerp(wrk_ctx* ctx, const wrk_tensor* x, const wrk_tensor* y, const wrk_tensor* f, int32_t reversed) {
    ENTER(ctx);
    CHECK_T(x, "lerp x");
    CHECK_T(y, "lerp y");
    CHECK_T(f, "lerp f");
    WRK_ARG(ctx, SAME_SHAPE(x, y), "lerp: x/y shape mismatch");
    WRK_ARG(ctx, f->view.shape[0] == y->view.shape[0], "lerp: factor channel mismatch");
    wrk::lerp(ctx->op_stream(), make_dtensor(x), make_dtensor(y), make_dtensor(f), reversed);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_blit(wrk_ctx* ctx, const wrk_tensor* input, const wrk_tensor* output) {
    ENTER(ctx);
    CHECK_T(input, "blit input");
    CHECK_T(output, "blit output");
    WRK_ARG(ctx, SAME_SHAPE(input, output), "blit: shape mismatch");
    wrk::blit(ctx->op_stream(), make_dtensor(input), make_dtensor(output));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_affine(wrk_ctx* ctx, const wrk_tensor* x, float scale, float bias) {
    ENTER(ctx);
    CHECK_T(x, "affine x");
    wrk::affine(ctx->op_stream(), make_dtensor(x), scale, bias);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_activate(wrk_ctx* ctx, const wrk_tensor* x, uint32_t act) {
    ENTER(ctx);
    CHECK_T(x, "activate x");
    wrk::activate(ctx->op_stream(), make_dtensor(x), act);
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_control_k_v7(wrk_ctx* ctx, const wrk_buf* p, const wrk_tensor* a, const wrk_tensor* k) {
    ENTER(ctx);
    CHECK_T(a, "control_k a");
    CHECK_T(k, "control_k k");
    WRK_ARG(ctx, SAME_SHAPE(a, k), "control_k: a/k shape mismatch");
    WRK_ARG(ctx, p && p->bytes >= (size_t)k->view.shape[0] * 2, "control_k: p must hold C f16");
    wrk::control_k_v7(ctx->op_stream(), p->ptr, make_dtensor(a), make_dtensor(k));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_time_mix_v7(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* state, const wrk_tensor* r, const wrk_tensor* w,
                           const wrk_tensor* n, const wrk_tensor* x) {
    ENTER(ctx);
    CHECK_T(state, "time_mix state");
    CHECK_T(r, "time_mix r");
    CHECK_T(w, "time_mix w");
    CHECK_T(n, "time_mix n");
    CHECK_T(x, "time_mix x");
    const uint32_t S = r->view.shape[0], H = r->view.shape[1], T = r->view.shape[2];
    WRK_ARG(ctx, S == 64, "time_mix_v7: head size %u unsupported (64 only, as every RWKV-7 World model)", S);
    WRK_ARG(ctx, SAME_SHAPE(r, w) && SAME_SHAPE(r, x), "time_mix_v7: r/w/x shape mismatch");
    WRK_ARG(ctx, n->view.shape[0] == S && n->view.shape[1] == H && n->view.shape[2] == T && n->view.shape[3] == 4, "time_mix_v7: n must be [S, H, T, 4]");
    WRK_ARG(ctx, state->view.shape[0] == S * H && state->view.shape[1] == S + 1, "time_mix_v7: state must be [C, S+1, B]");
    WRK_ARG(ctx, state->dtype == WRK_F32, "time_mix_v7: state must be f32");
    WRK_ARG(ctx, cursors && cursors->bytes >= (size_t)T * 4, "time_mix_v7: cursors must hold T u32");
    wrk::time_mix_v7(ctx->op_stream(), (const uint32_t*)cursors->ptr, make_dtensor(state), make_dtensor(r), make_dtensor(w), make_dtensor(n), make_dtensor(x));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_time_first_v7(wrk_ctx* ctx, const wrk_buf* u, const wrk_tensor* r, const wrk_tensor* n, const wrk_tensor* x) {
    ENTER(ctx);
    CHECK_T(r, "time_first r");
    CHECK_T(n, "time_first n");
    CHECK_T(x, "time_first x");
    WRK_ARG(ctx, r->view.shape[0] == 64, "time_first_v7: head size must be 64");
    WRK_ARG(ctx, SAME_SHAPE(r, x), "time_first_v7: r/x shape mismatch");
    WRK_ARG(ctx, u && u->bytes >= (size_t)r->view.shape[0] * r->view.shape[1] * 2, "time_first_v7: u must hold S*H f16");
    wrk::time_first_v7(ctx->op_stream(), u->ptr, make_dtensor(r), make_dtensor(n), make_dtensor(x));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_channel_mix_v7(wrk_ctx* ctx, const wrk_buf* cursors, const wrk_tensor* state, const wrk_tensor* v, const wrk_tensor* x) {
    ENTER(ctx);
    CHECK_T(state, "channel_mix state");
    CHECK_T(v, "channel_mix v");
    CHECK_T(x, "channel_mix x");
    WRK_ARG(ctx, SAME_SHAPE(v, x), "channel_mix: v/x shape mismatch");
    WRK_ARG(ctx, state->view.shape[0] == x->view.shape[0] && state->view.shape[1] == 1, "channel_mix: state must be [C, 1, B]");
    WRK_ARG(ctx, cursors && cursors->bytes >= (size_t)x->view.shape[1] * 4, "channel_mix: cursors must hold T u32");
    wrk::channel_mix_v7(ctx->op_stream(), (const uint32_t*)cursors->ptr, make_dtensor(state), make_dtensor(v), make_dtensor(x));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

int32_t wrk_op_softmax(wrk_ctx* ctx, const wrk_tensor* x) {
    ENTER(ctx);
    CHECK_T(x, "softmax x");
    wrk::softmax(ctx->op_stream(), make_dtensor(x));
    WRK_LAUNCH_CHECK(ctx);
    return WRK_OK;
}

}  // extern "C"
