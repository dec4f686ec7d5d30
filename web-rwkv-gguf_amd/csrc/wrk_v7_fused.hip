// Fused RWKV-7 decode step (every sequence contributes exactly one token).
#include "wrk_device.h"
#include "wrk_v7.h"

namespace wrk {

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

}  // namespace wrk

void wrk_v7_model::drop_graphs() {
    for (auto& kv : graphs) wrk_program_destroy(kv.second);
    graphs.clear();
}

void wrk_v7_model::free_fused() {}

int32_t wrk_v7_model::enqueue_fused_decode(wrk_v7_state* st, uint32_t B, uint32_t NH, bool identity_headers) {
    return enqueue_ops(st, B, NH, identity_headers);
}
