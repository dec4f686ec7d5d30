// What is the 3 us "cold" first-touch latency made of?  Reads 24 KB with one workgroup (latency) and 8 MB with 1024 (ramp)
// after streaming F bytes through the chip, for F = 320 MB (beyond L2 + Infinity Cache), 1 GB, 4 GB, with the target either
// in its own 8 MB allocation or in the middle of one 6 GB arena that also holds the flush buffer (same translation
// fragments as its neighbours).  If the latency grows with F beyond the cache sizes, translation misses are part of it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void reader(const u32x4* __restrict__ p, size_t nvec, unsigned* sink, unsigned long long* stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}
__global__ void stream_read(const u32x4* __restrict__ p, size_t nvec, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = p[i]; acc += v.x ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t GB = (size_t)1 << 30;
    unsigned char* arena; u32x4* own; unsigned* sink; unsigned long long* st;
    CK(hipMalloc(&arena, 6 * GB)); CK(hipMalloc(&own, 8 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&st, 2 * 2048 * 8));
    CK(hipMemset(arena, 1, 6 * GB)); CK(hipMemset(own, 1, 8 << 20));
    u32x4* flush = (u32x4*)arena;                         // first 4 GB
    u32x4* inside = (u32x4*)(arena + 5 * GB);             // 8 MB in the arena's tail
    std::vector<unsigned long long> h(2 * 2048);
    const size_t flushes[] = {320u << 20, GB, 4 * GB};
    for (int where = 0; where < 2; ++where)
        for (size_t F : flushes)
            for (int big = 0; big < 2; ++big) {
                const u32x4* x = where ? inside : own;
                const size_t bytes = big ? (8 << 20) : (24 << 10);
                const int grid = big ? 1024 : 1;
                std::vector<double> first, span;
                for (int rep = 0; rep < 9; ++rep) {
                    stream_read<<<2048, 256, 0, s>>>(flush, F / 16, sink);
                    reader<<<1, 64, 0, s>>>((const u32x4*)sink, 1, sink, st);        // keeps the reader's code warm: the target's DATA is what is cold
                    reader<<<grid, 256, 0, s>>>(x, bytes / 16, sink, st);
                    CK(hipStreamSynchronize(s));
                    CK(hipMemcpy(h.data(), st, 2 * grid * 8, hipMemcpyDeviceToHost));
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int b = 0; b < grid; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); }
                    first.push_back((double)(h[1] - h[0]) * 10.0);
                    span.push_back((double)(hi - lo) * 10.0);
                }
                std::sort(first.begin(), first.end()); std::sort(span.begin(), span.end());
                printf("%-22s flush %4zu MB   %-22s workgroup 0: %6.0f ns   first start -> last end: %6.0f ns\n", where ? "inside the 6 GB arena" : "own 8 MB allocation",
                       F >> 20, big ? "8 MB, 1024 workgroups" : "24 KB, 1 workgroup", first[first.size() / 2], span[span.size() / 2]);
            }
    return 0;
}
