// Does data read (or written) by one kernel stay close to the chip for the NEXT kernel?  Times a reader kernel from inside
// (wall_clock64, 100 MHz) for three histories of the same buffer: cold (1 GB streamed since the last touch), touched by
// the previous kernel of the stream (read), and written by the previous kernel.  Sizes: 24 KB read by ONE workgroup (the
// LN-prologue operand set: latency) and 8 MB read by 1024 workgroups (one layer kernel's weights: ramp + bandwidth).
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
    if (acc == 0x12345678u) sink[0] = acc;          // keeps the loads alive
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}
// same bytes, but block b touches the slices block b + 1 touches in `reader`: under the round-robin workgroup -> XCD placement
// every slice is then fetched by a DIFFERENT XCD than the one that reads it next (an L2 hit becomes impossible; a hit in the
// memory-side Infinity Cache does not care)
__global__ void reader_shifted(const u32x4* __restrict__ p, size_t nvec, unsigned* sink) {
    unsigned acc = 0;
    const unsigned vb = (blockIdx.x + 1) % gridDim.x;
    for (size_t i = vb * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void xcc_of_blocks(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // HW_REG_XCC_ID[3:0]
}
__global__ void writer(u32x4* p, size_t nvec, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) p[i] = (u32x4){seed, (unsigned)i, 1u, 2u};
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t flush_bytes = (size_t)1 << 30;
    u32x4 *flush, *x; unsigned* sink; unsigned long long* st;
    CK(hipMalloc(&flush, flush_bytes)); CK(hipMalloc(&x, 8 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&st, 2 * 2048 * 8));
    CK(hipMemset(flush, 1, flush_bytes)); CK(hipMemset(x, 1, 8 << 20));
    std::vector<unsigned long long> h(2 * 2048);
    struct Case { const char* name; size_t bytes; int grid; };
    const Case cases[] = {{"24 KB, 1 workgroup", 24 << 10, 1}, {"2 MB, 256 workgroups", 2 << 20, 256}, {"8 MB, 1024 workgroups", 8 << 20, 1024}};
    for (const Case& c : cases) {
        for (int hist = 0; hist < 4; ++hist) {
            std::vector<double> first, span;
            for (int rep = 0; rep < 15; ++rep) {
                const size_t nvec = c.bytes / 16;
                writer<<<1024, 256, 0, s>>>(flush, flush_bytes / 16, rep);                  // evict everything
                if (hist == 1) reader<<<c.grid, 256, 0, s>>>(x, nvec, sink, st);           // previous kernel READ the buffer
                if (hist == 2) writer<<<c.grid, 256, 0, s>>>(x, nvec, rep);                // previous kernel WROTE the buffer
                if (hist == 3) reader_shifted<<<c.grid, 256, 0, s>>>(x, nvec, sink);       // previous kernel read it from other XCDs
                reader<<<c.grid, 256, 0, s>>>(x, nvec, sink, st);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h.data(), st, 2 * c.grid * 8, hipMemcpyDeviceToHost));
                unsigned long long lo = ~0ull, hi = 0;
                for (int b = 0; b < c.grid; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); }
                first.push_back((double)(h[1] - h[0]) * 10.0);
                span.push_back((double)(hi - lo) * 10.0);
            }
            std::sort(first.begin(), first.end()); std::sort(span.begin(), span.end());
            const char* hn[] = {"cold (1 GB streamed since)", "read by the previous kernel", "written by the previous kernel", "read by other XCDs just before"};
            printf("%-24s %-32s workgroup 0: %6.0f ns   first start -> last end: %6.0f ns\n", c.name, hn[hist], first[first.size() / 2], span[span.size() / 2]);
        }
    }
    // workgroup -> XCD placement over a sequence of launches of different sizes
    unsigned* xo; CK(hipMalloc(&xo, 2048 * 4));
    const int grids[] = {1024, 1024, 32, 1024, 100, 1024, 1024, 7, 1024};
    std::vector<unsigned> hx(2048);
    for (int g : grids) {
        xcc_of_blocks<<<g, 64, 0, s>>>(xo);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hx.data(), xo, g * 4, hipMemcpyDeviceToHost));
        int rr = 1;
        for (int b = 1; b < g; ++b) rr &= (hx[b] == (hx[0] + b) % 8);
        printf("grid %4d: XCC of blocks 0..9 =", g);
        for (int b = 0; b < 10 && b < g; ++b) printf(" %u", hx[b]);
        printf("   strict round-robin from block 0: %s\n", rr ? "yes" : "NO");
    }
    return 0;
}
