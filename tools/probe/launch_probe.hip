// Launch-latency probe: how long does one link of a dependent kernel chain cost on this box, as a function of
// kernarg size, grid size and a "last workgroup does the tail" epilogue.  Replayed from a hipGraph like the decode loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big { unsigned v[440]; };   // 1760 bytes, like MatvecParams

__global__ void k_small(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void k_big(Big b, float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += (float)b.v[blockIdx.x & 255]; }
__global__ void k_big_all(Big b, float* p) { if (threadIdx.x == 0) p[blockIdx.x] += (float)b.v[blockIdx.x & 255]; }
// tail: every WG bumps a counter; the last one does a small serial job (stand-in for LN over 2048 values)
__global__ void k_tail(float* p, unsigned* counter, unsigned n) {
    __shared__ unsigned last;
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
    __threadfence();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u);
    __syncthreads();
    if (last == n - 1) {
        float s = 0.0f;
        for (unsigned i = threadIdx.x; i < 2048; i += blockDim.x) s += __builtin_nontemporal_load(p + i);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x == 0) { p[4096] = s; *counter = 0; }
    }
}

template <class F>
static float time_graph(hipStream_t s, int chain, int reps, F enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; ++i) enqueue();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1000.0f / (reps * chain);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    unsigned* cnt; CK(hipMalloc(&cnt, 64)); CK(hipMemset(cnt, 0, 64));
    Big b; for (int i = 0; i < 440; ++i) b.v[i] = i & 1;
    const char* env = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG=%s\n", env ? env : "(unset)");
    const int chain = 170, reps = 50;
    for (int grid : {1, 256, 1024, 4096}) {
        float t0 = time_graph(s, chain, reps, [&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, p); });
        float t1 = time_graph(s, chain, reps, [&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, s, b, p); });
        float t2 = time_graph(s, chain, reps, [&] { hipLaunchKernelGGL(k_big_all, dim3(grid), dim3(256), 0, s, b, p); });
        float t3 = time_graph(s, chain, reps, [&] { hipLaunchKernelGGL(k_tail, dim3(grid), dim3(256), 0, s, p, cnt, (unsigned)grid); });
        printf("grid %5d: small-arg %.2f us | 1760B-arg (1 reader) %.2f us | 1760B-arg (all WGs read) %.2f us | last-WG tail %.2f us\n", grid, t0, t1, t2, t3);
    }
    return 0;
}
