// What does one all-to-all dependency cost on this box, by mechanism?  ("why launches").
//
// The decode layer is a chain of all-to-all steps: every workgroup produces a few elements of a 4 KB vector and every
// workgroup of the next step needs all of it.  This probe times that pattern three ways, one workgroup per CU (256 x 256 threads):
//   (a) kernel chain      : one launch per step, replayed from a hipGraph (what the decode path does);
//   (b) flat grid barrier : ONE persistent launch; per step: store slice -> release fence -> add to one counter -> poll -> acquire fence;
//   (c) XCD-hierarchical  : the same with a counter per XCC (census of workgroups per XCC first), the last arriver of an XCC adds to a
//                           top counter, the last XCC publishes the generation; everybody polls the generation word.
// Every spin is bounded (a stuck barrier sets a flag and the kernel exits), so a residency surprise cannot hang the box.
// Each step every workgroup re-reads the whole vector (plain loads behind the acquire) and checks it, so a stale read is counted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, VEC = 1024;            // 1024 floats = 4 KB, 4 per workgroup
constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }      // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned target, unsigned* fail) {
    for (unsigned s = 0; s < SPIN_MAX; ++s) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    *fail = 1;
    return false;
}

// (a) one step as its own kernel: read everything (checked), write my slice of the next vector
__global__ void __launch_bounds__(256) step_kernel(const float* in, float* out, unsigned step, unsigned* stale) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < VEC; i += 256) s += in[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    if (threadIdx.x == 0 && tot != (float)step * VEC) atomicAdd(stale, 1u);
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = (float)(step + 1);
}

// (b), (c): persistent kernel with `steps` barriers
template <bool HIER>
__global__ void __launch_bounds__(256) persistent_kernel(float* va, float* vb, unsigned steps, unsigned* flat, unsigned* xcnt, unsigned* top,
                                                          unsigned* gen, unsigned* census, unsigned* stale, unsigned* fail) {
    __shared__ float red[4];
    __shared__ unsigned ok;
    const unsigned x = xcc_id();
    if (HIER) {     // census: how many workgroups live on my XCC (placement is not ours to assume)
        if (threadIdx.x == 0) {
            atomicAdd(&census[x], 1u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(flat, 1u);
            ok = wait_ge(flat, NWG, fail);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!ok) return;
    }
    const unsigned mine = HIER ? __hip_atomic_load(&census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    unsigned nx = 0;
    if (HIER) for (int i = 0; i < 8; ++i) nx += __hip_atomic_load(&census[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    for (unsigned step = 0; step < steps; ++step) {
        const float* in = (step & 1) ? vb : va;
        float* out = (step & 1) ? va : vb;
        float s = 0.0f;
        for (int i = threadIdx.x; i < VEC; i += 256) s += in[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        if (threadIdx.x == 0 && tot != (float)step * VEC) atomicAdd(stale, 1u);
        if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = (float)(step + 1);
        __syncthreads();                                            // this workgroup's stores are issued
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // ... and written back before the arrival
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!HIER) {
                atomicAdd(flat, 1u);
                ok = wait_ge(flat, (unsigned)NWG * (step + 1 + (HIER ? 1 : 0)), fail);
            } else {
                const unsigned a = atomicAdd(&xcnt[x * 32], 1u);      // counters on lines of their own
                if (a + 1 == mine * (step + 1)) {                     // last arriver of this XCC
                    const unsigned t = atomicAdd(top, 1u);
                    if (t + 1 == nx * (step + 1)) __hip_atomic_store(gen, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // last XCC: release everybody
                }
                ok = wait_ge(gen, step + 1, fail);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!ok) return;
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *va, *vb; CK(hipMalloc(&va, VEC * 4)); CK(hipMalloc(&vb, VEC * 4));
    unsigned* w; CK(hipMalloc(&w, 4096));       // flat | xcnt[8 * 32] | top | gen | census[8] | stale | fail
    unsigned *flat = w, *xcnt = w + 32, *top = w + 32 + 256, *gen = top + 32, *census = gen + 32, *stale = census + 32, *fail = stale + 32;
    const unsigned steps = 200;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;

    // (a) kernel chain from a graph
    CK(hipMemset(va, 0, VEC * 4)); CK(hipMemset(w, 0, 4096));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (unsigned i = 0; i < steps; ++i) hipLaunchKernelGGL(step_kernel, dim3(NWG), dim3(256), 0, s, (i & 1) ? vb : va, (i & 1) ? va : vb, i, stale);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(va, 0, VEC * 4, s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    unsigned hs[2]; CK(hipMemcpy(hs, stale, 4, hipMemcpyDeviceToHost));
    printf("(a) kernel chain (hipGraph), 256 WGs, 4 KB all-to-all per step : %.2f us per step   (stale reads %u)\n", best * 1000.0f / steps, hs[0]);

    // (b), (c) persistent
    for (int hier = 0; hier < 2; ++hier) {
        best = 1e9f;
        unsigned st = 0, fl = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(va, 0, VEC * 4, s)); CK(hipMemsetAsync(w, 0, 4096, s));
            CK(hipEventRecord(a, s));
            if (hier) hipLaunchKernelGGL(persistent_kernel<true>, dim3(NWG), dim3(256), 0, s, va, vb, steps, flat, xcnt, top, gen, census, stale, fail);
            else hipLaunchKernelGGL(persistent_kernel<false>, dim3(NWG), dim3(256), 0, s, va, vb, steps, flat, xcnt, top, gen, census, stale, fail);
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
            CK(hipMemcpy(&st, stale, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fl, fail, 4, hipMemcpyDeviceToHost));
        }
        unsigned hc[8]; CK(hipMemcpy(hc, census, 32, hipMemcpyDeviceToHost));
        printf("(%c) persistent launch, %s : %.2f us per step   (stale reads %u, spin give-ups %u", hier ? 'c' : 'b',
               hier ? "XCD-hierarchical barrier " : "flat counter barrier     ", best * 1000.0f / steps, st, fl);
        if (hier) { printf(", workgroups per XCC"); for (int i = 0; i < 8; ++i) printf(" %u", hc[i]); }
        printf(")\n");
    }
    return 0;
}
