// Round 3: what does one all-to-all dependency cost INSIDE a persistent launch when nothing is fenced?
// (variant (d) of tools/probe/barrier_probe.hip, whose persistent variants (b), (c) all paid a release fence, a
// counter, a poll and an acquire fence per step.)
//
// Mechanism (cdna_hip_programming.md, Guideline 16 form R2): the data IS the flag.  Every element of the exchanged vector travels in an
// 8-byte granule {payload : 32, tag : 32} written by ONE `sc1` (write-through, agent-scope) store; consumers re-read their granules with
// `sc1` loads (L1 bypassed) until every tag equals the step's epoch.  No fence, no counter, no flag.
//
//   (d) 256 workgroups x 256 threads, one per CU; per step every workgroup sweeps the whole vector (NG granules: thread t owns granules
//       NG/256 * t .. as 16-byte loads), checks it, and publishes its own NG/256 granules of the next vector.  NG = 1024 (a 2048-element
//       f16 vector: D of the 1.5B model) and 4096 (8192 elements: the ffn hidden vector).
//   (e) the same with a weight stream: after each successful sweep every thread requests WB bytes of a 512 MB buffer (non-temporal,
//       consumed after the NEXT sweep), i.e. the weights of the next stage are in flight while the workgroup waits for its inputs --
//       the structure of a persistent decode layer whose weight stream is decoupled from the activation dependency.
// Every spin is bounded; a give-up sets a flag and the workgroup leaves.  Double-buffered vectors (a workgroup can be at most one step
// ahead of the slowest one), tags = step + 1 so a zeroed buffer never matches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256;
constexpr unsigned SPIN_MAX = 1u << 20;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LPT 16-byte loads per thread (two granules each), all in flight, one wait
template <int LPT>
__device__ __forceinline__ void sweep_once(const u32x4* p, u32x4 (&v)[LPT]) {
    static_assert(LPT == 2 || LPT == 8, "");
    if constexpr (LPT == 2) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %2, %8, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                     "global_load_dwordx4 %4, %8, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
                     "global_load_dwordx4 %6, %8, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:112 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(p) : "memory");
    }
}

template <int LPT, int WB>      // WB: 16-byte weight loads per thread and step (0: no stream)
__global__ void __launch_bounds__(256) granule_kernel(unsigned long long* va, unsigned long long* vb, const u32x4* __restrict__ wts, size_t wts_vecs,
                                                      unsigned steps, unsigned* stale, unsigned* fail, unsigned* sink, unsigned long long* stamps) {
    constexpr int NG = 256 * LPT * 2, PER_WG = NG / NWG;
    __shared__ unsigned red[4];
    __shared__ unsigned ok_all;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 w[WB ? WB : 1];
    unsigned acc = 0;
    size_t wpos = ((size_t)blockIdx.x * 256 + tid) & (wts_vecs - 1);
    if (WB) {
#pragma unroll
        for (int i = 0; i < WB; ++i) w[i] = __builtin_nontemporal_load(wts + ((wpos + (size_t)i * NWG * 256) & (wts_vecs - 1)));
    }
    for (unsigned step = 0; step < steps; ++step) {
        unsigned long long* in = (step & 1) ? vb : va;
        unsigned long long* out = (step & 1) ? va : vb;
        const unsigned epoch = step + 1;
        u32x4 v[LPT];
        bool good = false;
        if (step == 0) good = true;     // the first vector is the launch's input
        for (unsigned spins = 0; !good && spins < SPIN_MAX; ++spins) {
            sweep_once<LPT>((const u32x4*)in + (size_t)tid * LPT, v);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < LPT; ++i) ok &= (v[i].y == epoch) & (v[i].w == epoch);
            good = __all(ok);
            if (!good) __builtin_amdgcn_s_sleep(1);
        }
        if (!good) { if (lane == 0) atomicAdd(fail, 1u); }
        // consume: sum of the payloads must be step * NG (every payload of vector `step` is `step`)
        unsigned s = 0;
        if (step) {
#pragma unroll
            for (int i = 0; i < LPT; ++i) s += v[i].x + v[i].z;
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave] = s;
        if (tid == 0) ok_all = 1;
        __syncthreads();
        if (!good) ok_all = 0;
        const unsigned tot = red[0] + red[1] + red[2] + red[3];
        if (step && tid == 0 && tot != step * (unsigned)NG) atomicAdd(stale, 1u);
        // "stage compute": fold the prefetched weights, then request the next stage's
        if (WB) {
#pragma unroll
            for (int i = 0; i < WB; ++i) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
            // the fold is complete before the next stage's loads are issued: without this the compiler hoists the loads above the fold,
            // renames the registers and ends the loop with `s_waitcnt vmcnt(0)` + copies -- the stream would no longer span the wait
            asm volatile("" : "+v"(acc) :: "memory");
            wpos = (wpos + (size_t)WB * NWG * 256) & (wts_vecs - 1);
#pragma unroll
            for (int i = 0; i < WB; ++i) w[i] = __builtin_nontemporal_load(wts + ((wpos + (size_t)i * NWG * 256) & (wts_vecs - 1)));
            asm volatile("" ::: "memory");
        }
        // publish my granules of the next vector: payload = step + 1, tag = step + 2
        if (tid < PER_WG)
            __hip_atomic_store(out + (size_t)blockIdx.x * PER_WG + tid, ((unsigned long long)(epoch + 1) << 32) | (unsigned long long)(step + 1),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stamps && blockIdx.x == 0 && tid == 0 && step < 64) stamps[step] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (!ok_all) break;
    }
    if (WB && acc == 0x12345678u) *sink = acc;
}


// (f) roles: 4 compute waves + 1 GATHER wave per workgroup (320 threads).  The gather wave alone sweeps the granules (8 x 16-byte sc1 loads
// per lane for a 4 KB vector) and hands the payloads to the compute waves through LDS; the compute waves keep TWO stages of weights in
// flight in registers (requested two steps before use) and never poll, so no poll queues behind a weight load in a wave's in-order
// vmcnt queue.  One barrier per step (LDS vector double-buffered).
template <int WB>
__global__ void __launch_bounds__(320) roles_kernel(unsigned long long* va, unsigned long long* vb, const u32x4* __restrict__ wts, size_t wts_vecs,
                                                    unsigned steps, unsigned* stale, unsigned* fail, unsigned* sink, unsigned long long* stamps) {
    constexpr int NG = 1024, PER_WG = NG / NWG;
    __shared__ unsigned xs[2][NG];
    __shared__ unsigned ok_flag[2];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4 w0[WB], w1[WB];
    unsigned acc = 0;
    size_t wpos = ((size_t)blockIdx.x * 256 + (tid & 255)) & (wts_vecs - 1);
    auto request = [&](u32x4 (&w)[WB]) {
#pragma unroll
        for (int i = 0; i < WB; ++i) w[i] = __builtin_nontemporal_load(wts + ((wpos + (size_t)i * NWG * 256) & (wts_vecs - 1)));
        wpos = (wpos + (size_t)WB * NWG * 256) & (wts_vecs - 1);
    };
    auto fold = [&](const u32x4 (&w)[WB]) {
#pragma unroll
        for (int i = 0; i < WB; ++i) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
        asm volatile("" : "+v"(acc) :: "memory");
    };
    if (wave < 4) { request(w0); request(w1); }
    if (tid == 0) { ok_flag[0] = 1; ok_flag[1] = 1; }
    __syncthreads();
    // The two roles run SEPARATE loops (each with its own barrier instructions; the hardware only counts arrivals): with one loop and
    // `if (wave == 4) ... if (wave < 4) ...` inside, hipcc's waitcnt pass merges the gather path (which skips the weight requests) into
    // the compute path and then waits vmcnt(1) for a stage although nine younger loads are in flight -- the prefetch distance silently
    // collapses to one stage.
    if (wave == 4) {
        for (unsigned step = 0; step < steps; ++step) {
            const unsigned long long* in = (step & 1) ? vb : va;
            const unsigned epoch = step + 1;
            unsigned* x = xs[step & 1];
            u32x4 v[8];
            bool good = step == 0;
            if (step == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (u32x4){0, 0, 0, 0};
            }
            for (unsigned spins = 0; !good && spins < SPIN_MAX; ++spins) {
                sweep_once<8>((const u32x4*)in + (size_t)lane * 8, v);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) ok &= (v[i].y == epoch) & (v[i].w == epoch);
                good = __all(ok);
                if (!good) __builtin_amdgcn_s_sleep(1);
            }
            if (!good && lane == 0) { atomicAdd(fail, 1u); ok_flag[step & 1] = 0; }
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[lane * 16 + 2 * i] = v[i].x; x[lane * 16 + 2 * i + 1] = v[i].z; }
            __syncthreads();
            if (!ok_flag[step & 1]) break;
        }
    } else {
        auto one_step = [&](unsigned step, u32x4 (&w)[WB]) -> bool {
            unsigned long long* out = (step & 1) ? va : vb;
            const unsigned epoch = step + 1;
            const unsigned* x = xs[step & 1];
            __syncthreads();
            if (!ok_flag[step & 1]) return false;
            unsigned s = 0;
            if (step) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s += x[tid * 4 + i];
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (lane == 0 && s != step * 256u) atomicAdd(stale, 1u);
            }
            fold(w);
            request(w);
            asm volatile("" ::: "memory");
            if (tid < PER_WG)
                __hip_atomic_store(out + (size_t)blockIdx.x * PER_WG + tid, ((unsigned long long)(epoch + 1) << 32) | (unsigned long long)(step + 1),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stamps && blockIdx.x == 0 && tid == 0 && step < 64) stamps[step] = __builtin_amdgcn_s_memrealtime();
            return true;
        };
        for (unsigned step = 0; step + 1 < steps; step += 2) {
            if (!one_step(step, w0)) break;
            if (!one_step(step + 1, w1)) break;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int WB>
static void run_roles(const char* label, hipStream_t s, unsigned long long* va, unsigned long long* vb, const u32x4* wts, size_t wts_vecs, unsigned* w,
                      unsigned long long* stamps) {
    const unsigned steps = 400;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, ms;
    unsigned st = 0, fl = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(va, 0, 1024 * 8, s)); CK(hipMemsetAsync(vb, 0, 1024 * 8, s)); CK(hipMemsetAsync(w, 0, 4096, s));
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL((roles_kernel<WB>), dim3(NWG), dim3(320), 0, s, va, vb, wts, wts_vecs, steps, w, w + 32, w + 64, stamps);
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
        unsigned h[2]; CK(hipMemcpy(&h[0], w, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h[1], w + 32, 4, hipMemcpyDeviceToHost));
        st += h[0]; fl += h[1];
    }
    std::vector<unsigned long long> hs(64);
    CK(hipMemcpy(hs.data(), stamps, 64 * 8, hipMemcpyDeviceToHost));
    std::vector<double> d;
    for (int i = 9; i < 64; ++i) d.push_back((double)(hs[i] - hs[i - 1]) * 0.01);
    std::sort(d.begin(), d.end());
    const double per = best * 1000.0 / steps;
    printf("%-78s: %.2f us per step (in-kernel median %.2f, p90 %.2f)  stream %.2f TB/s   (stale %u, give-ups %u)\n", label, per, d[d.size() / 2],
           d[d.size() * 9 / 10], (double)WB * 16 * 256 * NWG / (per * 1e-6) / 1e12, st, fl);
}

template <int LPT, int WB>
static void run(const char* label, hipStream_t s, unsigned long long* va, unsigned long long* vb, const u32x4* wts, size_t wts_vecs, unsigned* w,
                unsigned long long* stamps) {
    const unsigned steps = 400;
    constexpr int NG = 256 * LPT * 2;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, ms;
    unsigned st = 0, fl = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(va, 0, NG * 8, s)); CK(hipMemsetAsync(vb, 0, NG * 8, s)); CK(hipMemsetAsync(w, 0, 4096, s));
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL((granule_kernel<LPT, WB>), dim3(NWG), dim3(256), 0, s, va, vb, wts, wts_vecs, steps, w, w + 32, w + 64, stamps);
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
        unsigned h[2]; CK(hipMemcpy(&h[0], w, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h[1], w + 32, 4, hipMemcpyDeviceToHost));
        st += h[0]; fl += h[1];
    }
    std::vector<unsigned long long> hs(64);
    CK(hipMemcpy(hs.data(), stamps, 64 * 8, hipMemcpyDeviceToHost));
    std::vector<double> d;
    for (int i = 9; i < 64; ++i) d.push_back((double)(hs[i] - hs[i - 1]) * 0.01);
    std::sort(d.begin(), d.end());
    const double per = best * 1000.0 / steps;
    printf("%-78s: %.2f us per step (in-kernel median %.2f, p90 %.2f)", label, per, d[d.size() / 2], d[d.size() * 9 / 10]);
    if (WB) printf("  stream %.2f TB/s", (double)WB * 16 * 256 * NWG / (per * 1e-6) / 1e12);
    printf("   (stale %u, give-ups %u)\n", st, fl);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned long long *va, *vb; CK(hipMalloc(&va, 4096 * 8)); CK(hipMalloc(&vb, 4096 * 8));
    unsigned* w; CK(hipMalloc(&w, 4096));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 64 * 8));
    const size_t wbytes = (size_t)512 << 20;
    u32x4* wts; CK(hipMalloc(&wts, wbytes)); CK(hipMemset(wts, 0x5a, wbytes));
    const size_t wv = wbytes / 16;
    run<2, 0>("(d) granules, sc1 store + sc1 sweep, no fence; 4 KB vector (1024 granules)", s, va, vb, wts, wv, w, stamps);
    run<8, 0>("(d) granules, sc1 store + sc1 sweep, no fence; 16 KB vector (4096 granules)", s, va, vb, wts, wv, w, stamps);
    run<2, 2>("(e) + 8 KB/CU/step weight stream in flight across the wait (2 MB/step)", s, va, vb, wts, wv, w, stamps);
    run<2, 9>("(e) + 36 KB/CU/step weight stream in flight across the wait (9.4 MB/step)", s, va, vb, wts, wv, w, stamps);
    run<8, 9>("(e) 16 KB vector + 36 KB/CU/step weight stream (9.4 MB/step)", s, va, vb, wts, wv, w, stamps);
    run<2, 18>("(e) + 72 KB/CU/step weight stream (18.9 MB/step)", s, va, vb, wts, wv, w, stamps);
    run_roles<2>("(f) gather wave + 4 compute waves, 2 stages of weights in flight:  8 KB/CU/step", s, va, vb, wts, wv, w, stamps);
    run_roles<9>("(f) gather wave + 4 compute waves, 2 stages of weights in flight: 36 KB/CU/step", s, va, vb, wts, wv, w, stamps);
    run_roles<12>("(f) gather wave + 4 compute waves, 2 stages of weights in flight: 48 KB/CU/step", s, va, vb, wts, wv, w, stamps);
    return 0;
}
