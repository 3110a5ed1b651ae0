// What does a grid-wide dependency cost INSIDE one persistent launch on this chip?  The decode step's seven dependent
// launches would become the seams of a persistent recurrence kernel; this measures such a seam with the
// data hand-off it exists for: every workgroup writes 1 KiB (write-through), all arrive at a barrier, every workgroup reads
// 1 KiB another workgroup (another XCD) wrote.  One workgroup per CU (256), plain launch (the grid fits: checked),
// two barrier forms: one monotonic counter; per-XCD counters + a top counter (XCD-hierarchical).  Every spin is BOUNDED:
// a barrier that does not complete in ~50 ms sets an error flag and every workgroup leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_floor.hip -o /tmp/grid_barrier_floor && /tmp/grid_barrier_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bar { unsigned* top; unsigned* xcd; unsigned* gen; int* err; };

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false when the barrier timed out (error flag set): the caller leaves the kernel
__device__ bool barrier_flat(const Bar& b, unsigned round, int nwg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (round + 1) * (unsigned)nwg;
        long spins = 0;
        while (ld_sc1(b.top) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000 || ld_sc1((const unsigned*)b.err)) { *b.err = 1; ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ok = __syncthreads_and(ok);
    return ok;
}

__device__ bool barrier_xcd(const Bar& b, unsigned round, int nwg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7, per = nwg >> 3;                 // workgroups are dealt round-robin over the 8 XCDs
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(b.xcd + 32 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        if (old == (round + 1) * (unsigned)per - 1) {                // last of its XCD: arrive at the top, wait for all 8, release the XCD
            __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_sc1(b.top) < (round + 1) * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000 || ld_sc1((const unsigned*)b.err)) { *b.err = 1; ok = false; break; }
            }
            __hip_atomic_store(b.gen + 32 * x, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_sc1(b.gen + 32 * x) < round + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000 || ld_sc1((const unsigned*)b.err)) { *b.err = 1; ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ok = __syncthreads_and(ok);
    return ok;
}

template <int KIND>
__global__ __launch_bounds__(256) void seams(Bar b, float* buf, int rounds, int nwg, float* sink) {
    float acc = 0.f;
    const int me = blockIdx.x, other = (me + 9) % nwg;             // another CU on another XCD
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((long)(r & 1) * nwg + me) * 256;
        __hip_atomic_store(mine + threadIdx.x, acc + (float)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through
        const bool ok = KIND == 0 ? barrier_flat(b, (unsigned)r, nwg) : barrier_xcd(b, (unsigned)r, nwg);
        if (!ok) return;
        const float* theirs = buf + ((long)(r & 1) * nwg + other) * 256;
        acc += theirs[threadIdx.x];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int nwg = p.multiProcessorCount;                          // one workgroup per CU
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, seams<0>, 256, 0));
    if (per_cu < 1 || nwg % 8) { printf("grid does not fit / CU count %d\n", nwg); return 1; }
    unsigned* words; int* err; float *buf, *sink;
    CK(hipMalloc(&words, 4 * (1 + 2 * 8 * 32))); CK(hipMalloc(&err, 4)); CK(hipMalloc(&buf, 2L * nwg * 256 * 4)); CK(hipMalloc(&sink, 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind)
        for (int rounds : {7 * 51, 5 * 51}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(words, 0, 4 * (1 + 2 * 8 * 32), st)); CK(hipMemsetAsync(err, 0, 4, st));
                Bar b{words, words + 32, words + 32 + 8 * 32, err};
                CK(hipEventRecord(e0, st));
                if (kind == 0) hipLaunchKernelGGL(seams<0>, dim3(nwg), dim3(256), 0, st, b, buf, rounds, nwg, sink);
                else           hipLaunchKernelGGL(seams<1>, dim3(nwg), dim3(256), 0, st, b, buf, rounds, nwg, sink);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                if (h) { printf("barrier timed out (kind %d)\n", kind); return 1; }
                if (ms < best) best = ms;
            }
            printf("%-28s %3d workgroups, %4d seams in one launch: %.2f us per seam (write 1 KiB -> barrier -> read 1 KiB)  -> a %d-seam decode step = %.1f us of seams\n",
                   kind == 0 ? "one counter" : "per-XCD counters + top", nwg, rounds, best * 1e3 / rounds, rounds / 51, best * 1e3 / 51);
        }
    return 0;
}
