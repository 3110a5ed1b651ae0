// Shared host/device helpers for libscnattn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>

namespace scn {

// ---- error plumbing: no exceptions / aborts cross the C boundary -------------------------------
void set_error(const char* fmt, ...);
const char* last_error();

#define SCN_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            scn::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return (int)_e;                                                               \
        }                                                                                 \
    } while (0)

#define SCN_LAUNCH_CHECK() SCN_HIP(hipGetLastError())

#define SCN_ARG(cond, msg)                                                     \
    do {                                                                       \
        if (!(cond)) {                                                         \
            scn::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, msg); \
            return -1;                                                         \
        }                                                                      \
    } while (0)

#define SCN_TRY(call)            \
    do {                         \
        int _rc = (call);        \
        if (_rc != 0) return _rc; \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Sum `nslab` partial slabs (split-K outputs of skinny_gemm) at element `idx`, fixed order.
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, long idx, int nslab, long stride) {
    float s = p[idx];
    for (int i = 1; i < nslab; ++i) s += p[idx + (long)i * stride];
    return s;
}

// row (0..31) of accumulator register r of lane l in a 32x32 f32 MFMA C/D tile; column is l & 31.
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace scn
