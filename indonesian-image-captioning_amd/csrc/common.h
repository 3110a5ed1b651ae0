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

// ---- bf16 STORAGE of streamed operands (BASELINE configs[4]: mixed precision) --------------------------
// The decode step is bound by the bytes it streams (recurrent weights, att1, the trunk map), not by arithmetic:
// in the bf16 mode those operands are stored as bf16 and widened to fp32 in registers; every product is
// accumulated in fp32 and softmax / LSTM state / master weights / gradients stay fp32.
typedef unsigned short bf16_t;      // raw bits (the upper half of an IEEE fp32)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_to_f32(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {     // 4 consecutive elements: one 8-byte load
    const u32x2 u = *reinterpret_cast<const u32x2*>(p);
    f32x4 r;
    r[0] = __builtin_bit_cast(float, u[0] << 16);
    r[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, u[1] << 16);
    r[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
    return r;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_to_f32(*p); }

// Sum `nslab` partial slabs (split-K outputs of skinny_gemm) at element `idx`, fixed order.
// All loads are issued before the first add (a run-time-bounded loop would serialise one memory
// latency per slab): loads beyond n re-read the last slab (a cache hit) and are discarded.
template <int MAXN>
__device__ __forceinline__ float slab_sum_u(const float* __restrict__ p, long idx, int n, long stride) {
    float v[MAXN];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) v[i] = p[idx + (long)min(i, n - 1) * stride];
    float s = v[0];
#pragma unroll
    for (int i = 1; i < MAXN; ++i) s += (i < n) ? v[i] : 0.f;
    return s;
}
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, long idx, int nslab, long stride) {
    if (nslab <= 1) return p[idx];
    if (nslab <= 2) return slab_sum_u<2>(p, idx, nslab, stride);
    if (nslab <= 4) return slab_sum_u<4>(p, idx, nslab, stride);
    if (nslab <= 8) return slab_sum_u<8>(p, idx, nslab, stride);
    return slab_sum_u<16>(p, idx, nslab, stride);
}

// ---- buffer loads with hardware range checking (out-of-range lanes read 0, no branches) -------------
// hipcc turns "load from a clamped address, then select" back into a branch around the load and waits
// vmcnt(0) after every one of them, which serialises a whole memory latency per load.  A raw buffer
// load cannot be predicated that way: lanes that must not contribute get the offset OOB_OFF.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_OFF = 0x80000000u;   // > any num_records we build (checked on the host)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// row (0..31) of accumulator register r of lane l in a 32x32 f32 MFMA C/D tile; column is l & 31.
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace scn
