// Fused BatchNorm2d (+ residual add) (+ ReLU) for the ResNet-152 trunk, channels-last.
//
// The reference's encoder (models/encoders/caption.py:17-22) is torchvision's ResNet-152, whose
// bottleneck block runs, after every convolution,  bn -> relu  or  bn -> (+identity) -> relu  as
// separate eager ops; in train mode each BatchNorm is 3 kernels forward and 3 backward, the ReLU and the
// residual add are one more each way.  On channels-last memory a feature map is a row-major
// [R = N*H*W, C] matrix, so batch statistics are plain column reductions and everything else is
// element-wise; these kernels do the whole group in two passes forward and two backward:
//   forward : bn_stats (column sums, shifted for conditioning) -> bn_finalize (mean, 1/std, running
//             stats)  ;  bn_apply : y = relu(gamma*(z-mean)/std + beta + residual)
//   backward: bn_bwd_reduce : dbeta = sum g, dgamma = sum g*xhat with g = dy*[y>0]
//             bn_bwd_dx     : dz = gamma/std * (g - dbeta/R - xhat*dgamma/R),  dresidual = g
// All are HBM-bound streaming kernels (16-byte loads along C, 4 independent row loads in flight).
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

// Storage type of the feature maps: float, or bf16 when the trunk runs under bf16 autocast (BASELINE
// config 5).  Arithmetic and statistics are fp32 either way.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
};
template <> struct IO<__bf16> {
    static __device__ __forceinline__ f32x4 ld(const __bf16* p) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void st(__bf16* p, f32x4 v) {
        *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
    static __device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
};

// partial[chunk][2][C]: shifted sums S1 = sum(x - s), S2 = sum((x - s)^2), s = x[0][c]
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(int R, int C, int rows_per_chunk, const T* __restrict__ x,
                                                       float* __restrict__ partial) {
    __shared__ float red[16][2][64 + 1];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < C) {
        const f32x4 sh = IO<T>::ld(x + c);
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = IO<T>::ld(x + (long)(r + 16 * j) * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[j][k] - sh[k];
                    s1[k] += d;
                    s2[k] = fmaf(d, d, s2[k]);
                }
        }
        for (; r < r1; r += 16) {
            const f32x4 v = IO<T>::ld(x + (long)r * C + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[k] - sh[k];
                s1[k] += d;
                s2[k] = fmaf(d, d, s2[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[rl][0][cl * 4 + k] = s1[k];
        red[rl][1][cl * 4 + k] = s2[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        if (blockIdx.x * 64 + cc < C) {
            float v = red[0][which][cc];
#pragma unroll
            for (int i = 1; i < 16; ++i) v += red[i][which][cc];
            partial[((long)blockIdx.y * 2 + which) * C + blockIdx.x * 64 + cc] = v;
        }
    }
}

// Sum the per-chunk partials of 16 channels per workgroup: 16 chunk-lanes per channel run in parallel
// (a serial loop over up to 256 chunks is one dependent memory latency per chunk: ~40 us).
__device__ __forceinline__ void reduce_partials16(const float* __restrict__ partial, int C, int nchunk, int c, int lane16,
                                                  float (*red)[2][17], float& s1, float& s2) {
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int i = lane16; i < nchunk; i += 16) {
            a += partial[((long)i * 2) * C + c];
            b += partial[((long)i * 2 + 1) * C + c];
        }
    }
    const int cc = threadIdx.x & 15;
    red[lane16][0][cc] = a;
    red[lane16][1][cc] = b;
    __syncthreads();
    s1 = 0.f;
    s2 = 0.f;
    if (lane16 == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s1 += red[i][0][cc];
            s2 += red[i][1][cc];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(int R, int C, int nchunk, const T* __restrict__ x,
                                                          const float* __restrict__ partial, float eps, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ ss_out) {
    __shared__ float red[16][2][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), lane16 = threadIdx.x >> 4;
    float s1, s2;
    reduce_partials16(partial, C, nchunk, c, lane16, red, s1, s2);
    if (lane16 != 0 || c >= C) return;
    const float inv_n = 1.f / (float)R;
    const float m1 = s1 * inv_n;
    const float mu = IO<T>::ld1(x + c) + m1;
    float var = s2 * inv_n - m1 * m1;
    if (var < 0.f) var = 0.f;
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
    if (ss_out) {      // folded {scale, shift} for a consumer that normalises on load (cgemm.hip prologues)
        const float sc = gamma[c] * rsqrtf(var + eps);
        ss_out[2 * c] = sc;
        ss_out[2 * c + 1] = fmaf(-mu, sc, beta[c]);
    }
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    if (run_var) {
        const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * unbiased;
    }
}

template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(long n4, int C, const T* __restrict__ z,
                                                       const T* __restrict__ res, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ y) {
    const int C4 = C >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const f32x4 v = IO<T>::ld(z + i * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
        if (RES) rr = IO<T>::ld(res + i * 4);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = fmaf((v[k] - mu[k]) * is[k], ga[k], be[k]);
            if (RES) t += rr[k];
            if (RELU) t = fmaxf(t, 0.f);
            o[k] = t;
        }
        IO<T>::st(y + i * 4, o);
    }
}

// partial[chunk][2][C]: sum g, sum g*xhat, g = dy * [y > 0] (RELU) or dy.
// MASKZ: the ReLU mask is recomputed from z with the forward's own expression, fma((z-mean)*invstd, gamma, beta) > 0,
// instead of being read back from y -- valid when no residual was added before the ReLU (bn1/bn2 of a bottleneck);
// the backward pass then never touches y (one 4-byte read per element less in each of its two passes).
// GOUT: the masked gradient g is also WRITTEN (it is the gradient of the residual branch, an output of the backward pass
// anyway): the element-wise second pass then reads g and z only -- not dy and y again -- and writes dz only: 7 map
// transfers per BatchNorm(+residual)+ReLU backward instead of 8.
template <typename T, bool RELU, bool MASKZ, bool GOUT = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(int R, int C, int rows_per_chunk, const T* __restrict__ dy,
                                                            const T* __restrict__ y, const T* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ partial, T* __restrict__ gout = nullptr,
                                                            int ldp = 0) {      // ldp > 0: channel-major partial [2][C][ldp]
    __shared__ float red[16][2][64 + 1];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float sg[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0};
    if (c < C) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        f32x4 ga = {0.f, 0.f, 0.f, 0.f}, be = {0.f, 0.f, 0.f, 0.f};
        if (MASKZ) {
            ga = *reinterpret_cast<const f32x4*>(gamma + c);
            be = *reinterpret_cast<const f32x4*>(beta + c);
        }
        for (int r = r0 + rl; r < r1; r += 32) {
            f32x4 g[2], yy[2], zz[2];
            bool ok[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rr = r + 16 * j;
                ok[j] = rr < r1;
                const long off = (long)min(rr, r1 - 1) * C + c;
                g[j] = IO<T>::ld(dy + off);
                zz[j] = IO<T>::ld(z + off);
                if (RELU && !MASKZ) yy[j] = IO<T>::ld(y + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 gm;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float gv = g[j][k];
                    const float xh = (zz[j][k] - mu[k]) * is[k];
                    if (RELU) {
                        const bool on = MASKZ ? (fmaf(xh, ga[k], be[k]) > 0.f) : (yy[j][k] > 0.f);
                        if (!on) gv = 0.f;
                    }
                    if (!ok[j]) gv = 0.f;
                    gm[k] = gv;
                    sg[k] += gv;
                    sx[k] = fmaf(gv, xh, sx[k]);
                }
                if (GOUT && ok[j]) IO<T>::st(gout + (long)(r + 16 * j) * C + c, gm);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[rl][0][cl * 4 + k] = sg[k];
        red[rl][1][cl * 4 + k] = sx[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        if (blockIdx.x * 64 + cc < C) {
            float v = red[0][which][cc];
#pragma unroll
            for (int i = 1; i < 16; ++i) v += red[i][which][cc];
            if (ldp > 0) partial[((long)which * C + blockIdx.x * 64 + cc) * ldp + blockIdx.y] = v;
            else partial[((long)blockIdx.y * 2 + which) * C + blockIdx.x * 64 + cc] = v;
        }
    }
}

// ================= finalize on load: channel-major partials [2][C][ldp] ==========================================
// The statistics epilogues of csrc/cgemm.hip (and cstats / bn_bwd_reduce / the stem) leave one partial per (channel,
// 64-row block), channel-major.  A dependent launch costs ~8-9 us on this chip (4-5 us floor of a tiny kernel + the
// boundary), a Bottleneck had six tiny `finalize` launches on its critical path; here the CONSUMER of the statistics --
// an element-wise kernel that owns a block of 64 channels per workgroup -- sums the partials of its own channels while
// its first map loads are in flight: 4 lanes per channel read contiguous 16-byte pieces of a channel's row, fixed order
// (bit-reproducible, every workgroup of a column block computes the same bits).  The workgroups with blockIdx.y == 0
// also write the per-channel results (mean / invstd / running statistics, or d beta / d gamma) for later kernels.
__device__ __forceinline__ void reduce_partials_t(const float* __restrict__ partial, int C, int ldp, int nchunk, int c0,
                                                  float (*out)[64]) {
    const int ch = threadIdx.x >> 2, q = threadIdx.x & 3, c = c0 + ch;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(partial, (unsigned)(2L * C * ldp * 4));
    float a = 0.f, b = 0.f;
    const unsigned o1 = (unsigned)((long)c * ldp * 4), o2 = (unsigned)(((long)C + c) * ldp * 4);
    for (int i0 = 4 * q; i0 < nchunk; i0 += 64) {          // 4 x 2 sixteen-byte loads in flight per lane
        f32x4 u[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + 16 * j;
            const bool ok = c < C && i < nchunk;
            u[j] = buf_load4(rs, ok ? o1 + (unsigned)i * 4u : OOB_OFF);
            v[j] = buf_load4(rs, ok ? o2 + (unsigned)i * 4u : OOB_OFF);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + 16 * j;
#pragma unroll
            for (int e = 0; e < 4; ++e) {          // the row's padding past nchunk is never summed
                a += (i + e < nchunk) ? u[j][e] : 0.f;
                b += (i + e < nchunk) ? v[j][e] : 0.f;
            }
        }
    }
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
    b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
    if (q == 0) { out[0][ch] = a; out[1][ch] = b; }
    __syncthreads();
}

struct BnFin {      // what a forward finalize produces besides the normalised map
    const float* partial; int ldp, nchunk; const float* shift; float eps, momentum;
    float* mean; float* invstd; float* run_mean; float* run_var; float* ss_out;
};

// y = [relu](gamma*(z-mean)*invstd + beta [+ res]) with mean / invstd computed here from the producer's partials.
// grid (C/64 column blocks, row chunks).  `shift` must not alias anything this kernel writes (the caller hands over the
// PREVIOUS step's batch mean, not the running mean this kernel updates).
template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_apply_fin_kernel(long R, int C, int rows_per_chunk, const T* __restrict__ z,
                                                           const T* __restrict__ res, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y, BnFin f) {
    __shared__ float st[2][64];
    __shared__ float smu[64], sis[64];
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, cq = min(c0 + cl * 4, C - 4);
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    // the first rows of the map do not depend on the statistics: in flight while the partials are summed
    f32x4 v[4], rr[4];
    long r = r0 + rl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long off = min(r + 16 * j, r1 - 1) * C + cq;
        v[j] = IO<T>::ld(z + off);
        if (RES) rr[j] = IO<T>::ld(res + off);
    }
    reduce_partials_t(f.partial, C, f.ldp, f.nchunk, c0, st);
    if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
        const int c = c0 + threadIdx.x;
        const float inv_n = 1.f / (float)R;
        const float m1 = st[0][threadIdx.x] * inv_n;
        const float mu = (f.shift ? f.shift[c] : 0.f) + m1;
        float var = st[1][threadIdx.x] * inv_n - m1 * m1;
        if (var < 0.f) var = 0.f;
        const float is = rsqrtf(var + f.eps);
        smu[threadIdx.x] = mu;
        sis[threadIdx.x] = is;
        if (blockIdx.y == 0) {
            f.mean[c] = mu;
            f.invstd[c] = is;
            if (f.run_mean) f.run_mean[c] = (1.f - f.momentum) * f.run_mean[c] + f.momentum * mu;
            if (f.run_var) {
                const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
                f.run_var[c] = (1.f - f.momentum) * f.run_var[c] + f.momentum * unbiased;
            }
            if (f.ss_out) {
                const float sc = gamma[c] * is;
                f.ss_out[2 * c] = sc;
                f.ss_out[2 * c + 1] = fmaf(-mu, sc, beta[c]);
            }
        }
    }
    __syncthreads();
    const int c = c0 + cl * 4;
    if (c >= C) return;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(smu + cl * 4), is = *reinterpret_cast<const f32x4*>(sis + cl * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    while (r < r1) {
        f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = fmaf((v[j][k] - mu[k]) * is[k], ga[k], be[k]);      // the expression every mask recomputation uses
                if (RES) t += rr[j][k];
                if (RELU) t = fmaxf(t, 0.f);
                o[j][k] = t;
            }
        const long rn = r + 64;
        if (rn < r1) {        // next rows in flight before this batch is stored
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long off = min(rn + 16 * j, r1 - 1) * C + c;
                v[j] = IO<T>::ld(z + off);
                if (RES) rr[j] = IO<T>::ld(res + off);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r + 16 * j < r1) IO<T>::st(y + (r + 16 * j) * C + c, o[j]);
        r = rn;
    }
}

// statistics only (a BatchNorm whose consumer normalises on load: bn2 in front of conv3's prologue; the stem)
__global__ __launch_bounds__(256) void bn_finalize_t_kernel(long R, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, BnFin f) {
    __shared__ float st[2][64];
    const int c0 = blockIdx.x * 64;
    reduce_partials_t(f.partial, C, f.ldp, f.nchunk, c0, st);
    if (threadIdx.x >= 64 || c0 + threadIdx.x >= C) return;
    const int c = c0 + threadIdx.x;
    const float inv_n = 1.f / (float)R;
    const float m1 = st[0][threadIdx.x] * inv_n;
    const float mu = (f.shift ? f.shift[c] : 0.f) + m1;
    float var = st[1][threadIdx.x] * inv_n - m1 * m1;
    if (var < 0.f) var = 0.f;
    const float is = rsqrtf(var + f.eps);
    f.mean[c] = mu;
    f.invstd[c] = is;
    if (f.run_mean) f.run_mean[c] = (1.f - f.momentum) * f.run_mean[c] + f.momentum * mu;
    if (f.run_var) {
        const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
        f.run_var[c] = (1.f - f.momentum) * f.run_var[c] + f.momentum * unbiased;
    }
    if (f.ss_out) {
        const float sc = gamma[c] * is;
        f.ss_out[2 * c] = sc;
        f.ss_out[2 * c + 1] = fmaf(-mu, sc, beta[c]);
    }
}

// dz = gamma*invstd*(g - dbeta/R - xhat*dgamma/R) from an already masked g, with dbeta = sum g and dgamma = sum g*xhat
// summed here from the channel-major partials the mask pass / the reduce pass wrote
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dx_fin_kernel(long R, int C, int rows_per_chunk, const T* __restrict__ g,
                                                            const T* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ partial, int ldp, int nchunk,
                                                            float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                            T* __restrict__ dz) {
    __shared__ float st[2][64];
    const int c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = c0 + cl * 4, cq = min(c, C - 4);
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    f32x4 gg[4], zz[4];       // first rows in flight while the partials are summed
    long r = r0 + rl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long off = min(r + 16 * j, r1 - 1) * C + cq;
        gg[j] = IO<T>::ld(g + off);
        zz[j] = IO<T>::ld(z + off);
    }
    reduce_partials_t(partial, C, ldp, nchunk, c0, st);
    if (blockIdx.y == 0 && threadIdx.x < 64 && c0 + threadIdx.x < C) {
        dbeta[c0 + threadIdx.x] = st[0][threadIdx.x];
        dgamma[c0 + threadIdx.x] = st[1][threadIdx.x];
    }
    if (c >= C) return;
    const float inv_n = 1.f / (float)R;
    const f32x4 db = *reinterpret_cast<const f32x4*>(&st[0][cl * 4]), dg = *reinterpret_cast<const f32x4*>(&st[1][cl * 4]);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    while (r < r1) {
        f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (zz[j][k] - mu[k]) * is[k];
                o[j][k] = ga[k] * is[k] * (gg[j][k] - db[k] * inv_n - xh * dg[k] * inv_n);
            }
        const long rn = r + 64;
        if (rn < r1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long off = min(rn + 16 * j, r1 - 1) * C + c;
                gg[j] = IO<T>::ld(g + off);
                zz[j] = IO<T>::ld(z + off);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r + 16 * j < r1) IO<T>::st(dz + (r + 16 * j) * C + c, o[j]);      // dz may alias g: same element, read first
        r = rn;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int C, int nchunk, const float* __restrict__ partial,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma) {
    __shared__ float red[16][2][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), lane16 = threadIdx.x >> 4;
    float a, b;
    reduce_partials16(partial, C, nchunk, c, lane16, red, a, b);
    if (lane16 != 0 || c >= C) return;
    dbeta[c] = a;
    dgamma[c] = b;
}

// TRAIN: dz = gamma*invstd*(g - dbeta/R - xhat*dgamma/R); eval: dz = gamma*invstd*g.  dres = g.
template <typename T, bool RELU, bool TRAIN, bool MASKZ>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(long n4, int R, int C, const T* __restrict__ dy,
                                                        const T* __restrict__ y, const T* __restrict__ z,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, T* __restrict__ dz,
                                                        T* __restrict__ dres) {
    const int C4 = C >> 2;
    const float inv_n = 1.f / (float)R;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        f32x4 g = IO<T>::ld(dy + i * 4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        f32x4 zz = {0.f, 0.f, 0.f, 0.f}, mu = {0.f, 0.f, 0.f, 0.f};
        if (TRAIN || MASKZ) {
            zz = IO<T>::ld(z + i * 4);
            mu = *reinterpret_cast<const f32x4*>(mean + c);
        }
        if (RELU && MASKZ) {
            const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!(fmaf((zz[k] - mu[k]) * is[k], ga[k], be[k]) > 0.f)) g[k] = 0.f;
        } else if (RELU) {
            const f32x4 yy = IO<T>::ld(y + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!(yy[k] > 0.f)) g[k] = 0.f;
        }
        f32x4 o;
        if (TRAIN) {
            const f32x4 db = *reinterpret_cast<const f32x4*>(dbeta + c);
            const f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (zz[k] - mu[k]) * is[k];
                o[k] = ga[k] * is[k] * (g[k] - db[k] * inv_n - xh * dg[k] * inv_n);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ga[k] * is[k] * g[k];
        }
        if (dz) IO<T>::st(dz + i * 4, o);
        if (dres) IO<T>::st(dres + i * 4, g);
    }
}

}  // namespace
namespace {

inline int pick_chunks(int R, int C, int* rows_per_chunk) {
    const int colgroups = cdiv(C, 64);
    int nchunk = 2048 / colgroups;           // ~8 workgroups per CU in total
    const int maxchunk = cdiv(R, 64);        // at least 64 rows per chunk
    if (nchunk > maxchunk) nchunk = maxchunk;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 256) nchunk = 256;
    int rpc = cdiv(R, nchunk);
    rpc = (rpc + 15) & ~15;
    *rows_per_chunk = rpc;
    return cdiv(R, rpc);
}

inline unsigned ew_blocks(long n4) {
    long b = (n4 + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

}  // namespace

int bn_max_chunks() { return 256; }

template <typename T>
static int bn_stats_t(hipStream_t st, int R, int C, const T* x, float eps, float momentum, float* partial, float* mean,
                      float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, float* ss_out) {
    int rpc;
    const int nchunk = pick_chunks(R, C, &rpc);
    hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(cdiv(C, 64), nchunk), dim3(256), 0, st, R, C, rpc, x, partial);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3(cdiv(C, 16)), dim3(256), 0, st, R, C, nchunk, x, partial, eps,
                       momentum, mean, invstd, run_mean, run_var, gamma, beta, ss_out);
    SCN_LAUNCH_CHECK();
    return 0;
}

int bn_stats(hipStream_t st, int R, int C, const void* x, int bf16, float eps, float momentum, float* partial,
             float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta,
             float* ss_out) {
    SCN_ARG(R > 0 && C > 0 && C % 4 == 0 && x && partial && mean && invstd, "bn_stats: bad argument");
    SCN_ARG((reinterpret_cast<uintptr_t>(x) & (bf16 ? 7u : 15u)) == 0, "bn_stats: x is not vector aligned");
    SCN_ARG(!ss_out || (gamma && beta), "bn_stats: folded scale/shift need gamma and beta");
    if (bf16) return bn_stats_t<__bf16>(st, R, C, (const __bf16*)x, eps, momentum, partial, mean, invstd, run_mean, run_var, gamma, beta, ss_out);
    return bn_stats_t<float>(st, R, C, (const float*)x, eps, momentum, partial, mean, invstd, run_mean, run_var, gamma, beta, ss_out);
}

template <typename T>
static int bn_apply_t(hipStream_t st, int R, int C, const T* z, const T* res, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int relu, T* y) {
    const long n4 = (long)R * C / 4;
    dim3 grid(ew_blocks(n4)), block(256);
#define SCN_BN_APPLY(RELU_, RES_) \
    hipLaunchKernelGGL((bn_apply_kernel<T, RELU_, RES_>), grid, block, 0, st, n4, C, z, res, mean, invstd, gamma, beta, y)
    if (relu && res) SCN_BN_APPLY(true, true);
    else if (relu) SCN_BN_APPLY(true, false);
    else if (res) SCN_BN_APPLY(false, true);
    else SCN_BN_APPLY(false, false);
#undef SCN_BN_APPLY
    SCN_LAUNCH_CHECK();
    return 0;
}

int bn_apply(hipStream_t st, int R, int C, const void* z, const void* res, int bf16, const float* mean,
             const float* invstd, const float* gamma, const float* beta, int relu, void* y) {
    SCN_ARG(R > 0 && C > 0 && C % 4 == 0 && z && mean && invstd && gamma && beta && y, "bn_apply: bad argument");
    if (bf16) return bn_apply_t<__bf16>(st, R, C, (const __bf16*)z, (const __bf16*)res, mean, invstd, gamma, beta, relu, (__bf16*)y);
    return bn_apply_t<float>(st, R, C, (const float*)z, (const float*)res, mean, invstd, gamma, beta, relu, (float*)y);
}

template <typename T>
static int bn_bwd_t(hipStream_t st, int R, int C, const T* dy, const T* y, const T* z, const float* mean,
                    const float* invstd, const float* gamma, const float* beta, int relu, int train, float* partial,
                    float* dbeta, float* dgamma, T* dz, T* dres) {
    int rpc;
    const int nchunk = pick_chunks(R, C, &rpc);
    dim3 rgrid(cdiv(C, 64), nchunk), block(256);
    const bool maskz = relu && !y;       // ReLU mask recomputed from z (no residual in front of the ReLU)
    // fp32 maps with a residual branch: the reduction pass writes g = dy * [y > 0] as d residual, the second pass reads it
    // (bf16 maps keep the old form: their dz is computed from the unrounded g)
    const bool gfirst = relu && !maskz && dres && dz && sizeof(T) == 4;
    if (maskz)     hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true, true>), rgrid, block, 0, st, R, C, rpc, dy, y, z, mean, invstd, gamma, beta, partial);
    else if (relu && gfirst) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true, false, true>), rgrid, block, 0, st, R, C, rpc, dy, y, z, mean, invstd, gamma, beta, partial, dres);
    else if (relu) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true, false>), rgrid, block, 0, st, R, C, rpc, dy, y, z, mean, invstd, gamma, beta, partial);
    else           hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false, false>), rgrid, block, 0, st, R, C, rpc, dy, y, z, mean, invstd, gamma, beta, partial);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), block, 0, st, C, nchunk, partial, dbeta, dgamma);
    SCN_LAUNCH_CHECK();
    if (dz || dres) {
        const long n4 = (long)R * C / 4;
        dim3 grid(ew_blocks(n4));
#define SCN_BN_DX(RELU_, TRAIN_, MASKZ_)                                                                                  \
    hipLaunchKernelGGL((bn_bwd_dx_kernel<T, RELU_, TRAIN_, MASKZ_>), grid, block, 0, st, n4, R, C, dy, y, z, mean, invstd, \
                       gamma, beta, dbeta, dgamma, dz, dres)
        if (gfirst) {
            if (train) hipLaunchKernelGGL((bn_bwd_dx_kernel<T, false, true, false>), grid, block, 0, st, n4, R, C, dres, y, z, mean,
                                          invstd, gamma, beta, dbeta, dgamma, dz, (T*)nullptr);
            else       hipLaunchKernelGGL((bn_bwd_dx_kernel<T, false, false, false>), grid, block, 0, st, n4, R, C, dres, y, z, mean,
                                          invstd, gamma, beta, dbeta, dgamma, dz, (T*)nullptr);
        } else if (maskz && train) SCN_BN_DX(true, true, true);
        else if (maskz) SCN_BN_DX(true, false, true);
        else if (relu && train) SCN_BN_DX(true, true, false);
        else if (relu) SCN_BN_DX(true, false, false);
        else if (train) SCN_BN_DX(false, true, false);
        else SCN_BN_DX(false, false, false);
#undef SCN_BN_DX
        SCN_LAUNCH_CHECK();
    }
    return 0;
}

// ---- finalize-on-load entry points (fp32 maps, channel-major partials) -----------------------------------------------------
static inline int ew_chunks(long R, int C, int* rows_per_chunk) {       // ~1024 workgroups, >= 128 rows each: the finalize
    const int colgroups = cdiv(C, 64);                                    // prologue is paid once per workgroup
    long n = 1024 / colgroups;
    const long maxn = (R + 127) / 128;
    if (n > maxn) n = maxn;
    if (n < 1) n = 1;
    long rpc = (R + n - 1) / n;
    rpc = (rpc + 15) & ~15L;
    *rows_per_chunk = (int)rpc;
    return (int)((R + rpc - 1) / rpc);
}

int bn_apply_fin(hipStream_t st, long R, int C, const void* z, const void* res, int bf16, const float* partial, int ldp, int nchunk,
                 const float* shift, float eps, float momentum, const float* gamma, const float* beta, int relu, void* y,
                 float* mean, float* invstd, float* run_mean, float* run_var, float* ss_out) {
    SCN_ARG(R > 0 && C > 0 && C % 4 == 0 && z && y && partial && ldp >= nchunk && ldp % 4 == 0 && nchunk > 0 && gamma && beta &&
            mean && invstd && aligned16(partial), "bn_apply_fin: bad argument");
    SCN_ARG(((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & (bf16 ? 7u : 15u)) == 0,
            "bn_apply_fin: maps are not vector aligned");
    SCN_ARG(shift != run_mean || !run_mean, "bn_apply_fin: shift must not alias the running mean it updates");
    SCN_ARG(2L * C * ldp * 4 < 0x7fffffffL, "bn_apply_fin: partial too large");
    BnFin f{partial, ldp, nchunk, shift, eps, momentum, mean, invstd, run_mean, run_var, ss_out};
    int rpc;
    const int nch = ew_chunks(R, C, &rpc);
    dim3 grid(cdiv(C, 64), nch), block(256);
#define SCN_AF(T_, RELU_, RES_) hipLaunchKernelGGL((bn_apply_fin_kernel<T_, RELU_, RES_>), grid, block, 0, st, R, C, rpc, (const T_*)z, (const T_*)res, gamma, beta, (T_*)y, f)
    if (bf16) {
        if (relu && res) SCN_AF(__bf16, true, true); else if (relu) SCN_AF(__bf16, true, false); else if (res) SCN_AF(__bf16, false, true); else SCN_AF(__bf16, false, false);
    } else {
        if (relu && res) SCN_AF(float, true, true); else if (relu) SCN_AF(float, true, false); else if (res) SCN_AF(float, false, true); else SCN_AF(float, false, false);
    }
#undef SCN_AF
    SCN_LAUNCH_CHECK();
    return 0;
}

int bn_finalize_t(hipStream_t st, long R, int C, const float* partial, int ldp, int nchunk, const float* shift, float eps,
                  float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                  const float* beta, float* ss_out) {
    SCN_ARG(R > 0 && C > 0 && partial && ldp >= nchunk && ldp % 4 == 0 && nchunk > 0 && mean && invstd && aligned16(partial),
            "bn_finalize_t: bad argument");
    SCN_ARG(!ss_out || (gamma && beta), "bn_finalize_t: folded scale/shift need gamma and beta");
    SCN_ARG(2L * C * ldp * 4 < 0x7fffffffL, "bn_finalize_t: partial too large");
    BnFin f{partial, ldp, nchunk, shift, eps, momentum, mean, invstd, run_mean, run_var, ss_out};
    hipLaunchKernelGGL(bn_finalize_t_kernel, dim3(cdiv(C, 64)), dim3(256), 0, st, R, C, gamma, beta, f);
    SCN_LAUNCH_CHECK();
    return 0;
}

// g = dy * [y > 0] (relu with the forward output y) or dy; sums of g and g*xhat as channel-major partials [2][C][ldp];
// g is written to gout (it is the residual branch's gradient).  Returns the chunk count through nchunk_out.
int bn_bwd_reduce_t(hipStream_t st, int R, int C, const void* dy, const void* y, const void* z, int bf16, const float* mean,
                    const float* invstd, int relu, float* partial, int ldp_cap, void* gout, int* nchunk_out) {
    SCN_ARG(R > 0 && C > 0 && C % 4 == 0 && dy && z && mean && invstd && partial && (!relu || y), "bn_bwd_reduce_t: bad argument");
    int rpc;
    const int nchunk = pick_chunks(R, C, &rpc);
    const int ldp = (nchunk + 3) & ~3;
    SCN_ARG(ldp <= ldp_cap, "bn_bwd_reduce_t: partial buffer too small");
    dim3 grid(cdiv(C, 64), nchunk), block(256);
    const float* nf = nullptr;
#define SCN_BR(T_, RELU_, GOUT_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T_, RELU_, false, GOUT_>), grid, block, 0, st, R, C, rpc, (const T_*)dy, (const T_*)y, (const T_*)z, mean, invstd, nf, nf, partial, (T_*)gout, ldp)
    if (bf16) { if (relu && gout) SCN_BR(__bf16, true, true); else if (relu) SCN_BR(__bf16, true, false); else SCN_BR(__bf16, false, false); }
    else      { if (relu && gout) SCN_BR(float, true, true); else if (relu) SCN_BR(float, true, false); else SCN_BR(float, false, false); }
#undef SCN_BR
    SCN_LAUNCH_CHECK();
    if (nchunk_out) *nchunk_out = nchunk;
    return 0;
}

int bn_bwd_dx_fin(hipStream_t st, long R, int C, const void* g, const void* z, int bf16, const float* mean, const float* invstd,
                  const float* gamma, const float* partial, int ldp, int nchunk, float* dbeta, float* dgamma, void* dz) {
    SCN_ARG(R > 0 && C > 0 && C % 4 == 0 && g && z && mean && invstd && gamma && partial && ldp >= nchunk && ldp % 4 == 0 &&
            nchunk > 0 && dbeta && dgamma && dz && aligned16(partial), "bn_bwd_dx_fin: bad argument");
    SCN_ARG(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dz)) & (bf16 ? 7u : 15u)) == 0,
            "bn_bwd_dx_fin: maps are not vector aligned");
    SCN_ARG(2L * C * ldp * 4 < 0x7fffffffL, "bn_bwd_dx_fin: partial too large");
    int rpc;
    const int nch = ew_chunks(R, C, &rpc);
    if (bf16) hipLaunchKernelGGL(bn_bwd_dx_fin_kernel<__bf16>, dim3(cdiv(C, 64), nch), dim3(256), 0, st, R, C, rpc, (const __bf16*)g, (const __bf16*)z,
                                 mean, invstd, gamma, partial, ldp, nchunk, dbeta, dgamma, (__bf16*)dz);
    else      hipLaunchKernelGGL(bn_bwd_dx_fin_kernel<float>, dim3(cdiv(C, 64), nch), dim3(256), 0, st, R, C, rpc, (const float*)g, (const float*)z,
                                 mean, invstd, gamma, partial, ldp, nchunk, dbeta, dgamma, (float*)dz);
    SCN_LAUNCH_CHECK();
    return 0;
}

int bn_bwd(hipStream_t st, int R, int C, const void* dy, const void* y, const void* z, int bf16, const float* mean,
           const float* invstd, const float* gamma, const float* beta, int relu, int train, float* partial, float* dbeta,
           float* dgamma, void* dz, void* dres) {
    SCN_ARG(R > 0 && C > 0 && C % 4 == 0 && dy && z && mean && invstd && gamma && partial && dbeta && dgamma,
            "bn_bwd: bad argument");
    SCN_ARG(!relu || y || beta, "bn_bwd: relu needs the forward output y, or beta to recompute the mask from z");
    if (bf16) return bn_bwd_t<__bf16>(st, R, C, (const __bf16*)dy, (const __bf16*)y, (const __bf16*)z, mean, invstd, gamma, beta, relu,
                                      train, partial, dbeta, dgamma, (__bf16*)dz, (__bf16*)dres);
    return bn_bwd_t<float>(st, R, C, (const float*)dy, (const float*)y, (const float*)z, mean, invstd, gamma, beta, relu, train,
                           partial, dbeta, dgamma, (float*)dz, (float*)dres);
}

}  // namespace scn
