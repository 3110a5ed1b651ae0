// Data-movement and reduction helpers around the hot kernels: weight re-layout for the streaming
// GEMMs, column sums for bias gradients, embedding gather/scatter in time-major order, the
// dropout/transposition between the time-major recurrence and the batch-major `fc` GEMM, the
// encoder's AdaptiveAvgPool2d(14)+permute (models/encoders/caption.py:41-43), and the fused
// clamp(+-c) + Adam update (utils/optimizer.py:1-11 + torch.optim.Adam, trains/attention_scn.py:244-252).
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

__global__ __launch_bounds__(256) void transpose2d_kernel(int R, int C, const float* __restrict__ in, long ldi,
                                                          float* __restrict__ out, long ldo) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(long)r * ldi + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < R) out[(long)c * ldo + r] = tile[tx][ty + 8 * i];
    }
}

__global__ __launch_bounds__(256) void copy2d_kernel(int R, int C, const float* __restrict__ in, long ldi,
                                                     float* __restrict__ out, long ldo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)R * C) return;
    const int r = (int)(i / C), c = (int)(i - (long)r * C);
    out[(long)r * ldo + c] = in[(long)r * ldi + c];
}

// out[n] = beta*out[n] + sum_r X[r][n]; one workgroup per 16 columns x 16 row lanes, 4 independent
// loads in flight per thread, fixed summation order (deterministic).
template <bool MASK>
__global__ __launch_bounds__(256) void colsum_kernel(int R, int N, const float* __restrict__ X, long ld,
                                                     const float* __restrict__ rowmask, float* __restrict__ out,
                                                     float beta) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    const int nc = min(n, N - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rl;
    for (; r + 48 < R; r += 64) {
        float a0 = X[(long)r * ld + nc], a1 = X[(long)(r + 16) * ld + nc];
        float a2 = X[(long)(r + 32) * ld + nc], a3 = X[(long)(r + 48) * ld + nc];
        if (MASK) {      // rows with mask 0 do not contribute whatever they hold (select, not multiply)
            a0 = rowmask[r] != 0.f ? a0 : 0.f;
            a1 = rowmask[r + 16] != 0.f ? a1 : 0.f;
            a2 = rowmask[r + 32] != 0.f ? a2 : 0.f;
            a3 = rowmask[r + 48] != 0.f ? a3 : 0.f;
        }
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; r < R; r += 16) {
        const float a = X[(long)r * ld + nc];
        s0 += (!MASK || rowmask[r] != 0.f) ? a : 0.f;
    }
    part[rl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && n < N) {
        float v = part[0][c];
#pragma unroll
        for (int i = 1; i < 16; ++i) v += part[i][c];
        out[n] = (beta != 0.f ? beta * out[n] : 0.f) + v;
    }
}

// out_tm[t][b][:] = table[caps[b][t]][:]   (embedding lookup, attention_scn.py:124, time-major)
__global__ __launch_bounds__(256) void gather_rows_kernel(int B, int T, int L, int M, const long long* __restrict__ caps,
                                                          const float* __restrict__ table, int V,
                                                          float* __restrict__ out) {
    const int row = blockIdx.x;  // t*B + b
    const int t = row / B, b = row - t * B;
    long long tok = caps[(long)b * L + t];
    if (tok < 0) tok = 0;
    if (tok >= V) tok = V - 1;
    const float* src = table + tok * M;
    float* dst = out + (long)row * M;
    for (int m = threadIdx.x; m < M; m += 256) dst[m] = src[m];
}

// dtable[v][:] += sum over active cells (t < dl[b]) with caps[b][t] == v of demb_tm[t][b][:].
// One workgroup per vocabulary row scans the (small) token list in a fixed order, so the result is
// bitwise reproducible -- no float atomics (the reference's nn.Embedding backward is deterministic too).
__global__ __launch_bounds__(256) void mark_tokens_kernel(int B, int T, int L, const long long* __restrict__ caps,
                                                          const int* __restrict__ dl, int V, int* __restrict__ present) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= T * B) return;
    const int t = r / B, b = r - t * B;
    const long long tok = caps[(long)b * L + t];
    if (t < dl[b] && tok >= 0 && tok < V) present[tok] = 1;   // benign race: every writer stores 1
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int B, int T, int L, int M,
                                                               const long long* __restrict__ caps,
                                                               const int* __restrict__ dl,
                                                               const float* __restrict__ demb, int V,
                                                               const int* __restrict__ present,
                                                               float* __restrict__ dtable) {
    constexpr int MAXH = 1024;
    __shared__ int list[MAXH];
    __shared__ int nlist;
    const int v = blockIdx.x;
    const int TB = T * B;
    if (!present[v]) return;      // most vocabulary rows do not occur in the batch
    if (threadIdx.x == 0) nlist = 0;
    __syncthreads();
    // collect the cells that hold token v (append order is arbitrary) ...
    for (int r = threadIdx.x; r < TB; r += 256) {
        const int t = r / B, b = r - t * B;
        if (t < dl[b] && caps[(long)b * L + t] == v) {
            const int i = atomicAdd(&nlist, 1);
            if (i < MAXH) list[i] = r;
        }
    }
    __syncthreads();
    const int n = nlist;
    if (n > MAXH) {
        // a token filling more than MAXH cells (degenerate data): plain ordered scan, still deterministic
        for (int m = threadIdx.x; m < M; m += 256) {
            float acc = 0.f;
            for (int r = 0; r < TB; ++r) {
                const int t = r / B, b = r - t * B;
                if (t < dl[b] && caps[(long)b * L + t] == v) acc += demb[(long)r * M + m];
            }
            dtable[(long)v * M + m] += acc;
        }
        return;
    }
    // ... then sort them so that the summation order is fixed (bitwise reproducible result)
    if (threadIdx.x == 0) {
        for (int i = 1; i < n; ++i) {
            const int key = list[i];
            int j = i - 1;
            while (j >= 0 && list[j] > key) { list[j + 1] = list[j]; --j; }
            list[j + 1] = key;
        }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += 256) {
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += demb[(long)list[i] * M + m];
        dtable[(long)v * M + m] += acc;
    }
}

// out_bm[b][t][:] = (t < dl[b]) ? hs_tm[t][b][:] * mask[b][t][:] : 0 ; rowmask[b*T+t] = t < dl[b]
__global__ __launch_bounds__(256) void hidden_to_bm_kernel(int B, int T, int D, const int* __restrict__ dl,
                                                           const float* __restrict__ hs, const float* __restrict__ mask,
                                                           float* __restrict__ out, float* __restrict__ rowmask) {
    const int row = blockIdx.x;  // b*T + t
    const int b = row / T, t = row - b * T;
    const bool act = t < dl[b];
    if (threadIdx.x == 0 && rowmask) rowmask[row] = act ? 1.f : 0.f;
    const float* src = hs + ((long)t * B + b) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        if (act) v = src[d] * (mask ? mask[(long)row * D + d] : 1.f);
        out[(long)row * D + d] = v;
    }
}

__global__ __launch_bounds__(256) void hidden_from_bm_kernel(int B, int T, int D, const int* __restrict__ dl,
                                                             const float* __restrict__ dbm, const float* __restrict__ mask,
                                                             float* __restrict__ out) {
    const int row = blockIdx.x;  // t*B + b
    const int t = row / B, b = row - t * B;
    const bool act = t < dl[b];
    const long src = ((long)b * T + t) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        if (act) v = dbm[src + d] * (mask ? mask[src + d] : 1.f);
        out[(long)row * D + d] = v;
    }
}

__global__ __launch_bounds__(256) void add_bcast_rows_kernel(int B, int P, int E, const float* __restrict__ v,
                                                             float scale, float* __restrict__ x) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * P * E) return;
    const int e = (int)(i % E);
    const int b = (int)(i / ((long)P * E));
    x[i] += scale * v[(long)b * E + e];
}

// AdaptiveAvgPool2d window of output index o over n_in -> n_out: [floor(o*n_in/n_out), ceil((o+1)*n_in/n_out))
__device__ __forceinline__ void pool_window(int o, int n_in, int n_out, int& lo, int& hi) {
    lo = (o * n_in) / n_out;
    hi = ((o + 1) * n_in + n_out - 1) / n_out;
}

// y[b][oh][ow][c] = mean over the window of x[b][c][:, :]; x addressed through explicit strides so
// both NCHW-contiguous and channels-last trunks are read in place.  Threads run along c (the
// contiguous dimension of y, and of a channels-last x).
__global__ __launch_bounds__(256) void pool_permute_fwd_kernel(int B, int C, int Hin, int Win, int Ho, int Wo,
                                                               const float* __restrict__ x, long sxb, long sxc,
                                                               long sxh, long sxw, float* __restrict__ y) {
    const int cell = blockIdx.x;  // (b*Ho + oh)*Wo + ow
    const int ow = cell % Wo, oh = (cell / Wo) % Ho, b = cell / (Wo * Ho);
    int h0, h1, w0, w1;
    pool_window(oh, Hin, Ho, h0, h1);
    pool_window(ow, Win, Wo, w0, w1);
    const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) s += x[b * sxb + c * sxc + h * sxh + w * sxw];
        y[(long)cell * C + c] = s * inv;
    }
}

// dx[b][c][h][w] = sum over output cells whose window contains (h,w) of dy/|window|
__global__ __launch_bounds__(256) void pool_permute_bwd_kernel(int B, int C, int Hin, int Win, int Ho, int Wo,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               long sxb, long sxc, long sxh, long sxw) {
    const int cell = blockIdx.x;  // (b*Hin + h)*Win + w
    const int w = cell % Win, h = (cell / Win) % Hin, b = cell / (Win * Hin);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int oh = 0; oh < Ho; ++oh) {
            int h0, h1;
            pool_window(oh, Hin, Ho, h0, h1);
            if (h < h0 || h >= h1) continue;
            for (int ow = 0; ow < Wo; ++ow) {
                int w0, w1;
                pool_window(ow, Win, Wo, w0, w1);
                if (w < w0 || w >= w1) continue;
                s += dy[(((long)b * Ho + oh) * Wo + ow) * C + c] / (float)((h1 - h0) * (w1 - w0));
            }
        }
        dx[b * sxb + c * sxc + h * sxh + w * sxw] = s;
    }
}

// g <- clamp(g*gscale, -clip, clip); Adam(m, v, p) with bias correction, one pass over a flat buffer
__global__ __launch_bounds__(256) void clamp_adam_kernel(long n, float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, float lr,
                                                         float b1, float b2, float eps, float bc1, float sqrt_bc2,
                                                         float clip, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i] * gscale;
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// out[t][b][n] = x[t][b][n] * q[b][n]
__global__ __launch_bounds__(256) void mul_bcast_kernel(int T, int B, int N, const float* __restrict__ x,
                                                        const float* __restrict__ q, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * B * N) return;
    out[i] = x[i] * q[i % ((long)B * N)];
}

// fp32 -> bf16, round to nearest even (NaN kept quiet): the storage conversion of the bf16 decoder mode
__device__ __forceinline__ unsigned f2bf(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(long n4, const float* __restrict__ in, bf16_t* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + 4 * i);
        u32x2 o;
        o[0] = f2bf(v[0]) | (f2bf(v[1]) << 16);
        o[1] = f2bf(v[2]) | (f2bf(v[3]) << 16);
        *reinterpret_cast<u32x2*>(out + 4 * i) = o;
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(int rows, int N, Slabs s, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * N) return;
    const int r = (int)(i / N), c = (int)(i - (long)r * N);
    out[i] = slab_sum(s.p, (long)r * s.ld + c, s.n, s.stride);
}

}  // namespace

int mul_bcast(hipStream_t st, int T, int B, int N, const float* x, const float* q, float* out) {
    if (T <= 0 || B <= 0 || N <= 0) return 0;
    SCN_ARG(x && q && out, "mul_bcast: null operand");
    hipLaunchKernelGGL(mul_bcast_kernel, dim3(cdiv((long)T * B * N, 256)), dim3(256), 0, st, T, B, N, x, q, out);
    SCN_LAUNCH_CHECK();
    return 0;
}

int f32_to_bf16(hipStream_t st, long n, const float* in, void* out) {
    if (n <= 0) return 0;
    SCN_ARG(in && out && n % 4 == 0 && aligned16(in) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
            "f32_to_bf16: n must be a multiple of 4 and the buffers vector aligned");
    const long n4 = n / 4;
    const int blocks = (int)(n4 / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, st, n4, in, reinterpret_cast<bf16_t*>(out));
    SCN_LAUNCH_CHECK();
    return 0;
}

int reduce_slabs(hipStream_t st, int rows, int N, Slabs s, float* out) {
    if (rows <= 0 || N <= 0) return 0;
    SCN_ARG(s.p && out && s.n >= 1, "reduce_slabs: bad argument");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv((long)rows * N, 256)), dim3(256), 0, st, rows, N, s, out);
    SCN_LAUNCH_CHECK();
    return 0;
}

int transpose2d(hipStream_t st, int R, int C, const float* in, long ldi, float* out, long ldo) {
    if (R <= 0 || C <= 0) return 0;
    SCN_ARG(in && out, "transpose2d: null operand");
    hipLaunchKernelGGL(transpose2d_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, st, R, C, in, ldi, out, ldo);
    SCN_LAUNCH_CHECK();
    return 0;
}

int copy2d(hipStream_t st, int R, int C, const float* in, long ldi, float* out, long ldo) {
    if (R <= 0 || C <= 0) return 0;
    SCN_ARG(in && out, "copy2d: null operand");
    hipLaunchKernelGGL(copy2d_kernel, dim3(cdiv((long)R * C, 256)), dim3(256), 0, st, R, C, in, ldi, out, ldo);
    SCN_LAUNCH_CHECK();
    return 0;
}

int colsum(hipStream_t st, int R, int N, const float* X, long ld, float* out, float beta) {
    if (N <= 0) return 0;
    SCN_ARG(X && out && R >= 0, "colsum: bad argument");
    hipLaunchKernelGGL(colsum_kernel<false>, dim3(cdiv(N, 16)), dim3(256), 0, st, R, N, X, ld, nullptr, out, beta);
    SCN_LAUNCH_CHECK();
    return 0;
}

// out[n] = beta*out[n] + sum over rows with rowmask[r] != 0 of X[r][n]
int colsum_masked(hipStream_t st, int R, int N, const float* X, long ld, const float* rowmask, float* out, float beta) {
    if (N <= 0) return 0;
    SCN_ARG(X && out && rowmask && R >= 0, "colsum_masked: bad argument");
    hipLaunchKernelGGL(colsum_kernel<true>, dim3(cdiv(N, 16)), dim3(256), 0, st, R, N, X, ld, rowmask, out, beta);
    SCN_LAUNCH_CHECK();
    return 0;
}

int gather_rows_tm(hipStream_t st, int B, int T, int L, int M, const long long* caps, const float* table,
                   int V, float* out_tm) {
    if (B <= 0 || T <= 0) return 0;
    SCN_ARG(caps && table && out_tm && T <= L && V > 0, "gather_rows_tm: bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B * T), dim3(256), 0, st, B, T, L, M, caps, table, V, out_tm);
    SCN_LAUNCH_CHECK();
    return 0;
}

int scatter_add_rows_tm(hipStream_t st, int B, int T, int L, int M, const long long* caps, const int* dl,
                        const float* demb_tm, int V, float* dtable, int* present) {
    if (B <= 0 || T <= 0) return 0;
    SCN_ARG(caps && dl && demb_tm && dtable && present && T <= L, "scatter_add_rows_tm: bad argument");
    SCN_HIP(hipMemsetAsync(present, 0, sizeof(int) * V, st));
    hipLaunchKernelGGL(mark_tokens_kernel, dim3(cdiv((long)T * B, 256)), dim3(256), 0, st, B, T, L, caps, dl, V, present);
    SCN_LAUNCH_CHECK();
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(V), dim3(256), 0, st, B, T, L, M, caps, dl, demb_tm, V, present,
                       dtable);
    SCN_LAUNCH_CHECK();
    return 0;
}

int hidden_to_bm(hipStream_t st, int B, int T, int D, const int* dl, const float* hs_tm,
                 const float* mask_bm, float* out_bm, float* rowmask) {
    if (B <= 0 || T <= 0) return 0;
    SCN_ARG(dl && hs_tm && out_bm, "hidden_to_bm: null operand");
    hipLaunchKernelGGL(hidden_to_bm_kernel, dim3(B * T), dim3(256), 0, st, B, T, D, dl, hs_tm, mask_bm, out_bm, rowmask);
    SCN_LAUNCH_CHECK();
    return 0;
}

int hidden_from_bm(hipStream_t st, int B, int T, int D, const int* dl, const float* dbm,
                   const float* mask_bm, float* out_tm) {
    if (B <= 0 || T <= 0) return 0;
    SCN_ARG(dl && dbm && out_tm, "hidden_from_bm: null operand");
    hipLaunchKernelGGL(hidden_from_bm_kernel, dim3(B * T), dim3(256), 0, st, B, T, D, dl, dbm, mask_bm, out_tm);
    SCN_LAUNCH_CHECK();
    return 0;
}

int add_bcast_rows(hipStream_t st, int B, int P, int E, const float* v, float scale, float* x) {
    if (B <= 0) return 0;
    SCN_ARG(v && x, "add_bcast_rows: null operand");
    hipLaunchKernelGGL(add_bcast_rows_kernel, dim3(cdiv((long)B * P * E, 256)), dim3(256), 0, st, B, P, E, v, scale, x);
    SCN_LAUNCH_CHECK();
    return 0;
}

int pool_permute_fwd(hipStream_t st, int B, int C, int Hin, int Win, int Ho, int Wo, const float* x,
                     long sxb, long sxc, long sxh, long sxw, float* y) {
    if (B <= 0) return 0;
    SCN_ARG(x && y && C > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0, "pool_permute_fwd: bad argument");
    hipLaunchKernelGGL(pool_permute_fwd_kernel, dim3(B * Ho * Wo), dim3(256), 0, st, B, C, Hin, Win, Ho, Wo, x, sxb,
                       sxc, sxh, sxw, y);
    SCN_LAUNCH_CHECK();
    return 0;
}

int pool_permute_bwd(hipStream_t st, int B, int C, int Hin, int Win, int Ho, int Wo, const float* dy,
                     float* dx, long sxb, long sxc, long sxh, long sxw) {
    if (B <= 0) return 0;
    SCN_ARG(dy && dx && C > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0, "pool_permute_bwd: bad argument");
    hipLaunchKernelGGL(pool_permute_bwd_kernel, dim3(B * Hin * Win), dim3(256), 0, st, B, C, Hin, Win, Ho, Wo, dy, dx,
                       sxb, sxc, sxh, sxw);
    SCN_LAUNCH_CHECK();
    return 0;
}

int clamp_adam(hipStream_t st, long n, float* p, const float* g, float* m, float* v, double lr, double b1,
               double b2, double eps, int step, double clip, double gscale) {
    if (n <= 0) return 0;
    SCN_ARG(p && g && m && v && step >= 1, "clamp_adam: bad argument");
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double sqrt_bc2 = sqrt(1.0 - pow(b2, (double)step));
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clamp_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, p, g, m, v, (float)lr, (float)b1,
                       (float)b2, (float)eps, (float)bc1, (float)sqrt_bc2, (float)clip, (float)gscale);
    SCN_LAUNCH_CHECK();
    return 0;
}

// ---- fp32 master weights -> bf16 operand copies of the mixed-precision trunk (one launch per step) ---------------------------
namespace {
__global__ __launch_bounds__(256) void bf16_weights_kernel(int n, const WeightDesc* __restrict__ desc, const int* __restrict__ prefix) {
    __shared__ float tile[32][33];
    // which weight owns this 32 x 32 tile: binary search over the exclusive prefix sums (wave-uniform)
    int lo = 0, hi = n - 1;
    const int t = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const WeightDesc d = desc[lo];
    int r = t - prefix[lo];
    const int tci = d.cin >> 5, tco = d.cout >> 5;
    const int tap = r / (tco * tci);
    r -= tap * tco * tci;
    const int co0 = (r / tci) << 5, ci0 = (r % tci) << 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads, 4 rows each
    const long ld = (long)d.taps * d.cin, ldt = (long)d.taps * d.cout;
    unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
    unsigned short* dstt = reinterpret_cast<unsigned short*>(d.dst_t);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = co0 + ty + 8 * k;
        const float v = d.src[(long)co * ld + (long)tap * d.cin + ci0 + tx];
        tile[ty + 8 * k][tx] = v;
        if (dst) dst[(long)co * ld + (long)tap * d.cin + ci0 + tx] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    __syncthreads();
    if (dstt) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ci0 + ty + 8 * k;
            dstt[(long)ci * ldt + (long)tap * d.cout + co0 + tx] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][ty + 8 * k]);
        }
    }
}
}  // namespace

int bf16_weights(hipStream_t st, int n, const WeightDesc* desc, const int* tile_prefix, int total_tiles) {
    if (n <= 0 || total_tiles <= 0) return 0;
    SCN_ARG(desc && tile_prefix, "bf16_weights: null table");
    hipLaunchKernelGGL(bf16_weights_kernel, dim3(total_tiles), dim3(256), 0, st, n, desc, tile_prefix);
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
