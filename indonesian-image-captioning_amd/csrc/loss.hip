// The loss of the train step (trains/attention_scn.py:222-236) as five launches instead of ~130:
//     scores  = pack_padded_sequence(scores,  decode_lengths, batch_first=True).data
//     targets = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
//     loss    = CrossEntropyLoss()(scores, targets)                       # mean over N = sum(decode_lengths) rows
//     loss   += alpha_c * ((1. - alphas.sum(dim=1)) ** 2).mean()          # mean over B x P
// Packing only selects the rows (b, t < decode_length[b]) of the (B, T, V) score tensor; the mean does not
// care about their order.  So the cross-entropy runs on the tensor where it lies (no packed copy, whose
// backward alone is 51 row-block copies + a 65 MB zero fill at B=32, T=51, V=10000), one workgroup per
// (b, t) row with an online softmax (one read pass), and the backward writes d scores in place of the
// unpack: (softmax - onehot) * g / N for decoded rows, 0 elsewhere -- the (B*T, V) matrix the fc GEMMs of
// scnattn_seq_bwd consume.  Sums are formed in a fixed order (deterministic).  HBM-bound: forward reads
// 4*B*T*V bytes once, backward reads them once more and writes as many.
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void online(float& m, float& s, float x) {
    if (x > m) {
        s = s * expf(m - x) + 1.f;
        m = x;
    } else {
        s += expf(x - m);
    }
}

__device__ __forceinline__ void merge(float& m, float& s, float m2, float s2) {
    const float mm = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mm));
    m = mm;
}

// one workgroup per (b, t) row of scores
template <bool VEC>
__global__ __launch_bounds__(256) void ce_fwd_kernel(int T, int V, const float* __restrict__ scores,
                                                     const long long* __restrict__ targets, long ldt,
                                                     const int* __restrict__ dl, float* __restrict__ row_lse,
                                                     float* __restrict__ row_loss) {
    const int r = blockIdx.x, b = r / T, t = r - b * T;
    if (t >= dl[b]) {                         // not decoded: not part of the packed batch
        if (threadIdx.x == 0) { row_lse[r] = 0.f; row_loss[r] = 0.f; }
        return;
    }
    const float* x = scores + (long)r * V;
    float m = -INFINITY, s = 0.f;
    if (VEC) {
        for (int v = threadIdx.x * 4; v < V; v += 1024) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(x + v);
            online(m, s, q[0]); online(m, s, q[1]); online(m, s, q[2]); online(m, s, q[3]);
        }
    } else {
        for (int v = threadIdx.x; v < V; v += 256) online(m, s, x[v]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) merge(m, s, __shfl_xor(m, o), __shfl_xor(s, o));
    __shared__ float sm[4], ss[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = m; ss[w] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = sm[0]; s = ss[0];
        merge(m, s, sm[1], ss[1]); merge(m, s, sm[2], ss[2]); merge(m, s, sm[3], ss[3]);
        const float lse = m + logf(s);
        const long long tg = targets[(long)b * ldt + t];
        row_lse[r] = lse;
        row_loss[r] = (tg >= 0 && tg < V) ? lse - x[tg] : __builtin_nanf("");   // bad label poisons the loss, no fault
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void ce_bwd_kernel(int T, int V, const float* __restrict__ scores,
                                                     const long long* __restrict__ targets, long ldt,
                                                     const int* __restrict__ dl, const float* __restrict__ row_lse,
                                                     const float* __restrict__ gout, float inv_n,
                                                     float* __restrict__ dscores) {
    const int r = blockIdx.x, b = r / T, t = r - b * T;
    float* d = dscores + (long)r * V;
    const bool on = t < dl[b];
    const float* x = scores + (long)r * V;
    const float lse = on ? row_lse[r] : 0.f;
    const float g = on ? gout[0] * inv_n : 0.f;
    const int tg = on ? (int)targets[(long)b * ldt + t] : -1;
    if (VEC) {
        for (int v = threadIdx.x * 4; v < V; v += 1024) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (on) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(x + v);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (expf(q[j] - lse) - (v + j == tg ? 1.f : 0.f)) * g;
            }
            *reinterpret_cast<f32x4*>(d + v) = o;
        }
    } else {
        for (int v = threadIdx.x; v < V; v += 256) d[v] = on ? (expf(x[v] - lse) - (v == tg ? 1.f : 0.f)) * g : 0.f;
    }
}

// sm1[b][p] = sum_t alphas[b][t][p] - 1 ; reg_part[b] = sum_p sm1^2
__global__ __launch_bounds__(256) void alpha_reg_fwd_kernel(int T, int P, const float* __restrict__ alphas,
                                                            float* __restrict__ sm1, float* __restrict__ reg_part) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += alphas[((long)b * T + t) * P + p];
        s -= 1.f;
        sm1[(long)b * P + p] = s;
        acc += s * s;
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) reg_part[b] = (part[0] + part[1]) + (part[2] + part[3]);
}

// loss = sum(row_loss)/N + alpha_c * sum(reg_part)/(B*P); one workgroup, fixed order
__global__ __launch_bounds__(256) void loss_finalize_kernel(long nrows, const float* __restrict__ row_loss, float inv_n,
                                                            int B, const float* __restrict__ reg_part, float reg_scale,
                                                            float* __restrict__ loss) {
    float a = 0.f, c = 0.f;
    for (long i = threadIdx.x; i < nrows; i += 256) a += row_loss[i];
    if (reg_part)
        for (int i = threadIdx.x; i < B; i += 256) c += reg_part[i];
    a = wave_sum(a);
    c = wave_sum(c);
    __shared__ float pa[4], pc[4];
    if ((threadIdx.x & 63) == 0) { pa[threadIdx.x >> 6] = a; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0)
        loss[0] = ((pa[0] + pa[1]) + (pa[2] + pa[3])) * inv_n + ((pc[0] + pc[1]) + (pc[2] + pc[3])) * reg_scale;
}

// dalphas[b][t][p] = g * alpha_c * 2 * sm1[b][p] / (B*P)   (every t: alphas.sum(dim=1) runs over all T)
__global__ __launch_bounds__(256) void alpha_reg_bwd_kernel(long n, int T, int P, const float* __restrict__ sm1,
                                                            const float* __restrict__ gout, float scale,
                                                            float* __restrict__ dalphas) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long b = i / ((long)T * P);
    const int p = (int)(i % P);
    dalphas[i] = gout[0] * scale * sm1[b * P + p];
}

}  // namespace

int caption_loss_fwd(hipStream_t st, int B, int T, int V, int P, const float* scores, const long long* targets, long ldt,
                     const int* dl, long n_tokens, const float* alphas, float alpha_c, float* row_lse, float* row_loss,
                     float* sm1, float* reg_part, float* loss) {
    SCN_ARG(B > 0 && T > 0 && V > 0 && n_tokens > 0, "caption_loss_fwd: bad shape");
    SCN_ARG(scores && targets && dl && row_lse && row_loss && loss && ldt >= T, "caption_loss_fwd: bad argument");
    SCN_ARG(!alphas || (P > 0 && sm1 && reg_part), "caption_loss_fwd: alphas without workspace");
    const bool vec = (V % 4 == 0) && aligned16(scores);
    if (vec) hipLaunchKernelGGL(ce_fwd_kernel<true>, dim3(B * T), dim3(256), 0, st, T, V, scores, targets, ldt, dl, row_lse, row_loss);
    else     hipLaunchKernelGGL(ce_fwd_kernel<false>, dim3(B * T), dim3(256), 0, st, T, V, scores, targets, ldt, dl, row_lse, row_loss);
    SCN_LAUNCH_CHECK();
    if (alphas) {
        hipLaunchKernelGGL(alpha_reg_fwd_kernel, dim3(B), dim3(256), 0, st, T, P, alphas, sm1, reg_part);
        SCN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (long)B * T, row_loss, 1.f / (float)n_tokens, B,
                       alphas ? reg_part : nullptr, alphas ? alpha_c / ((float)B * (float)P) : 0.f, loss);
    SCN_LAUNCH_CHECK();
    return 0;
}

int caption_loss_bwd(hipStream_t st, int B, int T, int V, int P, const float* scores, const long long* targets, long ldt,
                     const int* dl, long n_tokens, const float* row_lse, const float* sm1, float alpha_c,
                     const float* gout, float* dscores, float* dalphas) {
    SCN_ARG(B > 0 && T > 0 && V > 0 && n_tokens > 0, "caption_loss_bwd: bad shape");
    SCN_ARG(scores && targets && dl && row_lse && gout && dscores && ldt >= T, "caption_loss_bwd: bad argument");
    SCN_ARG(!dalphas || (P > 0 && sm1), "caption_loss_bwd: dalphas without sm1");
    const bool vec = (V % 4 == 0) && aligned16(scores) && aligned16(dscores);
    const float inv_n = 1.f / (float)n_tokens;
    if (vec) hipLaunchKernelGGL(ce_bwd_kernel<true>, dim3(B * T), dim3(256), 0, st, T, V, scores, targets, ldt, dl, row_lse, gout, inv_n, dscores);
    else     hipLaunchKernelGGL(ce_bwd_kernel<false>, dim3(B * T), dim3(256), 0, st, T, V, scores, targets, ldt, dl, row_lse, gout, inv_n, dscores);
    SCN_LAUNCH_CHECK();
    if (dalphas) {
        const long n = (long)B * T * P;
        hipLaunchKernelGGL(alpha_reg_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, n, T, P, sm1, gout,
                           2.f * alpha_c / ((float)B * (float)P), dalphas);
        SCN_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace scn
