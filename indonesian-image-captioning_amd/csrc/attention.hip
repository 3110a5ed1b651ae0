// Soft-attention kernels (forward and backward) for the decode step.
// Reference math: models/attention.py:35-44 and the gate of models/decoders/attention_scn.py:147-148.
//
// The time-invariant projection att1 = encoder_att(enc) is computed ONCE per sequence by sgemm (the
// reference recomputes it every timestep); the kernels here are the HBM-bound streaming parts:
//   attn_scores   : reads att1 (B*P*A floats) once, one wave per pixel row, 16 B per lane
//   attn_context  : softmax + the weighted sum over the 14x14 grid; reads enc (B*P*E floats) once,
//                   a wave covers 1 KiB of one pixel row per load instruction, 8 waves interleave p
//   attn_dalpha   : backward mirror of attn_context (dot products of enc rows with dawe)
//   attn_softmax_bwd / attn_datt1_post : ReLU mask is RECOMPUTED from att1 + att2_t instead of
//                   storing the (b,196,512) activation per step as autograd does in the reference.
// None of these has data reuse across lanes, so operands go HBM -> VGPR directly; LDS only holds the
// per-row vectors every wave re-reads (att2, w, alpha, dawe).
#include "common.h"
#include "kernels.h"

namespace scn {

int g_attn_depth = 1;   // 1: deeper load batches in attn_context / attn_dalpha (option "attn_depth", A/B)

namespace {

constexpr int PC = 16;  // pixel rows per workgroup in the row-dot kernels (4 waves x 4 rows)

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    // red: >= 16 floats of LDS; all threads get the result
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// ------------------------------------------------------------------------------------------------
template <bool VEC, typename ET>
__global__ __launch_bounds__(256) void attn_scores_kernel(int rows, int P, int A, const ET* __restrict__ att1,
                                                          Slabs att2, const float* __restrict__ bd,
                                                          const float* __restrict__ w, const float* __restrict__ b0,
                                                          float* __restrict__ e, float* __restrict__ att2_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int A4 = (A + 3) & ~3;
    float* att2s = sm;
    float* ws = sm + A4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p0 = blockIdx.x * PC;
    // each wave: 4 pixel rows, independent accumulators (4 x 16 B loads in flight per lane)
    const ET* rowp[4];
    bool ok[4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + wave * 4 + j;
        ok[j] = p < P;
        rowp[j] = att1 + ((long)b * P + (ok[j] ? p : P - 1)) * A;
    }
    // The att1 rows do not depend on att2: with A <= 512 a lane's whole share of them (2 x 4 loads) goes in flight
    // BEFORE the att2 / w staging below, so the kernel pays one memory round trip, not two in a row.
    const bool early = VEC && A <= 512;
    f32x4 pre[2][4];
    if (early) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int a = min(lane * 4 + 256 * it, A - 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) pre[it][j] = ld4(rowp[j] + a);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep hipcc from sinking the loads below the staging loop
    }
    for (int a = tid; a < A4; a += 256) {
        float v = 0.f, wv = 0.f;
        if (a < A) {
            v = slab_sum(att2.p, (long)b * att2.ld + a, att2.n, att2.stride) + (bd ? bd[a] : 0.f);
            wv = w[a];
            if (blockIdx.x == 0 && att2_out) att2_out[(long)b * A + a] = v;
        }
        att2s[a] = v;
        ws[a] = wv;
    }
    __syncthreads();
    if (early) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int a = lane * 4 + 256 * it;
            if (a < A) {
                const f32x4 s2 = *reinterpret_cast<const f32x4*>(att2s + a);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(ws + a);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[j] = fmaf(fmaxf(pre[it][j][c] + s2[c], 0.f), ww[c], acc[j]);
            }
        }
    } else if (VEC) {
        for (int a = lane * 4; a < A; a += 256) {
            const f32x4 s2 = *reinterpret_cast<const f32x4*>(att2s + a);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(ws + a);
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ld4(rowp[j] + a);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[j] = fmaf(fmaxf(v[j][c] + s2[c], 0.f), ww[c], acc[j]);
        }
    } else {
        for (int a = lane; a < A; a += 64) {
            const float s2 = att2s[a], ww = ws[a];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(fmaxf(ld1(rowp[j] + a) + s2, 0.f), ww, acc[j]);
        }
    }
    const float bias0 = b0 ? b0[0] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float s = wave_sum(acc[j]);
        if (lane == 0 && ok[j]) e[(long)b * P + p0 + wave * 4 + j] = s + bias0;
    }
}

// ------------------------------------------------------------------------------------------------
// MODE 0: softmax-weighted sum (+ optional sigmoid gate).  MODE 1: plain mean over pixels.  MODE 2: sum with
// the given per-row weights `e` (not per batch row).
// POOLED (MODE 0): encoder_out is a fixed linear pooling of a smaller map x [B][Q][E] (scnattn_pool): the
// softmax still runs over the P pooled pixels, then alpha is folded back onto the Q source pixels,
// alphaq[q] = sum_k qtap_w[q][k] * alpha[qtap_idx[q][k]]  (deterministic gather), and the context is
// sum_q alphaq[q] * x[q] -- the same number by linearity, read from Q rows instead of P.
struct PoolQ {
    int Q, qtap_max;
    const int* qtap_idx;
    const float* qtap_w;
    float* alphaq_save;     // [rows][Q]
};

template <bool VEC, int MODE, int CU, bool POOLED, typename ET = float>
__global__ __launch_bounds__(512) void attn_context_kernel(int rows, int P, int E, const ET* __restrict__ enc,
                                                           const float* __restrict__ e, Slabs gpre,
                                                           const float* __restrict__ bbeta,
                                                           float* __restrict__ alpha_out, long alpha_ld,
                                                           float* __restrict__ alpha_save, float* __restrict__ awe,
                                                           float* __restrict__ gate, float* __restrict__ z, PoolQ pq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* part = sm;              // [8][256]
    float* red = sm + 8 * 256;     // [16]
    float* alph = red + 16;        // [P]   (then [Q] more when POOLED)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, e0 = blockIdx.x * 256;
    const int NR = POOLED ? pq.Q : P;          // rows of the map that is actually summed

    // The first batch of encoder rows does not depend on the softmax: put it in flight before the
    // (barrier-heavy) softmax prologue.  CU rows per wave per batch -> CU x 16 B loads in flight per lane;
    // a wave owns ceil(P/8) rows, and every extra batch is one more exposed memory latency, so the launcher
    // picks CU = 13 for P = 196 (25 rows per wave -> 2 batches instead of 4 with CU = 8).
    const int col = e0 + lane * 4;
    const ET* base = enc + (long)b * NR * E;
    const int cc = min(col, max(E - 4, 0));
    const bool cok = col < E;
    f32x4 v[CU];
    if (VEC) {
#pragma unroll
        for (int j = 0; j < CU; ++j) v[j] = ld4(base + (long)min(wave + 8 * j, NR - 1) * E + cc);
    }

    if (MODE == 0) {
        float m = -INFINITY;
        for (int p = tid; p < P; p += 512) m = fmaxf(m, e[(long)b * P + p]);
        m = block_reduce(m, red, true);
        float s = 0.f;
        for (int p = tid; p < P; p += 512) {
            const float ex = expf(e[(long)b * P + p] - m);
            alph[p] = ex;
            s += ex;
        }
        s = block_reduce(s, red, false);
        for (int p = tid; p < P; p += 512) {
            const float a = alph[p] / s;
            alph[p] = a;
            if (blockIdx.x == 0) {
                if (alpha_out) alpha_out[(long)b * alpha_ld + p] = a;
                if (alpha_save) alpha_save[(long)b * P + p] = a;
            }
        }
        __syncthreads();
    }
    if (MODE == 2) {
        for (int p = tid; p < NR; p += 512) alph[p] = e[p];
        __syncthreads();
    }
    const float* wts = alph;
    if (POOLED) {
        float* aq = alph + P;
        for (int q = tid; q < pq.Q; q += 512) {
            float a = 0.f;
            for (int k = 0; k < pq.qtap_max; ++k)      // unused taps carry weight 0: clamp the index, no branch
                a = fmaf(pq.qtap_w[q * pq.qtap_max + k], alph[max(pq.qtap_idx[q * pq.qtap_max + k], 0)], a);
            aq[q] = a;
            if (blockIdx.x == 0 && pq.alphaq_save) pq.alphaq_save[(long)b * pq.Q + q] = a;
        }
        __syncthreads();
        wts = aq;
    }

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
        for (int p = wave;;) {
            float al[CU];
#pragma unroll
            for (int j = 0; j < CU; ++j) {
                const int pp = p + 8 * j;
                al[j] = (pp < NR && cok) ? (MODE != 1 ? wts[min(pp, NR - 1)] : 1.f) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < CU; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(al[j], v[j][c], acc[c]);
            p += 8 * CU;
            if (p >= NR) break;
#pragma unroll
            for (int j = 0; j < CU; ++j)
                v[j] = ld4(base + (long)min(p + 8 * j, NR - 1) * E + cc);
        }
    } else {
        for (int p = wave; p < NR; p += 8) {
            const float al = MODE != 1 ? wts[p] : 1.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (col + c < E) acc[c] = fmaf(al, ld1(base + (long)p * E + col + c), acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[wave * 256 + lane * 4 + c] = acc[c];
    __syncthreads();
    if (tid < 256) {
        const int c = e0 + tid;
        if (c < E) {
            float a = part[tid];
#pragma unroll
            for (int w8 = 1; w8 < 8; ++w8) a += part[w8 * 256 + tid];
            if (MODE == 1) {
                awe[(long)b * E + c] = a / (float)P;
            } else if (MODE == 2) {
                awe[(long)b * E + c] = a;
            } else {
                awe[(long)b * E + c] = a;
                if (gpre.p) {
                    const float gp = slab_sum(gpre.p, (long)b * gpre.ld + c, gpre.n, gpre.stride) + (bbeta ? bbeta[c] : 0.f);
                    const float g = sigmoidf_(gp);
                    if (gate) gate[(long)b * E + c] = g;
                    if (z) z[(long)b * E + c] = g * a;
                } else if (z) {
                    z[(long)b * E + c] = a;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <bool VEC, int U, int RW, typename ET>
__global__ __launch_bounds__(256) void attn_dalpha_kernel(int rows, int P, int E, const ET* __restrict__ enc,
                                                          const float* __restrict__ dawe,
                                                          const float* __restrict__ dalpha_in, long din_ld,
                                                          float* __restrict__ dalpha) {
    // RW rows per wave, 4 waves: 4*RW rows per workgroup.  RW = 4 for the 196 pooled pixels; RW = 2 with
    // U = 8 for the 64 source pixels of the pooled path (twice the workgroups, a 2048-wide row in one batch).
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E4 = (E + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p0 = blockIdx.x * (4 * RW);
    const ET* rowp[RW];
    bool ok[RW];
    float acc[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int p = p0 + wave * RW + j;
        ok[j] = p < P;
        acc[j] = 0.f;
        rowp[j] = enc + ((long)b * P + (ok[j] ? p : P - 1)) * E;
    }
    if (VEC) {
        // U column chunks x RW rows = RW*U x 16 B loads in flight per lane; the first batch is issued before
        // the LDS staging of dawe (it does not depend on it)
        f32x4 v[U][RW];
        const int cmax = E - 4;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < RW; ++j) v[u][j] = ld4(rowp[j] + min(lane * 4 + 256 * u, cmax));
        for (int c = tid; c < E4; c += 256) sm[c] = c < E ? dawe[(long)b * E + c] : 0.f;
        __syncthreads();
        for (int c0 = lane * 4;;) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 256 * u;
                if (c < E) {
                    const f32x4 d = *reinterpret_cast<const f32x4*>(sm + c);
#pragma unroll
                    for (int j = 0; j < RW; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[j] = fmaf(v[u][j][k], d[k], acc[j]);
                }
            }
            c0 += 256 * U;
            if (c0 >= E) break;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < RW; ++j) v[u][j] = ld4(rowp[j] + min(c0 + 256 * u, cmax));
        }
    } else {
        for (int c = tid; c < E4; c += 256) sm[c] = c < E ? dawe[(long)b * E + c] : 0.f;
        __syncthreads();
        for (int c = lane; c < E; c += 64) {
            const float d = sm[c];
#pragma unroll
            for (int j = 0; j < RW; ++j) acc[j] = fmaf(ld1(rowp[j] + c), d, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const float s = wave_sum(acc[j]);
        const int p = p0 + wave * RW + j;
        if (lane == 0 && ok[j]) dalpha[(long)b * P + p] = s + (dalpha_in ? dalpha_in[(long)b * din_ld + p] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
struct DalphaTaps {      // pooled path: d alpha is gathered from the Q source-pixel dot products
    const int* tap_idx;  // [P][4] or NULL (dense dalpha)
    const float* tap_w;  // [P][4]
    const float* dalphaq;  // [rows][Q]
    int Q;
    const float* din;    // upstream d alphas [rows][din_ld] or NULL
    long din_ld;
};

template <bool VEC, typename ET>
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(int rows, int P, int A, const ET* __restrict__ att1,
                                                               const float* __restrict__ att2,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ dalpha,
                                                               float* __restrict__ de, float* __restrict__ datt2,
                                                               long datt2_ld, DalphaTaps dt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* part = sm;              // [16][64]
    float* red = sm + 16 * 64;     // [16]
    float* des = red + 16;         // [P]
    const int tid = threadIdx.x;
    const int b = blockIdx.y, a0 = blockIdx.x * 64;
    // The first batch of att1 rows depends on nothing the prologue computes: in flight before it (the softmax-backward
    // prologue is two block reductions deep), the next batch is requested while the current one is consumed.
    const ET* const base0 = att1 + (long)b * P * A;
    const int grp0 = tid >> 4, a_first = a0 + (tid & 15) * 4;
    f32x4 vpre[4];
    if (VEC) {
        const int ac0 = min(a_first, A - 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vpre[j] = ld4(base0 + (long)min(grp0 + 16 * j, P - 1) * A + ac0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float dot = 0.f;
    if (dt.tap_idx) {              // d alpha[p] = sum_k tap_w[p][k] * d alphaq[tap_idx[p][k]] (+ upstream d alphas)
        float* dq = part;          // [Q] staged first: the gather then needs no dependent global load
        for (int q = tid; q < dt.Q; q += 256) dq[q] = dt.dalphaq[(long)b * dt.Q + q];
        __syncthreads();
        for (int p = tid; p < P; p += 256) {
            const int4 ti = *reinterpret_cast<const int4*>(dt.tap_idx + p * 4);
            const f32x4 tw = *reinterpret_cast<const f32x4*>(dt.tap_w + p * 4);
            float da = dt.din ? dt.din[(long)b * dt.din_ld + p] : 0.f;
            da = fmaf(tw[0], dq[ti.x], da);
            da = fmaf(tw[1], dq[ti.y], da);
            da = fmaf(tw[2], dq[ti.z], da);
            da = fmaf(tw[3], dq[ti.w], da);
            des[p] = da;
            dot = fmaf(alpha[(long)b * P + p], da, dot);
        }
    } else {
        for (int p = tid; p < P; p += 256) {
            const float da = dalpha[(long)b * P + p];
            des[p] = da;
            dot = fmaf(alpha[(long)b * P + p], da, dot);
        }
    }
    dot = block_reduce(dot, red, false);
    for (int p = tid; p < P; p += 256) {
        const float d = alpha[(long)b * P + p] * (des[p] - dot);
        des[p] = d;
        if (blockIdx.x == 0 && de) de[(long)b * P + p] = d;
    }
    __syncthreads();
    const int grp = tid >> 4, q = tid & 15;
    const int a = a0 + q * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float s2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[c] = (a + c < A) ? att2[(long)b * A + a + c] : 0.f;
    const ET* base = att1 + (long)b * P * A;
    if (VEC) {
        const int ac = min(a, A - 4);
        const bool aok = a < A;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vpre[j];
        for (int p = grp; p < P; p += 64) {
            f32x4 nv[4];
            float d[4];
            const int pn = p + 64;
            if (pn < P) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nv[j] = ld4(base + (long)min(pn + 16 * j, P - 1) * A + ac);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pp = p + 16 * j;
                d[j] = (pp < P && aok) ? des[min(pp, P - 1)] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += (v[j][c] + s2[c] > 0.f) ? d[j] : 0.f;
            if (pn < P) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = nv[j];
            }
        }
    } else {
        for (int p = grp; p < P; p += 16) {
            const float d = des[p];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (a + c < A) acc[c] += (ld1(base + (long)p * A + a + c) + s2[c] > 0.f) ? d : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) part[grp * 64 + q * 4 + c] = acc[c];
    __syncthreads();
    if (tid < 64 && a0 + tid < A) {
        float s = part[tid];
#pragma unroll
        for (int g2 = 1; g2 < 16; ++g2) s += part[g2 * 64 + tid];
        datt2[(long)b * datt2_ld + a0 + tid] = w[a0 + tid] * s;
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int PC2 = 8;

template <typename ET>
__global__ __launch_bounds__(256) void attn_datt1_post_kernel(int B, int P, int A, int T, const int* __restrict__ dl,
                                                              const ET* __restrict__ att1,
                                                              const float* __restrict__ att2_all,
                                                              const float* __restrict__ de_all,
                                                              const float* __restrict__ w, float* __restrict__ datt1,
                                                              float* __restrict__ dwpart) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // de_s[T][PC2]
    const int tid = threadIdx.x;
    const int b = blockIdx.y, p0 = blockIdx.x * PC2;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    int Tb = dl[b];
    if (Tb > T) Tb = T;
    for (int i = tid; i < T * PC2; i += 256) {
        const int t = i / PC2, j = i - t * PC2;
        sm[i] = (t < Tb && p0 + j < P) ? de_all[((long)t * B + b) * P + p0 + j] : 0.f;
    }
    __syncthreads();
    for (int a = tid; a < A; a += 256) {
        float a1[PC2], acc[PC2];
#pragma unroll
        for (int j = 0; j < PC2; ++j) {
            a1[j] = ld1(att1 + ((long)b * P + min(p0 + j, P - 1)) * A + a);
            acc[j] = 0.f;
        }
        float dwacc = 0.f;
        for (int t = 0; t < Tb; ++t) {
            const float s2 = att2_all[((long)t * B + b) * A + a];
#pragma unroll
            for (int j = 0; j < PC2; ++j) {
                const float s = a1[j] + s2;
                const float d = sm[t * PC2 + j];
                if (s > 0.f) {
                    acc[j] += d;
                    dwacc = fmaf(d, s, dwacc);
                }
            }
        }
        const float wa = w[a];
#pragma unroll
        for (int j = 0; j < PC2; ++j)
            if (p0 + j < P) datt1[((long)b * P + p0 + j) * A + a] = wa * acc[j];
        dwpart[(long)blk * (A + 1) + a] = dwacc;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < Tb * PC2; ++i) s += sm[i];
        dwpart[(long)blk * (A + 1) + A] = s;
    }
}

}  // namespace

// ================================================================================================
// bf16 storage: vector path = 8-byte loads of 4 elements
static inline bool vec_ok(const void* p, int n, bool bf) {
    return n % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & (bf ? 7u : 15u)) == 0;
}
#define F32P(p) reinterpret_cast<const float*>(p)
#define BF16P(p) reinterpret_cast<const bf16_t*>(p)

int attn_scores(hipStream_t st, int rows, int P, int A, const void* att1, Slabs att2, const float* bd,
                const float* w, const float* b0, float* e, float* att2_out, bool bf) {
    if (rows <= 0) return 0;
    SCN_ARG(att1 && att2.p && w && e && P > 0 && A > 0, "attn_scores: bad argument");
    dim3 grid(cdiv(P, PC), rows), block(256);
    const size_t lds = 2 * ((A + 3) & ~3) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "attn_scores: attention_dim too large for the LDS staging");
    const bool vec = vec_ok(att1, A, bf);
    if (bf) {
        if (vec) hipLaunchKernelGGL((attn_scores_kernel<true, bf16_t>), grid, block, lds, st, rows, P, A, BF16P(att1), att2, bd, w, b0, e, att2_out);
        else     hipLaunchKernelGGL((attn_scores_kernel<false, bf16_t>), grid, block, lds, st, rows, P, A, BF16P(att1), att2, bd, w, b0, e, att2_out);
    } else {
        if (vec) hipLaunchKernelGGL((attn_scores_kernel<true, float>), grid, block, lds, st, rows, P, A, F32P(att1), att2, bd, w, b0, e, att2_out);
        else     hipLaunchKernelGGL((attn_scores_kernel<false, float>), grid, block, lds, st, rows, P, A, F32P(att1), att2, bd, w, b0, e, att2_out);
    }
    SCN_LAUNCH_CHECK();
    return 0;
}

int attn_context(hipStream_t st, int rows, int P, int E, const void* enc_, const float* e, Slabs gpre,
                 const float* bbeta, float* alpha_out, long alpha_ld, float* alpha_save, float* awe,
                 float* gate, float* z, bool bf) {
    if (bf) {     // bf16 storage: the dense map takes the pooled kernel's code path with an identity-free PoolQ
        if (rows <= 0) return 0;
        SCN_ARG(enc_ && e && awe && P > 0 && E > 0, "attn_context: bad argument");
        dim3 grid(cdiv(E, 256), rows), block(512);
        const size_t lds = (8 * 256 + 16 + P) * sizeof(float);
        SCN_ARG(lds <= 64 * 1024, "attn_context: num_pixels too large for the LDS staging");
        const PoolQ none{0, 0, nullptr, nullptr, nullptr};
        if (vec_ok(enc_, E, true))
            hipLaunchKernelGGL((attn_context_kernel<true, 0, 13, false, bf16_t>), grid, block, lds, st, rows, P, E, BF16P(enc_), e,
                               gpre, bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, none);
        else
            hipLaunchKernelGGL((attn_context_kernel<false, 0, 8, false, bf16_t>), grid, block, lds, st, rows, P, E, BF16P(enc_), e,
                               gpre, bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, none);
        SCN_LAUNCH_CHECK();
        return 0;
    }
    const float* enc = F32P(enc_);
    if (rows <= 0) return 0;
    SCN_ARG(enc && e && awe && P > 0 && E > 0, "attn_context: bad argument");
    dim3 grid(cdiv(E, 256), rows), block(512);
    const size_t lds = (8 * 256 + 16 + P) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "attn_context: num_pixels too large for the LDS staging");
    const int rpw = cdiv(P, 8);          // encoder rows per wave
    const bool deep = g_attn_depth && (rpw > 16 || (rpw > 8 && rpw <= 13));
    const PoolQ none{0, 0, nullptr, nullptr, nullptr};
    if (E % 4 == 0 && aligned16(enc)) {
        if (deep)
            hipLaunchKernelGGL((attn_context_kernel<true, 0, 13, false>), grid, block, lds, st, rows, P, E, enc, e, gpre,
                               bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, none);
        else
            hipLaunchKernelGGL((attn_context_kernel<true, 0, 8, false>), grid, block, lds, st, rows, P, E, enc, e, gpre,
                               bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, none);
    } else {
        hipLaunchKernelGGL((attn_context_kernel<false, 0, 8, false>), grid, block, lds, st, rows, P, E, enc, e, gpre,
                           bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, none);
    }
    SCN_LAUNCH_CHECK();
    return 0;
}

// attn_context over the un-pooled map x [rows][Q][E] (see PoolQ)
int attn_context_pooled(hipStream_t st, int rows, int P, int E, const void* x_, const PoolDesc& pool, const float* e,
                        Slabs gpre, const float* bbeta, float* alpha_out, long alpha_ld, float* alpha_save,
                        float* alphaq_save, float* awe, float* gate, float* z, bool bf) {
    const float* x = F32P(x_);
    if (rows <= 0) return 0;
    SCN_ARG(x && e && awe && P > 0 && E > 0 && pool.Q > 0 && pool.qtap_idx && pool.qtap_w, "attn_context_pooled: bad argument");
    dim3 grid(cdiv(E, 256), rows), block(512);
    const size_t lds = (8 * 256 + 16 + P + pool.Q) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "attn_context: num_pixels too large for the LDS staging");
    const PoolQ pq{pool.Q, pool.qtap_max, pool.qtap_idx, pool.qtap_w, alphaq_save};
    const int rpw = cdiv(pool.Q, 8);
    const bool deep = g_attn_depth && (rpw > 16 || (rpw > 8 && rpw <= 13));
    if (bf) {
        if (vec_ok(x_, E, true))
            hipLaunchKernelGGL((attn_context_kernel<true, 0, 8, true, bf16_t>), grid, block, lds, st, rows, P, E, BF16P(x_), e,
                               gpre, bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, pq);
        else
            hipLaunchKernelGGL((attn_context_kernel<false, 0, 8, true, bf16_t>), grid, block, lds, st, rows, P, E, BF16P(x_), e,
                               gpre, bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, pq);
    } else if (E % 4 == 0 && aligned16(x)) {
        if (deep)
            hipLaunchKernelGGL((attn_context_kernel<true, 0, 13, true>), grid, block, lds, st, rows, P, E, x, e, gpre,
                               bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, pq);
        else
            hipLaunchKernelGGL((attn_context_kernel<true, 0, 8, true>), grid, block, lds, st, rows, P, E, x, e, gpre,
                               bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, pq);
    } else {
        hipLaunchKernelGGL((attn_context_kernel<false, 0, 8, true>), grid, block, lds, st, rows, P, E, x, e, gpre,
                           bbeta, alpha_out, alpha_ld, alpha_save, awe, gate, z, pq);
    }
    SCN_LAUNCH_CHECK();
    return 0;
}

// out[b][:] = sum_q wts[q] * x[b][q][:]   (e.g. the pixel mean of the pooled map: wts = column sums of the pool / P)
int weighted_rows(hipStream_t st, int rows, int Q, int E, const float* x, const float* wts, float* out) {
    if (rows <= 0) return 0;
    SCN_ARG(x && wts && out && Q > 0 && E > 0, "weighted_rows: bad argument");
    dim3 grid(cdiv(E, 256), rows), block(512);
    const size_t lds = (8 * 256 + 16 + Q) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "weighted_rows: too many rows for the LDS staging");
    Slabs nos{nullptr, 0, 0, 0};
    const PoolQ none{0, 0, nullptr, nullptr, nullptr};
    if (E % 4 == 0 && aligned16(x))
        hipLaunchKernelGGL((attn_context_kernel<true, 2, 8, false>), grid, block, lds, st, rows, Q, E, x, wts, nos,
                           (const float*)nullptr, (float*)nullptr, 0L, (float*)nullptr, out, (float*)nullptr,
                           (float*)nullptr, none);
    else
        hipLaunchKernelGGL((attn_context_kernel<false, 2, 8, false>), grid, block, lds, st, rows, Q, E, x, wts, nos,
                           (const float*)nullptr, (float*)nullptr, 0L, (float*)nullptr, out, (float*)nullptr,
                           (float*)nullptr, none);
    SCN_LAUNCH_CHECK();
    return 0;
}

int mean_pixels(hipStream_t st, int rows, int P, int E, const float* enc, float* out) {
    if (rows <= 0) return 0;
    SCN_ARG(enc && out && P > 0 && E > 0, "mean_pixels: bad argument");
    dim3 grid(cdiv(E, 256), rows), block(512);
    const size_t lds = (8 * 256 + 16 + P) * sizeof(float);
    Slabs none{nullptr, 0, 0, 0};
    if (E % 4 == 0 && aligned16(enc))
        hipLaunchKernelGGL((attn_context_kernel<true, 1, 8, false>), grid, block, lds, st, rows, P, E, enc, (const float*)nullptr,
                           none, (const float*)nullptr, (float*)nullptr, 0L, (float*)nullptr, out, (float*)nullptr,
                           (float*)nullptr, PoolQ{0, 0, nullptr, nullptr, nullptr});
    else
        hipLaunchKernelGGL((attn_context_kernel<false, 1, 8, false>), grid, block, lds, st, rows, P, E, enc, (const float*)nullptr,
                           none, (const float*)nullptr, (float*)nullptr, 0L, (float*)nullptr, out, (float*)nullptr,
                           (float*)nullptr, PoolQ{0, 0, nullptr, nullptr, nullptr});
    SCN_LAUNCH_CHECK();
    return 0;
}

int attn_dalpha(hipStream_t st, int rows, int P, int E, const void* enc_, const float* dawe,
                const float* dalpha_in, long dalpha_in_ld, float* dalpha, bool bf) {
    const float* enc = F32P(enc_);
    if (rows <= 0) return 0;
    SCN_ARG(enc && dawe && dalpha && P > 0 && E > 0, "attn_dalpha: bad argument");
    const size_t lds = ((E + 3) & ~3) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "attn_dalpha: encoder_dim too large for the LDS staging");
    dim3 block(256);
    if (bf) {
        const bf16_t* eb = BF16P(enc_);
        if (vec_ok(enc_, E, true)) {
            if (E >= 2048 && P <= 128)
                hipLaunchKernelGGL((attn_dalpha_kernel<true, 8, 2, bf16_t>), dim3(cdiv(P, 8), rows), block, lds, st, rows, P, E, eb, dawe, dalpha_in, dalpha_in_ld, dalpha);
            else
                hipLaunchKernelGGL((attn_dalpha_kernel<true, 4, 4, bf16_t>), dim3(cdiv(P, PC), rows), block, lds, st, rows, P, E, eb, dawe, dalpha_in, dalpha_in_ld, dalpha);
        } else {
            hipLaunchKernelGGL((attn_dalpha_kernel<false, 2, 4, bf16_t>), dim3(cdiv(P, PC), rows), block, lds, st, rows, P, E, eb, dawe, dalpha_in, dalpha_in_ld, dalpha);
        }
    } else if (E % 4 == 0 && aligned16(enc)) {
        if (g_attn_depth && E >= 2048 && P <= 128)       // few rows (the un-pooled map): more, shallower workgroups
            hipLaunchKernelGGL((attn_dalpha_kernel<true, 8, 2, float>), dim3(cdiv(P, 8), rows), block, lds, st, rows, P, E, enc, dawe, dalpha_in, dalpha_in_ld, dalpha);
        else if (g_attn_depth && E >= 1024)
            hipLaunchKernelGGL((attn_dalpha_kernel<true, 4, 4, float>), dim3(cdiv(P, PC), rows), block, lds, st, rows, P, E, enc, dawe, dalpha_in, dalpha_in_ld, dalpha);
        else
            hipLaunchKernelGGL((attn_dalpha_kernel<true, 2, 4, float>), dim3(cdiv(P, PC), rows), block, lds, st, rows, P, E, enc, dawe, dalpha_in, dalpha_in_ld, dalpha);
    } else {
        hipLaunchKernelGGL((attn_dalpha_kernel<false, 2, 4, float>), dim3(cdiv(P, PC), rows), block, lds, st, rows, P, E, enc, dawe, dalpha_in, dalpha_in_ld, dalpha);
    }
    SCN_LAUNCH_CHECK();
    return 0;
}

int attn_softmax_bwd(hipStream_t st, int rows, int P, int A, const void* att1, const float* att2,
                     const float* w, const float* alpha, const float* dalpha, float* de, float* datt2,
                     long datt2_ld, bool bf) {
    if (rows <= 0) return 0;
    SCN_ARG(att1 && att2 && w && alpha && dalpha && datt2 && P > 0 && A > 0, "attn_softmax_bwd: bad argument");
    dim3 grid(cdiv(A, 64), rows), block(256);
    const size_t lds = (16 * 64 + 16 + P) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "attn_softmax_bwd: num_pixels too large for the LDS staging");
    const DalphaTaps none{nullptr, nullptr, nullptr, 0, nullptr, 0};
    const bool vec = vec_ok(att1, A, bf);
    if (bf) {
        if (vec) hipLaunchKernelGGL((attn_softmax_bwd_kernel<true, bf16_t>), grid, block, lds, st, rows, P, A, BF16P(att1), att2, w, alpha, dalpha, de, datt2, datt2_ld, none);
        else     hipLaunchKernelGGL((attn_softmax_bwd_kernel<false, bf16_t>), grid, block, lds, st, rows, P, A, BF16P(att1), att2, w, alpha, dalpha, de, datt2, datt2_ld, none);
    } else {
        if (vec) hipLaunchKernelGGL((attn_softmax_bwd_kernel<true, float>), grid, block, lds, st, rows, P, A, F32P(att1), att2, w, alpha, dalpha, de, datt2, datt2_ld, none);
        else     hipLaunchKernelGGL((attn_softmax_bwd_kernel<false, float>), grid, block, lds, st, rows, P, A, F32P(att1), att2, w, alpha, dalpha, de, datt2, datt2_ld, none);
    }
    SCN_LAUNCH_CHECK();
    return 0;
}

// softmax backward whose d alpha comes from the source-pixel dot products dalphaq [rows][Q] through the pool taps
int attn_softmax_bwd_pooled(hipStream_t st, int rows, int P, int A, const void* att1, const float* att2, const float* w,
                            const float* alpha, const PoolDesc& pool, const float* dalphaq, const float* dalpha_in,
                            long dalpha_in_ld, float* de, float* datt2, long datt2_ld, bool bf) {
    if (rows <= 0) return 0;
    SCN_ARG(att1 && att2 && w && alpha && dalphaq && datt2 && P > 0 && A > 0 && pool.tap_idx && pool.tap_w,
            "attn_softmax_bwd_pooled: bad argument");
    dim3 grid(cdiv(A, 64), rows), block(256);
    const size_t lds = (16 * 64 + 16 + P) * sizeof(float);
    SCN_ARG(lds <= 64 * 1024 && pool.Q <= 16 * 64, "attn_softmax_bwd: num_pixels too large for the LDS staging");
    const DalphaTaps dt{pool.tap_idx, pool.tap_w, dalphaq, pool.Q, dalpha_in, dalpha_in_ld};
    const bool vec = vec_ok(att1, A, bf);
    const float* nodal = nullptr;
    if (bf) {
        if (vec) hipLaunchKernelGGL((attn_softmax_bwd_kernel<true, bf16_t>), grid, block, lds, st, rows, P, A, BF16P(att1), att2, w, alpha, nodal, de, datt2, datt2_ld, dt);
        else     hipLaunchKernelGGL((attn_softmax_bwd_kernel<false, bf16_t>), grid, block, lds, st, rows, P, A, BF16P(att1), att2, w, alpha, nodal, de, datt2, datt2_ld, dt);
    } else {
        if (vec) hipLaunchKernelGGL((attn_softmax_bwd_kernel<true, float>), grid, block, lds, st, rows, P, A, F32P(att1), att2, w, alpha, nodal, de, datt2, datt2_ld, dt);
        else     hipLaunchKernelGGL((attn_softmax_bwd_kernel<false, float>), grid, block, lds, st, rows, P, A, F32P(att1), att2, w, alpha, nodal, de, datt2, datt2_ld, dt);
    }
    SCN_LAUNCH_CHECK();
    return 0;
}

// ---- small helpers of the pooled path ---------------------------------------------------------------
namespace {

// out[b][p][:] = sum_k tap_w[p][k] * y[b][tap_idx[p][k]][:] + bias      (att1 from y = x . We^T)
__global__ __launch_bounds__(256) void pool_expand_kernel(long n4, int P, int Q, int A4, const int* __restrict__ tap_idx,
                                                          const float* __restrict__ tap_w, const float* __restrict__ y,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int a4 = (int)(i % A4);
    const long bp = i / A4;
    const int p = (int)(bp % P);
    const long b = bp / P;
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + a4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = tap_w[p * 4 + k];
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + ((b * Q + tap_idx[p * 4 + k]) * A4 + a4) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(wk, v[c], acc[c]);
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = acc;
}

// out[b][q][:] = sum_k qtap_w[q][k] * in[b][qtap_idx[q][k]][:]           (transpose of the pooling, e.g. d y from d att1)
__global__ __launch_bounds__(256) void pool_transpose_kernel(long n4, int P, int Q, int A4, int qmax,
                                                             const int* __restrict__ qtap_idx,
                                                             const float* __restrict__ qtap_w, const float* __restrict__ in,
                                                             float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int a4 = (int)(i % A4);
    const long bq = i / A4;
    const int q = (int)(bq % Q);
    const long b = bq / Q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < qmax; ++k) {
        const int pi = qtap_idx[q * qmax + k];
        if (pi < 0) continue;
        const float wk = qtap_w[q * qmax + k];
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((b * P + pi) * A4 + a4) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(wk, v[c], acc[c]);
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = acc;
}

// out[b][q][:] += wts[q] * v[b][:]
__global__ __launch_bounds__(256) void add_bcast_rows_w_kernel(long n4, int Q, int E4, const float* __restrict__ wts,
                                                               const float* __restrict__ v, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int e4 = (int)(i % E4);
    const long bq = i / E4;
    const int q = (int)(bq % Q);
    const long b = bq / Q;
    const float wq = wts[q];
    const f32x4 x = *reinterpret_cast<const f32x4*>(v + (b * E4 + e4) * 4);
    f32x4 o = *reinterpret_cast<f32x4*>(out + i * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = fmaf(wq, x[c], o[c]);
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
}

}  // namespace

int pool_expand(hipStream_t st, int B, int P, int A, const PoolDesc& pool, const float* y, const float* bias, float* out) {
    SCN_ARG(B > 0 && P > 0 && A > 0 && A % 4 == 0 && y && out && pool.tap_idx && pool.tap_w && aligned16(y) && aligned16(out),
            "pool_expand: bad argument (A must be a multiple of 4)");
    const long n4 = (long)B * P * (A / 4);
    hipLaunchKernelGGL(pool_expand_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, n4, P, pool.Q, A / 4, pool.tap_idx,
                       pool.tap_w, y, bias, out);
    SCN_LAUNCH_CHECK();
    return 0;
}

int pool_transpose(hipStream_t st, int B, int P, int A, const PoolDesc& pool, const float* in, float* out) {
    SCN_ARG(B > 0 && P > 0 && A > 0 && A % 4 == 0 && in && out && pool.qtap_idx && pool.qtap_w && aligned16(in) &&
                aligned16(out), "pool_transpose: bad argument (A must be a multiple of 4)");
    const long n4 = (long)B * pool.Q * (A / 4);
    hipLaunchKernelGGL(pool_transpose_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, n4, P, pool.Q, A / 4, pool.qtap_max,
                       pool.qtap_idx, pool.qtap_w, in, out);
    SCN_LAUNCH_CHECK();
    return 0;
}

int add_bcast_rows_w(hipStream_t st, int B, int Q, int E, const float* wts, const float* v, float* out) {
    SCN_ARG(B > 0 && Q > 0 && E > 0 && E % 4 == 0 && wts && v && out && aligned16(v) && aligned16(out),
            "add_bcast_rows_w: bad argument (E must be a multiple of 4)");
    const long n4 = (long)B * Q * (E / 4);
    hipLaunchKernelGGL(add_bcast_rows_w_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, n4, Q, E / 4, wts, v, out);
    SCN_LAUNCH_CHECK();
    return 0;
}

int attn_datt1_post(hipStream_t st, int B, int P, int A, int T, const int* dl, const void* att1,
                    const float* att2_all, const float* de_all, const float* w, float* datt1,
                    float* dwpart, int* nblocks_out, bool bf) {
    SCN_ARG(B > 0 && P > 0 && A > 0 && T > 0, "attn_datt1_post: bad dims");
    SCN_ARG(dl && att1 && att2_all && de_all && w && datt1 && dwpart, "attn_datt1_post: null operand");
    dim3 grid(cdiv(P, PC2), B), block(256);
    const size_t lds = (size_t)T * PC2 * sizeof(float);
    SCN_ARG(lds <= 64 * 1024, "attn_datt1_post: too many timesteps for the LDS staging");
    if (nblocks_out) *nblocks_out = grid.x * grid.y;
    if (bf) hipLaunchKernelGGL(attn_datt1_post_kernel<bf16_t>, grid, block, lds, st, B, P, A, T, dl, BF16P(att1), att2_all, de_all, w, datt1, dwpart);
    else    hipLaunchKernelGGL(attn_datt1_post_kernel<float>, grid, block, lds, st, B, P, A, T, dl, F32P(att1), att2_all, de_all, w, datt1, dwpart);
    SCN_LAUNCH_CHECK();
    return 0;
}

int attn_datt1_post_blocks(int B, int P) { return cdiv(P, PC2) * B; }

}  // namespace scn
