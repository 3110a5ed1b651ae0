// General fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain).
//
//   C[z] = alpha * op(A[z]) . op(B[z]) + beta * C[z] + bias[n]      (row-major, z = batch index)
//   rows with rowmask[m] == 0 are written as exact zeros (used for the (b,t) cells past a caption's
//   decode length: the reference leaves those rows of `predictions` at 0, attention_scn.py:134-156).
//
// Replaces the dense ATen mm/addmm calls of the reference's hot path: encoder_att
// (models/attention.py:35), fc (attention_scn.py:154), the batched x-side projections
// (scn_cell.py:73-86) and every weight-gradient contraction autograd derives from them.
//
// Tiling: 128x128x16 block, 4 waves as 2x2, each wave a 64x64 sub-tile = 2x2 MFMA 32x32 tiles.
// Operands are staged k-major in LDS ([k][m], [k][n]) so that an MFMA fragment read is 32
// consecutive floats per half-wave (conflict free); global loads are 16 B per lane along the
// contiguous dimension and are prefetched into registers one k-step ahead (one barrier per k-step).
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDS_LD = BM + 4;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* rowmask;
    long lda, ldb, ldc, sA, sB, sC;
    int M, N, K;
    float alpha, beta;
};

// Load a (128 x 16) operand tile into registers.  CONTIG_K: element (r,k) at base[r*ld + k]
// (k contiguous) else at base[k*ld + r] (r contiguous).
template <bool CONTIG_K, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, long ld, int r0, int k0,
                                          int R, int K, int tid, float (&reg)[2][4]) {
    // Loads are unconditional from clamped (always valid) addresses and masked afterwards: a
    // conditional load makes hipcc branch around it and drain vmcnt per element.
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (CONTIG_K) {
            const int r = r0 + (tid >> 2) + 64 * i, k = k0 + (tid & 3) * 4;
            const int rc = min(r, R - 1);
            if (VEC) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)rc * ld + min(k, K - 4));
                const bool ok = r < R && k < K;
#pragma unroll
                for (int c = 0; c < 4; ++c) reg[i][c] = ok ? v[c] : 0.f;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = base[(long)rc * ld + min(k + c, K - 1)];
                    reg[i][c] = (r < R && k + c < K) ? v : 0.f;
                }
            }
        } else {
            const int k = k0 + (tid >> 5) + 8 * i, r = r0 + (tid & 31) * 4;
            const int kc = min(k, K - 1);
            if (VEC) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)kc * ld + min(r, R - 4));
                const bool ok = k < K && r < R;
#pragma unroll
                for (int c = 0; c < 4; ++c) reg[i][c] = ok ? v[c] : 0.f;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = base[(long)kc * ld + min(r + c, R - 1)];
                    reg[i][c] = (k < K && r + c < R) ? v : 0.f;
                }
            }
        }
    }
}

template <bool CONTIG_K>
__device__ __forceinline__ void store_tile(float (*lds)[LDS_LD], int tid, const float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (CONTIG_K) {
            const int r = (tid >> 2) + 64 * i, k = (tid & 3) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) lds[k + c][r] = reg[i][c];
        } else {
            const int k = (tid >> 5) + 8 * i, r = (tid & 31) * 4;
            *reinterpret_cast<f32x4*>(&lds[k][r]) = f32x4{reg[i][0], reg[i][1], reg[i][2], reg[i][3]};
        }
    }
}

// TA: A stored [K][M] (transposed).  TB: B stored [N][K] (transposed).
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void sgemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = g.A + (long)blockIdx.z * g.sA;
    const float* B = g.B + (long)blockIdx.z * g.sB;
    float* C = g.C + (long)blockIdx.z * g.sC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[2][4], rb[2][4];
    const int nk = (g.K + BK - 1) / BK;
    load_tile<!TA, VEC>(A, g.lda, m0, 0, g.M, g.K, tid, ra);
    load_tile<TB, VEC>(B, g.ldb, n0, 0, g.N, g.K, tid, rb);
    store_tile<!TA>(As[0], tid, ra);
    store_tile<TB>(Bs[0], tid, rb);
    __syncthreads();

    const int hh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<!TA, VEC>(A, g.lda, m0, (kt + 1) * BK, g.M, g.K, tid, ra);
            load_tile<TB, VEC>(B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][kk + hh][wm * 64 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[cur][kk + hh][wn * 64 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile<!TA>(As[cur ^ 1], tid, ra);
            store_tile<TB>(Bs[cur ^ 1], tid, rb);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                if (m >= g.M) continue;
                float* cp = C + (long)m * g.ldc + n;
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.beta != 0.f) v += g.beta * (*cp);
                if (g.rowmask && g.rowmask[m] == 0.f) v = 0.f;
                *cp = v;
            }
        }
}

}  // namespace

int sgemm(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda,
          const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
          const float* rowmask, int batch, long sA, long sB, long sC) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    SCN_ARG(A && B && C, "sgemm: null operand");
    SCN_ARG(K >= 1, "sgemm: K must be >= 1");
    GemmArgs g{A, B, C, bias, rowmask, lda, ldb, ldc, sA, sB, sC, M, N, K, alpha, beta};
    // 16-byte loads need: aligned bases/strides and the contiguous extent a multiple of 4
    const int contigA = tA ? M : K, contigB = tB ? K : N;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (sA % 4 == 0) &&
                     (sB % 4 == 0) && (contigA % 4 == 0) && (contigB % 4 == 0);
    dim3 grid(cdiv(N, BN), cdiv(M, BM), batch), block(256);
#define SCN_GEMM_LAUNCH(TA_, TB_)                                                        \
    do {                                                                                 \
        if (vec) hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, true>), grid, block, 0, st, g);  \
        else     hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, false>), grid, block, 0, st, g); \
    } while (0)
    if (!tA && !tB) SCN_GEMM_LAUNCH(false, false);
    else if (!tA && tB) SCN_GEMM_LAUNCH(false, true);
    else if (tA && !tB) SCN_GEMM_LAUNCH(true, false);
    else SCN_GEMM_LAUNCH(true, true);
#undef SCN_GEMM_LAUNCH
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
