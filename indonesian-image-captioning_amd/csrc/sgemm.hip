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
    int S, kper;      // split-K: S slices of kper (multiple of BK) k each; S == 1 -> direct epilogue
    float* ws;        // [batch][S][M][N] partial sums when S > 1
};

// Load a (128 x 16) operand tile into registers.  CONTIG_K: element (r,k) at base[r*ld + k]
// (k contiguous) else at base[k*ld + r] (r contiguous).  Range-checked buffer loads: lanes outside the
// matrix read 0 with no select and no branch, so nothing forces a wait before the MFMA block.
template <bool CONTIG_K, bool VEC>
__device__ __forceinline__ void load_tile(__amdgpu_buffer_rsrc_t rs, long ld, int r0, int k0, int R, int K,
                                          int tid, float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = CONTIG_K ? r0 + (tid >> 2) + 64 * i : r0 + (tid & 31) * 4;
        const int k = CONTIG_K ? k0 + (tid & 3) * 4 : k0 + (tid >> 5) + 8 * i;
        const long lin = CONTIG_K ? (long)r * ld + k : (long)k * ld + r;
        if (VEC) {
            const f32x4 v = buf_load4(rs, (r < R && k < K) ? (unsigned)(lin * 4) : OOB_OFF);
#pragma unroll
            for (int c = 0; c < 4; ++c) reg[i][c] = v[c];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = CONTIG_K ? (r < R && k + c < K) : (k < K && r + c < R);
                reg[i][c] = buf_load(rs, ok ? (unsigned)((lin + (CONTIG_K ? c : c)) * 4) : OOB_OFF);
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
__global__ __launch_bounds__(256, 3) void sgemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int zb = blockIdx.z / g.S, sp = blockIdx.z - zb * g.S;
    const float* A = g.A + (long)zb * g.sA;
    const float* B = g.B + (long)zb * g.sB;
    float* C = g.C + (long)zb * g.sC;
    const int kbeg = sp * g.kper, Kend = min(g.K, kbeg + g.kper);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[2][4], rb[2][4];
    const int nk = max(0, (Kend - kbeg + BK - 1) / BK);
    // descriptors from wave-uniform values (kernel args + blockIdx.z); sizes checked on the host
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(A, (unsigned)((TA ? ((long)(g.K - 1) * g.lda + g.M) : ((long)(g.M - 1) * g.lda + g.K)) * 4));
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(B, (unsigned)((TB ? ((long)(g.N - 1) * g.ldb + g.K) : ((long)(g.K - 1) * g.ldb + g.N)) * 4));
    load_tile<!TA, VEC>(ars, g.lda, m0, kbeg, g.M, Kend, tid, ra);
    load_tile<TB, VEC>(brs, g.ldb, n0, kbeg, g.N, Kend, tid, rb);
    store_tile<!TA>(As[0], tid, ra);
    store_tile<TB>(Bs[0], tid, rb);
    __syncthreads();

    const int hh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<!TA, VEC>(ars, g.lda, m0, kbeg + (kt + 1) * BK, g.M, Kend, tid, ra);
            load_tile<TB, VEC>(brs, g.ldb, n0, kbeg + (kt + 1) * BK, g.N, Kend, tid, rb);
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
        __builtin_amdgcn_sched_barrier(0);   // keep the LDS stores of the prefetched tile BEHIND the MFMAs
        if (kt + 1 < nk) {
            store_tile<!TA>(As[cur ^ 1], tid, ra);
            store_tile<TB>(Bs[cur ^ 1], tid, rb);
        }
        __syncthreads();
    }

    if (g.S > 1) {   // split-K: raw alpha-scaled partial tile; splitk_reduce_kernel applies the epilogue
        float* W = g.ws + ((long)blockIdx.z * g.M) * g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                    if (n < g.N && m < g.M) W[(long)m * g.N + n] = g.alpha * acc[i][j][r];
                }
            }
        return;
    }

    // Epilogue.  The optional reads (beta*C, rowmask) are issued as one batch of range-checked buffer
    // loads per tile and pinned with an empty asm, otherwise hipcc sinks each load next to its store
    // behind a vmcnt(0): 16 serialised memory latencies per tile.
    const bool use_c = g.beta != 0.f, use_m = g.rowmask != nullptr;
    const __amdgpu_buffer_rsrc_t cr = make_rsrc(C, use_c ? (unsigned)(((long)(g.M - 1) * g.ldc + g.N) * 4) : 0u);
    const __amdgpu_buffer_rsrc_t mr = make_rsrc(g.rowmask, use_m ? (unsigned)g.M * 4u : 0u);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            const bool nok = n < g.N;
            float cv[16], mk[16];
            if (use_c) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                    cv[r] = buf_load(cr, (nok && m < g.M) ? (unsigned)(((long)m * g.ldc + n) * 4) : OOB_OFF);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(cv[r]));
            }
            if (use_m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                    mk[r] = buf_load(mr, m < g.M ? (unsigned)m * 4u : OOB_OFF);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(mk[r]));
            }
            float bv = 0.f;
            if (g.bias && nok) bv = g.bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                float v = g.alpha * acc[i][j][r] + bv;
                if (use_c) v += g.beta * cv[r];
                if (use_m && mk[r] == 0.f) v = 0.f;
                if (nok && m < g.M) C[(long)m * g.ldc + n] = v;
            }
        }
}

// C = sum_s ws[z][s] + beta*C + bias, rows with rowmask == 0 -> 0 (slab order fixed: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)g.M * g.N) return;
    const int m = (int)(i / g.N), n = (int)(i - (long)m * g.N);
    const int zb = blockIdx.y;
    const float* W = g.ws + ((long)zb * g.S) * g.M * g.N;
    float v = slab_sum(W, i, g.S, (long)g.M * g.N);
    float* cp = g.C + (long)zb * g.sC + (long)m * g.ldc + n;
    if (g.bias) v += g.bias[n];
    if (g.beta != 0.f) v += g.beta * (*cp);
    if (g.rowmask && g.rowmask[m] == 0.f) v = 0.f;
    *cp = v;
}

}  // namespace

// split-K policy knobs (scnattn_set_option "gemm_target" / "gemm_kmin" / "gemm_kmin_small")
int g_gemm_gate = 256;        // split only when the tile grid alone has fewer workgroups than this
int g_gemm_target = 512;      // aim for this many workgroups (tiles x splits)
int g_gemm_kmin = 256;        // at least this much K per split ...
int g_gemm_kmin_small = 128;  // ... or this much when the product has <= 16 tiles (32-row operands)
int g_use_cgemm = 1;          // 16-byte-aligned products go to the LDS-DMA pipelined kernel of cgemm.hip

int sgemm_ws(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda,
             const float* B, long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask,
             int batch, long sA, long sB, long sC, float* ws, long ws_floats) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    SCN_ARG(A && B && C, "sgemm: null operand");
    SCN_ARG(K >= 1, "sgemm: K must be >= 1");
    if (g_use_cgemm && cgemm_supported(tA, tB, M, N, K, A, lda, B, ldb, sA, sB))
        return cgemm(st, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, rowmask, batch, sA, sB, sC, ws,
                     ws_floats, nullptr);
    {
        const long abytes = (tA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 4;
        const long bbytes = (tB ? ((long)(N - 1) * ldb + K) : ((long)(K - 1) * ldb + N)) * 4;
        SCN_ARG(abytes < 0x7fffffffL && bbytes < 0x7fffffffL, "sgemm: operand exceeds the 2 GiB buffer-descriptor range");
    }
    SCN_ARG(beta == 0.f || ((long)(M - 1) * ldc + N) * 4 < 0x7fffffffL, "sgemm: C too large for beta != 0");
    // split-K when the tile grid alone cannot fill the chip and K is deep enough to amortise the reduce
    // (measured on the decoder's shapes: one 128x128 tile per CU leaves the MFMA pipe ~45 % idle, and a
    //  32-row product with 16 tiles ran 85 us un-split)
    const long tiles = (long)cdiv(N, BN) * cdiv(M, BM) * batch;
    int S = 1;
    const int kmin = tiles <= 16 ? g_gemm_kmin_small : g_gemm_kmin;
    if (ws && tiles < g_gemm_gate && K >= 2 * kmin) {
        S = (int)((g_gemm_target + tiles - 1) / tiles);
        const int smax = K / kmin;
        if (S > smax) S = smax;
        if (S > SCN_MAX_KSPLIT) S = SCN_MAX_KSPLIT;
        while (S > 1 && (long)S * batch * M * N > ws_floats) --S;
        if (S < 1) S = 1;
    }
    int kper = cdiv(K, S);
    kper = (kper + BK - 1) / BK * BK;
    S = cdiv(K, kper);
    GemmArgs g{A, B, C, bias, rowmask, lda, ldb, ldc, sA, sB, sC, M, N, K, alpha, beta, S, kper, ws};
    // 16-byte loads need: aligned bases/strides and the contiguous extent a multiple of 4
    const int contigA = tA ? M : K, contigB = tB ? K : N;
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (sA % 4 == 0) &&
                     (sB % 4 == 0) && (contigA % 4 == 0) && (contigB % 4 == 0);
    dim3 grid(cdiv(N, BN), cdiv(M, BM), batch * S), block(256);
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
    if (S > 1) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv((long)M * N, 256), batch), dim3(256), 0, st, g);
        SCN_LAUNCH_CHECK();
    }
    return 0;
}

int sgemm(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda,
          const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
          const float* rowmask, int batch, long sA, long sB, long sC) {
    return sgemm_ws(st, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, rowmask, batch, sA, sB, sC,
                    nullptr, 0);
}

}  // namespace scn
