// Skinny GEMM for the recurrent part of the decode step: a handful of activation rows (the
// shrinking batch b_t <= 32 of models/decoders/attention_scn.py:143) against a large weight matrix
// that is streamed exactly once per launch.
//
//   Y[s][g][r][n] = sum_{k in K-slice s} X[r][g*xg + k] * W[g*wg + k*ldw + n]
//
// One launch serves every per-timestep contraction of the reference's step
// (attention.py:37, attention_scn.py:147, scn_cell.py:73-86 and :134-144) and their backward mirrors;
// `groups` = 4 runs the four gate blocks (i,f,o,c) of the factored SCN weights in one grid.
//
// Shape of the work: weight-bandwidth bound with a non-trivial MFMA floor (0.59 GFLOP per step at
// B=32), so the grid is (column tiles x groups) x K-slices ~ 2 workgroups per CU and every SIMD gets
// a slice of K:
//   * workgroup = 4 waves = one 32-column tile of one K-slice; the 4 waves split the slice's K range
//     and reduce their 32x32 accumulators through LDS (fixed order -> deterministic);
//   * the weight fragment of v_mfma_f32_32x32x2_f32 (lane l: B[k = l>>5][n = l&31]) is exactly two
//     full 128-byte lines of a row-major [K][N] matrix, so weights go HBM -> VGPR directly, no LDS;
//   * the tiny activation tile (32 x K-slice) is staged k-major in LDS once per workgroup;
//   * split-K partial sums are written as slabs and summed, in slab order, by the consumer kernel's
//     prologue (a launch boundary is cheaper than an in-kernel grid barrier on this chip).
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

constexpr int KC = 256;   // activation rows staged per LDS chunk (k extent)
constexpr int XLD = 33;   // padded leading dim of the k-major activation tile
constexpr int UN = 8;     // weight-fragment loads kept in flight per wave per batch

struct SkinnyArgs {
    const float* X; const float* W; float* Y;
    long ldx, xg, ldw, wg, ldy, yg, yslab;
    int rows, N, K, kslice, groups;
};

__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    __shared__ float Xs[KC * XLD];
    __shared__ float red[4][32 * XLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctiles = (a.N + 31) / 32;
    const int ct = blockIdx.x % ctiles, grp = blockIdx.x / ctiles;
    const int slice = blockIdx.y, r0 = blockIdx.z * 32;
    const int n0 = ct * 32;
    const int kbeg = slice * a.kslice;
    const int kend = min(a.K, kbeg + a.kslice);

    const float* X = a.X + (long)grp * a.xg;
    const float* W = a.W + (long)grp * a.wg;
    const int hh = lane >> 5, l31 = lane & 31;
    const int n = n0 + l31;
    const bool nok = n < a.N;
    const int nc = nok ? n : a.N - 1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int kc = kbeg; kc < kend; kc += KC) {
        const int kcn = min(KC, kend - kc);          // k extent of this chunk
        const int kcn8 = (kcn + 7) & ~7;             // padded so that each wave gets whole k-pairs
        if (kc != kbeg) __syncthreads();
        // stage X[r0..r0+31][kc..kc+kcn) k-major; lanes run along k (coalesced), LDS write is 2-way
        for (int idx = tid; idx < 32 * kcn8; idx += 256) {
            const int r = idx / kcn8, k = idx - r * kcn8;
            // unconditional load from a clamped (always valid) address, then select: keeps hipcc from
            // branching around the load
            const int rc = min(r0 + r, a.rows - 1), kcl = min(kc + k, a.K - 1);
            const float v = X[(long)rc * a.ldx + kcl];
            Xs[k * XLD + r] = (k < kcn && r0 + r < a.rows) ? v : 0.f;
        }
        __syncthreads();
        // this wave's k range inside the chunk: a contiguous quarter, whole k-pairs
        const int per = kcn8 / 4;                    // multiple of 2
        const int wk0 = wave * per;
        const int npair = per / 2;
        for (int p0 = 0; p0 < npair; p0 += UN) {
            float bf[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int k = kc + wk0 + 2 * (p0 + u) + hh;
                const float v = W[(long)min(k, a.K - 1) * a.ldw + nc];
                bf[u] = (p0 + u < npair && nok && k < kend) ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (p0 + u < npair) {
                    const float af = Xs[(wk0 + 2 * (p0 + u) + hh) * XLD + l31];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[u], acc, 0, 0, 0);
                }
            }
        }
    }

    // cross-wave reduction in wave order 0..3 (deterministic)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][mfma32_row(r, lane) * XLD + l31] = acc[r];
    __syncthreads();
    float* Y = a.Y + (long)slice * a.yslab + (long)grp * a.yg;
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const int r = idx >> 5, c = idx & 31;
        if (r0 + r < a.rows && n0 + c < a.N) {
            const float v = ((red[0][r * XLD + c] + red[1][r * XLD + c]) + red[2][r * XLD + c]) + red[3][r * XLD + c];
            Y[(long)(r0 + r) * a.ldy + n0 + c] = v;
        }
    }
}

}  // namespace

int skinny_pick_ksplit(int rows, int N, int K, int groups) {
    const int wgs = cdiv(N, 32) * groups * cdiv(rows > 0 ? rows : 1, 32);
    int ks = cdiv(512, wgs);                 // aim at ~2 workgroups per CU (256 CUs)
    const int kmax = K / 64 > 0 ? K / 64 : 1;  // keep >= 64 k per slice (>= 8 MFMAs per wave)
    if (ks > kmax) ks = kmax;
    if (ks > SCN_MAX_KSPLIT) ks = SCN_MAX_KSPLIT;
    if (ks < 1) ks = 1;
    return ks;
}

int skinny_gemm(hipStream_t st, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                const float* W, long ldw, long wg, float* Y, long ldy, long yg, long yslab, int ksplit) {
    if (rows <= 0 || N <= 0 || groups <= 0) return 0;
    SCN_ARG(X && W && Y, "skinny_gemm: null operand");
    SCN_ARG(K >= 1, "skinny_gemm: K must be >= 1");
    SCN_ARG(ksplit >= 1 && ksplit <= SCN_MAX_KSPLIT, "skinny_gemm: ksplit out of range");
    int kslice = cdiv(K, ksplit);
    kslice = (kslice + 7) & ~7;
    if (kslice < 8) kslice = 8;
    SkinnyArgs a{X, W, Y, ldx, xg, ldw, wg, ldy, yg, yslab, rows, N, K, kslice, groups};
    dim3 grid(cdiv(N, 32) * groups, ksplit, cdiv(rows, 32)), block(256);
    hipLaunchKernelGGL(skinny_kernel, grid, block, 0, st, a);
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
