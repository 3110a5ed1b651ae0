// fp32 GEMM core of the ResNet-152 1x1-convolution path (and of every 16-byte-aligned dense product of the
// decoder), gfx950 only:   C[M][N] = alpha * op(A) . op(B) (+ epilogues), exact fp32 (v_mfma_f32_32x32x2_f32).
//
// What it replaces: the `torch.nn.Conv2d(k=1)` calls of torchvision's Bottleneck behind the reference's
// models/encoders/caption.py:17-22 (forward, dgrad, wgrad -- on channels-last maps a 1x1 convolution is the GEMM
// [R = N*H*W, Cin] x [Cin, Cout]) together with the BatchNorm work that can ride on it, and the aten::mm/addmm calls
// of models/attention.py:35, attention_scn.py:154, scn_cell.py:73-86 that csrc/sgemm.hip served in round 1.
//
// Structure (MI355X_MICROARCH / cdna_hip_programming "Pipelining across barriers"):
//   * 128x128x16 block tile, 4 waves as 2x2, each wave 2x2 MFMA 32x32 tiles;
//   * operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, hardware
//     range checking: rows / k beyond the matrix land as zeros), 3-stage ring, ONE raw s_barrier per k-step and a
//     counted `s_waitcnt vmcnt(N)` that leaves the next tile's loads in flight across it;
//   * two LDS images, chosen per operand by which of its dimensions is contiguous in memory:
//       KC (k contiguous, e.g. activations [R][Cin], weights [Cout][Cin]):  [row][16 k], 64-byte rows, the four
//          16-byte granules of a row XOR-swizzled by (row>>2)&3 on the SOURCE address (the LDS-DMA destination is
//          lane-linear) so that a fragment read is two conflict-free ds_read_b128;
//       MC (m/n contiguous, e.g. dY [R][Cout] as the [K][M] operand of wgrad): [k][128], fragment = 8 ds_read_b32;
//     the MFMA k-slot of lane half h in MFMA j of a k-step is k0 + 8h + j for BOTH operands (any pairing is legal as
//     long as A and B agree), which is what makes the KC fragment 32 contiguous bytes;
//   * XCD-aware tile order: the 8 XCDs own contiguous runs of row panels, n fastest, so an activation panel is
//     fetched from HBM once per XCD and the weights stay in that XCD's L2;
//   * optional A-operand prologue on the fragment registers: a = relu(a*scale[k] + shift[k]) (the previous layer's
//     BatchNorm + ReLU folded to one fma, per input channel) -- the normalised map is never written to HBM;
//     scale/shift of the current k-step travel through the LDS ring with the tile;
//   * optional per-column prologue on a MC B operand (wgrad: B = relu(bn(z)) with the channel on the column);
//   * optional statistics epilogue: per (row-tile, column) sums of (y - s) and (y - s)^2 for the NEXT BatchNorm
//     (s = a per-channel shift for conditioning, e.g. the running mean), fixed order, no atomics;
//   * optional mask/statistics epilogue for dgrad: g = acc * [relu-mask recomputed from z], column sums of g and
//     g*xhat (the BatchNorm backward reductions), g stored;
//   * split-K into slabs reduced by a second launch in slab order (deterministic), for shapes whose tile grid alone
//     cannot fill 256 CUs.
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

constexpr int TM = 128, TN = 128, TK = 16, NSTAGE = 3;
constexpr int TILE_F = TM * TK;                 // floats per operand tile (8 KiB)
constexpr int AUX_F = 4 * 64;                   // per-wave [scale 16 | shift 16 | pad 32] of the current k-step (1 KiB)
constexpr int STAGE_F = 2 * TILE_F + AUX_F;     // 17 KiB per stage -> 51 KiB per workgroup, 3 workgroups per CU

typedef __attribute__((address_space(3))) void* lds_ptr;

struct CArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* rowmask;
    long lda, ldb, ldc, sA, sB, sC;
    int M, N, K;
    float alpha, beta;
    int S, kper;                  // split-K
    float* ws;
    int mt, nt;                   // tile grid
    // conv extras
    int gHi, gWi, gHo, gWo, gs;   // row gather (strided 1x1 convolution); gs == 0: none
    const float* pro_ss;          // interleaved {scale, shift} per channel: PRO 1 per k (A), PRO 2 per n (B)
    float* stat_partial; const float* stat_shift;       // EPI 1: [mt][2][N]
    const float* ez; const float* emean; const float* einvstd; const float* egamma; const float* ebeta;  // EPI 2
    long ldz;
};

// output row (n, ho, wo) of a strided 1x1 convolution -> input row (n, ho*s, wo*s)
__device__ __forceinline__ long gather_row(const CArgs& g, int r) {
    if (g.gs == 0) return r;
    const int hw = g.gHo * g.gWo;
    const int n = r / hw, rem = r - n * hw;
    const int ho = rem / g.gWo, wo = rem - ho * g.gWo;
    return (long)n * g.gHi * g.gWi + (long)(ho * g.gs) * g.gWi + wo * g.gs;
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_wave_base, 4, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

// A_MC / B_MC: operand stored with its m / n dimension contiguous ([K][M] / [K][N]); otherwise k contiguous.
// PRO: 0 none, 1 relu(a*scale[k]+shift[k]) on a KC A operand, 2 relu(b*scale[n]+shift[n]) on a MC B operand.
// EPI: 0 plain (alpha, beta, bias, rowmask), 1 plain store + column statistics, 2 relu-mask from z + BN-backward sums.
// AGATHER: rows of a KC A operand (PRO 0/1) or k-rows of a MC B operand are gathered (strided 1x1 convolution).
template <bool A_MC, bool B_MC, int PRO, int EPI, bool GATHER>
__global__ __launch_bounds__(256, 3) void cgemm_kernel(CArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE_F];
    constexpr int LPT = 4 + (PRO == 1 ? 1 : 0);       // LDS-DMA instructions per wave per tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int hh = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware tile order (speed only): blocks b, b+8, b+16 ... share an XCD ---------------------------
    const int ntiles = g.mt * g.nt;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / g.nt, tn = bid - tm * g.nt;
    const int m0 = tm * TM, n0 = tn * TN;
    const int zb = blockIdx.y / g.S, sp = blockIdx.y - zb * g.S;
    const float* A = g.A + (long)zb * g.sA;
    const float* B = g.B + (long)zb * g.sB;
    float* C = g.C + (long)zb * g.sC;
    const int kbeg = sp * g.kper, Kend = min(g.K, kbeg + g.kper);
    const int nk = (Kend - kbeg + TK - 1) / TK;

    const long a_elems = A_MC ? ((long)(g.K - 1) * g.lda + g.M)
                              : (GATHER ? (gather_row(g, g.M - 1) * g.lda + g.K) : ((long)(g.M - 1) * g.lda + g.K));
    const long b_elems = B_MC ? ((GATHER && A_MC ? gather_row(g, g.K - 1) : (long)(g.K - 1)) * g.ldb + g.N)
                              : ((long)(g.N - 1) * g.ldb + g.K);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(A, (unsigned)(a_elems * 4));
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(B, (unsigned)(b_elems * 4));
    const __amdgpu_buffer_rsrc_t srs = make_rsrc(g.pro_ss, PRO == 1 ? (unsigned)g.K * 8u : 0u);

    // ---- per-lane source offsets of the two chunks this wave stages per operand ------------------------------
    // KC: chunk = 16 rows x 64 B; lane -> row chunk*16 + lane/4, LDS granule lane&3 <- source granule (lane&3)^((row>>2)&3)
    // MC: chunk = 2 k-rows x 512 B; lane -> k-row chunk*2 + lane/32, columns 4*(lane&31)
    unsigned a_off[2], b_off[2];     // byte offsets at k = kbeg (KC: + k*4; MC: + k*ld*4 per step)
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int chunk = wave * 2 + c;
        if (!A_MC) {
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = m0 + row;
            a_ok[c] = grow < g.M;
            const long src = (GATHER ? gather_row(g, a_ok[c] ? grow : 0) : (long)grow) * g.lda + gsrc * 4;
            a_off[c] = (unsigned)(src * 4);
        } else {
            const int col = m0 + 4 * (lane & 31);
            a_ok[c] = col < g.M;
            a_off[c] = (unsigned)(((long)(chunk * 2 + (lane >> 5)) * g.lda + col) * 4);
        }
        if (!B_MC) {
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = n0 + row;
            b_ok[c] = grow < g.N;
            b_off[c] = (unsigned)(((long)grow * g.ldb + gsrc * 4) * 4);
        } else {
            const int col = n0 + 4 * (lane & 31);
            b_ok[c] = col < g.N;
            b_off[c] = (unsigned)(((long)(chunk * 2 + (lane >> 5)) * g.ldb + col) * 4);
        }
    }

    auto issue = [&](int kt, int stage) {     // LDS-DMA of k-step kt into ring slot `stage`
        float* sa = lds + stage * STAGE_F;
        float* sb = sa + TILE_F;
        const int k0 = kbeg + kt * TK;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int chunk = wave * 2 + c;
            unsigned va, vb;
            if (!A_MC) {
                const int kk = k0 + 4 * ((lane & 3) ^ (((chunk * 16 + (lane >> 2)) >> 2) & 3));
                va = (a_ok[c] && kk < Kend) ? a_off[c] + (unsigned)k0 * 4u : OOB_OFF;
            } else {
                const int kr = k0 + chunk * 2 + (lane >> 5);
                va = (a_ok[c] && kr < Kend) ? a_off[c] + (unsigned)((long)k0 * g.lda * 4) : OOB_OFF;
            }
            dma16(ars, sa + chunk * 256, va);
            if (!B_MC) {
                const int kk = k0 + 4 * ((lane & 3) ^ (((chunk * 16 + (lane >> 2)) >> 2) & 3));
                vb = (b_ok[c] && kk < Kend) ? b_off[c] + (unsigned)k0 * 4u : OOB_OFF;
            } else {
                const int kr = k0 + chunk * 2 + (lane >> 5);
                if (GATHER && A_MC) {     // wgrad of a strided convolution: k-rows of B = gathered rows of the input map
                    const long src = gather_row(g, kr < Kend ? kr : 0) * g.ldb + n0 + 4 * (lane & 31);
                    vb = (b_ok[c] && kr < Kend) ? (unsigned)(src * 4) : OOB_OFF;
                } else {
                    vb = (b_ok[c] && kr < Kend) ? b_off[c] + (unsigned)((long)k0 * g.ldb * 4) : OOB_OFF;
                }
            }
            dma16(brs, sb + chunk * 256, vb);
        }
        if (PRO == 1) {   // this wave's private copy of {scale, shift}[k0 .. k0+15] (32 floats, one instruction)
            float* sx = sa + 2 * TILE_F + wave * 64;
            const int kk = k0 + (lane >> 1);
            dma4(srs, sx, (lane < 32 && kk < Kend) ? (unsigned)(2 * k0 + lane) * 4u : OOB_OFF);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // PRO 2: per-column scale/shift of this lane's B columns (constant over k), loaded before any DMA is in flight
    float bsc[2] = {1.f, 1.f}, bsh[2] = {0.f, 0.f};
    if (PRO == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n < g.N) { bsc[j] = g.pro_ss[2 * n]; bsh[j] = g.pro_ss[2 * n + 1]; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(bsc[j])); asm volatile("" : "+v"(bsh[j])); }
    }

    if (nk > 0) issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) {
            int s2 = stage + 2; if (s2 >= NSTAGE) s2 -= NSTAGE;
            issue(kt + 2, s2);
        }
        const float* sa = lds + stage * STAGE_F;
        const float* sb = sa + TILE_F;
        float a[2][8], b[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!A_MC) {
                const int row = wm * 64 + i * 32 + l31, sw = (row >> 2) & 3;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sa + row * 16 + (((2 * hh) ^ sw) << 2));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sa + row * 16 + (((2 * hh + 1) ^ sw) << 2));
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[i][q] = v0[q]; a[i][4 + q] = v1[q]; }
            } else {
                const int col = wm * 64 + i * 32 + l31;
#pragma unroll
                for (int q = 0; q < 8; ++q) a[i][q] = sa[(8 * hh + q) * 128 + col];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!B_MC) {
                const int row = wn * 64 + j * 32 + l31, sw = (row >> 2) & 3;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sb + row * 16 + (((2 * hh) ^ sw) << 2));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + row * 16 + (((2 * hh + 1) ^ sw) << 2));
#pragma unroll
                for (int q = 0; q < 4; ++q) { b[j][q] = v0[q]; b[j][4 + q] = v1[q]; }
            } else {
                const int col = wn * 64 + j * 32 + l31;
#pragma unroll
                for (int q = 0; q < 8; ++q) b[j][q] = sb[(8 * hh + q) * 128 + col];
            }
        }
        if (PRO == 1) {
            const float* sx = sa + 2 * TILE_F + wave * 64 + 16 * hh;     // {scale, shift} of k = 8h .. 8h+7
            f32x4 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const f32x4*>(sx + 4 * q);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    a[i][q] = fmaxf(fmaf(a[i][q], t[q >> 1][(q & 1) * 2], t[q >> 1][(q & 1) * 2 + 1]), 0.f);
        }
        if (PRO == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) b[j][q] = fmaxf(fmaf(b[j][q], bsc[j], bsh[j]), 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
        // the next tile must have landed (this wave's part) before the barrier that publishes it to the others;
        // the tile after it stays in flight across the barrier
        if (kt + 2 < nk) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (++stage == NSTAGE) stage = 0;
    }

    // ================================== epilogues =====================================================
    if (g.S > 1) {   // split-K: raw alpha-scaled partial tile; creduce_kernel applies the epilogue
        float* W = g.ws + ((long)blockIdx.y * g.M) * g.N;
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

    if (EPI == 0) {
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
        return;
    }

    // EPI 1 / 2 share the column reduction: lanes l and l+32 hold the same column, the two row-halves (wm) meet in LDS
    // (the ring is dead by now: every wave passed the last barrier after its last fragment read)
    float* colsum = lds;                         // [wm][2][TN]
    if (EPI == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            const bool nok = n < g.N;
            const float sft = (g.stat_shift && nok) ? g.stat_shift[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                    const float v = g.alpha * acc[i][j][r];
                    if (nok && m < g.M) {
                        C[(long)m * g.ldc + n] = v;
                        const float d = v - sft;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                colsum[(wm * 2 + 0) * TN + wn * 64 + j * 32 + l31] = s1;
                colsum[(wm * 2 + 1) * TN + wn * 64 + j * 32 + l31] = s2;
            }
        }
    } else {   // EPI 2: g = acc * [fma((z-mean)*invstd, gamma, beta) > 0]; sums of g and g*xhat; store g
        const __amdgpu_buffer_rsrc_t zr = make_rsrc(g.ez, (unsigned)(((long)(g.M - 1) * g.ldz + g.N) * 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            const bool nok = n < g.N;
            float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
            if (nok) { mu = g.emean[n]; is = g.einvstd[n]; ga = g.egamma[n]; be = g.ebeta[n]; }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float zv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                    zv[r] = buf_load(zr, (nok && m < g.M) ? (unsigned)(((long)m * g.ldz + n) * 4) : OOB_OFF);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(zv[r]));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                    const float xh = (zv[r] - mu) * is;
                    float v = g.alpha * acc[i][j][r];
                    if (!(fmaf(xh, ga, be) > 0.f)) v = 0.f;
                    if (nok && m < g.M) {
                        C[(long)m * g.ldc + n] = v;
                        s1 += v;
                        s2 = fmaf(v, xh, s2);
                    }
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                colsum[(wm * 2 + 0) * TN + wn * 64 + j * 32 + l31] = s1;
                colsum[(wm * 2 + 1) * TN + wn * 64 + j * 32 + l31] = s2;
            }
        }
    }
    __syncthreads();
    if (tid < TN && n0 + tid < g.N) {
        float* p = g.stat_partial + ((long)tm * 2) * g.N + n0 + tid;
        p[0] = colsum[0 * TN + tid] + colsum[2 * TN + tid];
        p[g.N] = colsum[1 * TN + tid] + colsum[3 * TN + tid];
    }
}

// Sum split-K slabs in slab order and apply the epilogue.  Without statistics: element-wise (grid.x blocks of 256).
__global__ __launch_bounds__(256) void creduce_kernel(CArgs g) {
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

// Split-K reduce with the statistics epilogues (EPI 1 / 2): a workgroup owns 128 rows x 64 columns; thread (rl, cl)
// walks rows rl, rl+16, ... of 4 columns; partial[row-tile][2][N] exactly as the un-split kernel writes it.
template <int EPI>
__global__ __launch_bounds__(256) void creduce_stats_kernel(CArgs g) {
    __shared__ float red[16][2][64 + 1];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int r0 = blockIdx.y * TM, r1 = min(g.M, r0 + TM);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < g.N) {
        f32x4 sft = {0.f, 0.f, 0.f, 0.f}, mu = sft, is = sft, ga = sft, be = sft;
        if (EPI == 1 && g.stat_shift) sft = *reinterpret_cast<const f32x4*>(g.stat_shift + c);
        if (EPI == 2) {
            mu = *reinterpret_cast<const f32x4*>(g.emean + c);
            is = *reinterpret_cast<const f32x4*>(g.einvstd + c);
            ga = *reinterpret_cast<const f32x4*>(g.egamma + c);
            be = *reinterpret_cast<const f32x4*>(g.ebeta + c);
        }
        for (int r = r0 + rl; r < r1; r += 16) {
            const long idx = (long)r * g.N + c;
            f32x4 v = *reinterpret_cast<const f32x4*>(g.ws + idx);
            for (int s = 1; s < g.S; ++s) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(g.ws + (long)s * g.M * g.N + idx);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += w[k];
            }
            if (EPI == 2) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(g.ez + (long)r * g.ldz + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (zz[k] - mu[k]) * is[k];
                    if (!(fmaf(xh, ga[k], be[k]) > 0.f)) v[k] = 0.f;
                    s1[k] += v[k];
                    s2[k] = fmaf(v[k], xh, s2[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[k] - sft[k];
                    s1[k] += d;
                    s2[k] = fmaf(d, d, s2[k]);
                }
            }
            *reinterpret_cast<f32x4*>(g.C + (long)r * g.ldc + c) = v;
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
        if (blockIdx.x * 64 + cc < g.N) {
            float v = red[0][which][cc];
#pragma unroll
            for (int i = 1; i < 16; ++i) v += red[i][which][cc];
            g.stat_partial[((long)blockIdx.y * 2 + which) * g.N + blockIdx.x * 64 + cc] = v;
        }
    }
}

}  // namespace

int g_cgemm_target = 512;     // aim for this many workgroups (tiles x splits) when the tile grid alone is < 256
int g_cgemm_kmin = 128;       // at least this much K per split

bool cgemm_supported(bool tA, bool tB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                     long sA, long sB) {
    if (!(aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0 && sA % 4 == 0 && sB % 4 == 0 && K >= 1)) return false;
    // the k-contiguous image loads 16 bytes along k, the m/n-contiguous one along m/n
    if (!tA && K % 4) return false;
    if (tA && M % 4) return false;
    if (tB && K % 4) return false;
    if (!tB && N % 4) return false;
    const long abytes = (tA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 4;
    const long bbytes = (tB ? ((long)(N - 1) * ldb + K) : ((long)(K - 1) * ldb + N)) * 4;
    return abytes < 0x7fffffffL && bbytes < 0x7fffffffL;
}

// pro: 0 none, 1 A prologue (needs !tA), 2 B prologue (needs !tB);  epi: 0 plain, 1 stats, 2 mask+stats.
int cgemm(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda, const float* B,
          long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask, int batch, long sA, long sB,
          long sC, float* ws, long ws_floats, const ConvExtra* ex) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    SCN_ARG(A && B && C, "cgemm: null operand");
    SCN_ARG(cgemm_supported(tA, tB, M, N, K, A, lda, B, ldb, sA, sB), "cgemm: operand alignment / size not supported");
    SCN_ARG(beta == 0.f || ((long)(M - 1) * ldc + N) * 4 < 0x7fffffffL, "cgemm: C too large for beta != 0");
    const int pro = ex ? ex->pro : 0, epi = ex ? ex->epi : 0;
    const bool gather = ex && ex->stride > 1;
    SCN_ARG(pro == 0 || (pro == 1 && !tA) || (pro == 2 && !tB && tA), "cgemm: prologue / layout mismatch");
    SCN_ARG(epi == 0 || (batch == 1 && beta == 0.f && !bias && !rowmask && ex->stat_partial), "cgemm: statistics epilogue needs a plain product");
    SCN_ARG(epi != 2 || (ex->ez && ex->emean && ex->einvstd && ex->egamma && ex->ebeta && N % 4 == 0 && ex->ldz % 4 == 0),
            "cgemm: mask epilogue arguments");
    SCN_ARG(pro == 0 || ex->pro_ss, "cgemm: prologue table");
    SCN_ARG(!gather || (ex->Hi > 0 && ex->Wi > 0 && ex->Ho > 0 && ex->Wo > 0 && batch == 1), "cgemm: gather geometry");
    const int mt = cdiv(M, TM), nt = cdiv(N, TN);
    const long tiles = (long)mt * nt * batch;
    int S = 1;
    if (ws && tiles < 256 && K >= 2 * g_cgemm_kmin) {
        S = (int)((g_cgemm_target + tiles - 1) / tiles);
        const int smax = K / g_cgemm_kmin;
        if (S > smax) S = smax;
        if (S > SCN_MAX_KSPLIT) S = SCN_MAX_KSPLIT;
        while (S > 1 && (long)S * batch * M * N > ws_floats) --S;
        if (S < 1) S = 1;
    }
    if (ex && ex->force_split > 0) {
        S = ex->force_split;
        SCN_ARG(S == 1 || (ws && (long)S * batch * M * N <= ws_floats && S <= SCN_MAX_KSPLIT), "cgemm: forced split does not fit");
    }
    if (epi && S > 1) SCN_ARG(N % 4 == 0 && ldc % 4 == 0, "cgemm: split-K statistics epilogue needs N % 4 == 0");
    int kper = cdiv(K, S);
    kper = (kper + TK - 1) / TK * TK;
    S = cdiv(K, kper);
    CArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.rowmask = rowmask;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.S = S; g.kper = kper; g.ws = ws; g.mt = mt; g.nt = nt;
    if (ex) {
        g.gHi = ex->Hi; g.gWi = ex->Wi; g.gHo = ex->Ho; g.gWo = ex->Wo; g.gs = gather ? ex->stride : 0;
        g.pro_ss = ex->pro_ss;
        g.stat_partial = ex->stat_partial; g.stat_shift = ex->stat_shift;
        g.ez = ex->ez; g.emean = ex->emean; g.einvstd = ex->einvstd; g.egamma = ex->egamma; g.ebeta = ex->ebeta; g.ldz = ex->ldz;
    }
    dim3 grid(mt * nt, batch * S), block(256);
    const int kepi = S > 1 ? 0 : epi;     // with split-K the reduce kernel carries the epilogue
#define SCN_CG(AMC, BMC, PRO_, EPI_, G_) hipLaunchKernelGGL((cgemm_kernel<AMC, BMC, PRO_, EPI_, G_>), grid, block, 0, st, g)
    if (!tA && tB) {            // forward of a 1x1 convolution, nn.Linear: both k-contiguous
        if (gather) { if (pro == 1) { if (kepi == 1) SCN_CG(false, false, 1, 1, true); else SCN_CG(false, false, 1, 0, true); }
                      else { if (kepi == 1) SCN_CG(false, false, 0, 1, true); else SCN_CG(false, false, 0, 0, true); } }
        else if (pro == 1) { if (kepi == 1) SCN_CG(false, false, 1, 1, false); else SCN_CG(false, false, 1, 0, false); }
        else { if (kepi == 1) SCN_CG(false, false, 0, 1, false); else if (kepi == 2) SCN_CG(false, false, 0, 2, false); else SCN_CG(false, false, 0, 0, false); }
    } else if (!tA && !tB) {    // dgrad: A k-contiguous, B [K][N]
        SCN_ARG(!gather && pro == 0, "cgemm: NN product takes no gather / prologue");
        if (kepi == 2) SCN_CG(false, true, 0, 2, false); else if (kepi == 1) SCN_CG(false, true, 0, 1, false); else SCN_CG(false, true, 0, 0, false);
    } else if (tA && !tB) {     // wgrad: A [K][M], B [K][N]
        SCN_ARG(kepi == 0, "cgemm: TN product takes no statistics epilogue");
        if (gather) { if (pro == 2) SCN_CG(true, true, 2, 0, true); else SCN_CG(true, true, 0, 0, true); }
        else { if (pro == 2) SCN_CG(true, true, 2, 0, false); else SCN_CG(true, true, 0, 0, false); }
    } else {
        SCN_ARG(!gather && pro == 0 && kepi == 0, "cgemm: TT product is plain");
        SCN_CG(true, false, 0, 0, false);
    }
#undef SCN_CG
    SCN_LAUNCH_CHECK();
    if (S > 1) {
        if (epi == 1) hipLaunchKernelGGL(creduce_stats_kernel<1>, dim3(cdiv(N, 64), mt), block, 0, st, g);
        else if (epi == 2) hipLaunchKernelGGL(creduce_stats_kernel<2>, dim3(cdiv(N, 64), mt), block, 0, st, g);
        else hipLaunchKernelGGL(creduce_kernel, dim3(cdiv((long)M * N, 256), batch), block, 0, st, g);
        SCN_LAUNCH_CHECK();
    }
    return 0;
}

int cgemm_row_tiles(int M) { return cdiv(M, TM); }

}  // namespace scn
