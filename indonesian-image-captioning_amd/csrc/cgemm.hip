// fp32 GEMM core of the ResNet-152 1x1-convolution path (and of every 16-byte-aligned dense product of the
// decoder), gfx950 only:   C[M][N] = alpha * op(A) . op(B) (+ epilogues), exact fp32 (v_mfma_f32_32x32x2_f32).
//
// What it replaces: the `torch.nn.Conv2d(k=1)` calls of torchvision's Bottleneck behind the reference's
// models/encoders/caption.py:17-22 (forward, dgrad, wgrad -- on channels-last maps a 1x1 convolution is the GEMM
// [R = N*H*W, Cin] x [Cin, Cout]) together with the BatchNorm work that can ride on it, and the aten::mm/addmm calls
// of models/attention.py:35, attention_scn.py:154, scn_cell.py:73-86 that csrc/sgemm.hip served in round 1.
//
// Structure (MI355X_MICROARCH / cdna_hip_programming "Pipelining across barriers"):
//   * 128x128x16 block tile, 4 waves as 2x2, each wave 2x2 MFMA 32x32 tiles (MI = 2); a 64x128x16 variant (MI = 1,
//     each wave 1x2 tiles) for products whose 128-row tile grid cannot fill 256 CUs but whose 64-row grid can;
//   * operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, hardware
//     range checking: rows / k beyond the matrix land as zeros), 3-slot ring + two register sets of MFMA fragments;
//     fragment reads and LDS-DMA pieces are issued INSIDE the MFMA chain (one per gap), between two chains there is
//     ONE counted `s_waitcnt vmcnt(N)` and ONE raw s_barrier (round 3; see the k-loop);
//   * two LDS images, chosen per operand by which of its dimensions is contiguous in memory:
//       KC (k contiguous, e.g. activations [R][Cin], weights [Cout][Cin]):  [row][16 k], 64-byte rows, the four
//          16-byte granules of a row XOR-swizzled by (row>>2)&3 on the SOURCE address (the LDS-DMA destination is
//          lane-linear) so that a fragment read is two conflict-free ds_read_b128;
//       MC (m/n contiguous, e.g. dY [R][Cout] as the [K][M] operand of wgrad): [k][128], fragment = 8 ds_read_b32;
//     the MFMA k-slot of lane half h in MFMA j of a k-step is k0 + 8h + j for BOTH operands (any pairing is legal as
//     long as A and B agree), which is what makes the KC fragment 32 contiguous bytes;
//   * XCD-aware tile order: the 8 XCDs own contiguous runs of row panels, n fastest, so an activation panel is
//     fetched from HBM once per XCD and the weights stay in that XCD's L2;
//   * optional A-operand prologue: a = relu(a*scale[k] + shift[k]) (the previous layer's BatchNorm + ReLU folded to one
//     fma, per input channel) as an in-place pass over the landed tile, in two pieces inside the MFMA chain -- the
//     normalised map is never written to HBM; scale/shift of the k-step travel through the LDS ring with the tile;
//   * optional per-column prologue on a MC B operand (wgrad: B = relu(bn(z)) with the channel on the column);
//   * optional statistics epilogue: per (64-row block, column) sums of (y - s) and (y - s)^2 for the NEXT BatchNorm
//     (s = a per-channel shift for conditioning, e.g. the running mean), fixed order, no atomics, taken straight from
//     the accumulators before the stores are issued;
//   * optional mask epilogue for d-input products (EPI 2; round 3: inside the launch): g = dx * [relu-mask recomputed
//     from z], column sums of g and g*xhat (the BatchNorm backward reductions);  cstats_kernel<2> is the same as a second
//     launch for the layouts the in-launch form does not cover;
//   * C rows (and split-K slab rows) leave through a wave-private LDS transpose as 16-byte range-checked buffer stores
//     (16 store instructions per wave instead of 64);
//   * split-K into slabs, for shapes whose tile grid alone cannot fill 256 CUs; the slabs are summed in slab order
//     (deterministic) by the workgroup that arrives last at a tile, inside the launch, epilogue included (round 3; S <= 8),
//     or by a second launch (deeper splits, scalar stores).
#include "common.h"
#include "kernels.h"
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace scn {

int cgemm_stat_ld(int M);

namespace {

constexpr int TN = 128, TK = 16, NSTAGE = 3;
constexpr int TILE_F = 128 * TK;                // floats reserved per operand tile (8 KiB)
constexpr int AUX_F = 4 * 64;                   // per-wave {scale, shift} pairs of the current k-step (1 KiB)
constexpr int STAGE_F = 2 * TILE_F + AUX_F;     // 17 KiB per stage -> 51 KiB per workgroup, 3 workgroups per CU
constexpr int SROWS = 64;                       // rows per statistics partial

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
    int gHi, gWi, gHo, gWo, gs;   // row gather (strided 1x1 convolution); gs == 0: none.  3x3: source / destination maps
    int c3c;                      // 3x3 modes: channels per tap of the gathered operand (Cin forward / wgrad, Cout dgrad)
    long src_rows;                // 3x3 modes: rows of the gathered map (N * gHi * gWi)
    int dHi, dWi;                 // mode 4: extent of the d-input map the rows are scattered into
    const float* pro_ss;          // interleaved {scale, shift} per channel: PRO 1 per k (A), PRO 2 per n (B)
    float* stat_partial; const float* stat_shift;       // [2][N][ldp]: channel-major, one entry per 64-row block (chunk)
    int ldp;                      // leading dimension of stat_partial: cdiv(M, 64) rounded up to 4 (cgemm_stat_ld)
    const float* ez; const float* emean; const float* einvstd; const float* egamma; const float* ebeta;  // mask pass
    long ldz;
    int* cnt;                     // in-launch split-K combine: one arrival counter per (batch, tile), zero at rest; null -> two launches
    int comb;                     // 1: write-through (sc1) slab stores, 2: plain slab stores + agent release
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
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}

__device__ __forceinline__ void buf_store4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {   // write-through
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

// MI: 32-row MFMA tiles per wave along m (block tile = 64*MI x 128, waves 2 x 2).
// MI = 4 is a different WAVE LAYOUT, not a bigger register block: the four waves stack along m (4 x 1), each owning
// one 32-row block of ALL the columns the tile has -- a 128 x 64 tile for outputs that are at most 64 wide (the
// 64-channel maps of layer1).  With the 2 x 2 layout half of such a tile's MFMAs multiply zero columns.
// A_MC / B_MC: operand stored with its m / n dimension contiguous ([K][M] / [K][N]); otherwise k contiguous.
// PRO: 0 none, 1 relu(a*scale[k]+shift[k]) on a KC A operand, 2 relu(b*scale[n]+shift[n]) on a MC B operand.
// EPI: 0 plain (alpha, beta, bias, rowmask), 1 plain store + column statistics, 2 ReLU mask of the consumer BatchNorm
//      recomputed from its pre-activation z + the two column sums of its backward (g, g*xhat), g stored.
//      A split product (S > 1) with arrival counters (g.cnt) applies its epilogue IN THE SAME LAUNCH: every slice writes
//      its slab, takes a ticket, and the workgroup that draws the last ticket of a tile sums the S slabs in slab order
//      (deterministic) and finishes the tile -- no second launch, no second trip of the product through HBM.
// GATHER: rows of a KC A operand or k-rows of a MC B operand are gathered (strided 1x1 convolution).
// VEC: the output takes 16-byte row stores.
// C3: 3x3 convolution (pad 1) as an implicit GEMM -- no im2col buffer, the taps are a walk over K (forward, dgrad) or a
//     property of the column tile (wgrad), out-of-image taps are lanes whose LDS-DMA offset is out of range (zeros):
//     1 forward  Y[(n,ho,wo)][co] = sum_{tap,ci} X[(n, ho*s+dh-1, wo*s+dw-1)][ci] * W[co][tap][ci]     (A rows gathered per tap)
//     2 dgrad    dX[(n,hi,wi)][ci] = sum_{tap,co} dY[(n, hi-dh+1, wi-dw+1)][co] * W[co][tap][ci]       (stride 1; B rows (tap,co))
//     3 wgrad    dW[co][tap][ci]  = sum_r dY[r][co] * X[(n, ho*s+dh-1, wo*s+dw-1)][ci]                 (B k-rows gathered, tap per n-tile)
//     4 dgrad of a STRIDE-2 convolution, one parity class (ph, pw) = (hi & 1, wi & 1) per blockIdx.y: only the taps with
//       hi + 1 - dh even reach an output row, so the classes see 1 / 2 / 2 / 4 of the nine taps (K = taps * Cout) -- no
//       multiplication by the zeros a zero-inserted dY would carry.  Class row m = (n, ho', wo') of the Ho x Wo grid;
//       it gathers dY[(n, ho' + oh, wo' + ow)] (oh = 1 for the tap dh = 0 of an odd row, else 0) and is stored at d-input
//       row (n, 2 ho' + ph, 2 wo' + pw).
template <int MI, bool A_MC, bool B_MC, int PRO, int EPI, bool GATHER, bool VEC, int C3 = 0>
__global__ __launch_bounds__(256, MI == 2 ? 2 : 3) void cgemm_kernel(CArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE_F];
    constexpr bool W41 = (MI == 4);
    constexpr int RB = W41 ? 1 : MI;                   // 32-row MFMA blocks per wave
    constexpr int TM = W41 ? 128 : 64 * MI;
    constexpr int ACH = TM / 64;                       // A chunks (1 KiB LDS-DMA pieces) per wave per tile
    constexpr int LPT = ACH + 2 + (PRO == 1 ? 1 : 0);  // LDS-DMA instructions per wave per tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = W41 ? wave : wave >> 1, wn = W41 ? 0 : wave & 1;
    const int hh = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware tile order (speed only): blocks b, b+8, b+16 ... share an XCD ---------------------------
    const int ntiles = g.mt * g.nt;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / g.nt, tn = bid - tm * g.nt;
    const int m0 = tm * TM, n0 = tn * (W41 ? 64 : TN);      // 4 x 1 layout: 64-column tiles
    const int cls = C3 == 4 ? 3 - (int)blockIdx.y : 0, ph = cls >> 1, pw = cls & 1, ntw = 1 + pw;   // mode 4: parity class, the 4-tap one dispatched first
    const int zb = C3 == 4 ? 0 : blockIdx.y / g.S, sp = C3 == 4 ? 0 : blockIdx.y - zb * g.S;
    const float* A = g.A + (long)zb * g.sA;
    const float* B = g.B + (long)zb * g.sB;
    float* C = g.C + (long)zb * g.sC;
    const int kbeg = sp * g.kper, Kend = C3 == 4 ? (1 + ph) * ntw * g.c3c : min(g.K, kbeg + g.kper);
    const int nk = (Kend - kbeg + TK - 1) / TK;

    const long a_elems = (C3 == 1 || C3 == 2 || C3 == 4) ? g.src_rows * g.lda
                         : A_MC ? ((long)(g.K - 1) * g.lda + g.M)
                                : (GATHER ? (gather_row(g, g.M - 1) * g.lda + g.K) : ((long)(g.M - 1) * g.lda + g.K));
    const long b_elems = C3 == 3 ? g.src_rows * g.ldb
                         : (C3 == 2 || C3 == 4) ? (long)(g.K / 9) * g.ldb
                         : B_MC ? ((GATHER && A_MC ? gather_row(g, g.K - 1) : (long)(g.K - 1)) * g.ldb + g.N)
                                : ((long)(g.N - 1) * g.ldb + g.K);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(A, (unsigned)(a_elems * 4));
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(B, (unsigned)(b_elems * 4));
    const __amdgpu_buffer_rsrc_t srs = make_rsrc(g.pro_ss, PRO == 1 ? (unsigned)g.K * 8u : 0u);

    // ---- per-lane source offsets of the chunks this wave stages per operand ---------------------------------
    // KC: chunk = 16 rows x 64 B; lane -> row chunk*16 + lane/4, LDS granule lane&3 <- source granule (lane&3)^((row>>2)&3)
    // MC: chunk = 2 k-rows x 512 B (256 B when the tile is 64 wide: A with MI = 1); lane -> k-row, 4 columns
    unsigned a_off[ACH], b_off[2];   // byte offsets at k = kbeg (KC: + k*4; MC: + k*ld*4 per step)
    bool a_ok[ACH], b_ok[2];
    int a_nb[ACH], a_h0[ACH], a_w0[ACH];   // 3x3: image base row, top-left tap position of this lane's output pixel
#pragma unroll
    for (int c = 0; c < ACH; ++c) {
        const int chunk = wave * ACH + c;
        if (C3 == 1 || C3 == 2 || C3 == 4) {
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = m0 + row;
            a_ok[c] = grow < g.M;
            const int r = a_ok[c] ? grow : 0, hw = g.gHo * g.gWo;
            const int n = r / hw, rem = r - n * hw, hd = rem / g.gWo, wd = rem - hd * g.gWo;
            a_nb[c] = n * g.gHi * g.gWi;
            a_h0[c] = C3 == 4 ? hd : hd * g.gs - 1;
            a_w0[c] = C3 == 4 ? wd : wd * g.gs - 1;
            a_off[c] = (unsigned)(gsrc * 16);
        } else if (!A_MC) {
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = m0 + row;
            a_ok[c] = grow < g.M;
            const long src = (GATHER ? gather_row(g, a_ok[c] ? grow : 0) : (long)grow) * g.lda + gsrc * 4;
            a_off[c] = (unsigned)(src * 4);
        } else if (TM == 128) {
            const int col = m0 + 4 * (lane & 31);
            a_ok[c] = col < g.M;
            a_off[c] = (unsigned)(((long)(chunk * 2 + (lane >> 5)) * g.lda + col) * 4);
        } else {        // 64-wide [k][64] image: a chunk = 4 k-rows x 256 B
            const int col = m0 + 4 * (lane & 15);
            a_ok[c] = col < g.M;
            a_off[c] = (unsigned)(((long)(chunk * 4 + (lane >> 4)) * g.lda + col) * 4);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int chunk = wave * 2 + c;
        if (!B_MC) {
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
            const int grow = n0 + row;
            b_ok[c] = grow < g.N && (!W41 || row < 64);
            b_off[c] = (unsigned)(((long)grow * g.ldb + gsrc * 4) * 4);
        } else {
            const int col = n0 + 4 * (lane & 31);
            b_ok[c] = col < g.N && (!W41 || (lane & 31) < 16);
            b_off[c] = (unsigned)(((long)(chunk * 2 + (lane >> 5)) * g.ldb + col) * 4);
        }
    }

    // LDS-DMA of k-step kt into ring slot `stage`, one 1-KiB piece per call: P < ACH the A chunks, then the two B chunks,
    // then (PRO 1) the scale/shift table.  A k-step past the end (k0 >= Kend) zero-fills: the pipeline below never branches.
    auto issue_piece = [&](int kt, int stage, auto pc) {
        constexpr int P = decltype(pc)::value;
        float* sa = lds + stage * STAGE_F;
        float* sb = sa + TILE_F;
        const int k0 = kbeg + kt * TK;
        if constexpr (P < ACH) {
            constexpr int c = P;
            const int chunk = wave * ACH + c;
            unsigned va;
            if (C3 == 4) {                  // class tap t = (th, tw): source pixel (ho' + oh, wo' + ow)
                const int t = k0 / g.c3c, c0 = k0 - t * g.c3c, th = t / ntw, tw = t - th * ntw;
                const int hi = a_h0[c] + ((ph && th == 0) ? 1 : 0), wi = a_w0[c] + ((pw && tw == 0) ? 1 : 0);
                const bool ok = a_ok[c] && k0 < Kend && hi < g.gHi && wi < g.gWi;
                va = ok ? (unsigned)(((long)(a_nb[c] + hi * g.gWi + wi) * g.lda + c0) * 4) + a_off[c] : OOB_OFF;
            } else if (C3 == 1 || C3 == 2) {       // k-step -> one tap (c3c % 16 == 0), channels c0 .. c0+15 of it
                const int tap = k0 / g.c3c, c0 = k0 - tap * g.c3c;
                int dh = tap / 3, dw = tap - 3 * dh;
                if (C3 == 2) { dh = 2 - dh; dw = 2 - dw; }
                const int hi = a_h0[c] + dh, wi = a_w0[c] + dw;
                const bool ok = a_ok[c] && k0 < Kend && (unsigned)hi < (unsigned)g.gHi && (unsigned)wi < (unsigned)g.gWi;
                va = ok ? (unsigned)(((long)(a_nb[c] + hi * g.gWi + wi) * g.lda + c0) * 4) + a_off[c] : OOB_OFF;
            } else if (!A_MC) {
                const int kk = k0 + 4 * ((lane & 3) ^ (((chunk * 16 + (lane >> 2)) >> 2) & 3));
                va = (a_ok[c] && kk < Kend) ? a_off[c] + (unsigned)k0 * 4u : OOB_OFF;
            } else {
                const int kr = k0 + (TM == 128 ? chunk * 2 + (lane >> 5) : chunk * 4 + (lane >> 4));
                va = (a_ok[c] && kr < Kend) ? a_off[c] + (unsigned)((long)k0 * g.lda * 4) : OOB_OFF;
            }
            dma16(ars, sa + chunk * 256, va);
        }
        if constexpr (P >= ACH && P < ACH + 2) {
            constexpr int c = P - ACH;
            const int chunk = wave * 2 + c;
            unsigned vb;
            if (!B_MC) {
                const int kk = k0 + 4 * ((lane & 3) ^ (((chunk * 16 + (lane >> 2)) >> 2) & 3));
                vb = (b_ok[c] && kk < Kend) ? b_off[c] + (unsigned)k0 * 4u : OOB_OFF;
            } else {
                const int kr = k0 + chunk * 2 + (lane >> 5);
                if (C3 == 4) {            // B row k = (class tap t, co): the weight tap is (dh, dw) = (ph ? 2*th : 1, pw ? 2*tw : 1)
                    const int t = k0 / g.c3c, co = kr - t * g.c3c, th = t / ntw, tw = t - th * ntw;
                    const int tap = (ph ? 2 * th : 1) * 3 + (pw ? 2 * tw : 1);
                    vb = (b_ok[c] && kr < Kend) ? (unsigned)(((long)co * g.ldb + (long)tap * g.N + n0 + 4 * (lane & 31)) * 4) : OOB_OFF;
                } else if (C3 == 2) {            // B row k = (tap, co) of W[co][tap][ci]: co*ldb + tap*Cin + ci
                    const int tap = k0 / g.c3c, co = kr - tap * g.c3c;
                    vb = (b_ok[c] && kr < Kend) ? (unsigned)(((long)co * g.ldb + (long)tap * g.N + n0 + 4 * (lane & 31)) * 4) : OOB_OFF;
                } else if (C3 == 3) {     // B k-row r = output pixel; the column tile fixes the tap (c3c % 128 == 0)
                    const int tap = n0 / g.c3c, ci0 = n0 - tap * g.c3c, dh = tap / 3, dw = tap - 3 * dh;
                    const int r = kr < Kend ? kr : 0, hw = g.gHo * g.gWo;
                    const int n = r / hw, rem = r - n * hw, ho = rem / g.gWo, wo = rem - ho * g.gWo;
                    const int hi = ho * g.gs + dh - 1, wi = wo * g.gs + dw - 1;
                    const bool ok = b_ok[c] && kr < Kend && (unsigned)hi < (unsigned)g.gHi && (unsigned)wi < (unsigned)g.gWi;
                    vb = ok ? (unsigned)(((long)(n * g.gHi * g.gWi + hi * g.gWi + wi) * g.ldb + ci0 + 4 * (lane & 31)) * 4) : OOB_OFF;
                } else if (GATHER && A_MC) {     // wgrad of a strided convolution: k-rows of B = gathered rows of the input map
                    const long src = gather_row(g, kr < Kend ? kr : 0) * g.ldb + n0 + 4 * (lane & 31);
                    vb = (b_ok[c] && kr < Kend) ? (unsigned)(src * 4) : OOB_OFF;
                } else {
                    vb = (b_ok[c] && kr < Kend) ? b_off[c] + (unsigned)((long)k0 * g.ldb * 4) : OOB_OFF;
                }
            }
            dma16(brs, sb + chunk * 256, vb);
        }
        if constexpr (PRO == 1 && P == ACH + 2) {   // this wave's private copy of {scale, shift}[k0 .. k0+15] (32 floats, one instruction)
            float* sx = sa + 2 * TILE_F + wave * 64;
            const int kk = k0 + (lane >> 1);
            dma4(srs, sx, (lane < 32 && kk < Kend) ? (unsigned)(2 * k0 + lane) * 4u : OOB_OFF);
        }
    };

    auto issue = [&](int kt, int stage) {
        issue_piece(kt, stage, std::integral_constant<int, 0>{});
        if constexpr (LPT > 1) issue_piece(kt, stage, std::integral_constant<int, 1>{});
        if constexpr (LPT > 2) issue_piece(kt, stage, std::integral_constant<int, 2>{});
        if constexpr (LPT > 3) issue_piece(kt, stage, std::integral_constant<int, 3>{});
        if constexpr (LPT > 4) issue_piece(kt, stage, std::integral_constant<int, 4>{});
    };

    f32x16 acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
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

    // PRO 1: normalise-on-load as an in-place pass over the part of the A tile THIS wave staged, between the wait that
    // retires its DMA and the barrier that publishes the tile: every element once (not once per consuming wave) and off
    // the MFMA dependency chain.  Lane -> one 16-byte granule = 4 consecutive k of one row; its {scale, shift} pairs
    // sit in this wave's copy of the k-step's table.
    auto prologue_in_lds = [&](int stg) {
        float* sa = lds + stg * STAGE_F;
        const float* sx = sa + 2 * TILE_F + wave * 64;
#pragma unroll
        for (int c = 0; c < ACH; ++c) {
            const int chunk = wave * ACH + c;
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);     // source granule = k/4
            float* p = sa + chunk * 256 + lane * 4;
            f32x4 v = *reinterpret_cast<f32x4*>(p);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(sx + 8 * gsrc), t1 = *reinterpret_cast<const f32x4*>(sx + 8 * gsrc + 4);
            v[0] = fmaxf(fmaf(v[0], t0[0], t0[1]), 0.f);
            v[1] = fmaxf(fmaf(v[1], t0[2], t0[3]), 0.f);
            v[2] = fmaxf(fmaf(v[2], t1[0], t1[1]), 0.f);
            v[3] = fmaxf(fmaf(v[3], t1[2], t1[3]), 0.f);
            *reinterpret_cast<f32x4*>(p) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // ---- the k-loop: a three-slot LDS ring fed by LDS-DMA PLUS two register sets of MFMA fragments, and NOTHING between
    // two MFMA chains but one wait and one barrier.  K-step kt multiplies register set kt&1; INSIDE its MFMA chain, one
    // piece per MFMA gap, travel (a) the fragments of k-step kt+1, LDS -> the other register set, and (b) the LDS-DMA pieces
    // of k-step kt+3 into the slot k-step kt has just left (its fragments are in registers since the last barrier).  An
    // LDS-DMA piece holds the wave's issue for 60-180 cycles (MI355X_MICROARCH.md, per-instruction constants): issued
    // between the chains -- as a plain "prefetch, then compute" loop does -- three to five of them cost a workgroup that is
    // alone on its CU (one wave per SIMD: the 8192-row maps of layer3) a quarter of the k-step; inside the chain they run in
    // the shadow of the 64-cycle MFMAs.  End of k-step kt: this wave's pieces of k-step kt+2 have landed (kt+3 stays in
    // flight), barrier -> kt+2 is published and every wave is done reading the slot of kt+1.
    // The pipeline never branches: k-steps past the end are zero-fill DMAs into dead slots and fragment reads nobody uses.
    auto load_piece = [&](int stg, auto pc, float (&a)[RB][8], float (&b)[2][8]) {
        constexpr int P = decltype(pc)::value;
        const float* sa = lds + stg * STAGE_F;
        const float* sb = sa + TILE_F;
        if constexpr (P < RB) {
            constexpr int i = P;
            if (!A_MC) {
                const int row = wm * 32 * RB + i * 32 + l31, sw = (row >> 2) & 3;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sa + row * 16 + (((2 * hh) ^ sw) << 2));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sa + row * 16 + (((2 * hh + 1) ^ sw) << 2));
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[i][q] = v0[q]; a[i][4 + q] = v1[q]; }
            } else {
                const int col = wm * 32 * RB + i * 32 + l31;
#pragma unroll
                for (int q = 0; q < 8; ++q) a[i][q] = sa[(8 * hh + q) * TM + col];
            }
        } else {
            constexpr int j = P - RB;
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
    };
    auto load_frags = [&](int stg, float (&a)[RB][8], float (&b)[2][8]) {
        load_piece(stg, std::integral_constant<int, 0>{}, a, b);
        load_piece(stg, std::integral_constant<int, 1>{}, a, b);
        load_piece(stg, std::integral_constant<int, 2>{}, a, b);
        if constexpr (RB == 2) load_piece(stg, std::integral_constant<int, 3>{}, a, b);
    };
    auto slot = [](int kt) { return kt % NSTAGE; };
    // PRO 1 inside the chain: the in-place normalise pass over this wave's part of k-step kt+2 in two pieces -- its LDS
    // reads right behind the wait that retires those DMA pieces, the fma/max + write-back a few MFMAs later -- so that the
    // LDS round trip runs in the MFMA shadow instead of between the chain and the barrier.
    [[maybe_unused]] f32x4 pv[ACH], pt0[ACH], pt1[ACH];
    auto pro_read = [&](int stg) {
        const float* sa = lds + stg * STAGE_F;
        const float* sx = sa + 2 * TILE_F + wave * 64;
#pragma unroll
        for (int c = 0; c < ACH; ++c) {
            const int chunk = wave * ACH + c;
            const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
            pv[c] = *reinterpret_cast<const f32x4*>(sa + chunk * 256 + lane * 4);
            pt0[c] = *reinterpret_cast<const f32x4*>(sx + 8 * gsrc);
            pt1[c] = *reinterpret_cast<const f32x4*>(sx + 8 * gsrc + 4);
        }
    };
    auto pro_apply = [&](int stg) {
        float* sa = lds + stg * STAGE_F;
#pragma unroll
        for (int c = 0; c < ACH; ++c) {
            const int chunk = wave * ACH + c;
            f32x4 v = pv[c];
            v[0] = fmaxf(fmaf(v[0], pt0[c][0], pt0[c][1]), 0.f);
            v[1] = fmaxf(fmaf(v[1], pt0[c][2], pt0[c][3]), 0.f);
            v[2] = fmaxf(fmaf(v[2], pt1[c][0], pt1[c][1]), 0.f);
            v[3] = fmaxf(fmaf(v[3], pt1[c][2], pt1[c][3]), 0.f);
            *reinterpret_cast<f32x4*>(sa + chunk * 256 + lane * 4) = v;
        }
    };
    // pieces of a k-step, in chain order.  Plain: fragment loads, then DMA.  PRO 1: DMA, [wait + normalise reads], fragment
    // loads, [normalise + write-back].
    constexpr int NPL = RB + 2, NM = 16 * RB;                      // load pieces, MFMAs of a k-step
    constexpr int NPC = NPL + LPT + (PRO == 1 ? 2 : 0);            // all pieces
    float fa0[RB][8], fb0[2][8], fa1[RB][8], fb1[2][8];
    auto kstep = [&](int kt, float (&ac)[RB][8], float (&bc)[2][8], float (&an)[RB][8], float (&bn)[2][8]) {
        const int s_next = slot(kt + 1), s_dma = slot(kt), s_pro = slot(kt + 2);
        auto piece = [&](auto pc) {
            constexpr int P = decltype(pc)::value;
            if constexpr (PRO != 1) {
                if constexpr (P < NPL) load_piece(s_next, pc, an, bn);
                else if constexpr (P < NPC) issue_piece(kt + 3, s_dma, std::integral_constant<int, P - NPL>{});
            } else {
                if constexpr (P < LPT) issue_piece(kt + 3, s_dma, pc);
                else if constexpr (P == LPT) { wait_vmcnt<LPT>(); pro_read(s_pro); }
                else if constexpr (P < LPT + 1 + NPL) load_piece(s_next, std::integral_constant<int, P - LPT - 1>{}, an, bn);
                else if constexpr (P < NPC) pro_apply(s_pro);
            }
        };
        if (PRO == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) bc[j][q] = fmaxf(fmaf(bc[j][q], bsc[j], bsh[j]), 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int m = (q * RB + i) * 2 + j;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i][q], bc[j][q], acc[i][j], 0, 0, 0);
                    // piece p rides behind MFMA p * NM / NPC: spread evenly over the chain
                    if (m == 0 * NM / NPC) piece(std::integral_constant<int, 0>{});
                    if (m == 1 * NM / NPC) piece(std::integral_constant<int, 1>{});
                    if (m == 2 * NM / NPC) piece(std::integral_constant<int, 2>{});
                    if (m == 3 * NM / NPC) piece(std::integral_constant<int, 3>{});
                    if (m == 4 * NM / NPC) piece(std::integral_constant<int, 4>{});
                    if (m == 5 * NM / NPC) piece(std::integral_constant<int, 5>{});
                    if (m == 6 * NM / NPC) piece(std::integral_constant<int, 6>{});
                    if (m == 7 * NM / NPC) piece(std::integral_constant<int, 7>{});
                    if (m == 8 * NM / NPC) piece(std::integral_constant<int, 8>{});
                    if (m == 9 * NM / NPC) piece(std::integral_constant<int, 9>{});
                    if (m == 10 * NM / NPC) piece(std::integral_constant<int, 10>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
        static_assert(NPC <= 11 && NPC <= NM, "more pieces than slots");
        if (PRO != 1) wait_vmcnt<LPT>();         // k-step kt+2 landed (this wave's pieces); kt+3 stays in flight
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) -- the builtin, so that hipcc's wait-count pass knows the set is in
        __builtin_amdgcn_s_barrier();
    };

    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LPT) : "memory");
    if (PRO == 1) prologue_in_lds(0);
    __builtin_amdgcn_s_barrier();
    load_frags(0, fa0, fb0);
    wait_vmcnt<LPT>();
    if (PRO == 1) prologue_in_lds(1);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                                // k-step 1 published, slot 0 read out
    for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, fa0, fb0, fa1, fb1);
        if (kt + 1 < nk) kstep(kt + 1, fa1, fb1, fa0, fb0);
    }
    wait_vmcnt<0>();                                             // the zero-fill pieces of the last k-steps: the epilogues reuse the ring
    __builtin_amdgcn_s_barrier();

    // ================================== epilogues =====================================================
    // The ring is dead by now (every wave passed the last barrier after its last fragment read).  Nothing waits on the
    // stores: a workgroup's slot is free as soon as they are issued -- which is why the statistics come BEFORE them.
    // An opaque zero that the offsets depend on keeps hipcc from computing the ~40 loop-invariant row offsets before
    // the k-loop and spilling them around the MFMA chain.
    unsigned opq = 0;
    asm volatile("" : "+v"(opq));
    float* const lw = lds + wave * 2048;           // [32 rows][64 cols]; 256-byte rows are conflict-free both ways
    float* const colsum = lds + 4 * 2048;          // [wm][2][TN] (MI = 1 only), behind the four transpose regions
    const int rl = lane >> 4, c4 = (lane & 15) * 4;
    const int ncol = n0 + wn * 64 + c4;
    const bool cok = ncol < g.N;
    const long c_ld = g.S > 1 ? (long)g.N : g.ldc;
    float* const c_base = g.S > 1 ? g.ws + ((long)blockIdx.y * g.M) * g.N : C;
    const __amdgpu_buffer_rsrc_t ors = make_rsrc(c_base, VEC ? (unsigned)(((long)((C3 == 4 ? 4 * g.M : g.M) - 1) * c_ld + g.N) * 4) : 0u);
    auto row_off = [&](int m) -> unsigned {
        if (!(m < g.M && cok)) return OOB_OFF;
        long rr = m;
        if (C3 == 4) {      // class row (n, ho', wo') -> d-input row (n, 2 ho' + ph, 2 wo' + pw)
            const int hw = g.gHo * g.gWo, n = m / hw, rem = m - n * hw, hd = rem / g.gWo, wd = rem - hd * g.gWo;
            rr = ((long)n * g.dHi + 2 * hd + ph) * g.dWi + 2 * wd + pw;
        }
        return (unsigned)((rr * c_ld + ncol) * 4) + opq;
    };
    // one 32-row block of this wave from the accumulators into the transpose region
    auto dump_half = [&](int i, float add0, float add1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lw[mfma32_row(r, lane) * 64 + j * 32 + l31] = g.alpha * acc[i][j][r] + (j ? add1 : add0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave wrote it: only the LDS counter drains
    };
    const int mw0 = m0 + wm * 32 * RB;             // first row of this wave

    // ---- finishing a 32-row block in ROW layout (lane = columns c4 .. c4+3 of rows it*4 + rl, it = 0 .. 7): the in-launch
    // split-K combine (all epilogues) and the un-split mask epilogue (EPI 2) ------------------------------------------------
    [[maybe_unused]] f32x4 kv0 = {0.f, 0.f, 0.f, 0.f}, kv1 = kv0, kv2 = kv0, kv3 = kv0;   // per-column constants of the epilogue
    const bool folded = EPI == 2 && g.pro_ss != nullptr;   // the mask of the function the forward pass evaluated: relu(fma(z, scale, shift))
    [[maybe_unused]] float fs1[4] = {0.f, 0.f, 0.f, 0.f}, fs2[4] = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] f32x4 xv[8];            // EPI 0: C rows (beta != 0); EPI 2: z rows
    [[maybe_unused]] float mk[8];
    const bool f_use_c = EPI == 0 && g.beta != 0.f, f_use_m = EPI == 0 && g.rowmask != nullptr;
    const __amdgpu_buffer_rsrc_t f_crs = make_rsrc(C, VEC ? (unsigned)(((long)(g.M - 1) * g.ldc + g.N) * 4) : 0u);
    const __amdgpu_buffer_rsrc_t f_zrs = make_rsrc(g.ez, (VEC && EPI == 2) ? (unsigned)(((long)(g.M - 1) * g.ldz + g.N) * 4) : 0u);
    const __amdgpu_buffer_rsrc_t f_mrs = make_rsrc(g.rowmask, f_use_m ? (unsigned)g.M * 4u : 0u);
    auto out_off = [&](int m) -> unsigned { return (m < g.M && cok) ? (unsigned)(((long)m * g.ldc + ncol) * 4) + opq : OOB_OFF; };
    auto finish_setup = [&]() {
        if (!cok) return;
        if (EPI == 0 && g.bias) kv0 = *reinterpret_cast<const f32x4*>(g.bias + ncol);
        if (EPI == 1 && g.stat_shift) kv0 = *reinterpret_cast<const f32x4*>(g.stat_shift + ncol);
        if (EPI == 2) {
            kv0 = *reinterpret_cast<const f32x4*>(g.emean + ncol);
            kv1 = *reinterpret_cast<const f32x4*>(g.einvstd + ncol);
            if (folded) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(g.pro_ss + 2 * ncol), t1 = *reinterpret_cast<const f32x4*>(g.pro_ss + 2 * ncol + 4);
                kv2 = f32x4{t0[0], t0[2], t1[0], t1[2]};
                kv3 = f32x4{t0[1], t0[3], t1[1], t1[3]};
            } else {
                kv2 = *reinterpret_cast<const f32x4*>(g.egamma + ncol);
                kv3 = *reinterpret_cast<const f32x4*>(g.ebeta + ncol);
            }
        }
    };
    // NV rows (it0 .. it0+NV-1 of the block's eight) per call: 8 when the accumulators are dead (combine), 4 while the
    // other 32-row block still sits in them (un-split mask epilogue) -- register pressure
    auto finish_load = [&](int i, int it0, auto nv) {   // issue the side loads (they fly while the product is fetched / dumped)
        constexpr int NV = decltype(nv)::value;
        const int mb = mw0 + i * 32;
        if (f_use_c) {
#pragma unroll
            for (int u = 0; u < NV; ++u) xv[u] = buf_load4(f_crs, out_off(mb + (it0 + u) * 4 + rl));
        }
        if (f_use_m) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int m = mb + (it0 + u) * 4 + rl;
                mk[u] = buf_load(f_mrs, m < g.M ? (unsigned)m * 4u + opq : OOB_OFF);
            }
        }
        if (EPI == 2) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int m = mb + (it0 + u) * 4 + rl;
                xv[u] = buf_load4(f_zrs, (m < g.M && cok) ? (unsigned)(((long)m * g.ldz + ncol) * 4) + opq : OOB_OFF);
            }
        }
    };
    auto finish_block = [&](int i, int it0, auto& v) {
        constexpr int NV = sizeof(v) / sizeof(v[0]);
        const int mb = mw0 + i * 32;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int m = mb + (it0 + u) * 4 + rl;
            if (EPI == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] += kv0[q];
                if (f_use_c) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[u][q] = fmaf(g.beta, xv[u][q], v[u][q]);
                }
                if (f_use_m && mk[u] == 0.f) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if (EPI == 1) {
                if (m < g.M) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float d = v[u][q] - kv0[q];
                        fs1[q] += d;
                        fs2[q] = fmaf(d, d, fs2[q]);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = xv[u][q], xh = (z - kv0[q]) * kv1[q];
                    const bool on = fmaf(folded ? z : xh, kv2[q], kv3[q]) > 0.f;
                    v[u][q] = on ? v[u][q] : 0.f;
                    fs1[q] += v[u][q];
                    fs2[q] = fmaf(v[u][q], xh, fs2[q]);
                }
            }
            buf_store4(f_crs, out_off(m), v[u]);
        }
    };
    auto finish_stats = [&]() {              // column sums of this wave's rows -> one partial per 64-row block
        if (EPI == 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fs1[q] += __shfl_xor(fs1[q], 16, 64); fs1[q] += __shfl_xor(fs1[q], 32, 64);
            fs2[q] += __shfl_xor(fs2[q], 16, 64); fs2[q] += __shfl_xor(fs2[q], 32, 64);
        }
        if (MI == 2) {
            if (lane < 16 && cok && mw0 < g.M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* p = g.stat_partial + (long)(ncol + q) * g.ldp + (tm * 2 + wm);
                    p[0] = fs1[q];
                    p[(long)g.N * g.ldp] = fs2[q];
                }
            }
            return;
        }
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                colsum[(wm * 2 + 0) * TN + wn * 64 + c4 + q] = fs1[q];
                colsum[(wm * 2 + 1) * TN + wn * 64 + c4 + q] = fs2[q];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (MI == 1) {
            if (tid < TN && n0 + tid < g.N) {
                float* p = g.stat_partial + (long)(n0 + tid) * g.ldp + tm;
                p[0] = colsum[0 * TN + tid] + colsum[2 * TN + tid];
                p[(long)g.N * g.ldp] = colsum[1 * TN + tid] + colsum[3 * TN + tid];
            }
        } else {      // 4 x 1 waves: (0,1) form the tile's first 64-row statistics block, (2,3) its second
            const int blk = tid >> 6, col = tid & 63;
            if (tid < 128 && n0 + col < g.N && m0 + blk * 64 < g.M) {
                float* p = g.stat_partial + (long)(n0 + col) * g.ldp + (tm * 2 + blk);
                p[0] = colsum[((2 * blk) * 2 + 0) * TN + col] + colsum[((2 * blk + 1) * 2 + 0) * TN + col];
                p[(long)g.N * g.ldp] = colsum[((2 * blk) * 2 + 1) * TN + col] + colsum[((2 * blk + 1) * 2 + 1) * TN + col];
            }
        }
    };

    if (g.S > 1) {   // split-K: raw alpha-scaled partial tile; the combine below or the reduce kernels apply the epilogue
        if constexpr (VEC) {
            const bool wt = g.cnt && g.comb == 1;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                dump_half(i, 0.f, 0.f);
                if (wt) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = it * 4 + rl;
                        buf_store4_sc1(ors, row_off(mw0 + i * 32 + row), *reinterpret_cast<const f32x4*>(lw + row * 64 + c4));
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = it * 4 + rl;
                        buf_store4(ors, row_off(mw0 + i * 32 + row), *reinterpret_cast<const f32x4*>(lw + row * 64 + c4));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the next block overwrites
            }
            if (!g.cnt) return;
            // ---- in-launch combine (cdna_hip_programming.md, in-launch split-K reduction; MI355X_MICROARCH.md, valid forms):
            // every storing wave drains its stores, the workgroup meets, ONE lane publishes (write-through slabs need no
            // write-back; plain ones an agent release) and takes the ticket; the last arriver acquires and reads the slabs.
            int* const flag = reinterpret_cast<int*>(lds + 4 * 2048 + 8 * TN);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (g.comb != 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                int* const ticket = g.cnt + zb * ntiles + bid;
                const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == g.S - 1;
                if (last) {
                    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // zero at rest for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *flag = last;
            }
            __syncthreads();
            if (!*flag) return;
            const __amdgpu_buffer_rsrc_t srs = make_rsrc(g.ws + ((long)zb * g.S * g.M) * g.N, (unsigned)(((long)g.S * g.M * g.N) * 4));
            finish_setup();
#pragma unroll 1
            for (int i = 0; i < RB; ++i) {
                finish_load(i, 0, std::integral_constant<int, 8>{});
                f32x4 v[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int mb = mw0 + i * 32;
                unsigned so[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mb + it * 4 + rl;
                    so[it] = (m < g.M && cok) ? (unsigned)(((long)m * g.N + ncol) * 4) + opq : OOB_OFF;
                }
                const unsigned sstep = (unsigned)((long)g.M * g.N * 4);
                for (int s0 = 0; s0 < g.S; ++s0) {        // slab order: the sum does not depend on who arrived last
                    f32x4 t[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) t[it] = buf_load4(srs, so[it] == OOB_OFF ? OOB_OFF : so[it] + (unsigned)s0 * sstep);
#pragma unroll
                    for (int it = 0; it < 8; ++it)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[it][q] += t[it][q];
                }
                finish_block(i, 0, v);
            }
            finish_stats();
        } else {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mw0 + i * 32 + mfma32_row(r, lane);
                        if (n < g.N && m < g.M) c_base[(long)m * g.N + n] = g.alpha * acc[i][j][r];
                    }
                }
        }
        return;
    }

    if constexpr (VEC && EPI == 2) {      // un-split product feeding a BatchNorm backward: mask + sums in the epilogue
        finish_setup();
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            finish_load(i, 0, std::integral_constant<int, 4>{});
            dump_half(i, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(lw + ((h * 4 + u) * 4 + rl) * 64 + c4);
                finish_block(i, h * 4, v);
                if (h == 0) finish_load(i, 4, std::integral_constant<int, 4>{});
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        finish_stats();
        return;
    }

    if constexpr (VEC && EPI == 0) {
        const bool use_c = g.beta != 0.f, use_m = g.rowmask != nullptr;
        const __amdgpu_buffer_rsrc_t cr = make_rsrc(C, use_c ? (unsigned)(((long)(g.M - 1) * g.ldc + g.N) * 4) : 0u);
        const __amdgpu_buffer_rsrc_t mr = make_rsrc(g.rowmask, use_m ? (unsigned)g.M * 4u : 0u);
        float bv[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (g.bias && n < g.N) bv[j] = g.bias[n];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            f32x4 cv[8];
            float mk[8];
            if (use_c) {
#pragma unroll
                for (int it = 0; it < 8; ++it) cv[it] = buf_load4(cr, row_off(mw0 + i * 32 + it * 4 + rl));
            }
            if (use_m) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mw0 + i * 32 + it * 4 + rl;
                    mk[it] = buf_load(mr, m < g.M ? (unsigned)m * 4u + opq : OOB_OFF);
                }
            }
            dump_half(i, bv[0], bv[1]);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + rl;
                f32x4 v = *reinterpret_cast<const f32x4*>(lw + row * 64 + c4);
                if (use_c) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaf(g.beta, cv[it][q], v[q]);
                }
                if (use_m && mk[it] == 0.f) v = f32x4{0.f, 0.f, 0.f, 0.f};
                buf_store4(ors, row_off(mw0 + i * 32 + row), v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }

    if constexpr (VEC && EPI == 1) {
        // statistics first, straight from the accumulators (lane = column, registers = rows): one partial row per 64
        // rows.  MI = 2: a wave owns 64 rows of its columns outright -> no exchange at all; MI = 1: the two row halves
        // (wm) of the 64-row tile meet in LDS.
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            const float sft = (g.stat_shift && n < g.N) ? g.stat_shift[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw0 + i * 32 + mfma32_row(r, lane);
                    const float d = (m < g.M) ? g.alpha * acc[i][j][r] - sft : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (MI == 2) {
                if (lane < 32 && n < g.N && mw0 < g.M) {
                    float* p = g.stat_partial + (long)n * g.ldp + (tm * 2 + wm);
                    p[0] = s1;
                    p[(long)g.N * g.ldp] = s2;
                }
            } else if (lane < 32) {
                colsum[(wm * 2 + 0) * TN + wn * 64 + j * 32 + l31] = s1;
                colsum[(wm * 2 + 1) * TN + wn * 64 + j * 32 + l31] = s2;
            }
        }
        if (MI == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < TN && n0 + tid < g.N) {
                float* p = g.stat_partial + (long)(n0 + tid) * g.ldp + tm;
                p[0] = colsum[0 * TN + tid] + colsum[2 * TN + tid];
                p[(long)g.N * g.ldp] = colsum[1 * TN + tid] + colsum[3 * TN + tid];
            }
        }
        if (W41) {     // waves (0,1) form the tile's first 64-row statistics block, waves (2,3) its second
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int blk = tid >> 6, col = tid & 63;
            if (tid < 128 && n0 + col < g.N && m0 + blk * 64 < g.M) {
                float* p = g.stat_partial + (long)(n0 + col) * g.ldp + (tm * 2 + blk);
                p[0] = colsum[((2 * blk) * 2 + 0) * TN + col] + colsum[((2 * blk + 1) * 2 + 0) * TN + col];
                p[(long)g.N * g.ldp] = colsum[((2 * blk) * 2 + 1) * TN + col] + colsum[((2 * blk + 1) * 2 + 1) * TN + col];
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            dump_half(i, 0.f, 0.f);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + rl;
                buf_store4(ors, row_off(mw0 + i * 32 + row), *reinterpret_cast<const f32x4*>(lw + row * 64 + c4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }

    // ---- scalar epilogue for outputs that cannot take 16-byte row stores (odd N / ldc): EPI 0 only ------------------
    if constexpr (!VEC && EPI == 0) {
        const bool use_c = g.beta != 0.f, use_m = g.rowmask != nullptr;
        const __amdgpu_buffer_rsrc_t cr = make_rsrc(C, use_c ? (unsigned)(((long)(g.M - 1) * g.ldc + g.N) * 4) : 0u);
        const __amdgpu_buffer_rsrc_t mr = make_rsrc(g.rowmask, use_m ? (unsigned)g.M * 4u : 0u);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                const bool nok = n < g.N;
                float cv[16], mk[16];
                if (use_c) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mw0 + i * 32 + mfma32_row(r, lane);
                        cv[r] = buf_load(cr, (nok && m < g.M) ? (unsigned)(((long)m * g.ldc + n) * 4) : OOB_OFF);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(cv[r]));
                }
                if (use_m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mw0 + i * 32 + mfma32_row(r, lane);
                        mk[r] = buf_load(mr, m < g.M ? (unsigned)m * 4u : OOB_OFF);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(mk[r]));
                }
                float bv = 0.f;
                if (g.bias && nok) bv = g.bias[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw0 + i * 32 + mfma32_row(r, lane);
                    float v = g.alpha * acc[i][j][r] + bv;
                    if (use_c) v += g.beta * cv[r];
                    if (use_m && mk[r] == 0.f) v = 0.f;
                    if (nok && m < g.M) C[(long)m * g.ldc + n] = v;
                }
            }
    }
}

// Sum split-K slabs in slab order and apply the plain epilogue, 4 columns per thread when the output allows it.
template <bool V4>
__global__ __launch_bounds__(256) void creduce_kernel(CArgs g) {
    const long mn = (long)g.M * g.N;
    const int zb = blockIdx.y;
    const float* W = g.ws + ((long)zb * g.S) * mn;
    if (V4) {
        const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
        if (i4 >= mn) return;
        const int m = (int)(i4 / g.N), n = (int)(i4 - (long)m * g.N);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < g.S; s0 += 8) {       // 8 loads in flight per round, fixed order
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(W + (long)min(s0 + u, g.S - 1) * mn + i4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < g.S) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += t[u][q];
                }
        }
        float* cp = g.C + (long)zb * g.sC + (long)m * g.ldc + n;
        if (g.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += g.bias[n + q];
        }
        if (g.beta != 0.f) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaf(g.beta, c[q], v[q]);
        }
        if (g.rowmask && g.rowmask[m] == 0.f) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(cp) = v;
    } else {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= mn) return;
        const int m = (int)(i / g.N), n = (int)(i - (long)m * g.N);
        float v = 0.f;
        for (int s0 = 0; s0 < g.S; s0 += 16) v += slab_sum(W + (long)s0 * mn, i, min(16, g.S - s0), mn);
        float* cp = g.C + (long)zb * g.sC + (long)m * g.ldc + n;
        if (g.bias) v += g.bias[n];
        if (g.beta != 0.f) v += g.beta * (*cp);
        if (g.rowmask && g.rowmask[m] == 0.f) v = 0.f;
        *cp = v;
    }
}

// Statistics passes over a product, also the split-K reducer for them.  A workgroup owns 64 rows x 64 columns; thread
// (rl, cl) walks rows rl, rl+16, rl+32, rl+48 of 4 columns with all its loads in flight; partial[64-row block][2][N].
//   MODE 1: y = sum of S slabs -> C ; sums of (y - s), (y - s)^2.
//   MODE 2: g = (sum of S slabs, or C itself when S == 0) * [fma((z-mean)*invstd, gamma, beta) > 0] -> C ; sums of g,
//           g*xhat -- the ReLU mask and the two reductions of a BatchNorm backward folded into the dgrad that feeds it.
template <int MODE>
__global__ __launch_bounds__(256) void cstats_kernel(CArgs g) {
    __shared__ float red[16][2][64 + 1];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int r0 = blockIdx.y * SROWS;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < g.N) {
        f32x4 sft = {0.f, 0.f, 0.f, 0.f}, mu = sft, is = sft, ga = sft, be = sft;
        if (MODE == 1 && g.stat_shift) sft = *reinterpret_cast<const f32x4*>(g.stat_shift + c);
        bool folded = false;
        if (MODE == 2) {
            mu = *reinterpret_cast<const f32x4*>(g.emean + c);
            is = *reinterpret_cast<const f32x4*>(g.einvstd + c);
            if (g.pro_ss) {   // the mask of the function the forward pass actually evaluated: relu(fma(z, scale, shift))
                folded = true;
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(g.pro_ss + 2 * c), t1 = *reinterpret_cast<const f32x4*>(g.pro_ss + 2 * c + 4);
                ga = f32x4{t0[0], t0[2], t1[0], t1[2]};
                be = f32x4{t0[1], t0[3], t1[1], t1[3]};
            } else {
                ga = *reinterpret_cast<const f32x4*>(g.egamma + c);
                be = *reinterpret_cast<const f32x4*>(g.ebeta + c);
            }
        }
        const long mn = (long)g.M * g.N;
        f32x4 v[4], zz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(r0 + rl + 16 * u, g.M - 1);
            v[u] = g.S > 0 ? *reinterpret_cast<const f32x4*>(g.ws + (long)r * g.N + c)
                           : *reinterpret_cast<const f32x4*>(g.C + (long)r * g.ldc + c);
            if (MODE == 2) zz[u] = *reinterpret_cast<const f32x4*>(g.ez + (long)r * g.ldz + c);
        }
        for (int s = 1; s < g.S; ++s) {
            f32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = min(r0 + rl + 16 * u, g.M - 1);
                w[u] = *reinterpret_cast<const f32x4*>(g.ws + (long)s * mn + (long)r * g.N + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] += w[u][k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + rl + 16 * u;
            if (r >= g.M) continue;
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (zz[u][k] - mu[k]) * is[k];
                    const bool on = folded ? (fmaf(zz[u][k], ga[k], be[k]) > 0.f) : (fmaf(xh, ga[k], be[k]) > 0.f);
                    if (!on) v[u][k] = 0.f;
                    s1[k] += v[u][k];
                    s2[k] = fmaf(v[u][k], xh, s2[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[u][k] - sft[k];
                    s1[k] += d;
                    s2[k] = fmaf(d, d, s2[k]);
                }
            }
            *reinterpret_cast<f32x4*>(g.C + (long)r * g.ldc + c) = v[u];
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
            float t = red[0][which][cc];
#pragma unroll
            for (int i = 1; i < 16; ++i) t += red[i][which][cc];
            g.stat_partial[((long)which * g.N + blockIdx.x * 64 + cc) * g.ldp + blockIdx.y] = t;
        }
    }
}

}  // namespace

constexpr int CG_MAX_SPLIT = 128;   // wgrad of the early layers: 4 output tiles, K = 32768 rows
int g_cgemm_target = 512;     // aim for this many workgroups (tiles x splits) when the tile grid alone is < 256
int g_cgemm_kmin = 128;       // at least this much K per split
int g_cgemm_mi = 0;           // 0: pick the row tile (64 or 128) per shape; 1 / 2: force it (tuning)
int g_cgemm_combine = 1;      // split-K epilogue inside the launch: 0 off (second launch), 1 write-through slabs, 2 plain slabs + release
int g_cgemm_combine_max = 8;  // deepest split the last arriver sums alone; deeper (S = 16 at layer3 weight gradients: 48 vs 51 us) the reduce launch,
                              // which spreads the sum over the chip, wins (tools/cgemm_bench.py comb)

namespace {
// Arrival counters of the in-launch combine: one array per (device, stream), zero at rest (the last arriver of a tile
// resets its counter), so launches of ONE stream -- which never overlap -- share it and concurrent streams never do.
constexpr int CNT_N = 1 << 16;
std::mutex g_cnt_mu;
std::map<std::pair<int, hipStream_t>, int*> g_cnt;
int* stream_counters(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_cnt_mu);
    auto it = g_cnt.find({dev, st});
    if (it != g_cnt.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;   // no allocation inside a capture
    int* p = nullptr;
    if (hipMalloc(&p, CNT_N * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, CNT_N * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_cnt[{dev, st}] = p;
    return p;
}
}  // namespace
int* split_counters(hipStream_t st) { return stream_counters(st); }
static_assert(CNT_N == SPLIT_COUNTERS, "counter array size");

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

namespace {

template <int MI, bool AMC, bool BMC, int PRO, bool G>
void launch_ev(hipStream_t st, dim3 grid, const CArgs& g, int kepi, bool vec) {
    dim3 block(256);
    if (kepi == 2) {       // mask epilogue: the d-input products only (host-checked)
        if constexpr (!AMC && BMC && PRO == 0 && !G) hipLaunchKernelGGL((cgemm_kernel<MI, false, true, 0, 2, false, true>), grid, block, 0, st, g);
    } else if (kepi == 1) hipLaunchKernelGGL((cgemm_kernel<MI, AMC, BMC, PRO, 1, G, true>), grid, block, 0, st, g);
    else if (vec)  hipLaunchKernelGGL((cgemm_kernel<MI, AMC, BMC, PRO, 0, G, true>), grid, block, 0, st, g);
    else           hipLaunchKernelGGL((cgemm_kernel<MI, AMC, BMC, PRO, 0, G, false>), grid, block, 0, st, g);
}

template <int MI>
int launch_conv3(hipStream_t st, dim3 grid, const CArgs& g, int c3, int kepi) {
    dim3 block(256);
    if (c3 == 1) {
        if (kepi == 1) hipLaunchKernelGGL((cgemm_kernel<MI, false, false, 0, 1, false, true, 1>), grid, block, 0, st, g);
        else           hipLaunchKernelGGL((cgemm_kernel<MI, false, false, 0, 0, false, true, 1>), grid, block, 0, st, g);
    } else if (c3 == 2) {
        if (kepi == 2) hipLaunchKernelGGL((cgemm_kernel<MI, false, true, 0, 2, false, true, 2>), grid, block, 0, st, g);
        else           hipLaunchKernelGGL((cgemm_kernel<MI, false, true, 0, 0, false, true, 2>), grid, block, 0, st, g);
    } else if (c3 == 4) {
        hipLaunchKernelGGL((cgemm_kernel<MI, false, true, 0, 0, false, true, 4>), grid, block, 0, st, g);
    } else {
        hipLaunchKernelGGL((cgemm_kernel<MI, true, true, 0, 0, false, true, 3>), grid, block, 0, st, g);
    }
    return 0;
}

template <int MI>
int launch_layout(hipStream_t st, dim3 grid, const CArgs& g, bool tA, bool tB, int pro, int kepi, bool gather, bool vec) {
    if (!tA && tB) {            // forward of a 1x1 convolution, nn.Linear: both k-contiguous
        if (gather) { if (pro == 1) launch_ev<MI, false, false, 1, true>(st, grid, g, kepi, vec); else launch_ev<MI, false, false, 0, true>(st, grid, g, kepi, vec); }
        else        { if (pro == 1) launch_ev<MI, false, false, 1, false>(st, grid, g, kepi, vec); else launch_ev<MI, false, false, 0, false>(st, grid, g, kepi, vec); }
    } else if (!tA && !tB) {    // dgrad: A k-contiguous, B [K][N]
        SCN_ARG(!gather && pro == 0, "cgemm: NN product takes no gather / prologue");
        launch_ev<MI, false, true, 0, false>(st, grid, g, kepi, vec);
    } else if (tA && !tB) {     // wgrad: A [K][M], B [K][N]
        SCN_ARG(kepi == 0, "cgemm: TN product takes no statistics epilogue");
        if (gather) { if (pro == 2) launch_ev<MI, true, true, 2, true>(st, grid, g, 0, vec); else launch_ev<MI, true, true, 0, true>(st, grid, g, 0, vec); }
        else        { if (pro == 2) launch_ev<MI, true, true, 2, false>(st, grid, g, 0, vec); else launch_ev<MI, true, true, 0, false>(st, grid, g, 0, vec); }
    } else {
        SCN_ARG(!gather && pro == 0 && kepi == 0, "cgemm: TT product is plain");
        launch_ev<MI, true, false, 0, false>(st, grid, g, 0, vec);
    }
    return 0;
}

}  // namespace

// pro: 0 none, 1 A prologue (needs !tA), 2 B prologue (needs tA && !tB);  epi: 0 plain, 1 stats, 2 mask + BN-backward sums.
int cgemm(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda, const float* B,
          long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask, int batch, long sA, long sB,
          long sC, float* ws, long ws_floats, const ConvExtra* ex) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    SCN_ARG(A && B && C, "cgemm: null operand");
    SCN_ARG(cgemm_supported(tA, tB, M, N, K, A, lda, B, ldb, sA, sB), "cgemm: operand alignment / size not supported");
    SCN_ARG(beta == 0.f || ((long)(M - 1) * ldc + N) * 4 < 0x7fffffffL, "cgemm: C too large for beta != 0");
    const int pro = ex ? ex->pro : 0, epi = ex ? ex->epi : 0;
    const int c3 = ex ? ex->c3 : 0;
    const bool gather = ex && ex->stride > 1 && c3 == 0;
    if (c3) {
        SCN_ARG(c3 >= 1 && c3 <= 4 && pro == 0 && batch == 1 && beta == 0.f && !bias && !rowmask, "cgemm: 3x3 mode takes a plain product");
        SCN_ARG(ex->Hi > 0 && ex->Wi > 0 && ex->Ho > 0 && ex->Wo > 0 && ex->stride >= 1 && ex->c3c > 0 && ex->c3_src_rows > 0, "cgemm: 3x3 geometry");
        SCN_ARG((c3 == 1 && !tA && tB && K == 9 * ex->c3c && ex->c3c % 16 == 0) ||
                (c3 == 2 && !tA && !tB && K == 9 * ex->c3c && ex->c3c % 16 == 0 && ex->stride == 1) ||
                (c3 == 4 && !tA && !tB && K == 9 * ex->c3c && ex->c3c % 16 == 0 && ex->stride == 2 && epi == 0 &&
                 ex->Hi == 2 * ex->Ho && ex->Wi == 2 * ex->Wo && (long)4 * M * ldc * 4 < 0x7fffffffL) ||
                (c3 == 3 && tA && !tB && N == 9 * ex->c3c && ex->c3c % 128 == 0), "cgemm: 3x3 mode / layout / channel multiple");
        SCN_ARG(ex->c3_src_rows * (c3 == 3 ? ldb : lda) * 4 < 0x7fffffffL, "cgemm: 3x3 source map exceeds the descriptor range");
    }
    SCN_ARG(pro == 0 || (pro == 1 && !tA) || (pro == 2 && !tB && tA), "cgemm: prologue / layout mismatch");
    SCN_ARG(epi == 0 || (batch == 1 && beta == 0.f && !bias && !rowmask && ex->stat_partial), "cgemm: statistics epilogue needs a plain product");
    SCN_ARG(epi != 2 || (ex->ez && ex->emean && ex->einvstd && (ex->pro_ss || (ex->egamma && ex->ebeta)) && ex->ldz % 4 == 0),
            "cgemm: mask epilogue arguments");
    SCN_ARG(pro == 0 || ex->pro_ss, "cgemm: prologue table");
    SCN_ARG(!gather || (ex->Hi > 0 && ex->Wi > 0 && ex->Ho > 0 && ex->Wo > 0 && batch == 1), "cgemm: gather geometry");
    const bool vec = N % 4 == 0 && ldc % 4 == 0 && sC % 4 == 0 && aligned16(C) &&
                     ((long)(M - 1) * ldc + N) * 4 < 0x7fffffffL && (long)M * N * 4 < 0x7fffffffL;
    SCN_ARG(epi == 0 || vec, "cgemm: the statistics epilogues need N % 4 == 0, ldc % 4 == 0 and a 16-byte aligned C");
    // row tile: 64 rows when the 128-row grid alone cannot give every CU a workgroup but the 64-row grid can come closer
    int mi = 2;
    if ((long)cdiv(M, 128) * cdiv(N, TN) * batch < 256 && M > 64 && c3 != 1 && c3 != 2 && c3 != 4) mi = 1;
    // stride-2 d input: the four classes carry 1 / 2 / 2 / 4 taps, so the grid is uneven by construction; 64-row tiles
    // (twice the workgroups, all resident at once) let the dispatcher even it out when the 128-row grid is small
    if (c3 == 4 && (long)cdiv(M, 128) * cdiv(N, TN) * 4 < 768) mi = 1;
    if (N <= 64 && M >= 128 && c3 != 3) mi = 4;     // 128 x 64 tiles, waves 4 x 1 (layer1's 64-channel maps)
    if (g_cgemm_mi == 1 || g_cgemm_mi == 2) mi = g_cgemm_mi;
    if (ex && ex->force_mi > 0) mi = ex->force_mi;
    SCN_ARG(mi == 1 || mi == 2 || (mi == 4 && c3 != 3), "cgemm: bad tile selector");
    const int nt = cdiv(N, mi == 4 ? 64 : TN);
    const int tmrows = mi == 4 ? 128 : 64 * mi, mt = cdiv(M, tmrows);
    const long tiles = (long)mt * nt * batch;
    int S = 1;
    if (ws && tiles < 224 && K >= 2 * g_cgemm_kmin) {
        // 3x3 weight gradient: its gathered operand makes the k-loop latency-bound, a third resident workgroup per CU
        // pays (measured 115-120 us at ~768 workgroups against 127-142 at ~512; the 1x1 weight gradients are best at 512).
        // Very deep non-transposed products (the decoder's d hidden = d preds . fc.weight, K = vocabulary): ~1100
        // workgroups (1632 x 512 x 10000: 208 -> 158 us, tools/cgemm_bench.py gemmsweep).
        long target = g_cgemm_target;
        if (g_cgemm_target == 512) {
            if (c3 == 3) target = 768;
            else if (!tA && K >= 4096 && !c3) target = 1100;
        }
        S = (int)((target + tiles - 1) / tiles);
        const int smax = K / g_cgemm_kmin;
        if (S > smax) S = smax;
        if (S > CG_MAX_SPLIT) S = CG_MAX_SPLIT;
        while (S > 1 && (long)S * batch * M * N > ws_floats) --S;
        if (S < 1) S = 1;
    }
    // A grid of one to three residency rounds (224 .. 767 tiles of a deep product) ends in a long tail: the last round
    // runs with most CUs idle.  Splitting K to ~1200 workgroups evens it out (10000 x 512 x 1632: 208 -> 154 us,
    // 6272 x 512 x 2048: 143 -> 128 us).  K >= 1536 keeps the trunk's forward / d-input convolutions on their measured
    // setting (layer3's 8192 x 256 x 1024 loses 3-5 us to such a split: its statistics epilogue moves into the reduce pass).
    if (ws && !c3 && S == 1 && tiles >= 224 && tiles < 768 && K >= 1536 && g_cgemm_target == 512) {
        S = (int)((1200 + tiles / 2) / tiles);
        const int smax = K / (2 * g_cgemm_kmin);
        if (S > smax) S = smax;
        while (S > 1 && (long)S * batch * M * N > ws_floats) --S;
        if (S < 1) S = 1;
    }
    if (c3 == 4) S = 1;       // the four parity classes ride on grid.y; their K (1, 2, 2, 4 taps) is never split
    if ((c3 == 1 || c3 == 2) && ws) {   // deep K (9 taps): 128-row tiles, up to 4 K-slices to reach ~512 workgroups (measured)
        S = (int)(512 / (tiles > 0 ? tiles : 1));
        if (S > 4) S = 4;
        if (S < 1) S = 1;
        while (S > 1 && (long)S * M * N > ws_floats) --S;
    }
    if (ex && ex->force_split > 0 && c3 != 4) {
        S = ex->force_split;
        SCN_ARG(S == 1 || (ws && (long)S * batch * M * N <= ws_floats && S <= CG_MAX_SPLIT), "cgemm: forced split does not fit");
    }
    int kper = cdiv(K, S);
    kper = (kper + TK - 1) / TK * TK;
    S = cdiv(K, kper);
    CArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.rowmask = rowmask;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta; g.S = S; g.kper = kper; g.ws = ws; g.mt = mt; g.nt = nt;
    if (ex) {
        g.gHi = ex->Hi; g.gWi = ex->Wi; g.gHo = ex->Ho; g.gWo = ex->Wo; g.gs = (gather || c3) ? ex->stride : 0;
        g.c3c = ex->c3c; g.src_rows = ex->c3_src_rows;
        if (c3 == 4) { g.gHi = ex->Ho; g.gWi = ex->Wo; g.dHi = ex->Hi; g.dWi = ex->Wi; }   // gathered map = dY (Ho x Wo)
        g.pro_ss = ex->pro_ss;
        g.stat_partial = ex->stat_partial; g.stat_shift = ex->stat_shift; g.ldp = cgemm_stat_ld(M);
        g.ez = ex->ez; g.emean = ex->emean; g.einvstd = ex->einvstd; g.egamma = ex->egamma; g.ebeta = ex->ebeta; g.ldz = ex->ldz;
    }
    dim3 grid(mt * nt, c3 == 4 ? 4 : batch * S), block(256);
    // The epilogue runs inside the launch: from the accumulators of an un-split product, or -- split product -- by the
    // workgroup that arrives last at its tile (arrival counters of this stream).  Second launch (creduce / cstats) only
    // for what the in-launch forms do not cover: scalar stores, very deep splits, a mask epilogue on another layout.
    const bool epi_in = epi != 2 || (!tA && !tB && pro == 0 && !gather && c3 == 0) || c3 == 2;
    int* cnt = nullptr;
    if (S > 1 && g_cgemm_combine && S <= g_cgemm_combine_max && vec && epi_in && (epi == 0 || !tA) && tiles <= CNT_N &&
        (long)S * M * N * 4 < 0x7fffffffL)
        cnt = stream_counters(st);
    g.cnt = cnt; g.comb = g_cgemm_combine;
    const int kepi = (S > 1 && !cnt) ? 0 : (epi_in ? epi : 0);
    if (c3) {
        SCN_ARG(vec, "cgemm: 3x3 mode needs 16-byte row stores");
        if (mi == 4) SCN_TRY(launch_conv3<4>(st, grid, g, c3, kepi));
        else if (mi == 2) SCN_TRY(launch_conv3<2>(st, grid, g, c3, kepi));
        else SCN_TRY(launch_conv3<1>(st, grid, g, c3, kepi));
    } else if (mi == 4) SCN_TRY(launch_layout<4>(st, grid, g, tA, tB, pro, kepi, gather, vec));
    else if (mi == 2) SCN_TRY(launch_layout<2>(st, grid, g, tA, tB, pro, kepi, gather, vec));
    else SCN_TRY(launch_layout<1>(st, grid, g, tA, tB, pro, kepi, gather, vec));
    SCN_LAUNCH_CHECK();
    if ((S > 1 && !cnt) || (epi == 2 && !epi_in)) {
        if (epi == 0) {
            if (vec) hipLaunchKernelGGL(creduce_kernel<true>, dim3(cdiv((long)M * N / 4, 256), batch), block, 0, st, g);
            else     hipLaunchKernelGGL(creduce_kernel<false>, dim3(cdiv((long)M * N, 256), batch), block, 0, st, g);
        } else {
            if (S == 1) g.S = 0;      // un-split mask pass: read the product back from C
            dim3 sgrid(cdiv(N, 64), cdiv(M, SROWS));
            if (epi == 1) hipLaunchKernelGGL(cstats_kernel<1>, sgrid, block, 0, st, g);
            else          hipLaunchKernelGGL(cstats_kernel<2>, sgrid, block, 0, st, g);
        }
        SCN_LAUNCH_CHECK();
    }
    return 0;
}

int cgemm_row_tiles(int M) { return cdiv(M, SROWS); }
// Statistics partials are CHANNEL-MAJOR, [2][N][cgemm_stat_ld(M)]: entry (which, channel, 64-row block).  A consumer that
// needs the sums of a few channels (BatchNorm apply / backward kernels that finalize on load, csrc/batchnorm.hip) reads
// contiguous rows instead of a strided column.
int cgemm_stat_ld(int M) { return (cdiv(M, SROWS) + 3) & ~3; }

// C[M][N] (leading dimension ldc) = sum of S slabs [S][M][N] in slab order: the split-K reducer on its own, for kernels
// outside this file that write the same slab layout (csrc/conv3.hip).
int cgemm_reduce(hipStream_t st, const float* ws, int S, int M, int N, float* C, long ldc) {
    SCN_ARG(ws && C && S >= 1 && M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && aligned16(ws) && aligned16(C),
            "cgemm_reduce: arguments");
    CArgs g{};
    g.ws = const_cast<float*>(ws); g.C = C; g.M = M; g.N = N; g.S = S; g.ldc = ldc;
    hipLaunchKernelGGL(creduce_kernel<true>, dim3(cdiv((long)M * N / 4, 256), 1), dim3(256), 0, st, g);
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
