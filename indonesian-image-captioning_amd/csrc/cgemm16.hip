// bf16 GEMM core of the ResNet-152 convolution path (BASELINE configs[4]: mixed precision), gfx950 only:
//     C[M][N] = A[M][K] . B[N][K]^T (+ beta * C),  A / B bf16 in HBM, fp32 accumulation (v_mfma_f32_32x32x16_bf16),
//     C bf16 or fp32, statistics of C for the next BatchNorm taken from the fp32 accumulators.
//
// What it replaces: the same `nn.Conv2d` calls of torchvision's Bottleneck behind models/encoders/caption.py:17-22 that
// csrc/cgemm.hip serves in fp32 -- forward AND d input of every 1x1 and 3x3 convolution.  Both operands are always
// k-contiguous here:
//   forward   A = activations [R][Cin] (3x3: rows gathered tap by tap, K = 9 Cin),  B = weight [Cout][(9) Cin];
//   d input   A = dY [R][Cout],  B = the TRANSPOSED weight copy [Cin][(9) Cout] that the per-step fp32 -> bf16 conversion
//             writes beside the plain one (scnattn_bf16_weights): a stride-1 3x3 d input is then literally a forward
//             convolution of dY with flipped taps (`flip`), the stride-2 one walks the taps of its parity class (mode 4 as
//             in csrc/cgemm.hip);
// so the m/n-contiguous LDS image and its transposed reads are needed only by the weight gradients (csrc/wgrad16.hip).
//
// Structure = csrc/cgemm.hip's: 128 x 128 (or 64 x 128) block tile, 4 waves as 2 x 2, LDS-DMA 3-stage ring with ONE raw
// s_barrier per k-step and a counted vmcnt, XCD-aware tile order, [row][64-byte] LDS image with the 16-byte granules
// XOR-swizzled on the SOURCE address.  A k-step is 32 bf16 = the same 64 bytes per row, so every address computation is
// the fp32 kernel's in bytes; the fragment of v_mfma_f32_32x32x16_bf16 (lane (r, h): k = 8h .. 8h+7) is ONE 16-byte granule,
// one conflict-free ds_read_b128 per 32-row block and 16 k.  The matrix work per k-step is 8x shorter than in fp32 (8
// instructions of 32 cycles against 32 of 64), so this kernel is bound by its fill / epilogue / launch, not by the matrix
// pipes: what it buys is half the bytes of every map.
#include "common.h"
#include "kernels.h"

namespace scn {

int cgemm_stat_ld(int M);

namespace {

constexpr int TN = 128, TKE = 32, NSTAGE = 3;    // k ELEMENTS per stage
constexpr int TILE_F = 128 * 16;                 // floats reserved per operand tile: 128 rows x 64 bytes
constexpr int STAGE_F = 2 * TILE_F;              // 16 KiB per stage -> 48 KiB per workgroup
constexpr int SROWS = 64;

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct HArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    long lda, ldb, ldc;
    int M, N, K;
    float beta;
    int S, kper;
    float* ws;
    int mt, nt;
    int gHi, gWi, gHo, gWo, gs;   // row gather (strided 1x1) / 3x3 source and destination maps
    int c3c;                      // 3x3 modes: channels per tap of the gathered operand
    long src_rows;
    int dHi, dWi;                 // mode 4: extent of the d-input map the rows are scattered into
    int flip;                     // mode 1: weight tap = 8 - t (a stride-1 d input as a forward convolution of dY)
    float* stat_partial; const float* stat_shift; int ldp;
    const bf16_t* ez; long ldz; const float* emean; const float* einvstd; const float* egamma; const float* ebeta;   // EPI 2
};

__device__ __forceinline__ long gather_row(const HArgs& g, int r) {
    if (g.gs == 0) return r;
    const int hw = g.gHo * g.gWo;
    const int n = r / hw, rem = r - n * hw;
    const int ho = rem / g.gWo, wo = rem - ho * g.gWo;
    return (long)n * g.gHi * g.gWi + (long)(ho * g.gs) * g.gWi + wo * g.gs;
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {       // round to nearest even, NaN stays NaN (v_cvt_pk_bf16_f32)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

// MI: 32-row MFMA tiles per wave along m (block tile 64*MI x 128, waves 2 x 2).  EPI 1: + column statistics.
// EPI 2 (bf16 output, un-split d-input products): g = dx * [relu mask of the consumer BatchNorm recomputed from its bf16
// pre-activation z with the forward pass's own expression] stored, + column sums of g and g*xhat per 64-row block -- the
// reduce pass of that BatchNorm's backward inside the product that feeds it (as EPI 2 of csrc/cgemm.hip).
// GATHER: rows of A are gathered (strided 1x1 convolution).  OBF: C is bf16.  C3: 0 plain, 1 3x3 taps over K (forward, or a
// stride-1 d input with `flip`), 4 stride-2 d input, one parity class per blockIdx.y.
template <int MI, int EPI, bool GATHER, bool OBF, int C3>
__global__ __launch_bounds__(256, 3) void cgemm16_kernel(HArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE_F];
    constexpr int RB = MI, TM = 64 * MI, ACH = TM / 64, LPT = ACH + 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int hh = lane >> 5, l31 = lane & 31;

    const int ntiles = g.mt * g.nt;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / g.nt, tn = bid - tm * g.nt;
    const int m0 = tm * TM, n0 = tn * TN;
    const int cls = C3 == 4 ? 3 - (int)blockIdx.y : 0, ph = cls >> 1, pw = cls & 1, ntw = 1 + pw;
    const int sp = C3 == 4 ? 0 : blockIdx.y;
    const int kbeg = sp * g.kper, Kend = C3 == 4 ? (1 + ph) * ntw * g.c3c : min(g.K, kbeg + g.kper);
    const int nk = (Kend - kbeg + TKE - 1) / TKE;

    const long a_elems = (C3 == 1 || C3 == 4) ? g.src_rows * g.lda
                         : (GATHER ? (gather_row(g, g.M - 1) * g.lda + g.K) : ((long)(g.M - 1) * g.lda + g.K));
    const long b_elems = (long)(g.N - 1) * g.ldb + (C3 ? 9L * g.c3c : g.K);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(g.A, (unsigned)(a_elems * 2));
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(g.B, (unsigned)(b_elems * 2));

    // chunk = 16 rows x 64 B; lane -> row chunk*16 + lane/4, LDS granule lane&3 <- source granule (lane&3)^((row>>2)&3)
    unsigned a_off[ACH], b_off[2];
    bool a_ok[ACH], b_ok[2];
    int a_nb[ACH], a_h0[ACH], a_w0[ACH], a_g[ACH], b_g[2];
#pragma unroll
    for (int c = 0; c < ACH; ++c) {
        const int chunk = wave * ACH + c;
        const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
        const int grow = m0 + row;
        a_ok[c] = grow < g.M;
        a_g[c] = gsrc;
        if (C3 == 1 || C3 == 4) {
            const int r = a_ok[c] ? grow : 0, hw = g.gHo * g.gWo;
            const int n = r / hw, rem = r - n * hw, hd = rem / g.gWo, wd = rem - hd * g.gWo;
            a_nb[c] = n * g.gHi * g.gWi;
            a_h0[c] = C3 == 4 ? hd : hd * g.gs - 1;
            a_w0[c] = C3 == 4 ? wd : wd * g.gs - 1;
            a_off[c] = (unsigned)(gsrc * 16);
        } else {
            const long src = (GATHER ? gather_row(g, a_ok[c] ? grow : 0) : (long)grow) * g.lda * 2 + gsrc * 16;
            a_off[c] = (unsigned)src;
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int chunk = wave * 2 + c;
        const int row = chunk * 16 + (lane >> 2), gsrc = (lane & 3) ^ ((row >> 2) & 3);
        const int grow = n0 + row;
        b_ok[c] = grow < g.N;
        b_g[c] = gsrc;
        b_off[c] = (unsigned)((long)grow * g.ldb * 2 + gsrc * 16);
    }

    auto issue = [&](int kt, int stage) {
        float* sa = lds + stage * STAGE_F;
        float* sb = sa + TILE_F;
        const int k0 = kbeg + kt * TKE;
        // 3x3 modes: a k-step is 32 channels of ONE tap (c3c % 32 == 0)
        int t = 0, c0 = 0, wt = 0, oh = 0, ow = 0;
        if (C3 == 1) {
            t = k0 / g.c3c; c0 = k0 - t * g.c3c;
            oh = t / 3; ow = t - 3 * oh;                   // source pixel offset (oh - 1, ow - 1) from a_h0 / a_w0
            wt = g.flip ? 8 - t : t;
        } else if (C3 == 4) {
            t = k0 / g.c3c; c0 = k0 - t * g.c3c;
            const int th = t / ntw, tw = t - th * ntw;
            oh = (ph && th == 0) ? 1 : 0; ow = (pw && tw == 0) ? 1 : 0;
            wt = (ph ? 2 * th : 1) * 3 + (pw ? 2 * tw : 1);
        }
#pragma unroll
        for (int c = 0; c < ACH; ++c) {
            const int chunk = wave * ACH + c;
            unsigned va;
            if (C3 == 1 || C3 == 4) {
                const int hi = a_h0[c] + oh, wi = a_w0[c] + ow;
                const bool ok = a_ok[c] && k0 < Kend && (unsigned)hi < (unsigned)g.gHi && (unsigned)wi < (unsigned)g.gWi;
                va = ok ? (unsigned)(((long)(a_nb[c] + hi * g.gWi + wi) * g.lda + c0) * 2) + a_off[c] : OOB_OFF;
            } else {
                va = (a_ok[c] && k0 + 8 * a_g[c] < Kend) ? a_off[c] + (unsigned)k0 * 2u : OOB_OFF;
            }
            dma16(ars, sa + chunk * 256, va);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int chunk = wave * 2 + c;
            unsigned vb;
            if (C3) vb = (b_ok[c] && k0 < Kend) ? b_off[c] + (unsigned)((wt * g.c3c + c0) * 2) : OOB_OFF;
            else    vb = (b_ok[c] && k0 + 8 * b_g[c] < Kend) ? b_off[c] + (unsigned)k0 * 2u : OOB_OFF;
            dma16(brs, sb + chunk * 256, vb);
        }
    };

    f32x16 acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
        f32x4 a[RB][2], b[2][2];        // [block][16-k half]: one 16-byte granule = the 8 bf16 of a lane's fragment
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = wm * 32 * RB + i * 32 + l31, sw = (row >> 2) & 3;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) a[i][kb] = *reinterpret_cast<const f32x4*>(sa + row * 16 + (((2 * kb + hh) ^ sw) << 2));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + l31, sw = (row >> 2) & 3;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) b[j][kb] = *reinterpret_cast<const f32x4*>(sb + row * 16 + (((2 * kb + hh) ^ sw) << 2));
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][kb]),
                                                                        __builtin_bit_cast(bf16x8, b[j][kb]), acc[i][j], 0, 0, 0);
        if (kt + 2 < nk) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (++stage == NSTAGE) stage = 0;
    }

    // ================================== epilogues =====================================================
    unsigned opq = 0;
    asm volatile("" : "+v"(opq));
    float* const lw = lds + wave * 2048;           // [32 rows][64 cols] fp32 transpose region of this wave
    float* const colsum = lds + 4 * 2048;          // [wm][2][TN] (MI = 1 only)
    const long c_ld = g.S > 1 ? (long)g.N : g.ldc;
    const long out_rows = C3 == 4 ? 4L * g.M : g.M;
    auto row_of = [&](int m) -> long {
        if (C3 != 4) return m;
        const int hw = g.gHo * g.gWo, n = m / hw, rem = m - n * hw, hd = rem / g.gWo, wd = rem - hd * g.gWo;
        return ((long)n * g.dHi + 2 * hd + ph) * g.dWi + 2 * wd + pw;
    };
    auto dump_half = [&](int i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) lw[mfma32_row(r, lane) * 64 + j * 32 + l31] = acc[i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const int mw0 = m0 + wm * 32 * RB;

    if (g.S > 1) {   // split-K: raw fp32 partial tile into slab blockIdx.y; the reduce pass applies the epilogue
        float* const c_base = g.ws + ((long)blockIdx.y * g.M) * g.N;
        const __amdgpu_buffer_rsrc_t ors = make_rsrc(c_base, (unsigned)(((long)(g.M - 1) * g.N + g.N) * 4));
        const int rl = lane >> 4, c4 = (lane & 15) * 4, ncol = n0 + wn * 64 + c4;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            dump_half(i);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + rl, m = mw0 + i * 32 + row;
                buf_store4(ors, (m < g.M && ncol < g.N) ? (unsigned)(((long)m * g.N + ncol) * 4) + opq : OOB_OFF,
                           *reinterpret_cast<const f32x4*>(lw + row * 64 + c4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }

    if constexpr (EPI == 1) {       // statistics from the fp32 accumulators, before anything is rounded
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
                    const float d = (m < g.M) ? acc[i][j][r] - sft : 0.f;
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
    }

    if constexpr (OBF) {            // bf16 rows: 8 columns (16 bytes) per lane, 8 rows per pass
        bf16_t* const Cb = reinterpret_cast<bf16_t*>(g.C);
        const __amdgpu_buffer_rsrc_t ors = make_rsrc(Cb, (unsigned)(((out_rows - 1) * g.ldc + g.N) * 2));
        const int rl = lane >> 3, c8 = (lane & 7) * 8, ncol = n0 + wn * 64 + c8;
        const bool use_c = g.beta != 0.f;
        [[maybe_unused]] float mu[8], is[8], ga[8], be[8], s1[8], s2[8];
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t zrs = make_rsrc(g.ez, EPI == 2 ? (unsigned)(((long)(g.M - 1) * g.ldz + g.N) * 2) : 0u);
        if constexpr (EPI == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool ok = ncol + q < g.N;
                mu[q] = ok ? g.emean[ncol + q] : 0.f; is[q] = ok ? g.einvstd[ncol + q] : 0.f;
                ga[q] = ok ? g.egamma[ncol + q] : 0.f; be[q] = ok ? g.ebeta[ncol + q] : 0.f;
                s1[q] = 0.f; s2[q] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            u32x4 cv[4];
            [[maybe_unused]] u32x4 zv[4];
            if (use_c) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = mw0 + i * 32 + it * 8 + rl;
                    cv[it] = __builtin_amdgcn_raw_buffer_load_b128(ors, (m < g.M && ncol < g.N) ? (unsigned)((row_of(m) * g.ldc + ncol) * 2) + opq : OOB_OFF, 0, 0);
                }
            }
            if constexpr (EPI == 2) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = mw0 + i * 32 + it * 8 + rl;
                    zv[it] = __builtin_amdgcn_raw_buffer_load_b128(zrs, (m < g.M && ncol < g.N) ? (unsigned)(((long)m * g.ldz + ncol) * 2) + opq : OOB_OFF, 0, 0);
                }
            }
            dump_half(i);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + rl, m = mw0 + i * 32 + row;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(lw + row * 64 + c8), v1 = *reinterpret_cast<const f32x4*>(lw + row * 64 + c8 + 4);
                if (use_c) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        v0[2 * q] = fmaf(g.beta, bf16_to_f32(cv[it][q] & 0xffffu), v0[2 * q]);
                        v0[2 * q + 1] = fmaf(g.beta, bf16_to_f32(cv[it][q] >> 16), v0[2 * q + 1]);
                        v1[2 * q] = fmaf(g.beta, bf16_to_f32(cv[it][2 + q] & 0xffffu), v1[2 * q]);
                        v1[2 * q + 1] = fmaf(g.beta, bf16_to_f32(cv[it][2 + q] >> 16), v1[2 * q + 1]);
                    }
                }
                if constexpr (EPI == 2) {
                    float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned w = zv[it][q >> 1];
                        const float z = bf16_to_f32((q & 1) ? (w >> 16) : (w & 0xffffu));
                        const float xh = (z - mu[q]) * is[q];
                        const bool on = m < g.M && fmaf(xh, ga[q], be[q]) > 0.f;     // the forward pass's expression (csrc/batchnorm.hip)
                        // the sums are taken of the bf16 rounding of g: what the consumers will read
                        vv[q] = on ? bf16_to_f32(pack2(vv[q], 0.f) & 0xffffu) : 0.f;
                        s1[q] += vv[q];
                        s2[q] = fmaf(vv[q], xh, s2[q]);
                    }
                    v0 = f32x4{vv[0], vv[1], vv[2], vv[3]};
                    v1 = f32x4{vv[4], vv[5], vv[6], vv[7]};
                }
                const u32x4 o = {pack2(v0[0], v0[1]), pack2(v0[2], v0[3]), pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, ors, (m < g.M && ncol < g.N) ? (unsigned)((row_of(m) * g.ldc + ncol) * 2) + opq : OOB_OFF, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (EPI == 2) {        // column sums of this wave's RB*32 rows -> one partial per 64-row block
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                s1[q] += __shfl_xor(s1[q], 8, 64); s1[q] += __shfl_xor(s1[q], 16, 64); s1[q] += __shfl_xor(s1[q], 32, 64);
                s2[q] += __shfl_xor(s2[q], 8, 64); s2[q] += __shfl_xor(s2[q], 16, 64); s2[q] += __shfl_xor(s2[q], 32, 64);
            }
            if (MI == 2) {
                if (lane < 8 && mw0 < g.M) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (ncol + q < g.N) {
                            float* p = g.stat_partial + (long)(ncol + q) * g.ldp + (tm * 2 + wm);
                            p[0] = s1[q];
                            p[(long)g.N * g.ldp] = s2[q];
                        }
                }
            } else {
                if (lane < 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        colsum[(wm * 2 + 0) * TN + wn * 64 + c8 + q] = s1[q];
                        colsum[(wm * 2 + 1) * TN + wn * 64 + c8 + q] = s2[q];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (tid < TN && n0 + tid < g.N) {
                    float* p = g.stat_partial + (long)(n0 + tid) * g.ldp + tm;
                    p[0] = colsum[0 * TN + tid] + colsum[2 * TN + tid];
                    p[(long)g.N * g.ldp] = colsum[1 * TN + tid] + colsum[3 * TN + tid];
                }
            }
        }
    } else {                        // fp32 rows
        float* const Cf = reinterpret_cast<float*>(g.C);
        const __amdgpu_buffer_rsrc_t ors = make_rsrc(Cf, (unsigned)(((out_rows - 1) * g.ldc + g.N) * 4));
        const int rl = lane >> 4, c4 = (lane & 15) * 4, ncol = n0 + wn * 64 + c4;
        const bool use_c = g.beta != 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            f32x4 cv[8];
            if (use_c) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mw0 + i * 32 + it * 4 + rl;
                    cv[it] = buf_load4(ors, (m < g.M && ncol < g.N) ? (unsigned)((row_of(m) * g.ldc + ncol) * 4) + opq : OOB_OFF);
                }
            }
            dump_half(i);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + rl, m = mw0 + i * 32 + row;
                f32x4 v = *reinterpret_cast<const f32x4*>(lw + row * 64 + c4);
                if (use_c) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaf(g.beta, cv[it][q], v[q]);
                }
                buf_store4(ors, (m < g.M && ncol < g.N) ? (unsigned)((row_of(m) * g.ldc + ncol) * 4) + opq : OOB_OFF, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// Sum split-K slabs [S][M][N] (fp32) in slab order -> C (bf16 or fp32, + beta * C); 4 columns per thread.  STATS: also the
// column sums of (y - s), (y - s)^2 per 64-row block (the statistics epilogue of a split product), channel-major.
template <bool OBF, bool STATS>
__global__ __launch_bounds__(256) void creduce16_kernel(HArgs g) {
    __shared__ float red[16][2][64 + 1];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int r0 = blockIdx.y * SROWS;
    const long mn = (long)g.M * g.N;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < g.N) {
        f32x4 sft = {0.f, 0.f, 0.f, 0.f};
        if (STATS && g.stat_shift) sft = *reinterpret_cast<const f32x4*>(g.stat_shift + c);
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(r0 + rl + 16 * u, g.M - 1);
            v[u] = *reinterpret_cast<const f32x4*>(g.ws + (long)r * g.N + c);
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
            if (STATS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[u][k] - sft[k];
                    s1[k] += d;
                    s2[k] = fmaf(d, d, s2[k]);
                }
            }
            if (OBF) {
                bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long)r * g.ldc + c;
                if (g.beta != 0.f) {
                    const u32x2 o = *reinterpret_cast<const u32x2*>(cp);
                    v[u][0] = fmaf(g.beta, bf16_to_f32(o[0] & 0xffffu), v[u][0]); v[u][1] = fmaf(g.beta, bf16_to_f32(o[0] >> 16), v[u][1]);
                    v[u][2] = fmaf(g.beta, bf16_to_f32(o[1] & 0xffffu), v[u][2]); v[u][3] = fmaf(g.beta, bf16_to_f32(o[1] >> 16), v[u][3]);
                }
                *reinterpret_cast<u32x2*>(cp) = u32x2{pack2(v[u][0], v[u][1]), pack2(v[u][2], v[u][3])};
            } else {
                float* cp = reinterpret_cast<float*>(g.C) + (long)r * g.ldc + c;
                if (g.beta != 0.f) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = fmaf(g.beta, o[k], v[u][k]);
                }
                *reinterpret_cast<f32x4*>(cp) = v[u];
            }
        }
    }
    if (!STATS) return;
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

template <int MI, int C3>
void launch16(hipStream_t st, dim3 grid, const HArgs& g, int kepi, bool gather, bool obf) {
    dim3 block(256);
#define SCN_L16(EPI_, G_, O_) hipLaunchKernelGGL((cgemm16_kernel<MI, EPI_, G_, O_, C3>), grid, block, 0, st, g)
    if constexpr (C3 == 0) {
        if (gather) { if (kepi) { if (obf) SCN_L16(1, true, true); else SCN_L16(1, true, false); } else { if (obf) SCN_L16(0, true, true); else SCN_L16(0, true, false); } }
        else if (kepi == 2) SCN_L16(2, false, true);
        else        { if (kepi) { if (obf) SCN_L16(1, false, true); else SCN_L16(1, false, false); } else { if (obf) SCN_L16(0, false, true); else SCN_L16(0, false, false); } }
    } else {        // 3x3 modes always write bf16 maps
        if constexpr (C3 == 1) { if (kepi == 2) { SCN_L16(2, false, true); return; } }
        if (kepi) SCN_L16(1, false, true); else SCN_L16(0, false, true);
    }
#undef SCN_L16
}

}  // namespace

// C[M][N] = A . B^T (+ beta*C).  A bf16 [M][K] (lda), B bf16 [N][K] (ldb); C bf16 (out_bf16) or fp32, leading dimension ldc
// in ELEMENTS.  ex: epi 0 / 1 (statistics, channel-major partials), stride > 1 (row gather), c3 = 1 (3x3 taps over K;
// ex->force_mi == -1 ... no: `flip` is passed separately) or 4 (stride-2 d input).  ws: fp32 split-K slabs.
int cgemm16(hipStream_t st, int M, int N, int K, const void* A, long lda, const void* B, long ldb, float beta, void* C, long ldc,
            int out_bf16, float* ws, long ws_floats, const ConvExtra* ex, int flip) {
    if (M <= 0 || N <= 0) return 0;
    SCN_ARG(A && B && C && K >= 8, "cgemm16: null operand / K");
    SCN_ARG(aligned16(A) && aligned16(B) && aligned16(C) && lda % 8 == 0 && ldb % 8 == 0 && K % 8 == 0, "cgemm16: operands need 16-byte rows (multiples of 8 elements)");
    SCN_ARG(N % 8 == 0 && ldc % (out_bf16 ? 8 : 4) == 0, "cgemm16: N / ldc granularity");
    const int epi = ex ? ex->epi : 0, c3 = ex ? ex->c3 : 0;
    const bool gather = ex && ex->stride > 1 && c3 == 0;
    SCN_ARG(epi == 0 || ((epi == 1 || epi == 2) && ex->stat_partial && beta == 0.f), "cgemm16: statistics / mask epilogue needs a plain product");
    SCN_ARG(epi != 2 || (out_bf16 && !gather && c3 != 4 && ex->ez && ex->emean && ex->einvstd && ex->egamma && ex->ebeta && ex->ldz % 8 == 0 &&
                         aligned16(ex->ez) && ((long)(M - 1) * ex->ldz + N) * 2 < 0x7fffffffL),
            "cgemm16: mask epilogue needs a bf16 output, an un-gathered product and the consumer BatchNorm's z / mean / invstd / gamma / beta");
    SCN_ARG(!ex || ex->pro == 0, "cgemm16: no prologue in the bf16 path (the normalised map is materialised)");
    if (c3) {
        SCN_ARG((c3 == 1 || c3 == 4) && out_bf16 && beta == 0.f, "cgemm16: 3x3 modes 1 / 4, bf16 output");
        SCN_ARG(ex->Hi > 0 && ex->Wi > 0 && ex->Ho > 0 && ex->Wo > 0 && ex->c3c > 0 && ex->c3c % 32 == 0 && ex->c3_src_rows > 0 && K == 9 * ex->c3c,
                "cgemm16: 3x3 geometry / channel multiple of 32");
        SCN_ARG(c3 != 4 || (ex->stride == 2 && ex->Hi == 2 * ex->Ho && ex->Wi == 2 * ex->Wo && epi == 0), "cgemm16: stride-2 d input geometry");
        SCN_ARG(ex->c3_src_rows * lda * 2 < 0x7fffffffL, "cgemm16: 3x3 source map exceeds the descriptor range");
    }
    SCN_ARG(!gather || (ex->Hi > 0 && ex->Wi > 0 && ex->Ho > 0 && ex->Wo > 0), "cgemm16: gather geometry");
    const long out_rows = c3 == 4 ? 4L * M : M;
    SCN_ARG(((out_rows - 1) * ldc + N) * (out_bf16 ? 2 : 4) < 0x7fffffffL && (long)M * N * 4 < 0x7fffffffL, "cgemm16: C too large");
    int mi = 2;
    if ((long)cdiv(M, 128) * cdiv(N, TN) * (c3 == 4 ? 4 : 1) < 256 && M > 64) mi = 1;
    if (ex && (ex->force_mi == 1 || ex->force_mi == 2)) mi = ex->force_mi;
    const int nt = cdiv(N, TN), mt = cdiv(M, 64 * mi);
    const long tiles = (long)mt * nt;
    int S = 1;
    if (c3 != 4 && ws && tiles < 192 && K >= 512) {      // few tiles, deep K: split to ~512 workgroups, >= 256 k per slab
        S = (int)((512 + tiles - 1) / tiles);
        if (S > K / 256) S = K / 256;
        if (S > 16) S = 16;
        while (S > 1 && (long)S * M * N > ws_floats) --S;
        if (S < 1) S = 1;
    }
    if (epi == 2) S = 1;        // the mask epilogue lives in the product's own launch: un-split (only layer4's shapes would split)
    if (ex && ex->force_split > 0 && c3 != 4 && epi != 2) {
        S = ex->force_split;
        SCN_ARG(S == 1 || (ws && (long)S * M * N <= ws_floats && S <= 64), "cgemm16: forced split does not fit");
    }
    int kper = cdiv(K, S);
    kper = (kper + TKE - 1) / TKE * TKE;
    S = cdiv(K, kper);
    HArgs g{};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.beta = beta; g.S = S; g.kper = kper; g.ws = ws; g.mt = mt; g.nt = nt; g.flip = flip;
    if (ex) {
        g.gHi = ex->Hi; g.gWi = ex->Wi; g.gHo = ex->Ho; g.gWo = ex->Wo; g.gs = (gather || c3) ? ex->stride : 0;
        g.c3c = ex->c3c; g.src_rows = ex->c3_src_rows;
        if (c3 == 4) { g.gHi = ex->Ho; g.gWi = ex->Wo; g.dHi = ex->Hi; g.dWi = ex->Wi; }
        g.stat_partial = ex->stat_partial; g.stat_shift = ex->stat_shift; g.ldp = cgemm_stat_ld(M);
        g.ez = (const bf16_t*)ex->ez; g.ldz = ex->ldz; g.emean = ex->emean; g.einvstd = ex->einvstd; g.egamma = ex->egamma; g.ebeta = ex->ebeta;
    }
    dim3 grid(mt * nt, c3 == 4 ? 4 : S);
    const int kepi = S > 1 ? 0 : epi;
    const bool obf = out_bf16 != 0;
    if (c3 == 1)      { if (mi == 2) launch16<2, 1>(st, grid, g, kepi, false, true); else launch16<1, 1>(st, grid, g, kepi, false, true); }
    else if (c3 == 4) { if (mi == 2) launch16<2, 4>(st, grid, g, 0, false, true); else launch16<1, 4>(st, grid, g, 0, false, true); }
    else              { if (mi == 2) launch16<2, 0>(st, grid, g, kepi, gather, obf); else launch16<1, 0>(st, grid, g, kepi, gather, obf); }
    SCN_LAUNCH_CHECK();
    if (S > 1) {
        dim3 sgrid(cdiv(N, 64), cdiv(M, SROWS)), block(256);
        if (epi == 1) { if (obf) hipLaunchKernelGGL((creduce16_kernel<true, true>), sgrid, block, 0, st, g); else hipLaunchKernelGGL((creduce16_kernel<false, true>), sgrid, block, 0, st, g); }
        else          { if (obf) hipLaunchKernelGGL((creduce16_kernel<true, false>), sgrid, block, 0, st, g); else hipLaunchKernelGGL((creduce16_kernel<false, false>), sgrid, block, 0, st, g); }
        SCN_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace scn
