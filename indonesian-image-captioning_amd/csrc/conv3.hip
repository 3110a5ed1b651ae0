// 3x3 weight gradient of the ResNet-152 trunk with the activation halo staged ONCE, gfx950 only, exact fp32
// (v_mfma_f32_32x32x2_f32).
//
// What it replaces: the d-weight half of `torch.nn.Conv2d(k=3, padding=1)` in torchvision's Bottleneck.conv2 behind the
// reference's models/encoders/caption.py:17-22 (autograd -> MIOpen `igemm_wrw_*` in rounds 1-2):
//     dW[co][kh][kw][ci] = sum_{n,h,w} dY[n,h,w,co] * X[n, h+kh-1, w+kw-1, ci]          (stride 1, zero padding 1)
//
// Formulation.  The [K][M] x [K][N] gather form of csrc/cgemm.hip (mode C3 = 3) gives every tap its own column tile,
// so each activation row passes through LDS-DMA nine times and the output tile count (and with it the split-K slab
// traffic) is fixed by 128-wide tiles.  Here ONE WAVE owns a 32(co) x 32(ci) block of dW for ALL NINE taps -- nine
// 32x32 accumulators, 144 registers -- and walks output pixels:
//   * k = output pixel.  A = dY[pixel][co] is read once per k-step and feeds nine MFMAs, whose B operands are the SAME
//     staged activation rows read at nine shifted LDS addresses (rows h-1, h, h+1 x columns w-1, w, w+1);
//   * the map is cut into strips of SEG (16 or 8) columns; a wave walks a strip line by line with a 4-slot ring of
//     activation lines (SEG + 2 pixels x 32 channels each, the two halo columns included) and a 2-slot ring of dY lines:
//     going from row h to h+1 brings in ONE new activation line, so every activation byte is staged once per
//     (strip, co-tile), not nine times.  Out-of-image rows / columns are LDS-DMA lanes with an out-of-range offset
//     (hardware zero fill): no predicates anywhere in the k-loop;
//   * the four waves of a workgroup own the SAME output block and split K between them.  Their rings are private, so
//     the k-loop has NO s_barrier at all (only counted `s_waitcnt vmcnt` on a wave's own DMA); at the end the four
//     partial blocks meet in LDS and are summed in wave order (deterministic);
//   * a 32 x 32 x 9 block per workgroup means many small tiles: (Cout/32) x (Cin/32) = 16 / 64 / 256 of them for layer2 /
//     3 / 4, so a 512-workgroup grid needs only 32 / 8 / 2 K-slices whose slabs are 19 MB in all (the 128 x 128 tiling
//     needed ~50 MB), reduced by the same slab-order pass the GEMMs use.  layer4 can run with no slabs at all.
// LDS: 53 KB (SEG 16) / 29 KB (SEG 8) per workgroup, ~170 VGPRs -> two workgroups per CU, or one beside a cgemm workgroup
// of the main stream.
#include "common.h"
#include "kernels.h"

namespace scn {

int cgemm_reduce(hipStream_t st, const float* ws, int S, int M, int N, float* C, long ldc);

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;

struct W3Args {
    const float* dy;      // [N*H*W][Co]
    const float* x;       // [N*H*W][C]
    float* out;           // S == 1: dW [Co][9][C]; S > 1: slabs [S][Co][9*C]
    int N, H, W, C, Co;
    int nseg;             // column segments per map row
    int Q;                // segment lines = N * nseg * H (the K dimension in units of one strip row)
    int S;                // K slices at workgroup level (each is split again over the 4 waves)
    int tci, ntiles;      // ci tiles, (Co/32) * (C/32)
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_wave_base, 16, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

template <int SEG>
__global__ __launch_bounds__(256, 2) void conv3_wgrad_kernel(W3Args g) {
    constexpr int XPX = SEG + 2;                   // pixels of a staged activation line (halo columns included)
    constexpr int XSLOT = XPX * 32, YSLOT = SEG * 32;
    constexpr int NX = (XPX + 7) / 8, NY = SEG / 8;    // LDS-DMA instructions per activation / dY line (8 pixels each)
    constexpr int WAVE_F = 4 * XSLOT + 2 * YSLOT;
    constexpr int TPR = SEG == 16 ? 3 : 1;         // taps per reduction round (sized to fit the ring space)
    constexpr int LDS_F = (4 * WAVE_F > 4 * TPR * 1024) ? 4 * WAVE_F : 4 * TPR * 1024;
    __shared__ __attribute__((aligned(16))) float lds[LDS_F];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    // XCD-aware order (speed only): an XCD owns contiguous virtual ids = whole K slices, so a slice's dY / X lines are
    // fetched into ONE L2 and shared there by every (co, ci) tile
    const int total = g.ntiles * g.S;
    int v;
    {
        const int bid = blockIdx.x, q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ks = v / g.ntiles, tile = v - ks * g.ntiles;
    const int tco = tile / g.tci, tci = tile - tco * g.tci;
    const int co0 = tco * 32, ci0 = tci * 32;
    const long wsl = 4L * g.S;
    const int q0 = (int)(((long)g.Q * (4 * ks + wave)) / wsl), q1 = (int)(((long)g.Q * (4 * ks + wave + 1)) / wsl);

    const long rows = (long)g.N * g.H * g.W;
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(g.dy, (unsigned)(rows * g.Co * 4));
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(g.x, (unsigned)(rows * g.C * 4));
    float* const xring = lds + wave * WAVE_F;
    float* const yring = xring + 4 * XSLOT;
    const int pl = lane >> 3, gq = lane & 7;       // DMA lane -> pixel within a group of 8, 16-byte granule of its 32 channels

    // activation line hx of strip (n, cs) -> ring slot (hx + 1) & 3; rows outside the image are all-zero lines
    auto issue_x = [&](int n, int cs, int hx) {
        float* dst = xring + ((hx + 1) & 3) * XSLOT;
        const bool rowok = (unsigned)hx < (unsigned)g.H;
        const long rb = ((long)n * g.H + hx) * g.W;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int pp = 8 * j + pl, w = cs * SEG - 1 + pp;
            const bool ok = rowok && (unsigned)w < (unsigned)g.W;
            const unsigned voff = ok ? (unsigned)(((rb + w) * g.C + ci0 + 4 * gq) * 4) : OOB_OFF;
            if (8 * j + 8 <= XPX || pp < XPX) dma16(xrs, dst + j * 256, voff);     // last group: the lanes past the line stay off
        }
    };
    auto issue_y = [&](int n, int cs, int h, int slot) {
        float* dst = yring + slot * YSLOT;
        const long rb = ((long)n * g.H + h) * g.W + cs * SEG;
#pragma unroll
        for (int j = 0; j < NY; ++j) {       // pixels past the map's width (a ragged last strip) land as zeros: they add nothing
            const bool ok = cs * SEG + 8 * j + pl < g.W;
            dma16(yrs, dst + j * 256, ok ? (unsigned)(((rb + 8 * j + pl) * g.Co + co0 + 4 * gq) * 4) : OOB_OFF);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int q = q0; q < q1; ++q) {
        const int strip = q / g.H, h = q - strip * g.H;
        const int n = strip / g.nseg, cs = strip - n * g.nseg;
        asm volatile("" ::: "memory");
        if (q == q0 || h == 0) {      // first line of this wave / of a strip: stage the three lines and the dY line
            issue_x(n, cs, h - 1);
            issue_x(n, cs, h);
            issue_x(n, cs, h + 1);
            issue_y(n, cs, h, q & 1);
        }
        if (q + 1 < q1 && h + 1 < g.H) {       // next line continues the strip: ONE new activation line + its dY line
            issue_x(n, cs, h + 2);
            issue_y(n, cs, h + 1, (q + 1) & 1);
            wait_vm<NX + NY>();
        } else {
            wait_vm<0>();
        }
        const float* yb = yring + (q & 1) * YSLOT + hh * 32 + l31;
        const float* xb0 = xring + ((h + 0) & 3) * XSLOT + hh * 32 + l31;
        const float* xb1 = xring + ((h + 1) & 3) * XSLOT + hh * 32 + l31;
        const float* xb2 = xring + ((h + 2) & 3) * XSLOT + hh * 32 + l31;
#pragma unroll
        for (int kk = 0; kk < SEG / 2; ++kk) {
            const float a = yb[kk * 64];
            float b[9];
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                b[dw] = xb0[(2 * kk + dw) * 32];
                b[3 + dw] = xb1[(2 * kk + dw) * 32];
                b[6 + dw] = xb2[(2 * kk + dw) * 32];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of the slots the next DMAs overwrite has returned
    }

    // ---- the four K-partial blocks meet in LDS, TPR taps per round, summed in wave order ------------------------------
    __syncthreads();
    float* const outb = g.out + (g.S > 1 ? (long)ks * g.Co * 9 * g.C : 0L);
    const int rrow = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int r0 = 0; r0 < 9; r0 += TPR) {
#pragma unroll
        for (int t = 0; t < TPR; ++t) {
            float* w = lds + (wave * TPR + t) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) w[mfma32_row(r, lane) * 32 + l31] = acc[r0 + t][r];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPR; ++t) {
            f32x4 s = *reinterpret_cast<const f32x4*>(lds + (0 * TPR + t) * 1024 + rrow * 32 + c4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(lds + (w * TPR + t) * 1024 + rrow * 32 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += u[e];
            }
            *reinterpret_cast<f32x4*>(outb + (long)(co0 + rrow) * 9 * g.C + (long)(r0 + t) * g.C + ci0 + c4) = s;
        }
        __syncthreads();
    }
}

}  // namespace

int g_w3_target = 512;     // workgroups the halo-staged 3x3 weight gradient aims for (tuning: tools/cgemm_bench.py sweepw)

bool conv3x3_wgrad_halo_ok(int N, int H, int W, int C, int Co, const float* dy, const float* x, const float* dw) {
    return N > 0 && H > 0 && W > 0 && C % 32 == 0 && Co % 32 == 0 && aligned16(dy) && aligned16(x) &&
           aligned16(dw) && (long)N * H * W * (C > Co ? C : Co) * 4 < 0x7fffffffL;
}

// dW [Co][3][3][C] (channels-last weight layout) from dY [N*H*W][Co] and X [N*H*W][C]; stride 1, padding 1.
// force_split > 0 fixes the workgroup-level K split (tests / tuning).
int conv3x3_wgrad_halo(hipStream_t st, int N, int H, int W, int C, int Co, const float* dy, const float* x, float* dw,
                       float* ws, long ws_floats, int force_split) {
    SCN_ARG(conv3x3_wgrad_halo_ok(N, H, W, C, Co, dy, x, dw), "conv3x3_wgrad_halo: shape / alignment not supported");
    const int seg = (W % 16 == 0) ? 16 : 8;
    W3Args g{};
    g.dy = dy; g.x = x; g.N = N; g.H = H; g.W = W; g.C = C; g.Co = Co;
    g.nseg = (W + seg - 1) / seg;       // widths that are no multiple of 8: the last strip is ragged (zero-filled)
    g.Q = N * g.nseg * H;
    g.tci = C / 32;
    g.ntiles = (Co / 32) * g.tci;
    int S = force_split > 0 ? force_split : (g_w3_target + g.ntiles / 2) / g.ntiles;
    const int smax = g.Q / 16 > 0 ? g.Q / 16 : 1;       // at least 4 lines per wave
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    const long mn = (long)Co * 9 * C;
    while (S > 1 && (!ws || (long)S * mn > ws_floats)) --S;
    SCN_ARG(force_split <= 0 || S == force_split, "conv3x3_wgrad_halo: forced split does not fit the workspace / the map");
    g.S = S;
    g.out = S > 1 ? ws : dw;
    dim3 grid(g.ntiles * S), block(256);
    if (seg == 16) hipLaunchKernelGGL(conv3_wgrad_kernel<16>, grid, block, 0, st, g);
    else           hipLaunchKernelGGL(conv3_wgrad_kernel<8>, grid, block, 0, st, g);
    SCN_LAUNCH_CHECK();
    if (S > 1) SCN_TRY(cgemm_reduce(st, ws, S, Co, 9 * C, dw, 9L * C));
    return 0;
}

}  // namespace scn
