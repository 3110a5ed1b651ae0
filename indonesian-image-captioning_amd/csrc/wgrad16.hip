// bf16 weight gradients of the ResNet-152 convolution path (BASELINE configs[4]), gfx950 only:
//     dW[co][tap][ci] = sum_pixels dY[pixel][co] * X[shift_tap(pixel)][ci]      dY, X bf16 maps; dW fp32 (master precision)
// on v_mfma_f32_32x32x16_bf16 -- the d-weight half of every `nn.Conv2d` of torchvision's Bottleneck behind the reference's
// models/encoders/caption.py:17-22 when the trunk runs in mixed precision.
//
// Both operands are [pixel][channel] maps and the contraction runs over PIXELS, so both MFMA operands need eight
// consecutive k (pixels) of one channel per lane: a column of the LDS image.  gfx950 reads that with the transposing LDS
// load ds_read_b64_tr_b16 (a 4-row x 16-column block per 16 lanes, delivered column-major): the maps are staged as they
// lie in memory -- one LDS-DMA instruction moves 16 pixels x 32 channels -- and no transposed copy of any map exists.
//
// Structure = csrc/conv3.hip's (fp32 halo-staged weight gradient): a wave owns a block of dW, walks its share of the
// pixels line by line (16 pixels = ONE matrix instruction per tap) with wave-private LDS rings and NO s_barrier in the
// loop; the four waves of a workgroup own the same block, split K and meet in LDS at the end (fixed order).
//   w9: stride-1 3x3: 32(co) x 32(ci) x 9 taps per wave, activation lines staged once with their halo and read at nine
//       shifted addresses;
//   w1: everything else: a 64 x 64 block of one [Cout][Cin] slice over rows that may be GATHERED -- plain 1x1 weight
//       gradients, the strided 1x1 downsample, and a stride-2 3x3 (layerN.0) as nine launches, one per tap, whose source
//       pixel (2 ho + dh - 1, 2 wo + dw - 1) is a per-lane row index (out-of-image = out-of-range = zeros).
#include "common.h"
#include "kernels.h"

namespace scn {

int cgemm_reduce(hipStream_t st, const float* ws, int S, int M, int N, float* C, long ldc);

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v8i16 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_wave_base, 16, voff, 0, 0, 0);
}
// 8 consecutive k (rows of the LDS image, `rs` bytes apart) of this lane's column as one MFMA operand: two transposing
// reads of 4 rows each.  `p` = this lane's address for the first: row (k0 + q), columns 4p' .. 4p'+3 of its 16-column block.
__device__ __forceinline__ bf16x8 tr_frag(const char* p, int rs) {
    typedef __attribute__((address_space(3))) v4i16* lp;
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * rs));
    const v8i16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

struct W16Args {
    const bf16_t* dy;     // [rows][Co]
    const bf16_t* x;      // [src rows][C]
    float* out;           // S == 1: dW (ldo); S > 1: slabs [S][Mo][No] with Mo x No the full output matrix
    long ldo;             // leading dimension of dW (9*C for a 3x3 weight)
    long slab_ld, slab_stride;
    int N, H, W, C, Co;   // w9: the map; w1: H = W = 0
    int nseg, Q, S, tci, ntiles;
    // w1: rows / gather
    int R;                // output pixels (rows of dy)
    int gHi, gWi, gHo, gWo, gs, goh, gow;   // gs > 0: source pixel = (ho*gs + goh, wo*gs + gow) of a gHi x gWi map, else row itself
    long src_rows;
};

// ---------------------------------------------------------------------------------------------------------------------
// w9: stride-1 3x3, 32 x 32 x 9 per wave
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad16_w9_kernel(W16Args g) {
    constexpr int SEG = 16, XPX = SEG + 2;
    constexpr int XSLOT = XPX * 64, YSLOT = SEG * 64;      // bytes: 32 channels x 2 B per pixel
    constexpr int WAVE_B = 4 * XSLOT + 2 * YSLOT;          // 6656 B per wave
    constexpr int TPR = 3;
    constexpr int LDS_B = (4 * WAVE_B > 4 * TPR * 4096) ? 4 * WAVE_B : 4 * TPR * 4096;
    __shared__ __attribute__((aligned(16))) char lds[LDS_B];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
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
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(g.dy, (unsigned)(rows * g.Co * 2));
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(g.x, (unsigned)(rows * g.C * 2));
    char* const xring = lds + wave * WAVE_B;
    char* const yring = xring + 4 * XSLOT;
    const int pl = lane >> 2, gq = lane & 3;       // DMA lane -> pixel within a group of 16, 16-byte granule of its 64 bytes

    auto issue_x = [&](int n, int cs, int hx) {
        char* dst = xring + ((hx + 1) & 3) * XSLOT;
        const bool rowok = (unsigned)hx < (unsigned)g.H;
        const long rb = ((long)n * g.H + hx) * g.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = 16 * j + pl, w = cs * SEG - 1 + pp;
            const bool ok = rowok && (unsigned)w < (unsigned)g.W;
            const unsigned voff = ok ? (unsigned)(((rb + w) * g.C + ci0 + 8 * gq) * 2) : OOB_OFF;
            if (j == 0 || pp < XPX) dma16(xrs, dst + j * 1024, voff);
        }
    };
    auto issue_y = [&](int n, int cs, int h, int slot) {
        char* dst = yring + slot * YSLOT;
        const long rb = ((long)n * g.H + h) * g.W + cs * SEG;
        const bool ok = cs * SEG + pl < g.W;
        dma16(yrs, dst, ok ? (unsigned)(((rb + pl) * g.Co + co0 + 8 * gq) * 2) : OOB_OFF);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read address of this lane inside a line image: row (8h + q) [+ 4 for the second read], columns 4p .. 4p+3
    // of the 16-column block g1 = (lane >> 4) & 1
    const int i16 = lane & 15, qq = i16 >> 2, pp4 = i16 & 3, g1 = (lane >> 4) & 1, hh = lane >> 5;
    const int fo = (8 * hh + qq) * 64 + (16 * g1 + 4 * pp4) * 2;

    for (int q = q0; q < q1; ++q) {
        const int strip = q / g.H, h = q - strip * g.H;
        const int n = strip / g.nseg, cs = strip - n * g.nseg;
        asm volatile("" ::: "memory");
        if (q == q0 || h == 0) {
            issue_x(n, cs, h - 1);
            issue_x(n, cs, h);
            issue_x(n, cs, h + 1);
            issue_y(n, cs, h, q & 1);
        }
        if (q + 1 < q1 && h + 1 < g.H) {
            issue_x(n, cs, h + 2);
            issue_y(n, cs, h + 1, (q + 1) & 1);
            wait_vm<3>();
        } else {
            wait_vm<0>();
        }
        const bf16x8 a = tr_frag(yring + (q & 1) * YSLOT + fo, 64);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const char* xb = xring + ((h + dh) & 3) * XSLOT + fo;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw)
                acc[dh * 3 + dw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_frag(xb + dw * 64, 64), acc[dh * 3 + dw], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    __syncthreads();
    float* const red = reinterpret_cast<float*>(lds);
    float* const outb = g.out + (g.S > 1 ? (long)ks * g.slab_stride : 0L);
    const long ldo = g.S > 1 ? g.slab_ld : g.ldo;
    const int rrow = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int r0 = 0; r0 < 9; r0 += TPR) {
#pragma unroll
        for (int t = 0; t < TPR; ++t) {
            float* w = red + (wave * TPR + t) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) w[mfma32_row(r, lane) * 32 + l31] = acc[r0 + t][r];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPR; ++t) {
            f32x4 s = *reinterpret_cast<const f32x4*>(red + (0 * TPR + t) * 1024 + rrow * 32 + c4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(red + (w * TPR + t) * 1024 + rrow * 32 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += u[e];
            }
            *reinterpret_cast<f32x4*>(outb + (long)(co0 + rrow) * ldo + (long)(r0 + t) * g.C + ci0 + c4) = s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// w1: one [Cout][Cin] slice over (gathered) rows, 64 x 64 per wave (2 x 2 matrix tiles)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad16_w1_kernel(W16Args g) {
    constexpr int LINE = 16;                       // pixels per line = one k-block
    constexpr int SLOT = LINE * 128;               // bytes: 64 channels x 2 B per pixel
    constexpr int WAVE_B = 4 * SLOT;               // 2 dY + 2 X slots = 8 KiB per wave
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_B];      // 32 KiB (reduction: 16 KiB per round)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
    const int total = g.ntiles * g.S;
    int v;
    {
        const int bid = blockIdx.x, q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ks = v / g.ntiles, tile = v - ks * g.ntiles;
    const int tco = tile / g.tci, tci = tile - tco * g.tci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const long wsl = 4L * g.S;
    const int q0 = (int)(((long)g.Q * (4 * ks + wave)) / wsl), q1 = (int)(((long)g.Q * (4 * ks + wave + 1)) / wsl);

    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(g.dy, (unsigned)((long)g.R * g.Co * 2));
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(g.x, (unsigned)(g.src_rows * g.C * 2));
    char* const yring = lds + wave * WAVE_B;
    char* const xring = yring + 2 * SLOT;
    // DMA lane -> (pixel within a group of 8, LDS granule position gpos); the source granule is gpos ^ (((pixel >> 1) & 1) << 2):
    // rows k and k + 2 of a 128-byte-row image would hit the same banks in a transposing read, the swap of the two
    // 64-byte halves on every second pair of rows makes the four rows of a read block land in four different 16-dword windows
    const int pl = lane >> 3, gpos = lane & 7;
    // (channel tiles are whole: the host checks Co % 64 == 0 and C % 64 == 0)

    auto issue = [&](int line, int slot) {
        char* yd = yring + slot * SLOT;
        char* xd = xring + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = 8 * j + pl;                         // pixel of the line this lane stages
            const int gs_ = gpos ^ (((px >> 1) & 1) << 2);      // source granule
            const long r = (long)line * LINE + px;
            const bool ok = r < g.R;
            long src = r;
            bool sok = ok;
            if (g.gs > 0 && ok) {
                const int hw = g.gHo * g.gWo;
                const int n = (int)(r / hw), rem = (int)(r - (long)n * hw), ho = rem / g.gWo, wo = rem - ho * g.gWo;
                const int hi = ho * g.gs + g.goh, wi = wo * g.gs + g.gow;
                sok = (unsigned)hi < (unsigned)g.gHi && (unsigned)wi < (unsigned)g.gWi;
                src = ((long)n * g.gHi + hi) * g.gWi + wi;
            }
            dma16(yrs, yd + j * 1024, ok ? (unsigned)((r * g.Co + co0 + 8 * gs_) * 2) : OOB_OFF);
            dma16(xrs, xd + j * 1024, sok ? (unsigned)((src * g.C + ci0 + 8 * gs_) * 2) : OOB_OFF);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing-read address inside a [16 px][128 B] image: row k = 8h + q (+4), column block cb (32 channels) / g1 (16) / 4p:
    // 8-byte unit u = 8 cb + 4 g1 + p, granule u >> 1 swizzled by ((k >> 1) & 1) << 2 -- (k + 4) has the same swizzle as k
    const int i16 = lane & 15, qq = i16 >> 2, pp4 = i16 & 3, g1 = (lane >> 4) & 1, hh = lane >> 5;
    const int krow = 8 * hh + qq;
    int fo[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int u = 8 * cb + 4 * g1 + pp4;
        const int gran = (u >> 1) ^ (((krow >> 1) & 1) << 2);
        fo[cb] = krow * 128 + gran * 16 + (u & 1) * 8;
    }

    for (int q = q0; q < q1; ++q) {
        asm volatile("" ::: "memory");
        if (q == q0) issue(q, q & 1);
        if (q + 1 < q1) {
            issue(q + 1, (q + 1) & 1);
            wait_vm<4>();
        } else {
            wait_vm<0>();
        }
        const char* yb = yring + (q & 1) * SLOT;
        const char* xb = xring + (q & 1) * SLOT;
        bf16x8 a[2], b[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            a[cb] = tr_frag(yb + fo[cb], 128);
            b[cb] = tr_frag(xb + fo[cb], 128);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    __syncthreads();
    float* const red = reinterpret_cast<float*>(lds);
    float* const outb = g.out + (g.S > 1 ? (long)ks * g.slab_stride : 0L);
    const long ldo = g.S > 1 ? g.slab_ld : g.ldo;
    const int rrow = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* w = red + wave * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) w[mfma32_row(r, lane) * 32 + l31] = acc[i][j][r];
            __syncthreads();
            f32x4 s = *reinterpret_cast<const f32x4*>(red + rrow * 32 + c4);
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(red + w2 * 1024 + rrow * 32 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += u[e];
            }
            *reinterpret_cast<f32x4*>(outb + (long)(co0 + i * 32 + rrow) * ldo + ci0 + j * 32 + c4) = s;
            __syncthreads();
        }
}

}  // namespace

// dW [Co][3][3][C] fp32 from bf16 maps dy [N*H*W][Co], x [N*H*W][C]: stride 1, padding 1.  ws: slabs (S * Co*9*C floats).
int wgrad16_3x3(hipStream_t st, int N, int H, int W, int C, int Co, const void* dy, const void* x, float* dw, float* ws,
                long ws_floats, int force_split) {
    SCN_ARG(N > 0 && H > 0 && W > 0 && C % 32 == 0 && Co % 32 == 0 && dy && x && dw && aligned16(dy) && aligned16(x) && aligned16(dw),
            "wgrad16_3x3: shape / alignment");
    SCN_ARG((long)N * H * W * (C > Co ? C : Co) * 2 < 0x7fffffffL, "wgrad16_3x3: map exceeds the descriptor range");
    W16Args g{};
    g.dy = (const bf16_t*)dy; g.x = (const bf16_t*)x; g.N = N; g.H = H; g.W = W; g.C = C; g.Co = Co;
    g.nseg = (W + 15) / 16;
    g.Q = N * g.nseg * H;
    g.tci = C / 32;
    g.ntiles = (Co / 32) * g.tci;
    int S = force_split > 0 ? force_split : (512 + g.ntiles / 2) / g.ntiles;
    const int smax = g.Q / 16 > 0 ? g.Q / 16 : 1;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    const long mn = (long)Co * 9 * C;
    while (S > 1 && (!ws || (long)S * mn > ws_floats)) --S;
    SCN_ARG(force_split <= 0 || S == force_split, "wgrad16_3x3: forced split does not fit");
    g.S = S; g.ldo = 9L * C; g.slab_ld = 9L * C; g.slab_stride = mn;
    g.out = S > 1 ? ws : dw;
    hipLaunchKernelGGL(wgrad16_w9_kernel, dim3(g.ntiles * S), dim3(256), 0, st, g);
    SCN_LAUNCH_CHECK();
    if (S > 1) SCN_TRY(cgemm_reduce(st, ws, S, Co, 9 * C, dw, 9L * C));
    return 0;
}

// dw[co][ci] (leading dimension ldo, fp32) = sum_r dy[r][co] * x[src(r)][ci] over R output pixels; src(r) = r, or with
// gs > 0 the pixel (ho*gs + goh, wo*gs + gow) of a gHi x gWi map for r = (n, ho, wo) of a gHo x gWo grid (zeros outside).
int wgrad16_rows(hipStream_t st, int R, int C, int Co, const void* dy, const void* x, long src_rows, float* dw, long ldo,
                 int gs, int gHi, int gWi, int gHo, int gWo, int goh, int gow, float* ws, long ws_floats, int force_split) {
    SCN_ARG(R > 0 && C % 64 == 0 && Co % 64 == 0 && dy && x && dw && aligned16(dy) && aligned16(x) && aligned16(dw) && ldo % 4 == 0 && src_rows > 0,
            "wgrad16_rows: shape / alignment (channel counts in multiples of 64)");
    SCN_ARG((long)R * Co * 2 < 0x7fffffffL && src_rows * C * 2 < 0x7fffffffL, "wgrad16_rows: map exceeds the descriptor range");
    SCN_ARG(gs == 0 || (gHi > 0 && gWi > 0 && gHo > 0 && gWo > 0 && R % (gHo * gWo) == 0), "wgrad16_rows: gather geometry");
    W16Args g{};
    g.dy = (const bf16_t*)dy; g.x = (const bf16_t*)x; g.C = C; g.Co = Co; g.R = R; g.src_rows = src_rows;
    g.gs = gs; g.gHi = gHi; g.gWi = gWi; g.gHo = gHo; g.gWo = gWo; g.goh = goh; g.gow = gow;
    g.Q = (R + 15) / 16;
    g.tci = C / 64;
    g.ntiles = (Co / 64) * g.tci;
    int S = force_split > 0 ? force_split : (512 + g.ntiles / 2) / g.ntiles;
    const int smax = g.Q / 16 > 0 ? g.Q / 16 : 1;       // at least 4 lines per wave
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    const long mn = (long)Co * C;
    while (S > 1 && (!ws || (long)S * mn > ws_floats)) --S;
    SCN_ARG(force_split <= 0 || S == force_split, "wgrad16_rows: forced split does not fit");
    g.S = S; g.ldo = ldo; g.slab_ld = C; g.slab_stride = mn;
    g.out = S > 1 ? ws : dw;
    hipLaunchKernelGGL(wgrad16_w1_kernel, dim3(g.ntiles * S), dim3(256), 0, st, g);
    SCN_LAUNCH_CHECK();
    if (S > 1) SCN_TRY(cgemm_reduce(st, ws, S, Co, C, dw, ldo));
    return 0;
}

}  // namespace scn
