// The stem of the ResNet-152 trunk on hand-written kernels, gfx950 only, exact fp32:
//     conv1 7x7 / stride 2 / padding 3 (3 -> 64 channels)  ->  BatchNorm  ->  ReLU  ->  MaxPool 3x3 / stride 2 / padding 1
// = children 0..3 of `nn.Sequential(*list(resnet152.children())[:-2])` behind the reference's
// models/encoders/caption.py:17-22 (and models/encoders/tagger.py:18-24).  The stem is frozen in every configuration
// of the reference (caption.py:46-57 re-enables children[5:] only), so only the forward pass exists.
//
// stem_conv7_kernel: implicit GEMM  z[(n,oh,ow)][co] = sum_{kh,kw,c} x[n, c, 2oh+kh-3, 2ow+kw-3] * w[co][c][kh][kw],
//   M = N*Ho*Wo pixels, N = 64, K = 147.  A 3-channel pixel is 12 bytes, which no 16-byte LDS-DMA granule can address,
//   and its K is far too short for a staged k-loop -- so the roles are turned round:
//   * the WHOLE weight matrix lives in LDS for the lifetime of a (persistent) workgroup, transposed to [k][64] with each
//     kernel row padded from 21 to 22 taps (k = kh*22 + kw*3 + c; the pad tap has weight 0), K' = 154 = 77 MFMA steps;
//   * per 8 x 16 output tile the workgroup stages the 21 x 37 x 3 input patch ONCE (read with hardware range checking:
//     the zero padding costs nothing) as [row][col][c]; the A fragment of pixel (oh, ow) for k = (kh, kw*3 + c) is then the
//     LDS word  222*oh + 6*ow + 111*kh + (kw*3 + c)  -- an immediate offset per k-step, no im2col, no repack of the image;
//     the even row padding is what makes the two k of an MFMA step (k0, k0 + 1) always neighbours in that row;
//   * epilogue: per-tile column sums of (z - s), (z - s)^2 for the BatchNorm that follows (the statistics epilogue of
//     csrc/cgemm.hip, same partial layout -> scnattn_bn_finalize), then z leaves as 128-byte row segments.
// stem_bn_relu_maxpool_kernel: out = maxpool3x3s2(relu(z * scale[c] + shift[c])) in one pass (the normalised 128 x 128 map
//   is never written): 134 MB read + 33.5 MB written at batch 32 instead of (134 r + 134 w) + (134 r + 33.5 w).
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

constexpr int S_TH = 8, S_TW = 16;                       // output tile
constexpr int S_PR = 2 * S_TH + 5, S_PC = 2 * S_TW + 5;  // input patch 21 x 37
constexpr int S_ROW = S_PC * 3;                          // 111 floats per patch row
constexpr int S_PATCH = S_PR * S_ROW;                    // 2331
constexpr int S_PATCH_F = 2368;                          // + slack read by the pad tap of the last pixel (kept zero)
constexpr int S_KP = 7 * 22;                             // padded K = 154
constexpr int S_BT_F = S_KP * 64;
constexpr int S_RED_F = 4 * 2 * 64;

struct StemArgs {
    const float* x; long sn, sc, sh, sw; long x_elems;   // element strides of the (N,3,H,W) input, whatever its memory format
    const float* w; long wn, wc, wh, ww;                 // element strides of the (64,3,7,7) weight
    float* z;                                            // [N*Ho*Wo][64]
    float* partial; const float* stat_shift;             // channel-major [2][64][ldp], one entry per workgroup; nullptr: no statistics
    int N, H, W, Ho, Wo, tx, ty, ntiles, ldp;
};

__global__ __launch_bounds__(256, 3) void stem_conv7_kernel(StemArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[S_PATCH_F + S_BT_F + S_RED_F];
    float* const patch = lds;
    float* const bt = lds + S_PATCH_F;
    float* const red = bt + S_BT_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;

    // ---- weights -> LDS, once: bt[k = kh*22 + kw*3 + c][co], pad tap (kw*3 + c == 21) = 0 -------------------------
    for (int e = tid; e < S_BT_F; e += 256) {
        const int k = e >> 6, co = e & 63, kh = k / 22, kk = k - kh * 22, kw = kk / 3, c = kk - 3 * kw;
        bt[e] = kk < 21 ? g.w[co * g.wn + c * g.wc + kh * g.wh + kw * g.ww] : 0.f;
    }
    for (int e = S_PATCH + tid; e < S_PATCH_F; e += 256) patch[e] = 0.f;

    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(g.x, (unsigned)(g.x_elems * 4));
    // this lane's pixel inside the tile (rows of the 32 x 32 MFMA block of this wave) and its patch origin
    const int pi = wave * 32 + l31, ohl = pi >> 4, owl = pi & 15;
    const float* const abase = patch + ohl * (2 * S_ROW) + owl * 6 + hh;
    const float* const bbase = bt + hh * 64 + l31;
    float stat_run = 0.f;        // threads 0..127: this workgroup's running sum for (which = tid >> 6, channel = tid & 63)

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int n = tile / (g.tx * g.ty), trem = tile - n * (g.tx * g.ty), tyi = trem / g.tx, txi = trem - tyi * g.tx;
        const int oh0 = tyi * S_TH, ow0 = txi * S_TW;
        __syncthreads();                               // every wave is done with the previous patch / red
        {
            constexpr int PER = (S_PATCH + 255) / 256; // 10 elements per thread, all loads in flight before the first LDS write
            float v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + 256 * u;
                const int r = e / S_ROW, rem = e - r * S_ROW, c = rem / 3, ch = rem - 3 * c;
                const int hi = 2 * oh0 - 3 + r, wi = 2 * ow0 - 3 + c;
                const bool ok = e < S_PATCH && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                v[u] = buf_load(xrs, ok ? (unsigned)((n * g.sn + ch * g.sc + hi * g.sh + wi * g.sw) * 4) : OOB_OFF);
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + 256 * u;
                if (e < S_PATCH) patch[e] = v[u];
            }
        }
        __syncthreads();

        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int kk = 0; kk < 22; kk += 2) {
                const float a = abase[kh * S_ROW + kk];
                const float b0 = bbase[(kh * 22 + kk) * 64], b1 = bbase[(kh * 22 + kk) * 64 + 32];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }

        // ---- statistics of this tile (valid pixels only), fixed order ------------------------------------------------
        if (g.partial) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float sft = g.stat_shift ? g.stat_shift[j * 32 + l31] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = wave * 32 + mfma32_row(r, lane);
                    const bool ok = oh0 + (p >> 4) < g.Ho && ow0 + (p & 15) < g.Wo;
                    const float d = ok ? acc[j][r] - sft : 0.f;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    red[(wave * 2 + 0) * 64 + j * 32 + l31] = s1;
                    red[(wave * 2 + 1) * 64 + j * 32 + l31] = s2;
                }
            }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, co = tid & 63;
                stat_run += ((red[(0 * 2 + which) * 64 + co] + red[(1 * 2 + which) * 64 + co]) + red[(2 * 2 + which) * 64 + co]) +
                            red[(3 * 2 + which) * 64 + co];
            }
        }
        // ---- z: lane = channel, register = pixel: two 128-byte row segments per store instruction ---------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = wave * 32 + mfma32_row(r, lane);
            const int oh = oh0 + (p >> 4), ow = ow0 + (p & 15);
            if (oh < g.Ho && ow < g.Wo) {
                float* zp = g.z + (((long)n * g.Ho + oh) * g.Wo + ow) * 64 + l31;
                zp[0] = acc[0][r];
                zp[32] = acc[1][r];
            }
        }
    }
    // one partial row per (persistent) workgroup, tiles summed in the order this workgroup walked them
    if (g.partial && tid < 128) g.partial[((long)(tid >> 6) * 64 + (tid & 63)) * g.ldp + blockIdx.x] = stat_run;   // channel-major [2][64][ldp]
}

// out[(n,oh,ow)][c] = max over the 3x3 / stride 2 / padding 1 window of relu(z*scale[c] + shift[c]); 4 channels per thread
template <bool OBF>
__global__ __launch_bounds__(256) void stem_bn_relu_maxpool_kernel(long total4, int C, int Hz, int Wz, int Ho, int Wo,
                                                                   const float* __restrict__ z, const float* __restrict__ ss,
                                                                   void* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4n = C >> 2;
    const int cq = (int)(i % c4n);
    const long pix = i / c4n;
    const int ow = (int)(pix % Wo);
    const long t = pix / Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    const int c = cq * 4;
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(ss + 2 * c), t1 = *reinterpret_cast<const f32x4*>(ss + 2 * c + 4);
    const float sc[4] = {t0[0], t0[2], t1[0], t1[2]}, sh[4] = {t0[1], t0[3], t1[1], t1[3]};
    f32x4 m = {0.f, 0.f, 0.f, 0.f};          // relu >= 0 and the window always holds its centre: max(0, ...) IS the relu
    f32x4 v[9];
    bool ok[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int hi = 2 * oh - 1 + dh, wi = 2 * ow - 1 + dw;
            ok[dh * 3 + dw] = (unsigned)hi < (unsigned)Hz && (unsigned)wi < (unsigned)Wz;
            const int hc = min(max(hi, 0), Hz - 1), wc = min(max(wi, 0), Wz - 1);
            v[dh * 3 + dw] = *reinterpret_cast<const f32x4*>(z + (((long)n * Hz + hc) * Wz + wc) * C + c);
        }
#pragma unroll
    for (int q = 0; q < 9; ++q)
        if (ok[q]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], fmaf(v[q][e], sc[e], sh[e]));
        }
    if (OBF) {        // bf16 map for the mixed-precision trunk
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + pix * C + c) = bf16x4{(__bf16)m[0], (__bf16)m[1], (__bf16)m[2], (__bf16)m[3]};
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + pix * C + c) = m;
    }
}

}  // namespace

// rows of the statistics partial = workgroups of the (persistent) convolution launch
int stem_tiles(int N, int H, int W) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int t = N * cdiv(Ho, S_TH) * cdiv(Wo, S_TW);
    return t < 768 ? t : 768;
}

// x: (N,3,H,W) with element strides (sn,sc,sh,sw); w: (64,3,7,7) with element strides (wn,wc,wh,ww);
// z [N*Ho*Wo][64], Ho = (H-1)/2+1; partial [2][64][(stem_tiles(N,H,W) + 3) & ~3] (channel-major, as csrc/cgemm.hip writes them) or NULL.
int stem_conv7(hipStream_t st, int N, int H, int W, const float* x, long sn, long sc, long sh, long sw, const float* w,
               long wn, long wc, long wh, long ww, float* z, float* partial, const float* stat_shift) {
    SCN_ARG(N > 0 && H > 0 && W > 0 && x && w && z, "stem_conv7: arguments");
    const long xe = (long)(N - 1) * sn + 2 * sc + (long)(H - 1) * sh + (long)(W - 1) * sw + 1;
    SCN_ARG(sn >= 0 && sc >= 0 && sh >= 0 && sw >= 0 && xe * 4 < 0x7fffffffL, "stem_conv7: input strides / size");
    StemArgs g{};
    g.x = x; g.sn = sn; g.sc = sc; g.sh = sh; g.sw = sw; g.x_elems = xe;
    g.w = w; g.wn = wn; g.wc = wc; g.wh = wh; g.ww = ww;
    g.z = z; g.partial = partial; g.stat_shift = stat_shift;
    g.N = N; g.H = H; g.W = W; g.Ho = (H - 1) / 2 + 1; g.Wo = (W - 1) / 2 + 1;
    g.ty = cdiv(g.Ho, S_TH); g.tx = cdiv(g.Wo, S_TW); g.ntiles = N * g.tx * g.ty;
    const int grid = g.ntiles < 768 ? g.ntiles : 768;      // persistent: 3 workgroups per CU keep the weights in LDS
    g.ldp = (grid + 3) & ~3;
    hipLaunchKernelGGL(stem_conv7_kernel, dim3(grid), dim3(256), 0, st, g);
    SCN_LAUNCH_CHECK();
    return 0;
}

// z [N*Hz*Wz][C] -> out [N*Ho*Wo][C], Ho = (Hz-1)/2+1; ss [C][2] = {scale, shift}
int stem_bn_relu_maxpool(hipStream_t st, int N, int Hz, int Wz, int C, const float* z, const float* ss, void* out, int out_bf16) {
    SCN_ARG(N > 0 && Hz > 0 && Wz > 0 && C % 4 == 0 && aligned16(z) && aligned16(ss) && aligned16(out), "stem_bn_relu_maxpool: arguments");
    const int Ho = (Hz - 1) / 2 + 1, Wo = (Wz - 1) / 2 + 1;
    const long total4 = (long)N * Ho * Wo * (C / 4);
    if (out_bf16) hipLaunchKernelGGL(stem_bn_relu_maxpool_kernel<true>, dim3(cdiv(total4, 256)), dim3(256), 0, st, total4, C, Hz, Wz, Ho, Wo, z, ss, out);
    else          hipLaunchKernelGGL(stem_bn_relu_maxpool_kernel<false>, dim3(cdiv(total4, 256)), dim3(256), 0, st, total4, C, Hz, Wz, Ho, Wo, z, ss, out);
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
