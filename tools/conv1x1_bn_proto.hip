// Prototype for the next round, NOT part of the product: a 1x1 convolution on an NHWC
// feature map, i.e. Y[R][Cout] = f(Z[R][Cin]) . W[Cout][Cin]^T, on the fp32 MFMA tiles of csrc/sgemm.hip with
//   * an A-operand PROLOGUE f = relu(z * scale[k] + shift[k]) -- the previous layer's BatchNorm + ReLU folded to one
//     fma per element with per-input-channel scale = gamma/std, shift = beta - mean*gamma/std -- so that the
//     normalised map is never written to or read from HBM, and
//   * an EPILOGUE that leaves per-(row-tile, channel) partial sums of y and y^2 (the statistics of the NEXT
//     BatchNorm), fixed order, no atomics.
// It times the plain GEMM, prologue only, and prologue + epilogue on the trunk's 1x1 shapes and checks a few
// outputs against a host computation.  Build: hipcc -O3 --offload-arch=gfx950 -o conv1x1_bn_proto conv1x1_bn_proto.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr unsigned OOB_OFF = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

constexpr int BM = 128, BN = 128, BK = 16, LDS_LD = BM + 4;

// both operands k-contiguous: element (r, k) at base[r*ld + k]
__device__ __forceinline__ void load_tile(__amdgpu_buffer_rsrc_t rs, long ld, int r0, int k0, int R, int K, int tid,
                                          float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = r0 + (tid >> 2) + 64 * i, k = k0 + (tid & 3) * 4;
        const f32x4 v = buf_load4(rs, (r < R && k < K) ? (unsigned)(((long)r * ld + k) * 4) : OOB_OFF);
#pragma unroll
        for (int c = 0; c < 4; ++c) reg[i][c] = v[c];
    }
}
__device__ __forceinline__ void store_tile(float (*lds)[LDS_LD], int tid, const float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 2) + 64 * i, k = (tid & 3) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) lds[k + c][r] = reg[i][c];
    }
}

// PRO: apply relu(z*scale+shift) to A while staging.  EPI: write column partial sums of the output tile.
template <bool PRO, bool EPI>
__global__ __launch_bounds__(256, 3) void conv1x1_kernel(int R, int Cin, int Cout, const float* __restrict__ Z,
                                                         const float* __restrict__ W, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ Y,
                                                         float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDS_LD];
    __shared__ float colsum[2][2][BN];          // [wm][sum | sumsq][column]
    extern __shared__ float ss[];               // PRO: scale[Cin] | shift[Cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (PRO) {
        for (int k = tid; k < Cin; k += 256) { ss[k] = scale[k]; ss[Cin + k] = shift[k]; }
        __syncthreads();
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[2][4], rb[2][4];
    const int nk = (Cin + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(Z, (unsigned)((long)R * Cin * 4));
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(W, (unsigned)((long)Cout * Cin * 4));
    auto prologue = [&](float (&reg)[2][4], int k0) {
        if (!PRO) return;
        const int k = k0 + (tid & 3) * 4;
        if (k >= Cin) return;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + k);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + Cin + k);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) reg[i][c] = fmaxf(fmaf(reg[i][c], sc[c], sh[c]), 0.f);
    };
    load_tile(ars, Cin, m0, 0, R, Cin, tid, ra);
    load_tile(brs, Cin, n0, 0, Cout, Cin, tid, rb);
    prologue(ra, 0);
    store_tile(As[0], tid, ra);
    store_tile(Bs[0], tid, rb);
    __syncthreads();
    const int hh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile(ars, Cin, m0, (kt + 1) * BK, R, Cin, tid, ra);
            load_tile(brs, Cin, n0, (kt + 1) * BK, Cout, Cin, tid, rb);
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
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            prologue(ra, (kt + 1) * BK);
            store_tile(As[cur ^ 1], tid, ra);
            store_tile(Bs[cur ^ 1], tid, rb);
        }
        __syncthreads();
    }
    // output + (EPI) per-column sums over the tile's 128 rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                const float v = acc[i][j][r];
                if (n < Cout && m < R) {
                    Y[(long)m * Cout + n] = v;
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
            }
        if (EPI) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) { colsum[wm][0][wn * 64 + j * 32 + l31] = s1; colsum[wm][1][wn * 64 + j * 32 + l31] = s2; }
        }
    }
    if (EPI) {
        __syncthreads();
        if (tid < BN && n0 + tid < Cout) {
            float* p = partial + ((long)blockIdx.y * 2) * Cout + n0 + tid;
            p[0] = colsum[0][0][tid] + colsum[1][0][tid];
            p[Cout] = colsum[0][1][tid] + colsum[1][1][tid];
        }
    }
}

// ---- variant 2: LDS layout [k&1][row][k>>1] for the k-contiguous operands ------------------------------------
// lane (row, hh) of an MFMA operand needs k = kk + hh for kk = 0, 2, 4, ...: with this layout those 8 values of a
// 16-k step are 32 contiguous bytes, so the fragment reads are two ds_read_b128 per 32-row tile per k-step instead of
// eight ds_read_b32, and staging a thread's 4 consecutive k is two ds_write_b64 instead of four ds_write_b32.
constexpr int KP = BK / 2;                     // 8 k-pairs per step; KPL = padded row length (8: none, 12: 48-byte rows)

template <int KPL>
__device__ __forceinline__ void store_tile_v2(float (*lds)[BM][KPL], int tid, const float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 2) + 64 * i, q = (tid & 3) * 2;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(&lds[0][r][q]) = f32x2{reg[i][0], reg[i][2]};
        *reinterpret_cast<f32x2*>(&lds[1][r][q]) = f32x2{reg[i][1], reg[i][3]};
    }
}

template <bool PRO, bool EPI, int KPL>
__global__ __launch_bounds__(256, 3) void conv1x1_v2_kernel(int R, int Cin, int Cout, const float* __restrict__ Z,
                                                            const float* __restrict__ W, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ Y,
                                                            float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As[2][2][BM][KPL];
    __shared__ __attribute__((aligned(16))) float Bs[2][2][BN][KPL];
    __shared__ float colsum[2][2][BN];
    extern __shared__ float ss[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (PRO) {
        for (int k = tid; k < Cin; k += 256) { ss[k] = scale[k]; ss[Cin + k] = shift[k]; }
        __syncthreads();
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[2][4], rb[2][4];
    const int nk = (Cin + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(Z, (unsigned)((long)R * Cin * 4));
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(W, (unsigned)((long)Cout * Cin * 4));
    auto prologue = [&](float (&reg)[2][4], int k0) {
        if (!PRO) return;
        const int k = k0 + (tid & 3) * 4;
        if (k >= Cin) return;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + k);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + Cin + k);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) reg[i][c] = fmaxf(fmaf(reg[i][c], sc[c], sh[c]), 0.f);
    };
    load_tile(ars, Cin, m0, 0, R, Cin, tid, ra);
    load_tile(brs, Cin, n0, 0, Cout, Cin, tid, rb);
    prologue(ra, 0);
    store_tile_v2<KPL>(As[0], tid, ra);
    store_tile_v2<KPL>(Bs[0], tid, rb);
    __syncthreads();
    const int hh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile(ars, Cin, m0, (kt + 1) * BK, R, Cin, tid, ra);
            load_tile(brs, Cin, n0, (kt + 1) * BK, Cout, Cin, tid, rb);
        }
        f32x4 a[2][2], b[2][2];      // [tile][half of the 8 k-pairs]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) a[i][h] = *reinterpret_cast<const f32x4*>(&As[cur][hh][wm * 64 + i * 32 + l31][4 * h]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) b[j][h] = *reinterpret_cast<const f32x4*>(&Bs[cur][hh][wn * 64 + j * 32 + l31][4 * h]);
#pragma unroll
        for (int p = 0; p < KP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][p >> 2][p & 3], b[j][p >> 2][p & 3], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            prologue(ra, (kt + 1) * BK);
            store_tile_v2<KPL>(As[cur ^ 1], tid, ra);
            store_tile_v2<KPL>(Bs[cur ^ 1], tid, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lane);
                const float v = acc[i][j][r];
                if (n < Cout && m < R) {
                    Y[(long)m * Cout + n] = v;
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
            }
        if (EPI) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) { colsum[wm][0][wn * 64 + j * 32 + l31] = s1; colsum[wm][1][wn * 64 + j * 32 + l31] = s2; }
        }
    }
    if (EPI) {
        __syncthreads();
        if (tid < BN && n0 + tid < Cout) {
            float* p = partial + ((long)blockIdx.y * 2) * Cout + n0 + tid;
            p[0] = colsum[0][0][tid] + colsum[1][0][tid];
            p[Cout] = colsum[0][1][tid] + colsum[1][1][tid];
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool PRO, bool EPI, int V = 1>
float run(int R, int Cin, int Cout, const float* Z, const float* W, const float* sc, const float* sh, float* Y, float* part,
          int iters) {
    dim3 grid((Cout + BN - 1) / BN, (R + BM - 1) / BM), block(256);
    const size_t dyn = PRO ? 2 * Cin * sizeof(float) : 0;
    auto launch = [&]() {
        if (V == 2) hipLaunchKernelGGL((conv1x1_v2_kernel<PRO, EPI, 8>), grid, block, dyn, 0, R, Cin, Cout, Z, W, sc, sh, Y, part);
        else if (V == 3) hipLaunchKernelGGL((conv1x1_v2_kernel<PRO, EPI, 12>), grid, block, dyn, 0, R, Cin, Cout, Z, W, sc, sh, Y, part);
        else        hipLaunchKernelGGL((conv1x1_kernel<PRO, EPI>), grid, block, dyn, 0, R, Cin, Cout, Z, W, sc, sh, Y, part);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3f / iters;
}

int main() {
    struct Shape { const char* name; int R, Cin, Cout; };
    const Shape shapes[] = {{"l1.conv3", 131072, 64, 256}, {"l2.conv1", 32768, 512, 128}, {"l2.conv3", 32768, 128, 512},
                            {"l3.conv1", 8192, 1024, 256}, {"l3.conv3", 8192, 256, 1024}, {"l4.conv1", 2048, 2048, 512},
                            {"l4.conv3", 2048, 512, 2048}, {"sq4096", 4096, 4096, 4096}};
    printf("%-10s %8s %6s %6s | %9s %7s | %9s %7s | %9s %7s | check\n", "layer", "R", "Cin", "Cout", "plain us", "TF",
           "+pro us", "TF", "+pro+epi", "TF");
    for (const Shape& s : shapes) {
        const long nz = (long)s.R * s.Cin, nw = (long)s.Cout * s.Cin, ny = (long)s.R * s.Cout;
        std::vector<float> hz(nz), hw(nw), hsc(s.Cin), hsh(s.Cin);
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        for (auto& v : hz) v = rnd();
        for (auto& v : hw) v = rnd() * 0.1f;
        for (int k = 0; k < s.Cin; ++k) { hsc[k] = 1.f + rnd(); hsh[k] = 0.2f * rnd(); }
        float *Z, *W, *sc, *sh, *Y, *part;
        const int mt = (s.R + BM - 1) / BM;
        CK(hipMalloc(&Z, nz * 4)); CK(hipMalloc(&W, nw * 4)); CK(hipMalloc(&sc, s.Cin * 4)); CK(hipMalloc(&sh, s.Cin * 4));
        CK(hipMalloc(&Y, ny * 4)); CK(hipMalloc(&part, (long)mt * 2 * s.Cout * 4));
        CK(hipMemcpy(Z, hz.data(), nz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(sc, hsc.data(), s.Cin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hsh.data(), s.Cin * 4, hipMemcpyHostToDevice));
        const double flop = 2.0 * s.R * s.Cin * s.Cout;
        const float t0 = run<false, false>(s.R, s.Cin, s.Cout, Z, W, sc, sh, Y, part, 20);
        const float t1 = run<true, false>(s.R, s.Cin, s.Cout, Z, W, sc, sh, Y, part, 20);
        const float t2 = run<true, true>(s.R, s.Cin, s.Cout, Z, W, sc, sh, Y, part, 20);
        const float w0 = run<false, false, 3>(s.R, s.Cin, s.Cout, Z, W, sc, sh, Y, part, 20);
        const float u0 = run<false, false, 2>(s.R, s.Cin, s.Cout, Z, W, sc, sh, Y, part, 20);
        const float u2 = run<true, true, 2>(s.R, s.Cin, s.Cout, Z, W, sc, sh, Y, part, 20);   // last run: checked below
        // check: a few outputs and one column's statistics of the prologue+epilogue run
        std::vector<float> hy(ny), hp((long)mt * 2 * s.Cout);
        CK(hipMemcpy(hy.data(), Y, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 64; ++t) {
            const long m = ((long)t * 2654435761u) % s.R; const int n = (t * 97) % s.Cout;
            double ref = 0;
            for (int k = 0; k < s.Cin; ++k) {
                const float a = fmaxf(fmaf(hz[m * s.Cin + k], hsc[k], hsh[k]), 0.f);
                ref += (double)a * hw[(long)n * s.Cin + k];
            }
            worst = fmax(worst, fabs(ref - hy[m * s.Cout + n]) / (fabs(ref) + 1e-3));
        }
        const int n = 5 % s.Cout;
        double cs = 0, ps = 0;
        for (long m = 0; m < s.R; ++m) cs += hy[m * s.Cout + n];
        for (int i = 0; i < mt; ++i) ps += hp[((long)i * 2) * s.Cout + n];
        const double serr = fabs(cs - ps) / (fabs(cs) + 1e-3);
        printf("%-10s %8d %6d %6d | %9.1f %7.1f | %9.1f %7.1f | %9.1f %7.1f | v2 plain %6.1f %6.1f  fused %6.1f %6.1f  pad12 plain %6.1f %6.1f | out %.1e stats %.1e\n",
               s.name, s.R, s.Cin, s.Cout, t0, flop / t0 / 1e6, t1, flop / t1 / 1e6, t2, flop / t2 / 1e6, u0, flop / u0 / 1e6, u2,
               flop / u2 / 1e6, w0, flop / w0 / 1e6, worst, serr);
        CK(hipFree(Z)); CK(hipFree(W)); CK(hipFree(sc)); CK(hipFree(sh)); CK(hipFree(Y)); CK(hipFree(part));
    }
    return 0;
}
