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
// B=32) and, above all, LATENCY bound (a step is a chain of ~6 dependent launches), so a workgroup
// puts every global load it will ever need in flight at once:
//   * workgroup = 4 waves = one 32-column tile of one K-slice (<= 256 k per chunk); the 4 waves split
//     the slice's K range and reduce their 32x32 accumulators through LDS (fixed order ->
//     deterministic);
//   * the weight fragment of v_mfma_f32_32x32x2_f32 (lane l: B[k = l>>5][n = l&31]) is exactly two
//     full 128-byte lines of a row-major [K][N] matrix, so weights go HBM/L2 -> VGPR directly, all
//     (up to 32) fragments of the wave's K range issued back to back before anything waits;
//   * the activation fragment is one 16-byte load per lane per 8-k block straight from global memory
//     (the tile is tiny and L2-resident): no LDS staging and no barrier in front of the MFMA chain;
//   * split-K partial sums are written as slabs and summed, in slab order, by the consumer kernel's
//     prologue (a launch boundary is cheaper than an in-kernel grid barrier on this chip).
#include "common.h"
#include "kernels.h"
#include "scn_elem.h"

namespace scn {

namespace {

constexpr int KW = 64;        // k extent per wave per chunk
constexpr int NBW = KW / 8;   // 8-k blocks per wave per chunk
constexpr int XLD = 33;       // padded leading dim of the reduction tile

struct SkinnyArgs {
    const float* X; const void* W; float* Y;
    long ldx, xg, ldw, wg, ldy, yg, yslab;
    int rows, N, K, kslice, groups, xvec;
    int* cnt;                 // arrival counters (tail.kind != 0)
    SkinnyTail tail;
};

// The element-wise consumer on one 32-row x 32-column unit (u = unit column tile, rb = first row); thread -> row tid>>3,
// columns 4*(tid&7) .. +3; the arithmetic is csrc/scn_elem.h's cell_unit, shared with the stand-alone kernels.
__device__ __forceinline__ void skinny_tail(const SkinnyArgs& a, int nslab, int u, int ug, int rb) {
    const int tid = threadIdx.x;
    const OwnSlabs own{a.Y, nslab, a.yslab, a.ldy, a.yg};
    cell_unit<8>(a.tail, own, ug, rb + (tid >> 3), u * 32 + (tid & 7) * 4);
}

// Operand mapping of v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][kk = l>>5] and B[kk][n = l&31].
// The sum over k is order-free, so within each 8-k block the four MFMAs pair rows (k0+c, k0+4+c),
// c = 0..3: lane (i, kk) then needs X[i][k0 + 4kk .. k0 + 4kk + 3] -- ONE 16-byte load -- and
// W[k0 + 4kk + c][n] for c = 0..3.  No LDS staging, no barrier before the MFMA chain.
// NB = 8-k blocks per wave per chunk (compile-time so that every loop below is branch-free: a run-time
// bound makes hipcc sink the loads of the optional blocks next to their MFMAs, each behind a vmcnt(0)).
// WBF: the weights are stored as bf16 (2-byte loads, widened in registers); fp32 MFMA and accumulation either way.
// BFM (with WBF): the activation fragment is rounded to bf16 in registers as well and the products run on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation): 4 matrix instructions per 64 k instead of 32 -- the usual mixed-precision
// contract (both operands bf16).  Lane (r = l&31, h = l>>5) holds X[r][k0 + 8h .. +7] and W[k0 + 8h + j][n0 + r].
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int NB, bool XVEC, bool WBF, bool BFM = false>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    __shared__ float red[4][32 * XLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctiles = (a.N + 31) / 32;
    const int ct = blockIdx.x % ctiles, grp = blockIdx.x / ctiles;
    const int slice = blockIdx.y, r0 = blockIdx.z * 32;
    const int n0 = ct * 32;
    const int kbeg = slice * a.kslice;                 // kslice is a multiple of 32
    const int kend = min(a.K, kbeg + a.kslice);
    const int per = a.kslice >> 2;                     // k per wave (multiple of 8)

    const int kk = lane >> 5, l31 = lane & 31;
    const int n = n0 + l31;
    const bool nok = n < a.N;
    const int row = r0 + l31;
    const bool rok = row < a.rows;
    // descriptors are built from wave-uniform values only (kernel args, blockIdx): rows >= kend of W and
    // rows >= a.rows of X are out of range and read as 0; columns / k beyond the tile get OOB_OFF.
    constexpr unsigned WB = WBF ? 2u : 4u;             // bytes per weight element
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(reinterpret_cast<const char*>(a.W) + (long)grp * a.wg * WB,
                                                (unsigned)((long)kend * a.ldw * WB));
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.X + (long)grp * a.xg, (unsigned)((long)a.rows * a.ldx * 4));
    const unsigned wcol = nok ? (unsigned)n * WB : OOB_OFF;
    const unsigned xrow = rok ? (unsigned)((long)row * a.ldx * 4) : OOB_OFF;
    const unsigned ldw4 = (unsigned)a.ldw * WB;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if constexpr (BFM) {      // NB even (host): NB/2 blocks of 16 k
        for (int c0 = 0; c0 < per; c0 += 8 * NB) {
            const int kw0 = kbeg + wave * per + c0;
            unsigned short wr16[NB / 2][8];
            f32x4 xv[NB / 2][2];
#pragma unroll
            for (int j = 0; j < NB / 2; ++j) {
                const int k = kw0 + 16 * j + 8 * kk;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int kq = k + 4 * q;
                    if (XVEC) {
                        xv[j][q] = buf_load4(xr, (rok && kq < kend) ? xrow + (unsigned)kq * 4u : OOB_OFF);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            xv[j][q][c] = buf_load(xr, (rok && kq + c < kend) ? xrow + (unsigned)(kq + c) * 4u : OOB_OFF);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB / 2; ++j) {
                const unsigned k = (unsigned)(kw0 + 16 * j + 8 * kk);
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    wr16[j][c] = __builtin_amdgcn_raw_buffer_load_b16(wr, nok ? (k + c) * ldw4 + wcol : OOB_OFF, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB / 2; ++j) {
                bf16x8_t af, bfv;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    af[c] = (__bf16)xv[j][c >> 2][c & 3];
                    bfv[c] = __builtin_bit_cast(__bf16, wr16[j][c]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfv, acc, 0, 0, 0);
            }
        }
    } else
    for (int c0 = 0; c0 < per; c0 += 8 * NB) {         // per is a multiple of 8*NB (host guarantees it)
        const int kw0 = kbeg + wave * per + c0;        // first k of this wave's chunk
        // every load of the chunk is issued before the first MFMA
        // The activation fragments go FIRST: loads retire in order, so with the (small, L2-resident) X loads at the head of
        // the queue the k-th group of matrix instructions only waits for the k-th group of weight loads
        // (s_waitcnt vmcnt(4 * (NB - 1 - k))) and the ~1 us matrix chain runs under the arrival of the weight stream;
        // with them at the tail the first instruction waited for EVERY weight load (vmcnt(NB - 1)).
        float xf[NB][4], wf[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int k = kw0 + 8 * j + 4 * kk;
            if (XVEC) {
                const f32x4 v = buf_load4(xr, (rok && k < kend) ? xrow + (unsigned)k * 4u : OOB_OFF);
#pragma unroll
                for (int c = 0; c < 4; ++c) xf[j][c] = v[c];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    xf[j][c] = buf_load(xr, (rok && k + c < kend) ? xrow + (unsigned)(k + c) * 4u : OOB_OFF);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const unsigned k = (unsigned)(kw0 + 8 * j + 4 * kk);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned off = nok ? (k + c) * ldw4 + wcol : OOB_OFF;
                wf[j][c] = WBF ? bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(wr, off, 0, 0)) : buf_load(wr, off);
            }
        }
        // keep the scheduler from re-interleaving loads with the MFMA chain (it would hold only ~8 loads
        // in flight to save registers, i.e. pay a memory latency per 8 MFMAs)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j][c], wf[j][c], acc, 0, 0, 0);
    }

    // cross-wave reduction in wave order 0..3 (deterministic)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][mfma32_row(r, lane) * XLD + l31] = acc[r];
    __syncthreads();
    float* Y = a.Y + (long)slice * a.yslab + (long)grp * a.yg;
    if (a.tail.kind == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 5, c = idx & 31;
            if (r0 + r < a.rows && n0 + c < a.N) {
                const float v = ((red[0][r * XLD + c] + red[1][r * XLD + c]) + red[2][r * XLD + c]) + red[3][r * XLD + c];
                Y[(long)(r0 + r) * a.ldy + n0 + c] = v;
            }
        }
        return;
    }
    // ---- fused element-wise consumer (host: N % 4 == 0, ldy % 4 == 0, 16-byte aligned Y): write-through slab stores, every
    // wave drains them, the workgroup meets, one lane takes the unit's ticket; the last arriver acquires and runs the
    // consumer on the unit (cdna_hip_programming.md, in-launch split-K reduction; the same protocol as csrc/cgemm.hip).
    {
        const int r = tid >> 3, c = (tid & 7) * 4;
        if (r0 + r < a.rows && n0 + c < a.N) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = ((red[0][r * XLD + c + q] + red[1][r * XLD + c + q]) + red[2][r * XLD + c + q]) + red[3][r * XLD + c + q];
            const __amdgpu_buffer_rsrc_t yr = make_rsrc(Y, (unsigned)(((long)(a.rows - 1) * a.ldy + a.N) * 4));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, (unsigned)(((long)(r0 + r) * a.ldy + n0 + c) * 4), 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // also: every thread is done reading `red`
    int* const flag = reinterpret_cast<int*>(&red[0][0]);
    // unit = the 32 columns of the consumer's index space this tile feeds; arrivals = tiles x slices that feed it
    int unit = ct, ug = 0, expected = (int)gridDim.y;
    if (a.tail.kind == 1) expected *= 4;                                   // four gate blocks of the same hidden units
    if (a.tail.kind == 3) { const int ft = (a.tail.dim / 4) / 32; ug = grp; unit = ct % ft; expected *= 2; }   // [dmx | dmh] halves
    const int units_per_rb = a.tail.kind == 3 ? 4 * ((a.tail.dim / 4) / 32) : ctiles;
    const int uidx = blockIdx.z * units_per_rb + (a.tail.kind == 3 ? ug * ((a.tail.dim / 4) / 32) + unit : unit);
    if (tid == 0) {
        int* const ticket = a.cnt + uidx;
        const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == expected - 1;
        if (last) {
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    skinny_tail(a, (int)gridDim.y, unit, ug, r0);
}

}  // namespace

int g_dec_tail = 0;   // 1: the cell's element-wise kernels run inside the skinny launches that feed them -- measured SLOWER than their own
                      // launches (fwd 46.4 vs 41.3 us per step, bwd 47.8 vs 44.4: profiles/r03_decode_step_fused_cell_kernels_A_B.txt), kept
                      // as the measured answer to VERDICT r02 item 3.i, bit-identical by test

int skinny_pick_ksplit(int rows, int N, int K, int groups) {
    const int wgs = cdiv(N, 32) * groups * cdiv(rows > 0 ? rows : 1, 32);
    int ks = cdiv(K, 4 * KW);                // one chunk per wave: every load in flight at once
    const int ks_min = cdiv(256, wgs);       // at least one workgroup per CU
    if (ks < ks_min) ks = ks_min;
    const int kmax = K / 32 > 0 ? K / 32 : 1;  // keep >= 32 k per slice (>= 4 MFMAs per wave)
    if (ks > kmax) ks = kmax;
    // A grid just over one workgroup per CU (257 .. 383) makes a few CUs run two workgroups' matrix chains back to back
    // while the rest idle: the kernel then lasts two chains.  Twice the slices, each half as long, end sooner
    // (h -> [att2|gpre|ph], 288 workgroups: 6.1 -> 5.3 us stand-alone).
    if ((long)wgs * ks > 256 && (long)wgs * ks < 384 && ks * 2 <= kmax) ks *= 2;
    if (ks > SCN_MAX_KSPLIT) ks = SCN_MAX_KSPLIT;
    if (ks < 1) ks = 1;
    return ks;
}

int skinny_gemm(hipStream_t st, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                const void* W, long ldw, long wg, float* Y, long ldy, long yg, long yslab, int ksplit, int wbf,
                const SkinnyTail* tail, bool* fused) {
    if (fused) *fused = false;
    if (rows <= 0 || N <= 0 || groups <= 0) return 0;
    SCN_ARG(X && W && Y, "skinny_gemm: null operand");
    SCN_ARG(K >= 1, "skinny_gemm: K must be >= 1");
    SCN_ARG(ksplit >= 1 && ksplit <= SCN_MAX_KSPLIT, "skinny_gemm: ksplit out of range");
    SCN_ARG((long)K * ldw * (wbf ? 2 : 4) < 0x7fffffffL && (long)rows * ldx * 4 < 0x7fffffffL,
            "skinny_gemm: operand exceeds the 2 GiB buffer-descriptor range");
    int per = cdiv(cdiv(K, ksplit), 4);      // k per wave
    per = (per + 7) & ~7;                    // whole 8-k blocks
    if (wbf == 2) per = (per + 15) & ~15;    // bf16 matrix instruction: whole 16-k blocks
    if (per > KW) per = (per + KW - 1) / KW * KW;   // several full chunks
    const int nb = per > KW ? NBW : per / 8;
    const int kslice = 4 * per;
    const int xvec = (aligned16(X) && ldx % 4 == 0 && xg % 4 == 0 && K % 4 == 0 && K >= 4) ? 1 : 0;
    SkinnyArgs a{X, W, Y, ldx, xg, ldw, wg, ldy, yg, yslab, rows, N, K, kslice, groups, xvec, nullptr, SkinnyTail{}};
    dim3 grid(cdiv(N, 32) * groups, ksplit, cdiv(rows, 32)), block(256);
    // the fused consumer: 16-byte rows everywhere it touches, whole 32-column units, one row block per unit of rows
    if (tail && tail->kind >= 1 && tail->kind <= 5 && g_dec_tail && N % 4 == 0 && ldy % 4 == 0 && yg % 4 == 0 && yslab % 4 == 0 &&
        aligned16(Y) && tail->dim % 4 == 0 && (long)grid.x * grid.z <= SPLIT_COUNTERS &&
        ((long)(rows - 1) * ldy + N) * 4 < 0x7fffffffL && (long)ksplit * yslab * 4 < 0x7fffffffL && cdiv(tail->rows, 32) == (int)grid.z) {
        const int want_groups = (tail->kind == 1 || tail->kind == 3) ? 4 : 1;
        const int want_n = tail->kind == 3 ? tail->dim / 2 : tail->dim;
        bool ok = groups == want_groups && N == want_n;
        if (tail->kind == 2 || tail->kind == 3) ok = ok && (tail->dim / 4) % 4 == 0;      // four columns never straddle a gate block
        if (tail->kind == 3) ok = ok && (tail->dim / 4) % 32 == 0 && tail->l0 % 4 == 0;   // [dmx | dmh] halves are whole tiles
        if (tail->kind == 4) ok = ok && tail->l0 % 4 == 0;
        if (tail->kind == 2) ok = ok && tail->sx.p && tail->sx.ld % 4 == 0 && tail->sx.stride % 4 == 0 && aligned16(tail->sx.p);
        for (int i = 0; i < 4; ++i) ok = ok && aligned16(tail->ci[i]) && aligned16(tail->co[i]);
        if (ok) {
            a.cnt = split_counters(st);
            if (a.cnt) {
                a.tail = *tail;
                if (fused) *fused = true;
            }
        }
    }
#define SCN_SKINNY_CASE(NB_)                                                                     \
    case NB_:                                                                                     \
        if (wbf == 2 && (NB_ % 2) == 0) {                                                         \
            if (xvec) hipLaunchKernelGGL((skinny_kernel<(NB_ % 2 ? NB_ + 1 : NB_), true, true, true>), grid, block, 0, st, a);   \
            else      hipLaunchKernelGGL((skinny_kernel<(NB_ % 2 ? NB_ + 1 : NB_), false, true, true>), grid, block, 0, st, a);  \
        } else if (wbf) {                                                                         \
            if (xvec) hipLaunchKernelGGL((skinny_kernel<NB_, true, true>), grid, block, 0, st, a);    \
            else      hipLaunchKernelGGL((skinny_kernel<NB_, false, true>), grid, block, 0, st, a);   \
        } else {                                                                                  \
            if (xvec) hipLaunchKernelGGL((skinny_kernel<NB_, true, false>), grid, block, 0, st, a);   \
            else      hipLaunchKernelGGL((skinny_kernel<NB_, false, false>), grid, block, 0, st, a);  \
        }                                                                                         \
        break;
    switch (nb) {
        SCN_SKINNY_CASE(1) SCN_SKINNY_CASE(2) SCN_SKINNY_CASE(3) SCN_SKINNY_CASE(4)
        SCN_SKINNY_CASE(5) SCN_SKINNY_CASE(6) SCN_SKINNY_CASE(7) SCN_SKINNY_CASE(8)
        default: SCN_ARG(false, "skinny_gemm: internal blocking error");
    }
#undef SCN_SKINNY_CASE
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
