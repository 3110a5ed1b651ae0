// The arithmetic of the SCN-LSTM cell's element-wise steps (reference models/scn_cell.py:73-91 and :134-152, the gate of
// models/decoders/attention_scn.py:147-150), ONE definition for the stand-alone kernels (csrc/scn_cell.hip) and for the
// same steps run inside the skinny launches (csrc/skinny.hip, SkinnyTail): floating-point contraction is off inside
// these functions, so both callers round every product and sum at the same place and the two paths agree bit for bit.
#pragma once
#include "common.h"
#include "kernels.h"

namespace scn {

struct LstmFwdOut { float ig, fg, og, cg, c, tc, h; };
__device__ __forceinline__ LstmFwdOut lstm_fwd_math(float p0, float p1, float p2, float p3, float c_prev) {
#pragma clang fp contract(off)
    LstmFwdOut o;
    o.ig = sigmoidf_(p0); o.fg = sigmoidf_(p1); o.og = sigmoidf_(p2); o.cg = tanhf(p3);
    const float a = o.fg * c_prev, b = o.ig * o.cg;
    o.c = a + b;
    o.tc = tanhf(o.c);
    o.h = o.og * o.tc;
    return o;
}

struct LstmBwdOut { float d0, d1, d2, d3, dc; };
__device__ __forceinline__ LstmBwdOut lstm_bwd_math(float dh, float dcn, float ig, float fg, float og, float cg, float tc,
                                                    float c_prev) {
#pragma clang fp contract(off)
    LstmBwdOut o;
    const float dO = dh * tc;
    const float t2 = tc * tc, om = 1.f - t2;
    const float x = dh * og, y = x * om;
    const float dcc = dcn + y;
    o.d0 = ((dcc * cg) * ig) * (1.f - ig);
    o.d1 = ((dcc * c_prev) * fg) * (1.f - fg);
    o.d2 = (dO * og) * (1.f - og);
    const float c2 = cg * cg;
    o.d3 = (dcc * ig) * (1.f - c2);
    o.dc = dcc * fg;
    return o;
}

__device__ __forceinline__ void mix_bwd_math(float dmx, float dmh, float qx, float qh, float pa, float ph, float& dpx,
                                             float& dph, float& acc_x, float& acc_h) {
#pragma clang fp contract(off)
    dpx = dmx * qx;
    dph = dmh * qh;
    const float a = dmx * pa, b = dmh * ph;
    acc_x = acc_x + a;
    acc_h = acc_h + b;
}

__device__ __forceinline__ void gate_bwd_math(float d, float awe, float g, float& dawe, float& dgpre) {
#pragma clang fp contract(off)
    dawe = d * g;
    dgpre = ((d * awe) * g) * (1.f - g);
}


// ---- the cell's element-wise steps on four consecutive columns of one row (16-byte accesses), shared by the stand-alone
// kernels (csrc/scn_cell.hip) and the fused tails (csrc/skinny.hip) ---------------------------------------------------------
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// sum of n slabs `stride` apart at p, slab order; R 16-byte loads in flight per round, clamped to the last slab, no
// branch (a run-time-bounded loop of dependent adds pays one memory latency per slab; a branch per sum keeps the sums of
// the four gates from overlapping)
template <int R>
__device__ __forceinline__ f32x4 slab_sum4(const float* p, int n, long stride) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < n; s0 += R) {
        f32x4 t[R];
#pragma unroll
        for (int u = 0; u < R; ++u) t[u] = ld4(p + (long)min(s0 + u, n - 1) * stride);
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (s0 + u < n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (s0 + u == 0) ? t[u][q] : v[q] + t[u][q];
            }
    }
    return v;
}
struct OwnSlabs { const float* p; int n; long stride, ld, gstride; };     // the product the step consumes: [slab][group][row][col]
template <int R>
__device__ __forceinline__ f32x4 own_sum(const OwnSlabs& o, int g, int b, int c) {
    return slab_sum4<R>(o.p + (long)g * o.gstride + (long)b * o.ld + c, o.n, o.stride);
}

// t.kind (csrc/kernels.h, SkinnyTail) on row b, columns c .. c+3 (kind 3: columns of gate block ug)
template <int R>
__device__ __forceinline__ void cell_unit(const SkinnyTail& t, const OwnSlabs& own, int ug, int b, int c) {
    if (b >= t.rows) return;
    if (t.kind == 1) {            // lstm_fwd: c = hidden unit j
        const int H = t.dim;
        if (c >= H) return;
        f32x4 pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pre[g] = own_sum<R>(own, g, b, c);
            if (t.ci[0]) { const f32x4 v = ld4(t.ci[0] + g * H + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) pre[g][q] += v[q]; }
            if (t.ci[1]) { const f32x4 v = ld4(t.ci[1] + g * H + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) pre[g][q] += v[q]; }
        }
        const long i = (long)b * H + c;
        const f32x4 cp = ld4(t.ci[2] + i);
        f32x4 ig, fg, og, cg, cn, hn, tc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const LstmFwdOut o = lstm_fwd_math(pre[0][q], pre[1][q], pre[2][q], pre[3][q], cp[q]);
            ig[q] = o.ig; fg[q] = o.fg; og[q] = o.og; cg[q] = o.cg; cn[q] = o.c; tc[q] = o.tc; hn[q] = o.h;
        }
        float* gp = t.co[0] + (long)b * 4 * H + c;
        st4(gp, ig); st4(gp + H, fg); st4(gp + 2 * H, og); st4(gp + 3 * H, cg);
        st4(t.co[1] + i, cn);
        st4(t.co[2] + i, hn);
        if (t.co[3]) st4(t.co[3] + i, tc);
    } else if (t.kind == 2) {     // scn_mix_fwd: c = column of [b][4F]
        const int F4 = t.dim, F = F4 / 4;
        if (c >= F4) return;
        const long i = (long)b * F4 + c;
        const int g = c / F, f = c - g * F;
        f32x4 av = {0.f, 0.f, 0.f, 0.f};
        if (t.ci[0]) av = ld4(t.ci[0] + i);
        if (own.p) {
            const f32x4 pz = own_sum<R>(own, 0, b, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] += pz[q];
        }
        const f32x4 hv = slab_sum4<R>(t.sx.p + (long)b * t.sx.ld + c, t.sx.n, t.sx.stride);
        const f32x4 qx = ld4(t.ci[1] + i), qh = ld4(t.ci[2] + i);
        st4(t.co[0] + i, av);
        st4(t.co[1] + i, hv);
        float* xc = t.co[2] + ((long)b * 4 + g) * 2 * F + f;
        f32x4 mx, mh;
#pragma unroll
        for (int q = 0; q < 4; ++q) { mx[q] = av[q] * qx[q]; mh[q] = hv[q] * qh[q]; }
        st4(xc, mx);
        st4(xc + F, mh);
    } else if (t.kind == 3) {     // scn_mix_bwd: unit = gate ug, f = c
        const int F4 = t.dim, F = F4 / 4;
        if (c >= F) return;
        const int cc = ug * F + c;
        const long i = (long)b * F4 + cc;
        const f32x4 dmx = own_sum<R>(own, ug, b, c), dmh = own_sum<R>(own, ug, b, F + c);
        const f32x4 qx = ld4(t.ci[0] + i), qh = ld4(t.ci[1] + i), pa = ld4(t.ci[2] + i), ph = ld4(t.ci[3] + i);
        f32x4 ax = ld4(t.co[2] + i), ah = ld4(t.co[3] + i), o1, o2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float r1, r2, a1 = ax[q], a2 = ah[q];
            mix_bwd_math(dmx[q], dmh[q], qx[q], qh[q], pa[q], ph[q], r1, r2, a1, a2);
            o1[q] = r1; o2[q] = r2; ax[q] = a1; ah[q] = a2;
        }
        st4(t.co[0] + i, o1);
        st4(t.co[1] + (long)b * t.l0 + cc, o2);
        st4(t.co[2] + i, ax);
        st4(t.co[3] + i, ah);
    } else if (t.kind == 4) {     // gate_bwd
        const int E = t.dim;
        if (c >= E) return;
        const long i = (long)b * E + c;
        const f32x4 d = own_sum<R>(own, 0, b, c);
        const f32x4 awe = ld4(t.ci[0] + i), gt = ld4(t.ci[1] + i);
        f32x4 o1, o2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float r1, r2;
            gate_bwd_math(d[q], awe[q], gt[q], r1, r2);
            o1[q] = r1; o2[q] = r2;
        }
        st4(t.co[0] + i, o1);
        st4(t.co[1] + (long)b * t.l0 + c, o2);
    } else if (t.kind == 5) {     // lstm_bwd of the previous step; this launch's slabs are its dh_next
        const int H = t.dim;
        if (c >= H) return;
        const long i = (long)b * H + c;
        f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dcn = dh;
        if (t.ci[0]) dh = ld4(t.ci[0] + i);
        if (b < t.rows_next) {
            const f32x4 v = own_sum<R>(own, 0, b, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) dh[q] += v[q];
            dcn = ld4(t.co[0] + i);
        }
        const float* gp = t.ci[1] + (long)b * 4 * H + c;
        const f32x4 ig = ld4(gp), fg = ld4(gp + H), og = ld4(gp + 2 * H), cg = ld4(gp + 3 * H);
        const f32x4 tc = ld4(t.ci[3] + i), cp = ld4(t.ci[2] + i);
        f32x4 d0, d1, d2, d3, dco;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const LstmBwdOut o = lstm_bwd_math(dh[q], dcn[q], ig[q], fg[q], og[q], cg[q], tc[q], cp[q]);
            d0[q] = o.d0; d1[q] = o.d1; d2[q] = o.d2; d3[q] = o.d3; dco[q] = o.dc;
        }
        float* drp = t.co[1] + (long)b * 4 * H + c;
        st4(drp, d0); st4(drp + H, d1); st4(drp + 2 * H, d2); st4(drp + 3 * H, d3);
        st4(t.co[0] + i, dco);
    }
}


}  // namespace scn
