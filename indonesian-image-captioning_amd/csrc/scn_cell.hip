// Element-wise halves of the SCN LSTM step (models/scn_cell.py:73-91 and :134-152) and their
// gradients.  The contractions around them are skinny_gemm launches; these kernels are the
// prologues/epilogues: they sum the split-K slabs in slab order, apply the tag projections
// (the Wb / Hb factors), the biases and the gate non-linearities.
//
// Layouts (b = batch row, g = gate block in the reference's order i,f,o,c):
//   pa, ph, qx, qh : [b][4F]        gate g occupies columns [gF, (g+1)F)   (utils/tensor.py:37-42)
//   xcat           : [b][g][2F]     = [ (u.Wa_g) * (s.Wb_g) | (h.Ha_g) * (s.Hb_g) ]
//   r slabs        : [slab][g][b][H]
//   gates          : [b][4H]        post-activation i, f, o, c~
#include "common.h"
#include "kernels.h"
#include "scn_elem.h"

namespace scn {

namespace {

__global__ __launch_bounds__(256) void scn_mix_fwd_kernel(int rows, int F4, Slabs pz, const float* __restrict__ ex,
                                                          Slabs ph, const float* __restrict__ qx,
                                                          const float* __restrict__ qh, float* __restrict__ pa,
                                                          float* __restrict__ phs, float* __restrict__ xcat) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * F4) return;
    const int b = (int)(i / F4), c = (int)(i - (long)b * F4);
    const int F = F4 / 4, g = c / F, f = c - g * F;
    float a = ex ? ex[i] : 0.f;
    if (pz.p) a += slab_sum(pz.p, (long)b * pz.ld + c, pz.n, pz.stride);
    const float h = slab_sum(ph.p, (long)b * ph.ld + c, ph.n, ph.stride);
    pa[i] = a;
    phs[i] = h;
    float* xc = xcat + ((long)b * 4 + g) * 2 * F;
    xc[f] = a * qx[i];
    xc[F + f] = h * qh[i];
}

__global__ __launch_bounds__(256) void lstm_fwd_kernel(int rows, int H, Slabs r, long r_g, const float* __restrict__ bih,
                                                       const float* __restrict__ bhh, const float* __restrict__ c_prev,
                                                       float* __restrict__ gates, float* __restrict__ c_new,
                                                       float* __restrict__ h_new, float* __restrict__ tanhc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * H) return;
    const int b = (int)(i / H), j = (int)(i - (long)b * H);
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = slab_sum(r.p, (long)g * r_g + (long)b * r.ld + j, r.n, r.stride);
        if (bih) v += bih[g * H + j];
        if (bhh) v += bhh[g * H + j];
        pre[g] = v;
    }
    const LstmFwdOut o = lstm_fwd_math(pre[0], pre[1], pre[2], pre[3], c_prev[i]);
    float* gp = gates + (long)b * 4 * H;
    gp[j] = o.ig; gp[H + j] = o.fg; gp[2 * H + j] = o.og; gp[3 * H + j] = o.cg;
    c_new[i] = o.c;
    h_new[i] = o.h;
    if (tanhc) tanhc[i] = o.tc;
}

__global__ __launch_bounds__(256) void lstm_bwd_kernel(int rows, int rows_next, int H, const float* __restrict__ dh_fc,
                                                       Slabs dh_next, float* __restrict__ dc,
                                                       const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                       const float* __restrict__ tanhc, float* __restrict__ dr) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * H) return;
    const int b = (int)(i / H), j = (int)(i - (long)b * H);
    float dh = dh_fc ? dh_fc[i] : 0.f;
    float dcn = 0.f;
    if (b < rows_next) {   // rows that were still decoding at t+1 carry gradient back through (h, c)
        if (dh_next.p) dh += slab_sum(dh_next.p, (long)b * dh_next.ld + j, dh_next.n, dh_next.stride);
        dcn = dc[i];
    }
    const float* gp = gates + (long)b * 4 * H;
    const float ig = gp[j], fg = gp[H + j], og = gp[2 * H + j], cg = gp[3 * H + j];
    const LstmBwdOut o = lstm_bwd_math(dh, dcn, ig, fg, og, cg, tanhc[i], c_prev[i]);
    float* drp = dr + (long)b * 4 * H;
    drp[j] = o.d0;
    drp[H + j] = o.d1;
    drp[2 * H + j] = o.d2;
    drp[3 * H + j] = o.d3;
    dc[i] = o.dc;
}

__global__ __launch_bounds__(256) void scn_mix_bwd_kernel(int rows, int F4, Slabs dx, long dx_g, const float* __restrict__ qx,
                                                          const float* __restrict__ qh, const float* __restrict__ pa,
                                                          const float* __restrict__ phs, float* __restrict__ dpx,
                                                          float* __restrict__ dph, long dph_ld,
                                                          float* __restrict__ dqx_acc, float* __restrict__ dqh_acc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * F4) return;
    const int b = (int)(i / F4), c = (int)(i - (long)b * F4);
    const int F = F4 / 4, g = c / F, f = c - g * F;
    const long base = (long)g * dx_g + (long)b * dx.ld;
    const float dmx = slab_sum(dx.p, base + f, dx.n, dx.stride);
    const float dmh = slab_sum(dx.p, base + F + f, dx.n, dx.stride);
    float o1, o2, ax = dqx_acc[i], ah = dqh_acc[i];
    mix_bwd_math(dmx, dmh, qx[i], qh[i], pa[i], phs[i], o1, o2, ax, ah);
    dpx[i] = o1;
    dph[(long)b * dph_ld + c] = o2;
    dqx_acc[i] = ax;
    dqh_acc[i] = ah;
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(int rows, int E, Slabs dz, const float* __restrict__ awe,
                                                       const float* __restrict__ gate, float* __restrict__ dawe,
                                                       float* __restrict__ dgpre, long dgpre_ld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * E) return;
    const int b = (int)(i / E), c = (int)(i - (long)b * E);
    const float d = slab_sum(dz.p, (long)b * dz.ld + c, dz.n, dz.stride);
    float o1, o2;
    gate_bwd_math(d, awe[i], gate[i], o1, o2);
    dawe[i] = o1;
    dgpre[(long)b * dgpre_ld + c] = o2;
}

}  // namespace

int scn_mix_fwd(hipStream_t st, int rows, int F4, Slabs pz, const float* ex, Slabs ph, const float* qx,
                const float* qh, float* pa, float* phs, float* xcat) {
    if (rows <= 0) return 0;
    SCN_ARG(F4 > 0 && F4 % 4 == 0 && ph.p && qx && qh && pa && phs && xcat, "scn_mix_fwd: bad argument");
    hipLaunchKernelGGL(scn_mix_fwd_kernel, dim3(cdiv((long)rows * F4, 256)), dim3(256), 0, st, rows, F4, pz, ex, ph,
                       qx, qh, pa, phs, xcat);
    SCN_LAUNCH_CHECK();
    return 0;
}

int lstm_fwd(hipStream_t st, int rows, int H, Slabs r, long r_g, const float* bih, const float* bhh,
             const float* c_prev, float* gates, float* c_new, float* h_new, float* tanhc) {
    if (rows <= 0) return 0;
    SCN_ARG(H > 0 && r.p && c_prev && gates && c_new && h_new, "lstm_fwd: bad argument");
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(cdiv((long)rows * H, 256)), dim3(256), 0, st, rows, H, r, r_g, bih, bhh,
                       c_prev, gates, c_new, h_new, tanhc);
    SCN_LAUNCH_CHECK();
    return 0;
}

int lstm_bwd(hipStream_t st, int rows, int rows_next, int H, const float* dh_fc, Slabs dh_next,
             float* dc, const float* gates, const float* c_prev, const float* tanhc, float* dr) {
    if (rows <= 0) return 0;
    SCN_ARG(H > 0 && dc && gates && c_prev && tanhc && dr, "lstm_bwd: bad argument");
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(cdiv((long)rows * H, 256)), dim3(256), 0, st, rows, rows_next, H, dh_fc,
                       dh_next, dc, gates, c_prev, tanhc, dr);
    SCN_LAUNCH_CHECK();
    return 0;
}

int scn_mix_bwd(hipStream_t st, int rows, int F4, Slabs dxcat, long dxcat_g, const float* qx,
                const float* qh, const float* pa, const float* phs, float* dpx, float* dph, long dph_ld,
                float* dqx_acc, float* dqh_acc) {
    if (rows <= 0) return 0;
    SCN_ARG(F4 > 0 && F4 % 4 == 0 && dxcat.p && qx && qh && pa && phs && dpx && dph && dqx_acc && dqh_acc,
            "scn_mix_bwd: bad argument");
    hipLaunchKernelGGL(scn_mix_bwd_kernel, dim3(cdiv((long)rows * F4, 256)), dim3(256), 0, st, rows, F4, dxcat,
                       dxcat_g, qx, qh, pa, phs, dpx, dph, dph_ld, dqx_acc, dqh_acc);
    SCN_LAUNCH_CHECK();
    return 0;
}

int gate_bwd(hipStream_t st, int rows, int E, Slabs dz, const float* awe, const float* gate,
             float* dawe, float* dgpre, long dgpre_ld) {
    if (rows <= 0) return 0;
    SCN_ARG(E > 0 && dz.p && awe && gate && dawe && dgpre, "gate_bwd: bad argument");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(cdiv((long)rows * E, 256)), dim3(256), 0, st, rows, E, dz, awe, gate,
                       dawe, dgpre, dgpre_ld);
    SCN_LAUNCH_CHECK();
    return 0;
}

}  // namespace scn
