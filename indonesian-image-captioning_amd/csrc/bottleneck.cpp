// Whole-block drivers for the ResNet-152 trunk: ONE C call enqueues every kernel of a Bottleneck's forward pass, ONE
// its backward pass (torchvision's Bottleneck behind the reference's models/encoders/caption.py:17-22), the way
// sequence.cpp does for the decoder's time loop.  The block runs ~15 kernels forward and ~25 backward, each 5-100 us on
// an MI355X; enqueued from Python (ctypes calls, tensor allocations, view objects) the host needs ~220 / ~400 us per
// block and the train step turns host-bound as soon as anything else (data-parallel hooks, a profiler) takes host time.
//
// Data flow (fp32, channels-last maps as [R = N*H*W, C] matrices; all kernels in cgemm.hip / batchnorm.hip):
//   forward   z1 = conv1(x) (+ bn1 statistics epilogue) -> finalize -> a1 = relu(bn1(z1))
//             z2 = conv2_3x3(a1) as an implicit GEMM (+ bn2 statistics epilogue) -> finalize (+ folded scale/shift)
//             z3 = conv3(relu(bn2(z2)) applied ON LOAD) (+ bn3 statistics epilogue) -> finalize
//             [zd = conv_d(x) strided gather (+ statistics) -> finalize -> idn = bn_d(zd)]
//             out = relu(bn3(z3) + identity)
//   backward  bn3: dz3, d identity (-> dx buffer) ; conv3 wgrad (side stream, a2 recomputed on load)
//             conv3 dgrad + mask / reduction pass -> bn2 finalize -> dz2 (in place)
//             conv2 dgrad (implicit GEMM, stride 1) -> bn1 backward (dz1 in place) ; [conv2 wgrad: caller, side stream]
//             conv1 wgrad (side stream) ; W1^T ; dx += dz1 . W1   (accumulated in place, no residual-add kernel)
// The backward driver covers identity blocks (no downsample, stride 1: 44 of the 47 trainable blocks); the three
// strided blocks keep the Python orchestration of scnattn/conv.py.
#include <hip/hip_runtime.h>
#include <mutex>
#include "../../include/scnattn.h"
#include "common.h"
#include "kernels.h"

namespace scn {

namespace {

struct Carve {
    float* base; size_t off = 0;
    explicit Carve(float* b) : base(b) {}
    float* take(size_t n) { float* p = base ? base + off : nullptr; off += (n + 63) & ~size_t(63); return p; }
};

struct Saved { float *z1, *a1, *z2, *z3, *zd, *st1, *st2, *ss2, *st3, *std_; };
struct Scratch { float *dz3, *g2, *da1, *w1t, *gb; };

struct Geo { int Rin, Rout, Ho, Wo, C4; };
inline Geo geo(const scnattn_block& b) {
    Geo g;
    g.Ho = (b.Hi - 1) / b.stride + 1;
    g.Wo = (b.Wi - 1) / b.stride + 1;
    g.Rin = b.N * b.Hi * b.Wi;
    g.Rout = b.N * g.Ho * g.Wo;
    g.C4 = 4 * b.P;
    return g;
}

size_t carve_saved(const scnattn_block& b, float* base, Saved& s) {
    const Geo g = geo(b);
    Carve c(base);
    s.z1 = c.take((size_t)g.Rin * b.P);
    s.a1 = c.take((size_t)g.Rin * b.P);
    s.z2 = c.take((size_t)g.Rout * b.P);
    s.z3 = c.take((size_t)g.Rout * g.C4);
    s.zd = b.has_down ? c.take((size_t)g.Rout * g.C4) : nullptr;
    s.st1 = c.take(2 * (size_t)b.P);
    s.st2 = c.take(2 * (size_t)b.P);
    s.ss2 = c.take(2 * (size_t)b.P);
    s.st3 = c.take(2 * (size_t)g.C4);
    s.std_ = b.has_down ? c.take(2 * (size_t)g.C4) : nullptr;
    return c.off;
}

size_t carve_scratch(const scnattn_block& b, float* base, Scratch& s) {
    const Geo g = geo(b);
    Carve c(base);
    s.dz3 = c.take((size_t)g.Rout * g.C4);
    s.g2 = c.take((size_t)g.Rout * b.P);       // g2 (masked d a2), then dz2 in place
    s.da1 = c.take((size_t)g.Rin * b.P);       // d a1, then dz1 in place
    s.w1t = c.take((size_t)b.Cin * b.P);
    s.gb = c.take(2 * (size_t)(2 * b.P + g.C4));   // {dbeta, dgamma} x (bn1, bn2, bn3)
    return c.off;
}

int check_block(const scnattn_block* b) {
    SCN_ARG(b, "block is NULL");
    SCN_ARG(b->N > 0 && b->Hi > 0 && b->Wi > 0 && b->Cin > 0 && b->P > 0 && b->stride >= 1, "block: geometry");
    SCN_ARG(b->Cin % 16 == 0 && b->P % 16 == 0, "block: channel counts must be multiples of 16");
    SCN_ARG(b->w1 && b->g1 && b->b1 && b->w2 && b->g2 && b->b2 && b->w3 && b->g3 && b->b3, "block: NULL parameter");
    SCN_ARG(b->rm1 && b->rv1 && b->rm2 && b->rv2 && b->rm3 && b->rv3, "block: NULL running statistics");
    SCN_ARG(!b->has_down || (b->wd && b->gd && b->bd && b->rmd && b->rvd), "block: downsample parameters");
    SCN_ARG(b->has_down || (b->stride == 1 && b->Cin == 4 * b->P), "block: an identity shortcut needs stride 1 and Cin == 4*planes");
    return 0;
}

// fork / join helpers for the side stream: one cached event pair per device is not enough (several forks may be
// pending), so events come from a small ring and are never destroyed while possibly in flight
constexpr int NEV = 64;
std::mutex g_ev_mu;
hipEvent_t g_ev[16][NEV] = {};
int g_ev_next[16] = {};

int next_event(hipEvent_t* out) {
    int dev = 0;
    SCN_HIP(hipGetDevice(&dev));
    SCN_ARG(dev >= 0 && dev < 16, "device index out of range");
    std::lock_guard<std::mutex> lk(g_ev_mu);
    const int i = g_ev_next[dev];
    g_ev_next[dev] = (i + 1) % NEV;
    if (!g_ev[dev][i]) SCN_HIP(hipEventCreateWithFlags(&g_ev[dev][i], hipEventDisableTiming));
    *out = g_ev[dev][i];
    return 0;
}

int fork_to(hipStream_t main, hipStream_t side) {
    hipEvent_t e;
    SCN_TRY(next_event(&e));
    SCN_HIP(hipEventRecord(e, main));
    SCN_HIP(hipStreamWaitEvent(side, e, 0));
    return 0;
}

}  // namespace

}  // namespace scn

using namespace scn;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int scnattn_block_sizes(const scnattn_block* b, size_t* saved_floats, size_t* scratch_floats, long offsets[8]) {
    SCN_TRY(check_block(b));
    Saved s;
    Scratch k;
    const size_t sv = carve_saved(*b, nullptr, s);
    const size_t sc = carve_scratch(*b, nullptr, k);
    if (saved_floats) *saved_floats = sv;
    if (scratch_floats) *scratch_floats = sc;
    if (offsets) {   // what the caller needs views of: a1 (saved), dz2 (scratch) for the 3x3 weight gradient
        Saved s2;
        Scratch k2;
        float* const z = reinterpret_cast<float*>(uintptr_t(1024));
        carve_saved(*b, z, s2);
        carve_scratch(*b, z, k2);
        offsets[0] = s2.a1 - z;
        offsets[1] = k2.g2 - z;
        offsets[2] = s2.z2 - z;
        offsets[3] = s2.z1 - z;
        offsets[4] = s2.z3 - z;
        offsets[5] = k2.gb - z;
        offsets[6] = k2.dz3 - z;
        offsets[7] = k2.da1 - z;
    }
    return 0;
}

int scnattn_block_fwd(void* stream, const scnattn_block* b, const float* x, float* saved, float* out, float* ws,
                      long ws_floats, float* part, float* bnpart) {
    SCN_TRY(check_block(b));
    SCN_ARG(x && saved && out && ws && part, "block_fwd: NULL argument");
    hipStream_t st = ST(stream);
    const Geo g = geo(*b);
    Saved s;
    carve_saved(*b, saved, s);
    const int P = b->P, C4 = g.C4, Cin = b->Cin;
    const int nchunk_in = cgemm_row_tiles(g.Rin), nchunk_out = cgemm_row_tiles(g.Rout);
    // conv1 (+ bn1 statistics) -> a1
    {
        ConvExtra e; e.epi = 1; e.stat_partial = part; e.stat_shift = b->rm1;
        SCN_TRY(cgemm(st, false, true, g.Rin, P, Cin, 1.f, x, Cin, b->w1, Cin, 0.f, s.z1, P, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, &e));
        SCN_TRY(bn_finalize(st, g.Rin, P, nchunk_in, part, b->rm1, b->eps1, b->mom1, s.st1, s.st1 + P, b->rm1, b->rv1, nullptr, nullptr, nullptr));
        SCN_TRY(bn_apply(st, g.Rin, P, s.z1, nullptr, 0, s.st1, s.st1 + P, b->g1, b->b1, 1, s.a1));
    }
    // conv2 3x3 as an implicit GEMM (+ bn2 statistics, folded scale/shift for conv3's prologue)
    {
        ConvExtra e; e.epi = 1; e.stat_partial = part; e.stat_shift = b->rm2;
        e.c3 = 1; e.c3c = P; e.c3_src_rows = g.Rin; e.Hi = b->Hi; e.Wi = b->Wi; e.Ho = g.Ho; e.Wo = g.Wo; e.stride = b->stride;
        SCN_TRY(cgemm(st, false, true, g.Rout, P, 9 * P, 1.f, s.a1, P, b->w2, 9L * P, 0.f, s.z2, P, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, &e));
        SCN_TRY(bn_finalize(st, g.Rout, P, nchunk_out, part, b->rm2, b->eps2, b->mom2, s.st2, s.st2 + P, b->rm2, b->rv2, b->g2, b->b2, s.ss2));
    }
    // conv3 with the bn2 + relu prologue (+ bn3 statistics)
    {
        ConvExtra e; e.pro = 1; e.pro_ss = s.ss2; e.epi = 1; e.stat_partial = part; e.stat_shift = b->rm3;
        SCN_TRY(cgemm(st, false, true, g.Rout, C4, P, 1.f, s.z2, P, b->w3, P, 0.f, s.z3, C4, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, &e));
        SCN_TRY(bn_finalize(st, g.Rout, C4, nchunk_out, part, b->rm3, b->eps3, b->mom3, s.st3, s.st3 + C4, b->rm3, b->rv3, nullptr, nullptr, nullptr));
    }
    const float* idn = x;
    if (b->has_down) {
        ConvExtra e; e.epi = 1; e.stat_partial = part; e.stat_shift = b->rmd;
        e.stride = b->stride; e.Hi = b->Hi; e.Wi = b->Wi; e.Ho = g.Ho; e.Wo = g.Wo;
        SCN_TRY(cgemm(st, false, true, g.Rout, C4, Cin, 1.f, x, Cin, b->wd, Cin, 0.f, s.zd, C4, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, &e));
        SCN_TRY(bn_finalize(st, g.Rout, C4, nchunk_out, part, b->rmd, b->epsd, b->momd, s.std_, s.std_ + C4, b->rmd, b->rvd, nullptr, nullptr, nullptr));
        // idn = bn_d(zd), written over zd's own statistics input?  no: zd is needed by the backward pass -> out holds idn
        // first and bn3's apply then adds it in place (res == y is an element-wise read-modify-write)
        SCN_TRY(bn_apply(st, g.Rout, C4, s.zd, nullptr, 0, s.std_, s.std_ + C4, b->gd, b->bd, 0, out));
        idn = out;
    }
    SCN_TRY(bn_apply(st, g.Rout, C4, s.z3, idn, 0, s.st3, s.st3 + C4, b->g3, b->b3, 1, out));
    (void)bnpart;
    return 0;
}

int scnattn_block_bwd(void* stream, void* side_stream, const scnattn_block* b, const float* x, const float* saved,
                      const float* out, const float* dout, float* scratch, float* dx, const scnattn_block_grads* gr,
                      float* ws, float* ws_side, long ws_floats, float* part, float* bnpart, int phase) {
    SCN_TRY(check_block(b));
    SCN_ARG(!b->has_down && b->stride == 1, "block_bwd: identity blocks only (no downsample, stride 1)");
    SCN_ARG(x && saved && out && dout && scratch && gr && ws && part && bnpart, "block_bwd: NULL argument");
    SCN_ARG(!gr->dw1 && !gr->dw3 ? true : (ws_side != nullptr || side_stream == nullptr), "block_bwd: side workspace");
    hipStream_t st = ST(stream), sd = side_stream ? ST(side_stream) : st;
    float* wsd = side_stream ? ws_side : ws;
    const Geo g = geo(*b);
    Saved s;
    Scratch k;
    carve_saved(*b, const_cast<float*>(saved), s);
    carve_scratch(*b, scratch, k);
    const int P = b->P, C4 = g.C4, Cin = b->Cin, R = g.Rout;
    float *dgb1 = k.gb, *dgb2 = k.gb + 2 * P, *dgb3 = k.gb + 4 * P;     // each {dbeta [C], dgamma [C]}
    if (phase != 2) {
    // bn3 (+ identity + relu): dz3 and the identity branch's gradient, straight into the dx buffer
    SCN_TRY(bn_bwd(st, R, C4, dout, out, s.z3, 0, s.st3, s.st3 + C4, b->g3, nullptr, 1, 1, bnpart, dgb3, dgb3 + C4, k.dz3, dx));
    // conv3 weight gradient on the side stream, a2 = relu(bn2(z2)) recomputed on load
    if (gr->dw3) {
        if (side_stream) SCN_TRY(fork_to(st, sd));
        ConvExtra e; e.pro = 2; e.pro_ss = s.ss2;
        SCN_TRY(cgemm(sd, true, false, C4, P, R, 1.f, k.dz3, C4, s.z2, P, 0.f, gr->dw3, P, nullptr, nullptr, 1, 0, 0, 0, wsd, ws_floats, &e));
    }
    // conv3 d input + ReLU mask + the two bn2 reductions, then bn2's element-wise half in place
    {
        ConvExtra e; e.epi = 2; e.stat_partial = part; e.ez = s.z2; e.emean = s.st2; e.einvstd = s.st2 + P;
        e.egamma = b->g2; e.ebeta = b->b2; e.ldz = P; e.pro_ss = s.ss2;
        SCN_TRY(cgemm(st, false, false, R, P, C4, 1.f, k.dz3, C4, b->w3, P, 0.f, k.g2, P, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, &e));
        SCN_TRY(bn_bwd_finalize(st, P, cgemm_row_tiles(R), part, dgb2, dgb2 + P));
        SCN_TRY(bn_bwd_dx(st, R, P, k.g2, s.z2, s.st2, s.st2 + P, b->g2, dgb2, dgb2 + P, k.g2));
    }
    }   // phase != 2
    if (phase == 1) return 0;      // dz2 is ready: the caller forks conv2's weight gradient here, then calls phase 2
    // conv2 d input (implicit GEMM, stride 1); its weight gradient is the caller's (side stream) -- dz2 = k.g2, a1 saved
    {
        ConvExtra e; e.c3 = 2; e.c3c = P; e.c3_src_rows = R; e.Hi = b->Hi; e.Wi = b->Wi; e.Ho = b->Hi; e.Wo = b->Wi; e.stride = 1;
        SCN_TRY(cgemm(st, false, false, R, P, 9 * P, 1.f, k.g2, P, b->w2, 9L * P, 0.f, k.da1, P, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, &e));
    }
    // bn1 (+ relu, mask recomputed from z1), dz1 in place over d a1
    SCN_TRY(bn_bwd(st, R, P, k.da1, nullptr, s.z1, 0, s.st1, s.st1 + P, b->g1, b->b1, 1, 1, bnpart, dgb1, dgb1 + P, k.da1, nullptr));
    if (gr->dw1) {
        if (side_stream) SCN_TRY(fork_to(st, sd));
        SCN_TRY(cgemm(sd, true, false, P, Cin, R, 1.f, k.da1, P, x, Cin, 0.f, gr->dw1, Cin, nullptr, nullptr, 1, 0, 0, 0, wsd, ws_floats, nullptr));
    }
    if (dx) {   // dx (holds d identity) += dz1 . W1, with W1 transposed so that both operands are k-contiguous
        SCN_TRY(transpose2d(st, P, Cin, b->w1, Cin, k.w1t, P));
        SCN_TRY(cgemm(st, false, true, R, Cin, P, 1.f, k.da1, P, k.w1t, P, 1.f, dx, Cin, nullptr, nullptr, 1, 0, 0, 0, ws, ws_floats, nullptr));
    }
    return 0;
}

}  // extern "C"
