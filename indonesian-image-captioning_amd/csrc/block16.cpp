// One mixed-precision Bottleneck per call and direction: the launch sequences of scnattn/conv16.py (which remains the
// readable definition and the fallback) enqueued from C, because the bf16 step is bound by the host's cost per launch.
// Reference: torchvision's Bottleneck inside the trunk built at models/encoders/caption.py:17-22.  Every kernel is reached
// through the library's own C entry points (include/scnattn.h); nothing is allocated here.
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/scnattn.h"
#include "common.h"
#include "kernels.h"

namespace {

struct Carve {
    long Rin, Rout, C4;
    long z1, a1, z2, a2, z3, out, zd, idn, save_elems;          // save offsets (bf16 elements)
    long dres, dz3, dz2, dz1, dzd, dx, dxd, tmp_elems;           // tmp offsets
};

int carve(const scnattn_block16* b, Carve& c) {
    SCN_ARG(b && b->N > 0 && b->Cin > 0 && b->Hi > 0 && b->Wi > 0 && b->p > 0 && (b->stride == 1 || b->stride == 2),
            "block16: geometry");
    SCN_ARG(b->p % 64 == 0 && b->Cin % 64 == 0, "block16: channel counts must be multiples of 64");
    SCN_ARG(b->stride == 1 || (b->Hi % 2 == 0 && b->Wi % 2 == 0), "block16: a stride-2 block needs even maps");
    SCN_ARG(b->has_down || (b->stride == 1 && b->Cin == 4 * b->p), "block16: an identity block keeps its shape");
    const long Ho = (b->Hi - 1) / b->stride + 1, Wo = (b->Wi - 1) / b->stride + 1;
    c.Rin = (long)b->N * b->Hi * b->Wi; c.Rout = (long)b->N * Ho * Wo; c.C4 = 4L * b->p;
    long o = 0;
    auto take = [&](long n) { const long at = o; o += (n + 7) & ~7L; return at; };      // 16-byte aligned pieces
    c.z1 = take(c.Rin * b->p); c.a1 = take(c.Rin * b->p); c.z2 = take(c.Rout * b->p); c.a2 = take(c.Rout * b->p);
    c.z3 = take(c.Rout * c.C4); c.out = take(c.Rout * c.C4);
    c.zd = c.idn = -1;
    if (b->has_down) { c.zd = take(c.Rout * c.C4); c.idn = take(c.Rout * c.C4); }
    c.save_elems = o;
    o = 0;
    c.dres = take(c.Rout * c.C4); c.dz3 = take(c.Rout * c.C4); c.dz2 = take(c.Rout * b->p); c.dz1 = take(c.Rin * b->p);
    c.dzd = c.dx = c.dxd = -1;
    if (b->has_down) { c.dzd = take(c.Rout * c.C4); c.dx = take(c.Rin * b->Cin); c.dxd = take(c.Rout * b->Cin); }
    c.tmp_elems = o;
    return 0;
}

inline char* at16(void* base, long elems) { return static_cast<char*>(base) + 2 * elems; }
inline const char* at16(const void* base, long elems) { return static_cast<const char*>(base) + 2 * elems; }

// a small ring of events for the forks onto the weight-gradient stream (re-recording an event a stream already waits
// for is well defined: the wait was bound to the earlier record)
hipEvent_t fork_event() {
    static std::mutex mu;
    static hipEvent_t ring[64];
    static int n = 0, next = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (n < 64) {
        if (hipEventCreateWithFlags(&ring[n], hipEventDisableTiming) != hipSuccess) return nullptr;
        return ring[n++];
    }
    next = (next + 1) & 63;
    return ring[next];
}

int fork_to(void* side, void* main_stream) {
    hipEvent_t e = fork_event();
    SCN_ARG(e, "block16: event");
    SCN_HIP(hipEventRecord(e, static_cast<hipStream_t>(main_stream)));
    SCN_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(side), e, 0));
    return 0;
}

}  // namespace

extern "C" {

int scnattn_block16_sizes(const scnattn_block16* b, long* save_elems, long* out_offset, long* stats_floats, long* tmp_elems,
                          long* dgb_floats) {
    Carve c;
    SCN_TRY(carve(b, c));
    if (save_elems) *save_elems = c.save_elems;
    if (out_offset) *out_offset = c.out;
    if (stats_floats) *stats_floats = 8 * c.C4;
    if (tmp_elems) *tmp_elems = c.tmp_elems;
    if (dgb_floats) *dgb_floats = 8 * c.C4;
    return 0;
}

int scnattn_block16_fwd(void* st, const scnattn_block16* b) {
    Carve c;
    SCN_TRY(carve(b, c));
    SCN_ARG(b->x && b->save && b->stats && b->ws && b->part && b->bnpart, "block16_fwd: null buffer");
    const int p = b->p, C4 = (int)c.C4, Cin = b->Cin, s = b->stride;
    const int Ho = (b->Hi - 1) / s + 1, Wo = (b->Wi - 1) / s + 1;
    const int Rin = (int)c.Rin, Rout = (int)c.Rout;
    auto S = [&](long off) { return static_cast<void*>(at16(b->save, off)); };
    auto mean = [&](int i) { return b->stats + (2L * i) * C4; };
    auto istd = [&](int i) { return b->stats + (2L * i + 1) * C4; };
    auto apply = [&](int i, long R, int Cn, long z, const void* res, const float* part, int relu, long y) -> int {
        return scnattn_bn_apply_fin(st, R, Cn, S(z), res, 1, part, scnattn_cgemm_stat_ld((int)R), scnattn_cgemm_row_tiles((int)R),
                                    b->shift[i], b->eps[i], b->momentum[i], b->gamma[i], b->beta[i], relu, S(y), mean(i), istd(i),
                                    b->run_mean[i], b->run_var[i], nullptr);
    };
    scnattn_conv_extra ex{};
    ex.epi = 1; ex.stat_partial = b->part;
    // conv1 (+ bn1 statistics from the fp32 accumulators) -> bn1 + relu
    ex.stat_shift = b->shift[0];
    SCN_TRY(scnattn_cgemm16(st, Rin, p, Cin, b->x, Cin, b->w[0], Cin, 0.f, S(c.z1), p, 1, b->ws, b->ws_floats, &ex));
    SCN_TRY(apply(0, Rin, p, c.z1, nullptr, b->part, 1, c.a1));
    // conv2 3x3 (strided for layerN.0) -> bn2 + relu
    ex.stat_shift = b->shift[1];
    SCN_TRY(scnattn_conv3x3_fwd16(st, b->N, b->Hi, b->Wi, p, p, s, S(c.a1), b->w[1], S(c.z2), &ex, b->ws, b->ws_floats));
    SCN_TRY(apply(1, Rout, p, c.z2, nullptr, b->part, 1, c.a2));
    // conv3
    ex.stat_shift = b->shift[2];
    SCN_TRY(scnattn_cgemm16(st, Rout, C4, p, S(c.a2), p, b->w[2], p, 0.f, S(c.z3), C4, 1, b->ws, b->ws_floats, &ex));
    const void* idn = b->x;
    if (b->has_down) {
        scnattn_conv_extra ed{};
        ed.epi = 1; ed.stat_partial = b->bnpart; ed.stat_shift = b->shift[3];
        ed.stride = s; ed.Hi = b->Hi; ed.Wi = b->Wi; ed.Ho = Ho; ed.Wo = Wo;
        SCN_TRY(scnattn_cgemm16(st, Rout, C4, Cin, b->x, Cin, b->w[3], Cin, 0.f, S(c.zd), C4, 1, b->ws, b->ws_floats, &ed));
        SCN_TRY(apply(3, Rout, C4, c.zd, nullptr, b->bnpart, 0, c.idn));
        idn = S(c.idn);
    }
    SCN_TRY(apply(2, Rout, C4, c.z3, idn, b->part, 1, c.out));
    return 0;
}

int scnattn_block16_bwd(void* st, const scnattn_block16* b, void** dx_out, void** dxd_out) {
    Carve c;
    SCN_TRY(carve(b, c));
    SCN_ARG(b->x && b->save && b->stats && b->ws && b->bnpart && b->dout && b->tmp && b->dgb, "block16_bwd: null buffer");
    const int p = b->p, C4 = (int)c.C4, Cin = b->Cin, s = b->stride;
    const int Ho = (b->Hi - 1) / s + 1, Wo = (b->Wi - 1) / s + 1;
    const int Rin = (int)c.Rin, Rout = (int)c.Rout;
    auto S = [&](long off) { return static_cast<void*>(at16(b->save, off)); };
    auto T = [&](long off) { return static_cast<void*>(at16(b->tmp, off)); };
    auto mean = [&](int i) { return b->stats + (2L * i) * C4; };
    auto istd = [&](int i) { return b->stats + (2L * i + 1) * C4; };
    auto dbeta = [&](int i) { return b->dgb + (2L * i) * C4; };
    auto dgamma = [&](int i) { return b->dgb + (2L * i + 1) * C4; };
    const int cap = (int)(b->bnpart_floats / (2L * C4));       // chunk capacity of the backward partials at the widest map
    // weight gradients: their own stream when the caller gave one, forked once the operands exist
    auto wstream = [&](void** ws_out, long* wsn) -> void* {
        if (b->side_stream) { *ws_out = b->side_ws; *wsn = b->side_ws_floats; return b->side_stream; }
        *ws_out = b->ws; *wsn = b->ws_floats; return st;
    };
    auto fork = [&]() -> int { return b->side_stream ? fork_to(b->side_stream, st) : 0; };
    auto reduce = [&](int i, int R, int Cn, const void* dy, const void* y, const void* z, int relu, void* g, int* nch) -> int {
        return scnattn_bn_bwd_reduce(st, R, Cn, dy, y, z, 1, mean(i), istd(i), relu, b->bnpart, (int)(b->bnpart_floats / (2L * Cn)), g, nch);
    };
    auto dxfin = [&](int i, long R, int Cn, const void* g, const void* z, int nch, void* dz) -> int {
        return scnattn_bn_bwd_dx_fin(st, R, Cn, g, z, 1, mean(i), istd(i), b->gamma[i], b->bnpart, (nch + 3) & ~3, nch, dbeta(i),
                                     dgamma(i), dz);
    };
    (void)cap;
    int nch = 0;
    void* wsw = nullptr; long wsn = 0; void* sw = nullptr;
    // ---- bn3 (+ identity + relu): dres = dout * [out > 0] = d identity; dz3 -------------------------------------------------
    SCN_TRY(reduce(2, Rout, C4, b->dout, S(c.out), S(c.z3), 1, T(c.dres), &nch));
    SCN_TRY(dxfin(2, Rout, C4, T(c.dres), S(c.z3), nch, T(c.dz3)));
    if (b->dw[2]) {
        SCN_TRY(fork());
        sw = wstream(&wsw, &wsn);
        SCN_TRY(scnattn_wgrad16_rows(sw, Rout, p, C4, T(c.dz3), S(c.a2), Rout, b->dw[2], p, 0, 0, 0, 0, 0, 0, 0,
                                     static_cast<float*>(wsw), wsn, 0));
    }
    // ---- conv3 d input, bn2 --------------------------------------------------------------------------------------------------
    // the d-input product writes g2 = d a2 * [a2 > 0] and the two sums of bn2's backward itself (mask epilogue): no reduce pass
    auto mask_ex = [&](int i, const void* z, int Cn) {
        scnattn_conv_extra e{};
        e.epi = 2; e.stat_partial = b->bnpart; e.ez = static_cast<const float*>(z); e.ldz = Cn;
        e.emean = mean(i); e.einvstd = istd(i); e.egamma = b->gamma[i]; e.ebeta = b->beta[i];
        return e;
    };
    SCN_ARG(2L * p * scnattn_cgemm_stat_ld(Rin) <= b->bnpart_floats, "block16_bwd: bnpart too small for the mask epilogue's partials");
    {
        const scnattn_conv_extra e2 = mask_ex(1, S(c.z2), p);
        SCN_TRY(scnattn_cgemm16(st, Rout, p, C4, T(c.dz3), C4, b->wt[2], C4, 0.f, T(c.dz2), p, 1, b->ws, b->ws_floats, &e2));
        nch = scnattn_cgemm_row_tiles(Rout);
        SCN_TRY(scnattn_bn_bwd_dx_fin(st, Rout, p, T(c.dz2), S(c.z2), 1, mean(1), istd(1), b->gamma[1], b->bnpart,
                                      scnattn_cgemm_stat_ld(Rout), nch, dbeta(1), dgamma(1), T(c.dz2)));
    }
    // ---- conv2: weight gradient, d input, bn1 --------------------------------------------------------------------------------
    if (b->dw[1]) {
        SCN_TRY(fork());
        sw = wstream(&wsw, &wsn);
        if (s == 1) {
            SCN_TRY(scnattn_wgrad16_3x3(sw, b->N, b->Hi, b->Wi, p, p, T(c.dz2), S(c.a1), b->dw[1], static_cast<float*>(wsw), wsn, 0));
        } else {       // a stride-2 3x3 tap by tap: source pixel (2 ho + dh - 1, 2 wo + dw - 1) gathered per row
            for (int tap = 0; tap < 9; ++tap)
                SCN_TRY(scnattn_wgrad16_rows(sw, Rout, p, p, T(c.dz2), S(c.a1), Rin, b->dw[1] + (long)tap * p, 9L * p, s, b->Hi,
                                             b->Wi, Ho, Wo, tap / 3 - 1, tap % 3 - 1, static_cast<float*>(wsw), wsn, 0));
        }
    }
    if (s == 1) {       // mask epilogue with bn1
        const scnattn_conv_extra e1 = mask_ex(0, S(c.z1), p);
        SCN_TRY(scnattn_conv3x3_dgrad16(st, b->N, b->Hi, b->Wi, p, p, 1, T(c.dz2), b->wt[1], T(c.dz1), &e1, b->ws, b->ws_floats));
        nch = scnattn_cgemm_row_tiles(Rin);
        SCN_TRY(scnattn_bn_bwd_dx_fin(st, Rin, p, T(c.dz1), S(c.z1), 1, mean(0), istd(0), b->gamma[0], b->bnpart,
                                      scnattn_cgemm_stat_ld(Rin), nch, dbeta(0), dgamma(0), T(c.dz1)));
    } else {            // parity classes write scattered rows: the reduce pass stays
        SCN_TRY(scnattn_conv3x3_dgrad16(st, b->N, b->Hi, b->Wi, p, p, s, T(c.dz2), b->wt[1], T(c.dz1), nullptr, b->ws, b->ws_floats));
        SCN_TRY(reduce(0, Rin, p, T(c.dz1), S(c.a1), S(c.z1), 1, T(c.dz1), &nch));
        SCN_TRY(dxfin(0, Rin, p, T(c.dz1), S(c.z1), nch, T(c.dz1)));
    }
    if (b->dw[0]) {
        SCN_TRY(fork());
        sw = wstream(&wsw, &wsn);
        SCN_TRY(scnattn_wgrad16_rows(sw, Rin, Cin, p, T(c.dz1), b->x, Rin, b->dw[0], Cin, 0, 0, 0, 0, 0, 0, 0,
                                     static_cast<float*>(wsw), wsn, 0));
    }
    // ---- identity branch and d x ---------------------------------------------------------------------------------------------
    void* dx = nullptr;
    void* dxd = nullptr;
    if (b->has_down) {
        SCN_TRY(reduce(3, Rout, C4, T(c.dres), nullptr, S(c.zd), 0, nullptr, &nch));
        SCN_TRY(dxfin(3, Rout, C4, T(c.dres), S(c.zd), nch, T(c.dzd)));
        if (b->dw[3]) {
            SCN_TRY(fork());
            sw = wstream(&wsw, &wsn);
            SCN_TRY(scnattn_wgrad16_rows(sw, Rout, Cin, C4, T(c.dzd), b->x, Rin, b->dw[3], Cin, s, b->Hi, b->Wi, Ho, Wo, 0, 0,
                                         static_cast<float*>(wsw), wsn, 0));
        }
        if (b->need_dx) {
            dx = T(c.dx); dxd = T(c.dxd);
            SCN_TRY(scnattn_cgemm16(st, Rin, Cin, p, T(c.dz1), p, b->wt[0], p, 0.f, dx, Cin, 1, b->ws, b->ws_floats, nullptr));
            SCN_TRY(scnattn_cgemm16(st, Rout, Cin, C4, T(c.dzd), C4, b->wt[3], C4, 0.f, dxd, Cin, 1, b->ws, b->ws_floats, nullptr));
        }
    } else if (b->need_dx) {
        dx = T(c.dres);      // d x = d identity + dz1 . W1, accumulated in place (beta = 1)
        SCN_TRY(scnattn_cgemm16(st, Rin, Cin, p, T(c.dz1), p, b->wt[0], p, 1.f, dx, Cin, 1, b->ws, b->ws_floats, nullptr));
    }
    if (dx_out) *dx_out = dx;
    if (dxd_out) *dxd_out = dxd;
    return 0;
}

}  // extern "C"
