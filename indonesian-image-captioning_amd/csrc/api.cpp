// extern "C" surface of libscnattn (declared in include/scnattn.h).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/scnattn.h"
#include "common.h"
#include "kernels.h"

namespace scn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

extern int g_ksplit_scale;
extern int g_profile;
extern int g_attn_depth;
extern int g_dec_bf16;
extern int g_gemm_target, g_gemm_kmin, g_gemm_kmin_small, g_gemm_gate, g_use_cgemm, g_cgemm_target, g_cgemm_kmin, g_cgemm_mi, g_cgemm_combine, g_cgemm_combine_max, g_dec_tail;
int profile_collect(double* out);
int seq_workspace(const scnattn_dims* d, const scnattn_pool* pool, size_t* saved_bytes, size_t* scratch_bytes);
int seq_fwd(hipStream_t st, const scnattn_dims* d, const scnattn_params* w, const float* enc, const float* tags,
            const int64_t* caps, const int32_t* dl_dev, const int32_t* bt, const float* drop_mask, float* saved,
            float* scratch, float* preds, float* alphas, const scnattn_pool* pool);
int seq_bwd(hipStream_t st, hipStream_t wst, const scnattn_dims* d, const scnattn_params* w, const float* enc,
            const float* tags, const int64_t* caps, const int32_t* dl_dev, const int32_t* bt, const float* drop_mask,
            const float* saved, float* scratch, const float* dpreds, const float* dalphas, const scnattn_params* g,
            float* denc, float* dtags, const scnattn_pool* pool);

}  // namespace scn

using namespace scn;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int scnattn_version(void) { return SCNATTN_VERSION; }
const char* scnattn_last_error(void) { return last_error(); }

// Process-wide knobs.  Two kinds only (the experiment switches of rounds 1-2 -- fuse_attn, chains,
// attn_handoff, cgemm_stagger, cgemm_w41, cgemm_vec, bn_gfirst, skinny_tail -- are gone with the code paths they selected):
//   * what a CALLER chooses per process: "decoder_bf16" (storage mode of the decode step, BASELINE configs[4]), "profile"
//     (HIP-event timing of the recurrence loops for bench.py);
//   * split-K / tile policy overrides that only tools/ (sweeps) and the tests of both policies set.
int scnattn_set_option(const char* name, int value) {
    struct Opt { const char* name; int* var; int lo, hi; };
    static const Opt opts[] = {
        {"decoder_bf16", &g_dec_bf16, 0, 2},   {"profile", &g_profile, 0, 2},
        {"ksplit", &g_ksplit_scale, 0, SCN_MAX_KSPLIT}, {"attn_depth", &g_attn_depth, 0, 1},
        {"use_cgemm", &g_use_cgemm, 0, 1},     {"cgemm_mi", &g_cgemm_mi, 0, 2},
        {"dec_tail", &g_dec_tail, 0, 1},
        {"cgemm_combine", &g_cgemm_combine, 0, 2}, {"cgemm_combine_max", &g_cgemm_combine_max, 1, 128},
        {"cgemm_target", &g_cgemm_target, 1, 1 << 20}, {"cgemm_kmin", &g_cgemm_kmin, 16, 1 << 20},
        {"gemm_target", &g_gemm_target, 1, 1 << 20},   {"gemm_gate", &g_gemm_gate, 1, 1 << 20},
        {"gemm_kmin", &g_gemm_kmin, 16, 1 << 20},      {"gemm_kmin_small", &g_gemm_kmin_small, 16, 1 << 20},
    };
    for (const Opt& o : opts)
        if (name && std::strcmp(name, o.name) == 0) {
            if (value < o.lo || value > o.hi) { set_error("scnattn_set_option: %s out of range [%d, %d]", o.name, o.lo, o.hi); return -1; }
            *o.var = value;
            return 0;
        }
    set_error("scnattn_set_option: unknown option '%s'", name ? name : "(null)");
    return -1;
}

int scnattn_profile_collect(double* out6) {
    if (!out6) { set_error("scnattn_profile_collect: NULL"); return -1; }
    return profile_collect(out6);
}

int scnattn_seq_workspace(const scnattn_dims* d, const scnattn_pool* pool, size_t* saved_bytes, size_t* scratch_bytes) {
    return seq_workspace(d, pool, saved_bytes, scratch_bytes);
}

int scnattn_seq_fwd(void* stream, const scnattn_dims* d, const scnattn_params* w, const float* enc,
                    const float* tags, const int64_t* caps, const int32_t* dl_dev, const int32_t* bt_host,
                    const float* drop_mask, float* saved, float* scratch, float* preds, float* alphas,
                    const scnattn_pool* pool) {
    return seq_fwd(ST(stream), d, w, enc, tags, caps, dl_dev, bt_host, drop_mask, saved, scratch, preds, alphas, pool);
}

int scnattn_seq_bwd(void* stream, const scnattn_dims* d, const scnattn_params* w, const float* enc,
                    const float* tags, const int64_t* caps, const int32_t* dl_dev, const int32_t* bt_host,
                    const float* drop_mask, const float* saved, float* scratch, const float* dpreds,
                    const float* dalphas, const scnattn_params* g, float* denc, float* dtags,
                    const scnattn_pool* pool) {
    return seq_bwd(ST(stream), nullptr, d, w, enc, tags, caps, dl_dev, bt_host, drop_mask, saved, scratch, dpreds,
                   dalphas, g, denc, dtags, pool);
}

int scnattn_seq_bwd_streams(void* stream, void* wgrad_stream, const scnattn_dims* d, const scnattn_params* w,
                            const float* enc, const float* tags, const int64_t* caps, const int32_t* dl_dev,
                            const int32_t* bt_host, const float* drop_mask, const float* saved, float* scratch,
                            const float* dpreds, const float* dalphas, const scnattn_params* g, float* denc,
                            float* dtags, const scnattn_pool* pool) {
    return seq_bwd(ST(stream), ST(wgrad_stream), d, w, enc, tags, caps, dl_dev, bt_host, drop_mask, saved, scratch,
                   dpreds, dalphas, g, denc, dtags, pool);
}

int scnattn_sgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                  long lda, const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                  const float* rowmask, int batch, long strideA, long strideB, long strideC) {
    return sgemm(ST(stream), transA != 0, transB != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, rowmask,
                 batch < 1 ? 1 : batch, strideA, strideB, strideC);
}

int scnattn_sgemm_ws(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                     long lda, const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                     const float* rowmask, int batch, long strideA, long strideB, long strideC, float* ws,
                     long ws_floats) {
    return sgemm_ws(ST(stream), transA != 0, transB != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, rowmask,
                    batch < 1 ? 1 : batch, strideA, strideB, strideC, ws, ws ? ws_floats : 0);
}

static ConvExtra to_extra(const scnattn_conv_extra* e) {
    ConvExtra x;
    if (!e) return x;
    x.pro = e->pro; x.epi = e->epi; x.pro_ss = e->pro_ss;
    x.stat_partial = e->stat_partial; x.stat_shift = e->stat_shift;
    x.ez = e->ez; x.emean = e->emean; x.einvstd = e->einvstd; x.egamma = e->egamma; x.ebeta = e->ebeta; x.ldz = e->ldz;
    x.stride = e->stride < 1 ? 1 : e->stride; x.Hi = e->Hi; x.Wi = e->Wi; x.Ho = e->Ho; x.Wo = e->Wo;
    x.force_split = e->force_split; x.force_mi = e->force_mi;
    return x;
}

int scnattn_cgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, long lda,
                  const float* B, long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask,
                  int batch, long strideA, long strideB, long strideC, float* ws, long ws_floats,
                  const scnattn_conv_extra* ex) {
    const ConvExtra x = to_extra(ex);
    return cgemm(ST(stream), transA != 0, transB != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, rowmask,
                 batch < 1 ? 1 : batch, strideA, strideB, strideC, ws, ws ? ws_floats : 0, ex ? &x : nullptr);
}

int scnattn_cgemm_row_tiles(int M) { return cgemm_row_tiles(M); }

int scnattn_conv1x1_fwd(void* stream, int R, int Cin, int Cout, const float* x, const float* w, float* y,
                        const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    const ConvExtra e = to_extra(ex);
    SCN_ARG(!ex || ex->pro != 2, "conv1x1_fwd: prologue must be on the input (pro = 1)");
    return cgemm(ST(stream), false, true, R, Cout, Cin, 1.f, x, Cin, w, Cin, 0.f, y, Cout, nullptr, nullptr, 1, 0, 0, 0,
                 ws, ws ? ws_floats : 0, ex ? &e : nullptr);
}

int scnattn_conv1x1_dgrad(void* stream, int R, int Cin, int Cout, const float* dy, const float* w, int w_transposed,
                          float beta, float* dx, const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    const ConvExtra e = to_extra(ex);
    SCN_ARG(!ex || (ex->pro == 0 && ex->stride <= 1), "conv1x1_dgrad: no prologue / stride here");
    if (w_transposed)   // w is [Cin][Cout]: dx = dy . (w^T)^T, both operands k-contiguous
        return cgemm(ST(stream), false, true, R, Cin, Cout, 1.f, dy, Cout, w, Cout, beta, dx, Cin, nullptr, nullptr, 1, 0,
                     0, 0, ws, ws ? ws_floats : 0, ex ? &e : nullptr);
    return cgemm(ST(stream), false, false, R, Cin, Cout, 1.f, dy, Cout, w, Cin, beta, dx, Cin, nullptr, nullptr, 1, 0, 0,
                 0, ws, ws ? ws_floats : 0, ex ? &e : nullptr);
}

int scnattn_conv1x1_wgrad(void* stream, int R, int Cin, int Cout, const float* dy, const float* x, float* dw,
                          const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    ConvExtra e = to_extra(ex);
    SCN_ARG(!ex || (ex->pro != 1 && ex->epi == 0), "conv1x1_wgrad: prologue must be on the input (pro = 2), no epilogue");
    return cgemm(ST(stream), true, false, Cout, Cin, R, 1.f, dy, Cout, x, Cin, 0.f, dw, Cin, nullptr, nullptr, 1, 0, 0, 0,
                 ws, ws ? ws_floats : 0, ex ? &e : nullptr);
}

int scnattn_conv3x3_fwd(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const float* x,
                        const float* w, float* y, const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    SCN_ARG(N > 0 && Hi > 0 && Wi > 0 && stride >= 1, "conv3x3_fwd: geometry");
    ConvExtra e = to_extra(ex);
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    e.c3 = 1; e.c3c = Cin; e.c3_src_rows = (long)N * Hi * Wi; e.Hi = Hi; e.Wi = Wi; e.Ho = Ho; e.Wo = Wo; e.stride = stride;
    return cgemm(ST(stream), false, true, N * Ho * Wo, Cout, 9 * Cin, 1.f, x, Cin, w, 9L * Cin, 0.f, y, Cout, nullptr,
                 nullptr, 1, 0, 0, 0, ws, ws ? ws_floats : 0, &e);
}

int scnattn_conv3x3_dgrad(void* stream, int N, int Hi, int Wi, int Cin, int Cout, const float* dy, const float* w,
                          float* dx, const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    SCN_ARG(N > 0 && Hi > 0 && Wi > 0, "conv3x3_dgrad: geometry");
    ConvExtra e = to_extra(ex);
    // stride 1: dy and dx maps have the same extent; the gathered (source) map is dy
    e.c3 = 2; e.c3c = Cout; e.c3_src_rows = (long)N * Hi * Wi; e.Hi = Hi; e.Wi = Wi; e.Ho = Hi; e.Wo = Wi; e.stride = 1;
    return cgemm(ST(stream), false, false, N * Hi * Wi, Cin, 9 * Cout, 1.f, dy, Cout, w, 9L * Cin, 0.f, dx, Cin, nullptr,
                 nullptr, 1, 0, 0, 0, ws, ws ? ws_floats : 0, &e);
}

int scnattn_conv3x3_dgrad_strided(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const float* dy,
                                  const float* w, float* dx, float* ws, long ws_floats) {
    SCN_ARG(N > 0 && Hi > 0 && Wi > 0 && stride == 2 && Hi % 2 == 0 && Wi % 2 == 0, "conv3x3_dgrad_strided: stride 2, even map");
    ConvExtra e;
    const int Ho = Hi / 2, Wo = Wi / 2;
    // one launch, four parity classes of d-input pixels (grid.y), each a product over the taps that reach it
    e.c3 = 4; e.c3c = Cout; e.c3_src_rows = (long)N * Ho * Wo; e.Hi = Hi; e.Wi = Wi; e.Ho = Ho; e.Wo = Wo; e.stride = 2;
    return cgemm(ST(stream), false, false, N * Ho * Wo, Cin, 9 * Cout, 1.f, dy, Cout, w, 9L * Cin, 0.f, dx, Cin, nullptr,
                 nullptr, 1, 0, 0, 0, ws, ws ? ws_floats : 0, &e);
}

int scnattn_conv3x3_wgrad(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const float* dy,
                          const float* x, float* dw, float* ws, long ws_floats, int k_slices) {
    SCN_ARG(N > 0 && Hi > 0 && Wi > 0 && stride >= 1, "conv3x3_wgrad: geometry");
    // stride 1: the halo-staged kernel (csrc/conv3.hip); strided (layerN.0) or odd widths: the gather form of cgemm
    if (stride == 1 && k_slices >= 0 && conv3x3_wgrad_halo_ok(N, Hi, Wi, Cin, Cout, dy, x, dw))
        return conv3x3_wgrad_halo(ST(stream), N, Hi, Wi, Cin, Cout, dy, x, dw, ws, ws ? ws_floats : 0, k_slices);
    ConvExtra e;
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    e.c3 = 3; e.c3c = Cin; e.c3_src_rows = (long)N * Hi * Wi; e.Hi = Hi; e.Wi = Wi; e.Ho = Ho; e.Wo = Wo; e.stride = stride;
    return cgemm(ST(stream), true, false, Cout, 9 * Cin, N * Ho * Wo, 1.f, dy, Cout, x, Cin, 0.f, dw, 9L * Cin, nullptr,
                 nullptr, 1, 0, 0, 0, ws, ws ? ws_floats : 0, &e);
}

int scnattn_stem_tiles(int N, int H, int W) { return stem_tiles(N, H, W); }

int scnattn_stem_conv7(void* stream, int N, int H, int W, const float* x, long sn, long sc, long sh, long sw,
                       const float* w, long wn, long wc, long wh, long ww, float* z, float* stat_partial,
                       const float* stat_shift) {
    return stem_conv7(ST(stream), N, H, W, x, sn, sc, sh, sw, w, wn, wc, wh, ww, z, stat_partial, stat_shift);
}

int scnattn_stem_bn_relu_maxpool(void* stream, int N, int Hz, int Wz, int C, const float* z, const float* ss, void* out,
                                 int out_bf16) {
    return stem_bn_relu_maxpool(ST(stream), N, Hz, Wz, C, z, ss, out, out_bf16);
}

int scnattn_cgemm_stat_ld(int M) { return cgemm_stat_ld(M); }

int scnattn_bn_finalize(void* stream, long R, int C, const float* partial, int ldp, int nchunk, const float* shift,
                        float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var,
                        const float* gamma, const float* beta, float* ss_out) {
    return bn_finalize_t(ST(stream), R, C, partial, ldp, nchunk, shift, eps, momentum, mean, invstd, run_mean, run_var, gamma,
                         beta, ss_out);
}

int scnattn_bn_apply_fin(void* stream, long R, int C, const void* z, const void* res, int bf16, const float* partial, int ldp,
                         int nchunk, const float* shift, float eps, float momentum, const float* gamma, const float* beta,
                         int relu, void* y, float* mean, float* invstd, float* run_mean, float* run_var, float* ss_out) {
    return bn_apply_fin(ST(stream), R, C, z, res, bf16, partial, ldp, nchunk, shift, eps, momentum, gamma, beta, relu, y, mean,
                        invstd, run_mean, run_var, ss_out);
}

int scnattn_bn_bwd_reduce(void* stream, int R, int C, const void* dy, const void* y, const void* z, int bf16, const float* mean,
                          const float* invstd, int relu, float* partial, int ldp_cap, void* gout, int* nchunk_out) {
    return bn_bwd_reduce_t(ST(stream), R, C, dy, y, z, bf16, mean, invstd, relu, partial, ldp_cap, gout, nchunk_out);
}

int scnattn_bn_bwd_dx_fin(void* stream, long R, int C, const void* g, const void* z, int bf16, const float* mean,
                          const float* invstd, const float* gamma, const float* partial, int ldp, int nchunk, float* dbeta,
                          float* dgamma, void* dz) {
    return bn_bwd_dx_fin(ST(stream), R, C, g, z, bf16, mean, invstd, gamma, partial, ldp, nchunk, dbeta, dgamma, dz);
}

// ---- mixed-precision (bf16) convolution path ---------------------------------------------------------------------------------
int scnattn_cgemm16(void* stream, int M, int N, int K, const void* A, long lda, const void* B, long ldb, float beta, void* C,
                    long ldc, int out_bf16, float* ws, long ws_floats, const scnattn_conv_extra* ex) {
    const ConvExtra x = to_extra(ex);
    return cgemm16(ST(stream), M, N, K, A, lda, B, ldb, beta, C, ldc, out_bf16, ws, ws ? ws_floats : 0, ex ? &x : nullptr, 0);
}

int scnattn_conv3x3_fwd16(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const void* x, const void* w,
                          void* y, const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    SCN_ARG(N > 0 && Hi > 0 && Wi > 0 && stride >= 1, "conv3x3_fwd16: geometry");
    ConvExtra e = to_extra(ex);
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    e.c3 = 1; e.c3c = Cin; e.c3_src_rows = (long)N * Hi * Wi; e.Hi = Hi; e.Wi = Wi; e.Ho = Ho; e.Wo = Wo; e.stride = stride;
    return cgemm16(ST(stream), N * Ho * Wo, Cout, 9 * Cin, x, Cin, w, 9L * Cin, 0.f, y, Cout, 1, ws, ws ? ws_floats : 0, &e, 0);
}

// wt: the TRANSPOSED bf16 weight copy [Cin][3][3][Cout] (scnattn_bf16_weights).  stride 1: a forward convolution of dy with
// flipped taps; stride 2: four parity classes in one launch.
int scnattn_conv3x3_dgrad16(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const void* dy, const void* wt,
                            void* dx, const scnattn_conv_extra* ex, float* ws, long ws_floats) {
    SCN_ARG(N > 0 && Hi > 0 && Wi > 0 && (stride == 1 || (stride == 2 && Hi % 2 == 0 && Wi % 2 == 0)), "conv3x3_dgrad16: geometry");
    ConvExtra e = to_extra(ex);
    SCN_ARG(stride == 1 || e.epi == 0, "conv3x3_dgrad16: the mask epilogue serves the stride-1 form only");
    if (stride == 1) {
        e.c3 = 1; e.c3c = Cout; e.c3_src_rows = (long)N * Hi * Wi; e.Hi = Hi; e.Wi = Wi; e.Ho = Hi; e.Wo = Wi; e.stride = 1;
        return cgemm16(ST(stream), N * Hi * Wi, Cin, 9 * Cout, dy, Cout, wt, 9L * Cout, 0.f, dx, Cin, 1, ws, ws ? ws_floats : 0, &e, 1);
    }
    const int Ho = Hi / 2, Wo = Wi / 2;
    e.c3 = 4; e.c3c = Cout; e.c3_src_rows = (long)N * Ho * Wo; e.Hi = Hi; e.Wi = Wi; e.Ho = Ho; e.Wo = Wo; e.stride = 2;
    return cgemm16(ST(stream), N * Ho * Wo, Cin, 9 * Cout, dy, Cout, wt, 9L * Cout, 0.f, dx, Cin, 1, ws, ws ? ws_floats : 0, &e, 0);
}

int scnattn_wgrad16_3x3(void* stream, int N, int H, int W, int Cin, int Cout, const void* dy, const void* x, float* dw,
                        float* ws, long ws_floats, int k_slices) {
    return wgrad16_3x3(ST(stream), N, H, W, Cin, Cout, dy, x, dw, ws, ws ? ws_floats : 0, k_slices);
}

int scnattn_wgrad16_rows(void* stream, int R, int Cin, int Cout, const void* dy, const void* x, long src_rows, float* dw,
                         long ldo, int gs, int gHi, int gWi, int gHo, int gWo, int goh, int gow, float* ws, long ws_floats,
                         int k_slices) {
    return wgrad16_rows(ST(stream), R, Cin, Cout, dy, x, src_rows, dw, ldo, gs, gHi, gWi, gHo, gWo, goh, gow, ws,
                        ws ? ws_floats : 0, k_slices);
}

int scnattn_bf16_weights(void* stream, int n, const scnattn_weight_desc* desc, const int* tile_prefix, int total_tiles) {
    static_assert(sizeof(scnattn_weight_desc) == sizeof(WeightDesc), "descriptor layout");
    return bf16_weights(ST(stream), n, reinterpret_cast<const WeightDesc*>(desc), tile_prefix, total_tiles);
}

int scnattn_bn_stats_fold(void* stream, int R, int C, const void* x, float eps, float momentum, float* partial,
                          float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                          const float* beta, float* ss_out) {
    return bn_stats(ST(stream), R, C, x, 0, eps, momentum, partial, mean, invstd, run_mean, run_var, gamma, beta, ss_out);
}

int scnattn_skinny_gemm(void* stream, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                        const float* W, long ldw, long wg, float* Y, long ldy, long yg, long yslab,
                        int ksplit, int* ksplit_out) {
    if (ksplit <= 0) ksplit = skinny_pick_ksplit(rows, N, K, groups);
    if (ksplit_out) *ksplit_out = ksplit;
    return skinny_gemm(ST(stream), rows, N, K, groups, X, ldx, xg, W, ldw, wg, Y, ldy, yg, yslab, ksplit);
}

int scnattn_skinny_gemm_bf16w(void* stream, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                              const void* W_bf16, long ldw, long wg, float* Y, long ldy, long yg, long yslab,
                              int ksplit, int* ksplit_out) {
    if (ksplit <= 0) ksplit = skinny_pick_ksplit(rows, N, K, groups);
    if (ksplit_out) *ksplit_out = ksplit;
    return skinny_gemm(ST(stream), rows, N, K, groups, X, ldx, xg, W_bf16, ldw, wg, Y, ldy, yg, yslab, ksplit, 1);
}

int scnattn_skinny_gemm_bf16(void* stream, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                             const void* W_bf16, long ldw, long wg, float* Y, long ldy, long yg, long yslab,
                             int ksplit, int* ksplit_out) {
    if (ksplit <= 0) ksplit = skinny_pick_ksplit(rows, N, K, groups);
    if (ksplit_out) *ksplit_out = ksplit;
    return skinny_gemm(ST(stream), rows, N, K, groups, X, ldx, xg, W_bf16, ldw, wg, Y, ldy, yg, yslab, ksplit, 2);
}

int scnattn_f32_to_bf16(void* stream, long n, const float* in, void* out) { return f32_to_bf16(ST(stream), n, in, out); }

int scnattn_attn_scores(void* stream, int rows, int P, int A, const float* att1, const float* att2, int nslab,
                        long slab_stride, long att2_ld, const float* dec_bias, const float* w, const float* b0,
                        float* e, float* att2_out) {
    return attn_scores(ST(stream), rows, P, A, att1, Slabs{att2, nslab, slab_stride, att2_ld}, dec_bias, w, b0, e,
                       att2_out);
}

int scnattn_attn_context(void* stream, int rows, int P, int E, const float* enc, const float* e,
                         const float* gpre, int nslab, long slab_stride, long gpre_ld, const float* gate_bias,
                         float* alpha_out, long alpha_ld, float* alpha_save, float* awe, float* gate, float* z) {
    return attn_context(ST(stream), rows, P, E, enc, e, Slabs{gpre, nslab, slab_stride, gpre_ld}, gate_bias,
                        alpha_out, alpha_ld, alpha_save, awe, gate, z);
}

int scnattn_mean_pixels(void* stream, int rows, int P, int E, const float* enc, float* out) {
    return mean_pixels(ST(stream), rows, P, E, enc, out);
}

int scnattn_attn_dalpha(void* stream, int rows, int P, int E, const float* enc, const float* dawe,
                        const float* dalpha_in, long dalpha_in_ld, float* dalpha) {
    return attn_dalpha(ST(stream), rows, P, E, enc, dawe, dalpha_in, dalpha_in_ld, dalpha);
}

int scnattn_attn_softmax_bwd(void* stream, int rows, int P, int A, const float* att1, const float* att2,
                             const float* w, const float* alpha, const float* dalpha, float* de, float* datt2,
                             long datt2_ld) {
    return attn_softmax_bwd(ST(stream), rows, P, A, att1, att2, w, alpha, dalpha, de, datt2, datt2_ld);
}

int scnattn_attn_datt1_post_blocks(int B, int P) { return attn_datt1_post_blocks(B, P); }

int scnattn_attn_datt1_post(void* stream, int B, int P, int A, int T, const int32_t* dl, const float* att1,
                            const float* att2_all, const float* de_all, const float* w, float* datt1,
                            float* dwpart) {
    return attn_datt1_post(ST(stream), B, P, A, T, dl, att1, att2_all, de_all, w, datt1, dwpart, nullptr);
}

int scnattn_scn_mix_fwd(void* stream, int rows, int F4, const float* pz, int pz_nslab, long pz_stride, long pz_ld,
                        const float* ex, const float* ph, int ph_nslab, long ph_stride, long ph_ld,
                        const float* qx, const float* qh, float* pa, float* phs, float* xcat) {
    return scn_mix_fwd(ST(stream), rows, F4, Slabs{pz, pz_nslab, pz_stride, pz_ld}, ex,
                       Slabs{ph, ph_nslab, ph_stride, ph_ld}, qx, qh, pa, phs, xcat);
}

int scnattn_lstm_fwd(void* stream, int rows, int H, const float* r, int nslab, long slab_stride, long r_ld,
                     long r_gate_stride, const float* bih, const float* bhh, const float* c_prev, float* gates,
                     float* c_new, float* h_new, float* tanhc) {
    return lstm_fwd(ST(stream), rows, H, Slabs{r, nslab, slab_stride, r_ld}, r_gate_stride, bih, bhh, c_prev, gates,
                    c_new, h_new, tanhc);
}

int scnattn_lstm_bwd(void* stream, int rows, int rows_next, int H, const float* dh_fc, const float* dh_next,
                     int nslab, long slab_stride, long dh_ld, float* dc, const float* gates, const float* c_prev,
                     const float* tanhc, float* dr) {
    return lstm_bwd(ST(stream), rows, rows_next, H, dh_fc, Slabs{dh_next, nslab, slab_stride, dh_ld}, dc, gates,
                    c_prev, tanhc, dr);
}

int scnattn_scn_mix_bwd(void* stream, int rows, int F4, const float* dxcat, int nslab, long slab_stride, long dx_ld,
                        long dx_gate_stride, const float* qx, const float* qh, const float* pa, const float* phs,
                        float* dpx, float* dph, long dph_ld, float* dqx_acc, float* dqh_acc) {
    return scn_mix_bwd(ST(stream), rows, F4, Slabs{dxcat, nslab, slab_stride, dx_ld}, dx_gate_stride, qx, qh, pa, phs,
                       dpx, dph, dph_ld, dqx_acc, dqh_acc);
}

int scnattn_gate_bwd(void* stream, int rows, int E, const float* dz, int nslab, long slab_stride, long dz_ld,
                     const float* awe, const float* gate, float* dawe, float* dgpre, long dgpre_ld) {
    return gate_bwd(ST(stream), rows, E, Slabs{dz, nslab, slab_stride, dz_ld}, awe, gate, dawe, dgpre, dgpre_ld);
}

int scnattn_transpose2d(void* stream, int R, int C, const float* in, long ldi, float* out, long ldo) {
    return transpose2d(ST(stream), R, C, in, ldi, out, ldo);
}

int scnattn_colsum(void* stream, int R, int N, const float* X, long ld, float* out, float beta) {
    return colsum(ST(stream), R, N, X, ld, out, beta);
}

int scnattn_mul_bcast(void* stream, int T, int B, int N, const float* x, const float* q, float* out) {
    return mul_bcast(ST(stream), T, B, N, x, q, out);
}

int scnattn_pool_permute_fwd(void* stream, int B, int C, int Hin, int Win, int Ho, int Wo, const float* x,
                             long sxb, long sxc, long sxh, long sxw, float* y) {
    return pool_permute_fwd(ST(stream), B, C, Hin, Win, Ho, Wo, x, sxb, sxc, sxh, sxw, y);
}

int scnattn_pool_permute_bwd(void* stream, int B, int C, int Hin, int Win, int Ho, int Wo, const float* dy,
                             float* dx, long sxb, long sxc, long sxh, long sxw) {
    return pool_permute_bwd(ST(stream), B, C, Hin, Win, Ho, Wo, dy, dx, sxb, sxc, sxh, sxw);
}

int scnattn_caption_loss_fwd(void* stream, int B, int T, int V, int P, const float* scores, const int64_t* targets,
                             long ldt, const int32_t* decode_lengths, long n_tokens, const float* alphas, float alpha_c,
                             float* row_lse, float* row_loss, float* sm1, float* reg_part, float* loss) {
    return caption_loss_fwd(ST(stream), B, T, V, P, scores, (const long long*)targets, ldt, decode_lengths, n_tokens,
                            alphas, alpha_c, row_lse, row_loss, sm1, reg_part, loss);
}

int scnattn_caption_loss_bwd(void* stream, int B, int T, int V, int P, const float* scores, const int64_t* targets,
                             long ldt, const int32_t* decode_lengths, long n_tokens, const float* row_lse,
                             const float* sm1, float alpha_c, const float* grad_loss, float* dscores, float* dalphas) {
    return caption_loss_bwd(ST(stream), B, T, V, P, scores, (const long long*)targets, ldt, decode_lengths, n_tokens,
                            row_lse, sm1, alpha_c, grad_loss, dscores, dalphas);
}

int scnattn_u8_gather_normalize(void* stream, const uint8_t* src, long n_src, const int64_t* idx, long n_out, int C,
                                long HW, const float* lut, void* dst, int dst_bf16, int channels_last) {
    return u8_gather_normalize(ST(stream), src, n_src, (const long long*)idx, n_out, C, HW, lut, dst, dst_bf16,
                               channels_last);
}

int scnattn_bn_workspace_floats(int C) { return bn_max_chunks() * 2 * C; }

int scnattn_bn_stats(void* stream, int R, int C, const void* x, int bf16, float eps, float momentum, float* partial,
                     float* mean, float* invstd, float* run_mean, float* run_var) {
    return bn_stats(ST(stream), R, C, x, bf16, eps, momentum, partial, mean, invstd, run_mean, run_var);
}

int scnattn_bn_apply(void* stream, int R, int C, const void* z, const void* res, int bf16, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, int relu, void* y) {
    return bn_apply(ST(stream), R, C, z, res, bf16, mean, invstd, gamma, beta, relu, y);
}

int scnattn_bn_bwd(void* stream, int R, int C, const void* dy, const void* y, const void* z, int bf16,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, int train,
                   float* partial, float* dbeta, float* dgamma, void* dz, void* dres) {
    return bn_bwd(ST(stream), R, C, dy, y, z, bf16, mean, invstd, gamma, beta, relu, train, partial, dbeta, dgamma, dz, dres);
}

int scnattn_clamp_adam(void* stream, long n, float* p, const float* g, float* m, float* v, double lr,
                       double beta1, double beta2, double eps, int step, double clip, double gscale) {
    return clamp_adam(ST(stream), n, p, g, m, v, lr, beta1, beta2, eps, step, clip, gscale);
}

}  // extern "C"
