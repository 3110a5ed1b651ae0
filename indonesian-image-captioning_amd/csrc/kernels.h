// Internal launcher API of libscnattn: one function per kernel, all asynchronous on `st`,
// all returning 0 or an error code (message via scn::last_error()).  The extern "C" surface in
// api.cpp and the sequence drivers in sequence.cpp are thin layers over these.
#pragma once
#include <hip/hip_runtime.h>

#define SCN_MAX_KSPLIT 16

namespace scn {

// ---- sgemm.hip -------------------------------------------------------------------------------
int sgemm(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda,
          const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
          const float* rowmask, int batch, long sA, long sB, long sC);

// same, with a workspace: low-parallelism shapes (few output tiles, deep K) are split along K into
// partial slabs in `ws` and reduced in slab order by a second launch (deterministic)
int sgemm_ws(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda,
             const float* B, long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask,
             int batch, long sA, long sB, long sC, float* ws, long ws_floats);

// ---- cgemm.hip: LDS-DMA pipelined fp32 GEMM with the 1x1-convolution prologues / epilogues -------------------
struct ConvExtra {
    int pro = 0;              // 1: A = relu(A*scale[k]+shift[k]) (k-contiguous A); 2: B = relu(B*scale[n]+shift[n]) ([K][N] B)
    int epi = 0;              // 1: column sums of (y-s), (y-s)^2 per 64 rows; 2: relu mask from z + sums of g, g*xhat
    const float* pro_ss = nullptr;    // interleaved {scale, shift} per channel
    float* stat_partial = nullptr;    // [2][N][cgemm_stat_ld(M)], channel-major
    const float* stat_shift = nullptr;
    const float* ez = nullptr; const float* emean = nullptr; const float* einvstd = nullptr;
    const float* egamma = nullptr; const float* ebeta = nullptr; long ldz = 0;
    int stride = 1, Hi = 0, Wi = 0, Ho = 0, Wo = 0;   // stride > 1: rows of the activation operand are gathered
    int force_split = 0;      // tests / tuning: > 0 forces the split-K factor
    int force_mi = 0;         // tests / tuning: 1 / 2 forces the 64- / 128-row tile
    int c3 = 0;               // 3x3 convolution as an implicit GEMM: 1 forward, 2 dgrad (stride 1), 3 wgrad, 4 dgrad (stride 2)
    int c3c = 0;              // channels per tap of the gathered operand
    long c3_src_rows = 0;     // rows of the gathered map
};
bool cgemm_supported(bool tA, bool tB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                     long sA, long sB);
int cgemm(hipStream_t st, bool tA, bool tB, int M, int N, int K, float alpha, const float* A, long lda, const float* B,
          long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask, int batch, long sA, long sB,
          long sC, float* ws, long ws_floats, const ConvExtra* ex);
int cgemm_row_tiles(int M);
int cgemm_stat_ld(int M);      // leading dimension of the channel-major statistics partial [2][N][ld] a product of M rows writes
int cgemm_reduce(hipStream_t st, const float* ws, int S, int M, int N, float* C, long ldc);

// ---- conv3.hip: 3x3 weight gradient (stride 1) with the activation halo staged once per strip ---------------------
bool conv3x3_wgrad_halo_ok(int N, int H, int W, int C, int Co, const float* dy, const float* x, const float* dw);
int conv3x3_wgrad_halo(hipStream_t st, int N, int H, int W, int C, int Co, const float* dy, const float* x, float* dw,
                       float* ws, long ws_floats, int force_split);

// ---- stem.hip: conv 7x7 / 2 (+ statistics) and BatchNorm + ReLU + MaxPool 3x3 / 2 of the trunk's stem -----------------
int stem_tiles(int N, int H, int W);
int stem_conv7(hipStream_t st, int N, int H, int W, const float* x, long sn, long sc, long sh, long sw, const float* w,
               long wn, long wc, long wh, long ww, float* z, float* partial, const float* stat_shift);
int stem_bn_relu_maxpool(hipStream_t st, int N, int Hz, int Wz, int C, const float* z, const float* ss, void* out, int out_bf16);

// ---- skinny.hip ------------------------------------------------------------------------------
int skinny_pick_ksplit(int rows, int N, int K, int groups);
// wbf 1: W holds bf16 (raw 16-bit) elements, ldw / wg still count elements; 2: X is rounded to bf16 in registers too and
// the products run on the bf16 matrix instruction (fp32 accumulation)
// A split-K result: `n` slabs `stride` elements apart, row leading dimension `ld`.
struct Slabs {
    const float* p; int n; long stride; long ld;
};

// The element-wise consumer of a skinny product, run INSIDE the product's launch by the workgroup that arrives last at
// a 32-column unit of its output (arrival counters per stream, csrc/cgemm.hip): the decode step's cell kernels
// (csrc/scn_cell.hip) without their launches.  Same arithmetic in the same order as the stand-alone kernels -> same bits.
//   kind 1 lstm_fwd   (product: groups = 4, N = H)   ci = {bih, bhh, c_prev}         co = {gates, c_new, h_new, tanhc}
//   kind 2 mix_fwd    (groups = 1, N = 4F)           ci = {ex, qx, qh}, sx = ph     co = {pa, phs, xcat}
//   kind 3 mix_bwd    (groups = 4, N = 2F)           ci = {qx, qh, pa, phs}         co = {dpx, dph, dqx_acc, dqh_acc}, l0 = dph_ld
//   kind 4 gate_bwd   (groups = 1, N = E)            ci = {awe, gate}               co = {dawe, dgpre}, l0 = dgpre_ld
//   kind 5 lstm_bwd of the step BEFORE the product's (groups = 1, N = H): rows / rows_next are that step's
//                                                    ci = {dh_fc, gates, c_prev, tanhc}   co = {dc, dr}
struct SkinnyTail {
    int kind, rows, rows_next, dim;          // dim: H / 4F / 4F / E / H
    const float* ci[4]; float* co[4]; long l0; Slabs sx;
};
// in-launch split-K combine: arrival counters of a stream (zero at rest; csrc/cgemm.hip), null inside a graph capture
int* split_counters(hipStream_t st);
constexpr int SPLIT_COUNTERS = 1 << 16;
// `tail` (optional): fuse the element-wise consumer; *fused reports whether the launch took it (else the caller launches
// the stand-alone kernel).
int skinny_gemm(hipStream_t st, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                const void* W, long ldw, long wg, float* Y, long ldy, long yg, long yslab, int ksplit, int wbf = 0,
                const SkinnyTail* tail = nullptr, bool* fused = nullptr);

// ---- attention.hip ---------------------------------------------------------------------------
// e[b,p] = w . relu(att1[b,p,:] + att2[b,:]) + b0, att2 = sum(slabs) + bd   (attention.py:37-39)
// bf (here and below): the streamed operand (att1 / enc / x) holds bf16 elements
int attn_scores(hipStream_t st, int rows, int P, int A, const void* att1, Slabs att2, const float* bd,
                const float* w, const float* b0, float* e, float* att2_out, bool bf = false);
// alpha = softmax_p(e); awe = sum_p alpha*enc; z = sigmoid(gpre + bbeta) * awe   (attention.py:40-42,
// attention_scn.py:147-148).  gpre.p == nullptr -> no gate (z = awe, gate not written).
int attn_context(hipStream_t st, int rows, int P, int E, const void* enc, const float* e, Slabs gpre,
                 const float* bbeta, float* alpha_out, long alpha_ld, float* alpha_save, float* awe,
                 float* gate, float* z, bool bf = false);
int mean_pixels(hipStream_t st, int rows, int P, int E, const float* enc, float* out);
// dalpha[b,p] = enc[b,p,:] . dawe[b,:] + dalpha_in[b,p]
// Pooled path (include/scnattn.h, scnattn_pool): encoder_out = fixed linear pooling of x [B][Q][E]
struct PoolDesc {
    int Q, qtap_max;
    const int* tap_idx;     // [P][4]
    const float* tap_w;     // [P][4]
    const int* qtap_idx;    // [Q][qtap_max], -1 = unused
    const float* qtap_w;    // [Q][qtap_max]
    const float* col_w;     // [Q]: column sums of the pooling matrix / P (pixel mean of the pooled map)
};
int attn_context_pooled(hipStream_t st, int rows, int P, int E, const void* x, const PoolDesc& pool, const float* e,
                        Slabs gpre, const float* bbeta, float* alpha_out, long alpha_ld, float* alpha_save,
                        float* alphaq_save, float* awe, float* gate, float* z, bool bf = false);
int weighted_rows(hipStream_t st, int rows, int Q, int E, const float* x, const float* wts, float* out);
int attn_softmax_bwd_pooled(hipStream_t st, int rows, int P, int A, const void* att1, const float* att2, const float* w,
                            const float* alpha, const PoolDesc& pool, const float* dalphaq, const float* dalpha_in,
                            long dalpha_in_ld, float* de, float* datt2, long datt2_ld, bool bf = false);
int pool_expand(hipStream_t st, int B, int P, int A, const PoolDesc& pool, const float* y, const float* bias, float* out);
int pool_transpose(hipStream_t st, int B, int P, int A, const PoolDesc& pool, const float* in, float* out);
int add_bcast_rows_w(hipStream_t st, int B, int Q, int E, const float* wts, const float* v, float* out);
int attn_dalpha(hipStream_t st, int rows, int P, int E, const void* enc, const float* dawe,
                const float* dalpha_in, long dalpha_in_ld, float* dalpha, bool bf = false);
// de = alpha*(dalpha - sum(alpha*dalpha));  datt2[b,a] = w[a] * sum_p de[b,p]*[att1+att2 > 0]
int attn_softmax_bwd(hipStream_t st, int rows, int P, int A, const void* att1, const float* att2,
                     const float* w, const float* alpha, const float* dalpha, float* de, float* datt2,
                     long datt2_ld, bool bf = false);
// After the time loop: datt1[b,p,a] = w[a]*sum_t de_t[b,p]*[att1+att2_t>0]; per-block partials of
// dw[a] = sum de_t*relu(att1+att2_t) and db0 = sum de_t in dwpart[block][A+1].
int attn_datt1_post(hipStream_t st, int B, int P, int A, int T, const int* dl, const void* att1,
                    const float* att2_all, const float* de_all, const float* w, float* datt1,
                    float* dwpart, int* nblocks_out, bool bf = false);
int attn_datt1_post_blocks(int B, int P);

// ---- scn_cell.hip ----------------------------------------------------------------------------
int scn_mix_fwd(hipStream_t st, int rows, int F4, Slabs pz, const float* ex, Slabs ph, const float* qx,
                const float* qh, float* pa, float* phs, float* xcat);
int lstm_fwd(hipStream_t st, int rows, int H, Slabs r, long r_g, const float* bih, const float* bhh,
             const float* c_prev, float* gates, float* c_new, float* h_new, float* tanhc);
int lstm_bwd(hipStream_t st, int rows, int rows_next, int H, const float* dh_fc, Slabs dh_next,
             float* dc, const float* gates, const float* c_prev, const float* tanhc, float* dr);
int scn_mix_bwd(hipStream_t st, int rows, int F4, Slabs dxcat, long dxcat_g, const float* qx,
                const float* qh, const float* pa, const float* phs, float* dpx, float* dph, long dph_ld,
                float* dqx_acc, float* dqh_acc);
int gate_bwd(hipStream_t st, int rows, int E, Slabs dz, const float* awe, const float* gate,
             float* dawe, float* dgpre, long dgpre_ld);

// ---- misc.hip --------------------------------------------------------------------------------
int transpose2d(hipStream_t st, int R, int C, const float* in, long ldi, float* out, long ldo);
int copy2d(hipStream_t st, int R, int C, const float* in, long ldi, float* out, long ldo);
int colsum(hipStream_t st, int R, int N, const float* X, long ld, float* out, float beta);
int colsum_masked(hipStream_t st, int R, int N, const float* X, long ld, const float* rowmask, float* out, float beta);
int gather_rows_tm(hipStream_t st, int B, int T, int L, int M, const long long* caps, const float* table,
                   int V, float* out_tm);
int scatter_add_rows_tm(hipStream_t st, int B, int T, int L, int M, const long long* caps, const int* dl,
                        const float* demb_tm, int V, float* dtable, int* present);
int hidden_to_bm(hipStream_t st, int B, int T, int D, const int* dl, const float* hs_tm,
                 const float* mask_bm, float* out_bm, float* rowmask);
int hidden_from_bm(hipStream_t st, int B, int T, int D, const int* dl, const float* dbm,
                   const float* mask_bm, float* out_tm);
int add_bcast_rows(hipStream_t st, int B, int P, int E, const float* v, float scale, float* x);
// loss.hip: packed cross-entropy + doubly-stochastic attention term of the train step, in place
int caption_loss_fwd(hipStream_t st, int B, int T, int V, int P, const float* scores, const long long* targets, long ldt,
                     const int* dl, long n_tokens, const float* alphas, float alpha_c, float* row_lse, float* row_loss,
                     float* sm1, float* reg_part, float* loss);
int caption_loss_bwd(hipStream_t st, int B, int T, int V, int P, const float* scores, const long long* targets, long ldt,
                     const int* dl, long n_tokens, const float* row_lse, const float* sm1, float alpha_c,
                     const float* gout, float* dscores, float* dalphas);
// data.hip: uint8 image rows (gathered by index) -> normalised fp32/bf16 batch, NCHW or channels-last
int u8_gather_normalize(hipStream_t st, const uint8_t* src, long n_src, const long long* idx, long n_out, int C,
                        long HW, const float* lut, void* dst, int dst_bf16, int channels_last);
int pool_permute_fwd(hipStream_t st, int B, int C, int Hin, int Win, int Ho, int Wo, const float* x,
                     long sxb, long sxc, long sxh, long sxw, float* y);
int pool_permute_bwd(hipStream_t st, int B, int C, int Hin, int Win, int Ho, int Wo, const float* dy,
                     float* dx, long sxb, long sxc, long sxh, long sxw);
int clamp_adam(hipStream_t st, long n, float* p, const float* g, float* m, float* v, double lr, double b1,
               double b2, double eps, int step, double clip, double gscale);
int mul_bcast(hipStream_t st, int T, int B, int N, const float* x, const float* q, float* out);
int reduce_slabs(hipStream_t st, int rows, int N, Slabs s, float* out);
// fp32 -> bf16 (round to nearest even), n elements (n % 4 == 0, 16-byte aligned input)
int f32_to_bf16(hipStream_t st, long n, const float* in, void* out);

// ---- batchnorm.hip (channels-last feature maps as [R = N*H*W, C] matrices) --------------------------
int bn_max_chunks();
int bn_stats(hipStream_t st, int R, int C, const void* x, int bf16, float eps, float momentum, float* partial,
             float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma = nullptr,
             const float* beta = nullptr, float* ss_out = nullptr);
int bn_apply(hipStream_t st, int R, int C, const void* z, const void* res, int bf16, const float* mean,
             const float* invstd, const float* gamma, const float* beta, int relu, void* y);
int bn_bwd(hipStream_t st, int R, int C, const void* dy, const void* y, const void* z, int bf16, const float* mean,
           const float* invstd, const float* gamma, const float* beta, int relu, int train, float* partial, float* dbeta,
           float* dgamma, void* dz, void* dres);

// ---- finalize on load: channel-major partials [2][C][ldp] (cgemm statistics epilogues, cstats, bn_bwd_reduce_t, the stem) ----
int bn_finalize_t(hipStream_t st, long R, int C, const float* partial, int ldp, int nchunk, const float* shift, float eps,
                  float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                  const float* beta, float* ss_out);
// bf16 != 0: the maps (z, res, y / dy, y, z, gout / g, z, dz) hold bf16; statistics, partials and parameters stay fp32
int bn_apply_fin(hipStream_t st, long R, int C, const void* z, const void* res, int bf16, const float* partial, int ldp, int nchunk,
                 const float* shift, float eps, float momentum, const float* gamma, const float* beta, int relu, void* y,
                 float* mean, float* invstd, float* run_mean, float* run_var, float* ss_out);
int bn_bwd_reduce_t(hipStream_t st, int R, int C, const void* dy, const void* y, const void* z, int bf16, const float* mean,
                    const float* invstd, int relu, float* partial, int ldp_cap, void* gout, int* nchunk_out);
int bn_bwd_dx_fin(hipStream_t st, long R, int C, const void* g, const void* z, int bf16, const float* mean, const float* invstd,
                  const float* gamma, const float* partial, int ldp, int nchunk, float* dbeta, float* dgamma, void* dz);

// ---- bf16 convolution path (BASELINE configs[4]): cgemm16.hip, wgrad16.hip, misc.hip ----------------------------------------
int cgemm16(hipStream_t st, int M, int N, int K, const void* A, long lda, const void* B, long ldb, float beta, void* C, long ldc,
            int out_bf16, float* ws, long ws_floats, const ConvExtra* ex, int flip);
int wgrad16_3x3(hipStream_t st, int N, int H, int W, int C, int Co, const void* dy, const void* x, float* dw, float* ws,
                long ws_floats, int force_split);
int wgrad16_rows(hipStream_t st, int R, int C, int Co, const void* dy, const void* x, long src_rows, float* dw, long ldo,
                 int gs, int gHi, int gWi, int gHo, int gWo, int goh, int gow, float* ws, long ws_floats, int force_split);
// One launch converts every convolution weight of a trunk: fp32 master [Cout][taps][Cin] -> bf16 copy in the same layout
// and a TRANSPOSED bf16 copy [Cin][taps][Cout] (the B operand of the d-input products).  desc: device array of
// WeightDesc, tile_prefix[n+1]: exclusive prefix sums of each weight's 32 x 32 tile count (taps * Cout/32 * Cin/32).
struct WeightDesc { const float* src; void* dst; void* dst_t; int cout, taps, cin, pad; };
int bf16_weights(hipStream_t st, int n, const WeightDesc* desc, const int* tile_prefix, int total_tiles);

}  // namespace scn
