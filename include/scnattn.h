/* libscnattn -- C ABI of the MI355X (gfx950) SCN+Attention training path.
 *
 * The reference (rayandrew/indonesian-image-captioning) has no FFI layer: its hot path is eager
 * PyTorch.  This header is the boundary that sits UNDER the reference's nn.Module API
 * (SURVEY.md 8b); each entry point names the reference code it replaces (paths relative to the
 * reference checkout).  The Python binding a maintainer adds is in INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - plain pointers + sizes, fp32 row-major device memory, int64 token ids, no torch types;
 *   - every call is asynchronous on the hipStream_t passed as `void* stream` (0 = null stream);
 *   - the library allocates nothing: callers pass workspaces sized by scnattn_seq_workspace();
 *   - return 0 on success, <0 for invalid arguments, >0 = hipError_t; message (thread-local) from
 *     scnattn_last_error(); no exceptions or aborts cross this boundary; entry points are
 *     re-entrant (autograd calls backward from its own thread).
 */
#ifndef SCNATTN_H
#define SCNATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCNATTN_VERSION 108 /* 0.1.8: + bf16 trunk kernels (scnattn_cgemm16, _conv3x3_fwd16/_dgrad16, _wgrad16_*, _bf16_weights), split-K
                               epilogues inside the GEMM launch (options cgemm_combine, cgemm_combine_max), option dec_tail;
                               0.1.7: + halo-staged 3x3 weight gradient, strided 3x3 d input, the stem (scnattn_stem_*), BatchNorm
                               finalize on load; - whole-block drivers, scnattn_stream_*, the experiment options of rounds 1-2 */

int scnattn_version(void);
const char* scnattn_last_error(void);
/* Process-wide options.  Every entry point is re-entrant and takes its stream explicitly; what is left here is
 *   - what a caller chooses once per process: "decoder_bf16" (0 fp32; 1: the sequence drivers stream bf16 copies of the
 *     recurrent weights, att1 and the encoder map, fp32 accumulate / state / gradients; 2: also the bf16 matrix
 *     instruction in the per-step products -- BASELINE configs[4]; needs D, F, E, A multiples of 4) and "profile" (1: bracket
 *     the recurrence loops with HIP events on the caller's stream; 2: also every attn_context launch; see
 *     scnattn_profile_collect);
 *   - policy overrides used by tools/ sweeps and the tests that cover both sides of a policy: "ksplit" (split-K factor of the
 *     skinny GEMMs; 0 = auto), "attn_depth", "use_cgemm" (0: every dense product on the round-1 sgemm kernel), "cgemm_mi"
 *     (0 auto; 1 / 2 force the 64- / 128-row tile), "cgemm_target" (workgroups a split-K product aims for, 512),
 *     "cgemm_kmin" (smallest K per slab, 128), "gemm_target" / "gemm_gate" / "gemm_kmin" / "gemm_kmin_small" (the same for
 *     sgemm), "cgemm_combine" (1, default: a split product's epilogue is run inside the launch by the workgroup that arrives
 *     last at each tile -- write-through slabs, arrival counters per stream, bit-identical to the reduce launch; 2: plain
 *     slab stores + an agent release; 0: always the reduce launch), "cgemm_combine_max" (deepest split combined in-launch, 8),
 *     "dec_tail" (0, default; 1: the decode step's element-wise cell kernels run inside the skinny launches that feed
 *     them -- bit-identical, measured slower: DESIGN.md 6c).  The experiment switches of rounds 1-2 (fuse_attn, chains, attn_handoff, cgemm_stagger, cgemm_w41, cgemm_vec,
 *     bn_gfirst, skinny_tail) were removed together with the code paths they selected; DESIGN.md keeps their numbers.
 * Returns -1 for an unknown name or an out-of-range value. */
int scnattn_set_option(const char* name, int value);
/* Sums since the last call: out6 = {forward loop ms, forward steps, backward loop ms, backward steps,
 * attn_context ms, attn_context launches}.  Synchronises on the recorded events. */
int scnattn_profile_collect(double* out6);

/* ---- dimensions of one decoder (models/decoders/attention_scn.py:28-56, pure_scn.py:26-48) ------ */
typedef struct {
    int B;       /* batch rows                                                         */
    int P;       /* pixels per image (14*14)                                           */
    int E;       /* encoder_dim                                                        */
    int A;       /* attention_dim   (ignored when has_att == 0)                        */
    int D;       /* decoder_dim == SCNCell.hidden_size                                 */
    int F;       /* factored_dim                                                       */
    int M;       /* embed_dim                                                          */
    int S;       /* semantic_dim (tags)                                                */
    int V;       /* vocab_size                                                         */
    int T;       /* decode steps = max(decode_lengths)                                 */
    int L;       /* width of the caption tensor (>= T)                                 */
    int has_att; /* 1: AttentionSCN (SCNCell input = M+E), 0: PureSCN (input = M)      */
} scnattn_dims;

/* Parameter pointers, named after the reference's state_dict keys.  The same struct (with
 * writable memory behind it) receives the gradients in scnattn_seq_bwd. */
typedef struct {
    float* attention_encoder_att_weight; /* [A,E]   models/attention.py:18 */
    float* attention_encoder_att_bias;   /* [A]                            */
    float* attention_decoder_att_weight; /* [A,D]   models/attention.py:20 */
    float* attention_decoder_att_bias;   /* [A]                            */
    float* attention_full_att_weight;    /* [1,A]   models/attention.py:22 */
    float* attention_full_att_bias;      /* [1]                            */
    float* embedding_weight;             /* [V,M]   attention_scn.py:43    */
    float* decode_step_weight_ia;        /* [I,4F]  models/scn_cell.py:29  */
    float* decode_step_weight_ib;        /* [S,4F]                         */
    float* decode_step_weight_ic;        /* [D,4F]                         */
    float* decode_step_weight_ha;        /* [D,4F]                         */
    float* decode_step_weight_hb;        /* [S,4F]                         */
    float* decode_step_weight_hc;        /* [D,4F]                         */
    float* decode_step_bias_ih;          /* [4D]                           */
    float* decode_step_bias_hh;          /* [4D]                           */
    float* init_h_weight;                /* [D,E]   attention_scn.py:48    */
    float* init_h_bias;                  /* [D]                            */
    float* init_c_weight;                /* [D,E]                          */
    float* init_c_bias;                  /* [D]                            */
    float* f_beta_weight;                /* [E,D]   attention_scn.py:52    */
    float* f_beta_bias;                  /* [E]                            */
    float* fc_weight;                    /* [V,D]   attention_scn.py:55    */
    float* fc_bias;                      /* [V]                            */
} scnattn_params;

/* Optional: encoder_out described as a FIXED LINEAR POOLING of a smaller feature map x [B,Q,E] -- what
 * models/encoders/caption.py:41-43 produces (AdaptiveAvgPool2d(14) of the trunk's 8x8 map at 256x256 input,
 * an up-sampling 1-or-2-tap average, then permute):
 *     enc[b][p][:] = sum_{k<4} tap_w[p][k] * x[b][tap_idx[p][k]][:]        (unused taps: weight 0, index 0)
 * With it the sequence drivers take x in place of enc and never materialise the pooled map: by linearity
 * encoder_att runs on Q rows (att1 = pool(x.We^T) + be), the attention context and its gradient read Q rows
 * per image instead of P (sum_p alpha_p enc_p = sum_q alphaq_q x_q, alphaq = pool^T alpha), and d x comes out
 * directly.  SURVEY.md 8d names this shortcut.  All tables live in device memory.
 *   qtap_*: the transpose -- for source pixel q the (p, weight) pairs that read it, padded with index -1;
 *   col_w[q] = (sum_p weight(p,q)) / P, so that mean_p enc[b][p] = sum_q col_w[q] x[b][q]. */
typedef struct scnattn_pool {
    int Q;                   /* source pixels per image (Q <= P) */
    int qtap_max;            /* row length of qtap_idx / qtap_w (<= 64) */
    const int32_t* tap_idx;  /* [P][4] */
    const float* tap_w;      /* [P][4] */
    const int32_t* qtap_idx; /* [Q][qtap_max] */
    const float* qtap_w;     /* [Q][qtap_max] */
    const float* col_w;      /* [Q] */
} scnattn_pool;

/* Bytes of the two workspaces of the sequence drivers (pool may be NULL).  `saved` carries forward state to
 * the backward pass; `scratch` is free after each call.  Both must be zero-filled by the caller before
 * scnattn_seq_fwd (saved) / each call (scratch) unless every caption decodes at every step (bt_host[T-1] == B),
 * in which case every element is written before it is read. */
int scnattn_seq_workspace(const scnattn_dims* d, const scnattn_pool* pool, size_t* saved_bytes, size_t* scratch_bytes);

/* Teacher-forced decoder forward over all T steps: replaces the loop of
 * models/decoders/attention_scn.py:124-156 (pure_scn.py:114-138 when has_att == 0), i.e. per step
 * Attention.forward (models/attention.py:35-44), the f_beta gate, SCNCell.forward/recurrent_step
 * (models/scn_cell.py:62-154), dropout and fc.
 *   enc     [B,P,E]  encoder output, rows already permuted by sort_ind ([B,Q,E] = x when pool != NULL)
 *   tags    [B,S]    semantic input, NOT permuted (reference quirk, attention_scn.py:152)
 *   caps    [B,L]    int64 sorted captions;  dl_dev [B] int32 decode lengths (device)
 *   bt_host [T]      host array: active rows at step t (non-increasing)
 *   drop_mask [B,T,D] pre-scaled dropout mask or NULL
 *   preds   [B,T,V]  out (fully written; rows past a caption's length are 0)
 *   alphas  [B,T,P]  out, must be zero-filled by the caller (NULL when has_att == 0) */
int scnattn_seq_fwd(void* stream, const scnattn_dims* d, const scnattn_params* w, const float* enc,
                    const float* tags, const int64_t* caps, const int32_t* dl_dev, const int32_t* bt_host,
                    const float* drop_mask, float* saved, float* scratch, float* preds, float* alphas,
                    const scnattn_pool* pool);

/* Gradient of scnattn_seq_fwd (what autograd derives for the reference's loop).  `g` receives
 * d loss / d parameter (fields may be NULL to skip; embedding_weight must be zero-filled: rows
 * are accumulated).  denc [B,P,E] ([B,Q,E] = d x when pool != NULL) and dtags [B,S] may be NULL. */
int scnattn_seq_bwd(void* stream, const scnattn_dims* d, const scnattn_params* w, const float* enc,
                    const float* tags, const int64_t* caps, const int32_t* dl_dev, const int32_t* bt_host,
                    const float* drop_mask, const float* saved, float* scratch, const float* dpreds,
                    const float* dalphas, const scnattn_params* g, float* denc, float* dtags,
                    const scnattn_pool* pool);
/* The same with the weight gradients on a second stream.  Nothing needs d loss / d weight before the optimizer step,
 * while denc heads the encoder's whole backward pass (trains/attention_scn.py:238-240 runs them as one autograd
 * sweep): `wgrad_stream` carries d fc.weight beside the reverse recurrence and the post-loop weight-gradient GEMMs
 * beside whatever the caller enqueues next on `stream`.  Event-ordered inside; on return `stream` holds denc / dtags
 * and does NOT wait for `wgrad_stream`: the caller joins it before reading `g`, and keeps saved / scratch / dpreds /
 * enc / tags alive until then.  wgrad_stream == NULL or == stream: identical to scnattn_seq_bwd. */
int scnattn_seq_bwd_streams(void* stream, void* wgrad_stream, const scnattn_dims* d, const scnattn_params* w,
                            const float* enc, const float* tags, const int64_t* caps, const int32_t* dl_dev,
                            const int32_t* bt_host, const float* drop_mask, const float* saved, float* scratch,
                            const float* dpreds, const float* dalphas, const scnattn_params* g, float* denc,
                            float* dtags, const scnattn_pool* pool);

/* ---- primitives (each = one kernel launch); used by the stand-alone modules and the tests ------- */
/* C = alpha*op(A).op(B) + beta*C + bias[n]; rows with rowmask[m]==0 written as 0.  Replaces the
 * aten::mm / addmm calls behind nn.Linear and `@` on the path. */
int scnattn_sgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                  long lda, const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                  const float* rowmask, int batch, long strideA, long strideB, long strideC);
/* Same product with a caller-provided workspace of `ws_floats` floats for deterministic split-K partial sums:
 * when the 128x128 tile grid alone cannot fill the chip (few rows: beam search, batch-4 configurations) the K
 * range is divided over up to 16 workgroups per tile and reduced in a fixed order by a second launch.  ws may
 * be NULL (then identical to scnattn_sgemm).  The workspace must not be shared by calls on different streams. */
int scnattn_sgemm_ws(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A,
                     long lda, const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                     const float* rowmask, int batch, long strideA, long strideB, long strideC, float* ws,
                     long ws_floats);
/* Y[s][g][r][n] = sum_{k in slice s} X[r][g*xg+k] * W[g*wg + k*ldw + n]; ksplit<=0 picks one.
 * Returns the ksplit used through *ksplit_out. */
int scnattn_skinny_gemm(void* stream, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                        const float* W, long ldw, long wg, float* Y, long ldy, long yg, long yslab,
                        int ksplit, int* ksplit_out);
/* Mixed precision (BASELINE configs[4]): the same product with the weight matrix stored as bf16 (raw 16-bit elements,
 * ldw / wg in elements), widened to fp32 in registers, fp32 accumulation.  scnattn_f32_to_bf16 makes such a copy
 * (round to nearest even; n % 4 == 0).  The sequence drivers use both when option "decoder_bf16" is 1: the operands
 * the recurrence streams every step (recurrent weights, att1, the encoder map) are then read as bf16 copies made once
 * per call; softmax, LSTM state, master weights and every gradient stay fp32. */
int scnattn_skinny_gemm_bf16w(void* stream, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                              const void* W_bf16, long ldw, long wg, float* Y, long ldy, long yg, long yslab,
                              int ksplit, int* ksplit_out);
/* ... and with the activation rows rounded to bf16 in registers as well: v_mfma_f32_32x32x16_bf16, fp32 accumulation
 * (option "decoder_bf16" = 2 selects it inside the sequence drivers) */
int scnattn_skinny_gemm_bf16(void* stream, int rows, int N, int K, int groups, const float* X, long ldx, long xg,
                             const void* W_bf16, long ldw, long wg, float* Y, long ldy, long yg, long yslab,
                             int ksplit, int* ksplit_out);
int scnattn_f32_to_bf16(void* stream, long n, const float* in, void* out);

/* models/attention.py:37-39 */
int scnattn_attn_scores(void* stream, int rows, int P, int A, const float* att1, const float* att2, int nslab,
                        long slab_stride, long att2_ld, const float* dec_bias, const float* w, const float* b0,
                        float* e, float* att2_out);
/* models/attention.py:40-42 (+ gate of attention_scn.py:147-148 when gpre != NULL) */
int scnattn_attn_context(void* stream, int rows, int P, int E, const float* enc, const float* e,
                         const float* gpre, int nslab, long slab_stride, long gpre_ld, const float* gate_bias,
                         float* alpha_out, long alpha_ld, float* alpha_save, float* awe, float* gate, float* z);
int scnattn_mean_pixels(void* stream, int rows, int P, int E, const float* enc, float* out);
int scnattn_attn_dalpha(void* stream, int rows, int P, int E, const float* enc, const float* dawe,
                        const float* dalpha_in, long dalpha_in_ld, float* dalpha);
int scnattn_attn_softmax_bwd(void* stream, int rows, int P, int A, const float* att1, const float* att2,
                             const float* w, const float* alpha, const float* dalpha, float* de, float* datt2,
                             long datt2_ld);
int scnattn_attn_datt1_post_blocks(int B, int P);
int scnattn_attn_datt1_post(void* stream, int B, int P, int A, int T, const int32_t* dl, const float* att1,
                            const float* att2_all, const float* de_all, const float* w, float* datt1,
                            float* dwpart);
/* models/scn_cell.py:73-91 / 134-144 element-wise parts */
int scnattn_scn_mix_fwd(void* stream, int rows, int F4, const float* pz, int pz_nslab, long pz_stride, long pz_ld,
                        const float* ex, const float* ph, int ph_nslab, long ph_stride, long ph_ld,
                        const float* qx, const float* qh, float* pa, float* phs, float* xcat);
/* models/scn_cell.py:146-152 */
int scnattn_lstm_fwd(void* stream, int rows, int H, const float* r, int nslab, long slab_stride, long r_ld,
                     long r_gate_stride, const float* bih, const float* bhh, const float* c_prev, float* gates,
                     float* c_new, float* h_new, float* tanhc);
int scnattn_lstm_bwd(void* stream, int rows, int rows_next, int H, const float* dh_fc, const float* dh_next,
                     int nslab, long slab_stride, long dh_ld, float* dc, const float* gates, const float* c_prev,
                     const float* tanhc, float* dr);
int scnattn_scn_mix_bwd(void* stream, int rows, int F4, const float* dxcat, int nslab, long slab_stride, long dx_ld,
                        long dx_gate_stride, const float* qx, const float* qh, const float* pa, const float* phs,
                        float* dpx, float* dph, long dph_ld, float* dqx_acc, float* dqh_acc);
int scnattn_gate_bwd(void* stream, int rows, int E, const float* dz, int nslab, long slab_stride, long dz_ld,
                     const float* awe, const float* gate, float* dawe, float* dgpre, long dgpre_ld);
/* helpers */
int scnattn_transpose2d(void* stream, int R, int C, const float* in, long ldi, float* out, long ldo);
int scnattn_colsum(void* stream, int R, int N, const float* X, long ld, float* out, float beta);
int scnattn_mul_bcast(void* stream, int T, int B, int N, const float* x, const float* q, float* out);
/* models/encoders/caption.py:41-43: AdaptiveAvgPool2d((Ho,Wo)) + permute(0,2,3,1); x given by strides */
int scnattn_pool_permute_fwd(void* stream, int B, int C, int Hin, int Win, int Ho, int Wo, const float* x,
                             long sxb, long sxc, long sxh, long sxw, float* y);
int scnattn_pool_permute_bwd(void* stream, int B, int C, int Hin, int Win, int Ho, int Wo, const float* dy,
                             float* dx, long sxb, long sxc, long sxh, long sxw);
/* The loss of the train step, trains/attention_scn.py:222-236, without materialising the packed batch:
 *   loss = mean over rows (b, t < decode_lengths[b]) of  logsumexp(scores[b,t,:]) - scores[b,t,targets[b,t]]
 *        + alpha_c * mean over (b,p) of (1 - sum_t alphas[b,t,p])^2            (alphas may be NULL)
 * scores [B,T,V]; targets int64 with row stride ldt (pass caps_sorted + 1, ldt = L: `targets = caps_sorted[:, 1:]`);
 * n_tokens = sum_b min(decode_lengths[b], T) (the row count of pack_padded_sequence(...).data).
 * Workspaces (kept for the backward): row_lse, row_loss [B*T], sm1 [B*P], reg_part [B]; loss [1].
 * bwd: grad_loss is the DEVICE scalar d/d loss; dscores [B,T,V] is written everywhere (0 for rows that were
 * not decoded), dalphas [B,T,P] likewise.  A target outside [0,V) makes the loss NaN instead of faulting. */
int scnattn_caption_loss_fwd(void* stream, int B, int T, int V, int P, const float* scores, const int64_t* targets,
                             long ldt, const int32_t* decode_lengths, long n_tokens, const float* alphas, float alpha_c,
                             float* row_lse, float* row_loss, float* sm1, float* reg_part, float* loss);
int scnattn_caption_loss_bwd(void* stream, int B, int T, int V, int P, const float* scores, const int64_t* targets,
                             long ldt, const int32_t* decode_lengths, long n_tokens, const float* row_lse,
                             const float* sm1, float alpha_c, const float* grad_loss, float* dscores, float* dalphas);
/* Input assembly: replaces the per-sample host arithmetic of datasets/caption.py:51-53
 * (`torch.FloatTensor(imgs[i // cpi] / 255.)` + torchvision Normalize, trains/attention_scn.py:121-126).
 * src: n_src uint8 images [n_src][C][HW] in HBM (a staged batch or the whole dataset); idx: n_out int64
 * source rows on the device, or NULL for rows 0..n_out-1; lut: C*256 floats,
 * lut[c][v] = ((float)(v/255.0) - mean[c]) / std[c] computed by the caller with the reference's own
 * arithmetic, so the output is bit-identical to the reference's tensor.  dst: [n_out][C][HW]
 * (channels_last = 0) or [n_out][HW][C] (1), fp32 or bf16 (round-to-nearest-even of the fp32 value).
 * A row index outside [0, n_src) yields a NaN image instead of a fault. */
int scnattn_u8_gather_normalize(void* stream, const uint8_t* src, long n_src, const int64_t* idx, long n_out, int C,
                                long HW, const float* lut, void* dst, int dst_bf16, int channels_last);
/* Fused BatchNorm2d (+ residual) (+ ReLU) on channels-last maps viewed as [R = N*H*W, C] (C % 4 == 0):
 * the `bn -> relu` / `bn -> (+identity) -> relu` groups of torchvision's Bottleneck behind
 * models/encoders/caption.py:17-22.  `partial` needs scnattn_bn_workspace_floats(C) floats.
 *   bn_stats : batch mean / 1/sqrt(var+eps) per channel (+ running-stat update with `momentum`)
 *   bn_apply : y = relu?(gamma*(z-mean)*invstd + beta + res?)
 *   bn_bwd   : dbeta, dgamma, dz (batch-stat form when train != 0) and dres = dy*[y>0] */
int scnattn_bn_workspace_floats(int C);
/* bf16 != 0: the feature maps (x, z, res, y, dy, dz, dres) are bf16 in memory (trunk under bf16 autocast,
 * BASELINE config 5); statistics, parameters and arithmetic stay fp32. */
int scnattn_bn_stats(void* stream, int R, int C, const void* x, int bf16, float eps, float momentum, float* partial,
                     float* mean, float* invstd, float* run_mean, float* run_var);
int scnattn_bn_apply(void* stream, int R, int C, const void* z, const void* res, int bf16, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, int relu, void* y);
/* relu != 0: the mask is [y > 0]; with y == NULL and beta given it is recomputed from z with the forward's own
 * expression fma((z-mean)*invstd, gamma, beta) > 0 (only valid when no residual was added before the ReLU) and
 * the backward pass never reads y. */
int scnattn_bn_bwd(void* stream, int R, int C, const void* dy, const void* y, const void* z, int bf16,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, int relu, int train,
                   float* partial, float* dbeta, float* dgamma, void* dz, void* dres);
/* ---- 1x1 convolutions of the ResNet-152 trunk as fused GEMMs (csrc/cgemm.hip) --------------------------------
 * Replaces the `nn.Conv2d(kernel_size=1)` layers of torchvision's Bottleneck (conv1, conv3, downsample.0) behind
 * models/encoders/caption.py:17-22 -- forward, d input, d weight -- on channels-last maps, where such a convolution
 * is the product [R = N*Ho*Wo, Cin] x [Cin, Cout], together with the BatchNorm work that can ride on it:
 *   pro_ss [Cin][2] = {scale, shift} (fwd, wgrad): the input operand is taken as relu(x * scale[c] + shift[c]) --
 *       the previous BatchNorm + ReLU folded to one fma per element (scale = gamma*invstd, shift = beta - mean*scale,
 *       written interleaved by scnattn_bn_finalize) -- so the normalised map is never written to or read from HBM;
 *   stat_partial [2][Cout][scnattn_cgemm_stat_ld(R)] (fwd; channel-major): per output channel and 64-row block, sum(y - s)
 *       and sum((y - s)^2) with s = stat_shift[c] (or 0): the statistics pass of the NEXT BatchNorm, fixed order;
 *   ez / emean / einvstd / egamma / ebeta (dgrad): the result is masked with the ReLU mask recomputed from the
 *       BatchNorm input z ([R][Cin], leading dimension ldz), g = dx * [fma((z-mean)*invstd, gamma, beta) > 0] -- or,
 *       when pro_ss is given, * [fma(z, scale, shift) > 0], bit for bit the mask of what a forward prologue computed --
 *       and stat_partial receives sum(g), sum(g * xhat): the two reductions of that BatchNorm's backward pass.
 * stride > 1 (downsample.0): input rows are gathered / scattered at (n, ho*stride, wo*stride) of an Hi x Wi map.
 * ws / ws_floats: split-K partial sums (shapes whose 128x128 tile grid cannot fill the chip); may be NULL. */
typedef struct scnattn_conv_extra {
    int pro;                  /* 0 none, 1: A operand prologue (per k), 2: B operand prologue (per n) */
    int epi;                  /* 0 plain, 1: statistics of the output, 2: ReLU mask from z + BatchNorm-backward sums */
    const float* pro_ss;
    float* stat_partial; const float* stat_shift;
    const float* ez; const float* emean; const float* einvstd; const float* egamma; const float* ebeta; long ldz;
    int stride, Hi, Wi, Ho, Wo;
    int force_split;          /* > 0: force the split-K factor (tests, tuning) */
    int force_mi;             /* 1 / 2: force the 64- / 128-row tile (tests, tuning) */
} scnattn_conv_extra;
/* C = alpha*op(A).op(B) + beta*C + bias, rows with rowmask == 0 written as 0: same contract as scnattn_sgemm_ws, on
 * the LDS-DMA pipelined kernel; needs 16-byte aligned operands (returns -1 otherwise; scnattn_sgemm_ws picks the
 * kernel by itself).  ex may be NULL. */
int scnattn_cgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, long lda,
                  const float* B, long ldb, float beta, float* C, long ldc, const float* bias, const float* rowmask,
                  int batch, long strideA, long strideB, long strideC, float* ws, long ws_floats,
                  const scnattn_conv_extra* ex);
int scnattn_cgemm_row_tiles(int M);
/* y [R][Cout] = f(x) . w^T, w [Cout][Cin] (a channels-last 1x1 conv weight); R = output rows */
int scnattn_conv1x1_fwd(void* stream, int R, int Cin, int Cout, const float* x, const float* w, float* y,
                        const scnattn_conv_extra* ex, float* ws, long ws_floats);
/* dx [R][Cin] = dy [R][Cout] . w (+ beta * dx: the residual branch's gradient is accumulated in place).
 * w_transposed != 0: `w` is the weight already transposed to [Cin][Cout] (scnattn_transpose2d), which puts both
 * operands on the k-contiguous LDS image -- worth it when Cin is large and Cout small (conv1 of a bottleneck). */
int scnattn_conv1x1_dgrad(void* stream, int R, int Cin, int Cout, const float* dy, const float* w, int w_transposed,
                          float beta, float* dx, const scnattn_conv_extra* ex, float* ws, long ws_floats);
/* dw [Cout][Cin] = dy^T . f(x) */
int scnattn_conv1x1_wgrad(void* stream, int R, int Cin, int Cout, const float* dy, const float* x, float* dw,
                          const scnattn_conv_extra* ex, float* ws, long ws_floats);
/* 3x3 convolutions (padding 1, stride s) of the trunk -- conv2 of every Bottleneck -- as IMPLICIT GEMMs on the same
 * kernel: no im2col buffer; the nine taps are a walk over K (forward, d input) or a property of the column tile
 * (d weight); a tap that falls outside the image is a lane whose LDS-DMA offset is out of range, i.e. zeros.
 * x [N*Hi*Wi][Cin], y / dy [N*Ho*Wo][Cout] channels-last maps, w / dw [Cout][3][3][Cin] (a channels-last conv weight).
 * Cin, Cout multiples of 16.  ex may carry epi = 1 (fwd: statistics of y) or epi = 2 (dgrad: ReLU mask from z +
 * BatchNorm-backward sums). */
int scnattn_conv3x3_fwd(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const float* x,
                        const float* w, float* y, const scnattn_conv_extra* ex, float* ws, long ws_floats);
/* d input of a stride-1 convolution: the nine taps (flipped) walk K */
int scnattn_conv3x3_dgrad(void* stream, int N, int Hi, int Wi, int Cin, int Cout, const float* dy, const float* w,
                          float* dx, const scnattn_conv_extra* ex, float* ws, long ws_floats);
/* d input of a STRIDE-2 convolution (conv2 of layer2.0 / 3.0 / 4.0; Hi, Wi even): ONE launch whose grid.y runs over the
 * four parity classes (hi & 1, wi & 1) of d-input pixels; a class sees only the taps that reach it (1, 2, 2 or 4 of the
 * nine), so nothing is multiplied by the zeros a zero-inserted dy would carry.  dy [N*(Hi/2)*(Wi/2)][Cout]. */
int scnattn_conv3x3_dgrad_strided(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const float* dy,
                                  const float* w, float* dx, float* ws, long ws_floats);
/* d weight.  stride 1, Cin % 32 == 0, Cout % 32 == 0 (every identity Bottleneck of the trunk): the
 * HALO-STAGED kernel of csrc/conv3.hip -- a wave owns a 32 x 32 block of dw for all nine taps (nine accumulators), walks
 * output pixels strip by strip and reads each staged activation line at nine shifted LDS addresses, so an activation
 * byte is staged once instead of nine times; the four waves of a workgroup split K and meet in LDS; k_slices > 0 fixes
 * the workgroup-level K split (0: policy, ~512 workgroups), whose slabs (k_slices * |dw| floats in ws) a second launch
 * sums in slab order.  Otherwise (strided, or k_slices < 0): the gathered [K][M] x [K][N] form on scnattn_cgemm's
 * kernel (needs Cin % 128 == 0). */
int scnattn_conv3x3_wgrad(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const float* dy,
                          const float* x, float* dw, float* ws, long ws_floats, int k_slices);
/* ---- the stem: conv1 7x7 / 2 / pad 3 (3 -> 64) -> BatchNorm -> ReLU -> MaxPool 3x3 / 2 / pad 1 (csrc/stem.hip) ----------
 * children 0..3 of the trunk behind models/encoders/caption.py:17-22; frozen in every configuration of the reference, so
 * forward only.  x is the (N,3,H,W) image batch in ANY memory format (element strides sn, sc, sh, sw), w the (64,3,7,7)
 * weight in any format (wn, wc, wh, ww).  scnattn_stem_conv7 writes z [N*Ho*Wo][64] (Ho = (H-1)/2+1) and, when
 * stat_partial is given, per-workgroup sums {sum(z - s), sum((z - s)^2)} [2][64][(scnattn_stem_tiles(N,H,W)+3)&~3] for
 * scnattn_bn_finalize (s = stat_shift[c] or 0).  scnattn_stem_bn_relu_maxpool: out [N*Hp*Wp][C] = max over the 3x3 window
 * of relu(z*scale[c] + shift[c]) with ss [C][2] = {scale, shift} (scnattn_bn_finalize's ss_out), Hp = (Hz-1)/2+1. */
int scnattn_stem_tiles(int N, int H, int W);
int scnattn_stem_conv7(void* stream, int N, int H, int W, const float* x, long sn, long sc, long sh, long sw,
                       const float* w, long wn, long wc, long wh, long ww, float* z, float* stat_partial,
                       const float* stat_shift);
int scnattn_stem_bn_relu_maxpool(void* stream, int N, int Hz, int Wz, int C, const float* z, const float* ss, void* out,
                                 int out_bf16);       /* out_bf16 != 0: the pooled map is written as bf16 (mixed-precision trunk) */
/* ---- BatchNorm statistics that ride on the convolutions, finalized ON LOAD (csrc/batchnorm.hip) --------------------------
 * The statistics epilogues above (and scnattn_stem_conv7, scnattn_bn_bwd_reduce, the dgrad mask pass) leave CHANNEL-MAJOR
 * partials  partial[2][C][ldp]:  entry (which, channel, chunk), ldp = chunk count rounded up to 4
 * (scnattn_cgemm_stat_ld(R) for a product of R rows: one chunk per 64 rows).  A dependent launch costs ~8-9 us on MI355X,
 * so the tiny per-BatchNorm `finalize` launches are folded into the element-wise kernel that consumes the statistics:
 * each of its workgroups owns 64 channels and sums their partial rows itself (fixed order: every workgroup gets the same
 * bits), the workgroups of the first row chunk also write the per-channel results.
 *   scnattn_bn_finalize    statistics only: mean, 1/sqrt(var+eps), running-stat update (momentum; run_* may be NULL) and,
 *                          when ss_out is given, the folded {scale = gamma*invstd, shift = beta - mean*scale} pairs [C][2]
 *                          for a consumer that normalises on load (conv3's prologue, the stem's pooling kernel);
 *   scnattn_bn_apply_fin   y = [relu](gamma*(z-mean)*invstd + beta [+ res]) with the same finalize inside;
 *                          partial holds {sum(z - s), sum((z - s)^2)}, s = shift[c] (NULL: 0).  `shift` must be the vector the
 *                          producer's epilogue used and must NOT alias run_mean (this kernel updates it while other
 *                          workgroups still read shift): callers pass the previous step's batch mean;
 *   scnattn_bn_bwd_reduce  g = dy * [y > 0] (relu != 0) or dy -> gout (may be NULL); partial = {sum g, sum g*xhat};
 *                          *nchunk_out = chunk count (ldp = that rounded up to 4 <= ldp_cap);
 *   scnattn_bn_bwd_dx_fin  dz = gamma*invstd*(g - dbeta/R - xhat*dgamma/R) from an already masked g, dbeta = sum g and
 *                          dgamma = sum g*xhat summed from the partials inside (and written out as the parameter gradients). */
int scnattn_cgemm_stat_ld(int M);
int scnattn_bn_finalize(void* stream, long R, int C, const float* partial, int ldp, int nchunk, const float* shift,
                        float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var,
                        const float* gamma, const float* beta, float* ss_out);
/* bf16 != 0: every MAP argument (z, res, y / dy, y, z, gout / g, z, dz) holds bf16 elements; statistics, partials and
 * parameters are fp32 either way */
int scnattn_bn_apply_fin(void* stream, long R, int C, const void* z, const void* res, int bf16, const float* partial, int ldp,
                         int nchunk, const float* shift, float eps, float momentum, const float* gamma, const float* beta,
                         int relu, void* y, float* mean, float* invstd, float* run_mean, float* run_var, float* ss_out);
int scnattn_bn_bwd_reduce(void* stream, int R, int C, const void* dy, const void* y, const void* z, int bf16, const float* mean,
                          const float* invstd, int relu, float* partial, int ldp_cap, void* gout, int* nchunk_out);
int scnattn_bn_bwd_dx_fin(void* stream, long R, int C, const void* g, const void* z, int bf16, const float* mean,
                          const float* invstd, const float* gamma, const float* partial, int ldp, int nchunk, float* dbeta,
                          float* dgamma, void* dz);

/* ---- mixed-precision trunk (BASELINE configs[4]): bf16 maps and operand copies, fp32 accumulation / statistics / master
 * weights / weight gradients (csrc/cgemm16.hip, csrc/wgrad16.hip) ------------------------------------------------------------
 * The same convolutions of torchvision's Bottleneck (models/encoders/caption.py:17-22) on v_mfma_f32_32x32x16_bf16.
 *   scnattn_bf16_weights   ONE launch per step: every convolution weight [Cout][taps][Cin] fp32 -> a bf16 copy in the same
 *                          layout (forward B operand) and a transposed bf16 copy [Cin][taps][Cout] (d-input B operand);
 *                          desc / tile_prefix are device arrays (tile_prefix[i] = number of 32 x 32 tiles of weights 0..i-1,
 *                          a weight has taps * Cout/32 * Cin/32 of them; Cout, Cin multiples of 32);
 *   scnattn_cgemm16        C[M][N] = A[M][K] . B[N][K]^T (+ beta*C): A, B bf16 k-contiguous, C bf16 (out_bf16) or fp32;
 *                          ex: epi 1 (statistics of C from the fp32 accumulators, channel-major partials), stride (row
 *                          gather of a strided 1x1 convolution); a 1x1 d input is this with B = the transposed copy;
 *   scnattn_conv3x3_fwd16 / _dgrad16   implicit GEMMs as in fp32 (taps over K; stride-2 d input by parity classes);
 *                          epi 2 (scnattn_cgemm16 with a bf16 output, and the stride-1 _dgrad16; ex->ez is then the consumer
 *                          BatchNorm's bf16 pre-activation): the ReLU mask recomputed with the forward pass's expression,
 *                          g stored, column sums of g and g*xhat as channel-major partials -- that BatchNorm's backward
 *                          reduce pass inside the d-input product (the product then runs un-split);
 *   scnattn_wgrad16_3x3    dw [Cout][3][3][Cin] fp32 of a stride-1 3x3 convolution from bf16 maps: contraction over pixels
 *                          with both operands read through the transposing LDS load (ds_read_b64_tr_b16), halo staged once;
 *   scnattn_wgrad16_rows   dw[co][ci] (ldo) = sum_r dy[r][co] * x[src(r)][ci]: 1x1 weight gradients (gs = 0), the strided
 *                          downsample (gs = stride, goh = gow = 0) and ONE TAP of a stride-2 3x3 (gs = 2, goh = dh - 1,
 *                          gow = dw - 1, dw offset by tap*Cin, ldo = 9*Cin); Cin, Cout multiples of 64. */
typedef struct scnattn_weight_desc {
    const float* src; void* dst; void* dst_t; int cout, taps, cin, pad;
} scnattn_weight_desc;
int scnattn_bf16_weights(void* stream, int n, const scnattn_weight_desc* desc, const int* tile_prefix, int total_tiles);
int scnattn_cgemm16(void* stream, int M, int N, int K, const void* A, long lda, const void* B, long ldb, float beta, void* C,
                    long ldc, int out_bf16, float* ws, long ws_floats, const scnattn_conv_extra* ex);
int scnattn_conv3x3_fwd16(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const void* x, const void* w,
                          void* y, const scnattn_conv_extra* ex, float* ws, long ws_floats);
int scnattn_conv3x3_dgrad16(void* stream, int N, int Hi, int Wi, int Cin, int Cout, int stride, const void* dy, const void* wt,
                            void* dx, const scnattn_conv_extra* ex, float* ws, long ws_floats);
int scnattn_wgrad16_3x3(void* stream, int N, int H, int W, int Cin, int Cout, const void* dy, const void* x, float* dw,
                        float* ws, long ws_floats, int k_slices);
int scnattn_wgrad16_rows(void* stream, int R, int Cin, int Cout, const void* dy, const void* x, long src_rows, float* dw,
                         long ldo, int gs, int gHi, int gWi, int gHo, int gWo, int goh, int gow, float* ws, long ws_floats,
                         int k_slices);
/* ---- one mixed-precision Bottleneck from ONE call per direction (csrc/block16.cpp) -------------------------------------------
 * torchvision's Bottleneck.forward (conv1-bn1-relu-conv2-bn2-relu-conv3-bn3 [+ downsample] + identity, relu; the blocks of the
 * trunk built at models/encoders/caption.py:17-22) and its gradient as the launch sequences above, enqueued by the library:
 * the bf16 step is bound by what the HOST spends per launch (~25 calls per block and direction from Python), not by the GPU.
 * The library still allocates nothing: the caller sizes three buffers with scnattn_block16_sizes and owns them.
 *   save  (bf16): z1 a1 | z2 a2 | z3 out | [zd idn]   -- everything the backward pass re-reads; `out` (the block's result,
 *                 [N*Ho*Wo][4p]) starts at out_offset elements;
 *   stats (fp32): [4][2][4p]  mean, invstd of bn1, bn2, bn3, downsample.1 (written by the forward call);
 *   tmp   (bf16, backward): dres dz3 | dz2 | dz1 | [dzd dx dxd];  dgb (fp32, backward): [4][2][4p]  d beta, d gamma.
 * BatchNorm index: 0 bn1, 1 bn2, 2 bn3, 3 downsample.1; convolution index: 0 conv1, 1 conv2, 2 conv3, 3 downsample.0.
 * shift[i]: conditioning shift of BatchNorm i (scnattn_bn_apply_fin); w / wt: the bf16 copies of scnattn_bf16_weights.
 * Backward: weight gradients with a non-null dw[i] are written there (fp32, the parameter's layout); with side_stream they run
 * on it, forked from `stream` by events (the caller joins it and keeps save / tmp / x alive until then).  dx_out receives the
 * gradient of the block input: for an identity block it is dres (accumulated in place); with a downsample the caller adds
 * dxd_out (the [N*Ho*Wo][Cin] gradient through downsample.0) into the strided rows of dx_out itself. */
typedef struct scnattn_block16 {
    int N, Cin, Hi, Wi, p, stride, has_down, pad;
    const float* gamma[4]; const float* beta[4]; float* run_mean[4]; float* run_var[4]; const float* shift[4];
    float eps[4], momentum[4];
    const void* w[4]; const void* wt[4];
    float* ws; long ws_floats; float* part; float* bnpart; long bnpart_floats;
    const void* x; void* save; float* stats;
    /* backward only */
    const void* dout; void* tmp; float* dgb; float* dw[4]; int need_dx, pad2;
    void* side_stream; float* side_ws; long side_ws_floats;
} scnattn_block16;
int scnattn_block16_sizes(const scnattn_block16* b, long* save_elems, long* out_offset, long* stats_floats, long* tmp_elems,
                          long* dgb_floats);
int scnattn_block16_fwd(void* stream, const scnattn_block16* b);
int scnattn_block16_bwd(void* stream, const scnattn_block16* b, void** dx_out, void** dxd_out);
/* scnattn_bn_stats (fp32 maps) that also writes the folded {scale, shift} pairs [C][2] for a consumer's prologue */
int scnattn_bn_stats_fold(void* stream, int R, int C, const void* x, float eps, float momentum, float* partial,
                          float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                          const float* beta, float* ss_out);

/* ---- data-parallel gradient exchange (the reference has no distributed code) ---------------------
 * One process per GPU.  RCCL SUM all-reduce of gradient buckets on a library-owned communication stream, ordered
 * against the caller's compute stream by HIP events, so a bucket's reduction overlaps the rest of the backward pass:
 *   id    = scnattn_dp_unique_id() on rank 0, sent to the other ranks by any out-of-band channel (128 bytes);
 *   comm  = scnattn_dp_comm_create(id, world, rank) on every rank (current HIP device; collective);
 *   scnattn_dp_comm_allreduce_bucket(comm, compute_stream, buf, n): in-place SUM of n floats once everything enqueued
 *           on compute_stream so far has run; returns at once;
 *   scnattn_dp_comm_finish(comm, compute_stream): compute_stream waits for every bucket handed over so far
 *           (call before the optimizer reads the gradients; the 1/world scale is folded into scnattn_clamp_adam);
 *   scnattn_dp_comm_set_stream(comm, stream): optional -- reduce on a caller-owned stream instead of the library's
 *     (HIP maps streams onto a few hardware queues; a host that has probed which of its streams really run beside
 *     the compute stream hands that one over; NULL returns to the library's own stream);
 *   scnattn_dp_comm_destroy(comm).
 * Return codes >= 1000 are 1000 + ncclResult_t. */
typedef struct scnattn_dp_comm scnattn_dp_comm;
int scnattn_dp_unique_id(char out[128]);
int scnattn_dp_comm_create(const char id[128], int world, int rank, scnattn_dp_comm** out);
int scnattn_dp_comm_allreduce_bucket(scnattn_dp_comm* comm, void* compute_stream, float* buf, long n);
int scnattn_dp_comm_finish(scnattn_dp_comm* comm, void* compute_stream);
int scnattn_dp_comm_set_stream(scnattn_dp_comm* comm, void* stream);
int scnattn_dp_comm_world(const scnattn_dp_comm* comm);
int scnattn_dp_comm_destroy(scnattn_dp_comm* comm);

/* utils/optimizer.py:1-11 (element-wise clamp) fused with torch.optim.Adam's update
 * (trains/attention_scn.py:244-252); g is first scaled by gscale (1/world for data parallel). */
int scnattn_clamp_adam(void* stream, long n, float* p, const float* g, float* m, float* v, double lr,
                       double beta1, double beta2, double eps, int step, double clip, double gscale);

#ifdef __cplusplus
}
#endif
#endif /* SCNATTN_H */
