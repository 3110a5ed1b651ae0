// Whole-sequence drivers: one C call enqueues every kernel of the teacher-forced decoder forward
// (models/decoders/attention_scn.py:124-156 / pure_scn.py:114-138) or of its gradient, on the
// caller's stream.  The time loop lives here, not in Python, so the host cost per timestep is a
// handful of hipLaunchKernel calls and nothing else.
//
// Sequence-level restructuring (algebraically identical to the reference):
//   * time-invariant projections are hoisted out of the loop: att1 = encoder_att(enc),
//     qx = s.Wb, qh = s.Hb; the embedding half of u.Wa is batched over all steps (ex), and fc runs
//     once over all (b,t) rows after the loop;
//   * in the backward pass the per-step weight-gradient GEMMs collapse into one GEMM per weight over
//     the stacked (t,b) rows kept in `scratch`, and d att1 / d enc are formed once after the loop.
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/scnattn.h"
#include "common.h"
#include "kernels.h"

namespace scn {

int g_ksplit_scale = 0;  // 0 = auto; >0 forces ksplit for every skinny launch (tuning/testing)
int g_profile = 0;       // 1: bracket the recurrence loops with HIP events (scnattn_profile_collect)
int g_dec_bf16 = 0;          // 1 (2: + bf16 matrix instruction in the skinny GEMMs): BASELINE configs[4] flavour -- the operands the recurrence STREAMS every step (recurrent
                             //    weights, att1, the encoder map) are kept as bf16 copies, made once per call; products
                             //    accumulate in fp32, softmax / LSTM state / master weights / every gradient stay fp32

// ---- optional in-stream timing of the recurrence loops (bench.py's roofline figure) ------------------
struct LoopEvent { hipEvent_t a, b; int kind, steps; };
static std::mutex g_prof_mu;
static std::vector<LoopEvent> g_prof;

static hipEvent_t prof_begin(hipStream_t st) {
    if (!g_profile) return nullptr;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, st);
    return e;
}
static void prof_end(hipStream_t st, hipEvent_t a, int kind, int steps) {
    if (!a) return;
    hipEvent_t b = nullptr;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
    (void)hipEventRecord(b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(LoopEvent{a, b, kind, steps});
}

// out[0..5] = {fwd loop ms, fwd steps, bwd loop ms, bwd steps, attn_context ms, attn_context launches}
// summed since the last collect (the last pair only when option "profile" was 2)
int profile_collect(double* out) {
    std::vector<LoopEvent> ev;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ev.swap(g_prof);
    }
    for (int i = 0; i < 6; ++i) out[i] = 0.0;
    for (auto& e : ev) {
        float ms = 0.f;
        SCN_HIP(hipEventSynchronize(e.b));
        SCN_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        out[e.kind * 2] += ms;
        out[e.kind * 2 + 1] += e.steps;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    return 0;
}

namespace {

struct Carver {
    float* base;
    size_t off = 0;  // in floats
    explicit Carver(float* b) : base(b) {}
    float* take(size_t n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) & ~size_t(63);  // 256-byte granules keep every buffer 16-byte aligned
        return p;
    }
};

struct Saved {
    float *att1, *qx, *qh, *ex, *emb_tm, *mean_enc, *Hs, *Cs, *att2_all, *alpha_tm, *awe_all, *gate_all, *z_all,
        *pa_all, *ph_all, *gates_all, *tanhc_all, *Hd_bm, *rowmask, *alphaq_tm;
    float *att1h, *ench;      // bf16 copies (raw 16-bit elements) of att1 and of the encoder map, bf16 mode only
};

constexpr long GEMM_WS_FLOATS = 24L << 20;  // 96 MiB of split-K partials for the big GEMMs (d fc.weight: 4 slabs of 10000 x 512)

struct FwdScratch {
    float *WcatA, *WD, *slabA, *e, *slabC, *xcat, *slabD, *gws, *y;
    float *WcatAh, *WDh, *WaMh;      // bf16 copies of the per-step weight operands
};

struct BwdScratch {
    float *WDb, *WaTz, *WcatT, *dHd_bm, *dhfc_tm, *dr_all, *dpx_all, *dcat_all, *dawe_all, *de_all, *dalpha, *dc,
        *dqx_acc, *dqh_acc, *sDb, *sZ, *sH, *datt1, *dwpart, *dwtmp, *demb_tm, *dmean, *dh0, *mx_all, *gws, *present,
        *dalphaq, *dy, *gws2, *WDbh, *WaTzh, *WcatTh;
};

inline size_t sz(long a, long b = 1, long c = 1, long d = 1) { return (size_t)a * b * c * d; }

// Q > 0: pooled path (encoder_out given as its un-pooled source map x [B][Q][E])
size_t carve_saved(const scnattn_dims& d, int Q, float* base, Saved& s) {
    Carver c(base);
    const int B = d.B, P = d.P, E = d.E, A = d.A, D = d.D, F4 = 4 * d.F, T = d.T;
    s.att1 = d.has_att ? c.take(sz(B, P, A)) : nullptr;
    s.qx = c.take(sz(B, F4));
    s.qh = c.take(sz(B, F4));
    s.ex = c.take(sz(T, B, F4));
    s.emb_tm = c.take(sz(T, B, d.M));
    s.mean_enc = c.take(sz(B, E));
    s.Hs = c.take(sz(T + 1, B, D));
    s.Cs = c.take(sz(T + 1, B, D));
    s.att2_all = d.has_att ? c.take(sz(T, B, A)) : nullptr;
    s.alpha_tm = d.has_att ? c.take(sz(T, B, P)) : nullptr;
    s.awe_all = d.has_att ? c.take(sz(T, B, E)) : nullptr;
    s.gate_all = d.has_att ? c.take(sz(T, B, E)) : nullptr;
    s.z_all = d.has_att ? c.take(sz(T, B, E)) : nullptr;
    s.pa_all = c.take(sz(T, B, F4));
    s.ph_all = c.take(sz(T, B, F4));
    s.gates_all = c.take(sz(T, B, 4 * D));
    s.tanhc_all = c.take(sz(T, B, D));
    s.Hd_bm = c.take(sz(B, T, D));
    s.rowmask = c.take(sz(B, T));
    s.alphaq_tm = (d.has_att && Q > 0) ? c.take(sz(T, B, Q)) : nullptr;
    s.att1h = d.has_att ? c.take((sz(B, P, A) + 1) / 2) : nullptr;
    s.ench = d.has_att ? c.take((sz(B, Q > 0 ? Q : P, E) + 1) / 2) : nullptr;
    return c.off * sizeof(float);
}

inline int ncatA(const scnattn_dims& d) { return d.has_att ? d.A + d.E + 4 * d.F : 4 * d.F; }

size_t carve_fwd(const scnattn_dims& d, int Q, float* base, FwdScratch& s) {
    Carver c(base);
    const int B = d.B, D = d.D, F = d.F, NA = ncatA(d);
    s.WcatA = c.take(sz(D, NA));
    s.WD = c.take(sz(4, 2 * F, D));
    s.slabA = c.take(sz(SCN_MAX_KSPLIT, B, NA));
    s.e = d.has_att ? c.take(sz(B, d.P)) : nullptr;
    s.slabC = d.has_att ? c.take(sz(SCN_MAX_KSPLIT, B, 4 * F)) : nullptr;
    s.xcat = c.take(sz(B, 4, 2 * F));
    s.slabD = c.take(sz(SCN_MAX_KSPLIT, 4, B, D));
    s.gws = c.take(GEMM_WS_FLOATS);
    s.y = (d.has_att && Q > 0) ? c.take(sz(B, Q, d.A)) : nullptr;
    s.WcatAh = c.take((sz(D, NA) + 1) / 2);
    s.WDh = c.take((sz(4, 2 * F, D) + 1) / 2);
    s.WaMh = d.has_att ? c.take((sz(d.E, 4 * F) + 1) / 2) : nullptr;
    return c.off * sizeof(float);
}

size_t carve_bwd(const scnattn_dims& d, int Q, float* base, BwdScratch& s) {
    Carver c(base);
    const int B = d.B, P = d.P, E = d.E, A = d.A, D = d.D, F = d.F, F4 = 4 * d.F, T = d.T, NC = ncatA(d);
    s.WDb = c.take(sz(4, D, 2 * F));
    s.WaTz = d.has_att ? c.take(sz(F4, E)) : nullptr;
    s.WcatT = c.take(sz(NC, D));
    s.dHd_bm = c.take(sz(B, T, D));
    s.dhfc_tm = c.take(sz(T, B, D));
    s.dr_all = c.take(sz(T, B, 4 * D));
    s.dpx_all = c.take(sz(T, B, F4));
    s.dcat_all = c.take(sz(T, B, NC));
    s.dawe_all = d.has_att ? c.take(sz(T, B, E)) : nullptr;
    s.de_all = d.has_att ? c.take(sz(T, B, P)) : nullptr;
    s.dalpha = d.has_att ? c.take(sz(B, P)) : nullptr;
    s.dc = c.take(sz(B, D));
    s.dqx_acc = c.take(sz(B, F4));
    s.dqh_acc = c.take(sz(B, F4));
    s.sDb = c.take(sz(SCN_MAX_KSPLIT, 4, B, 2 * F));
    s.sZ = d.has_att ? c.take(sz(SCN_MAX_KSPLIT, B, E)) : nullptr;
    s.sH = c.take(sz(SCN_MAX_KSPLIT, B, D));
    s.datt1 = d.has_att ? c.take(sz(B, P, A)) : nullptr;
    s.dwpart = d.has_att ? c.take(sz(attn_datt1_post_blocks(B, P), A + 1)) : nullptr;
    s.dwtmp = d.has_att ? c.take(sz(A + 1)) : nullptr;
    s.demb_tm = c.take(sz(T, B, d.M));
    s.dmean = c.take(sz(B, E));
    s.dh0 = c.take(sz(B, D));
    s.mx_all = c.take(sz(T, B, F4));
    s.present = c.take(sz(d.V));
    s.gws = c.take(GEMM_WS_FLOATS);
    s.dalphaq = (d.has_att && Q > 0) ? c.take(sz(B, Q)) : nullptr;
    s.dy = (d.has_att && Q > 0) ? c.take(sz(B, Q, A)) : nullptr;
    s.gws2 = c.take(GEMM_WS_FLOATS);      // split-K partials of the weight-gradient stream
    s.WDbh = c.take((sz(4, D, 2 * F) + 1) / 2);
    s.WaTzh = d.has_att ? c.take((sz(F4, E) + 1) / 2) : nullptr;
    s.WcatTh = c.take((sz(NC, D) + 1) / 2);
    return c.off * sizeof(float);
}

int check_dims(const scnattn_dims* d) {
    SCN_ARG(d, "dims is NULL");
    SCN_ARG(d->B > 0 && d->P > 0 && d->E > 0 && d->D > 0 && d->F > 0 && d->M > 0 && d->S > 0 && d->V > 0,
            "dims must be positive");
    SCN_ARG(d->T > 0 && d->L >= d->T, "need 0 < T <= L");
    SCN_ARG(!d->has_att || d->A > 0, "attention_dim must be positive");
    return 0;
}

int check_bt(const scnattn_dims* d, const int32_t* bt) {
    SCN_ARG(bt, "bt_host is NULL");
    for (int t = 0; t < d->T; ++t) {
        SCN_ARG(bt[t] >= 1 && bt[t] <= d->B, "bt_host[t] must be in [1, B]");
        SCN_ARG(t == 0 || bt[t] <= bt[t - 1], "bt_host must be non-increasing (captions sorted by length)");
    }
    SCN_ARG(bt[0] == d->B, "bt_host[0] must equal B (every caption decodes at least one step)");
    return 0;
}

int check_pool(const scnattn_dims* d, const scnattn_pool* p, PoolDesc& out) {
    out = PoolDesc{0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!p) return 0;
    // (without attention only the initial state reads encoder_out: its pixel mean becomes a weighted mean of x)
    SCN_ARG(p->Q > 0 && p->Q <= d->P && p->qtap_max > 0 && p->qtap_max <= 64, "scnattn_pool: bad Q / qtap_max");
    SCN_ARG(p->tap_idx && p->tap_w && p->qtap_idx && p->qtap_w && p->col_w, "scnattn_pool: null table");
    SCN_ARG((!d->has_att || d->A % 4 == 0) && d->E % 4 == 0,
            "scnattn_pool: attention_dim and encoder_dim must be multiples of 4");
    out = PoolDesc{p->Q, p->qtap_max, p->tap_idx, p->tap_w, p->qtap_idx, p->qtap_w, p->col_w};
    return 0;
}

// bf16 storage mode: every converted buffer must be a whole number of 4-element groups
inline bool bf16_mode(const scnattn_dims& d) {
    return g_dec_bf16 && d.D % 4 == 0 && d.F % 4 == 0 && d.E % 4 == 0 && (!d.has_att || d.A % 4 == 0);
}
// element offset into a buffer that holds fp32 or bf16 elements
inline const void* eoff(const void* base, long elems, bool bf) {
    return reinterpret_cast<const char*>(base) + elems * (bf ? 2 : 4);
}

inline int pick(int rows, int N, int K, int groups) {
    if (g_ksplit_scale > 0) {
        int ks = g_ksplit_scale;
        const int kmax = K / 8 > 0 ? K / 8 : 1;
        if (ks > kmax) ks = kmax;
        if (ks > SCN_MAX_KSPLIT) ks = SCN_MAX_KSPLIT;
        return ks;
    }
    return skinny_pick_ksplit(rows, N, K, groups);
}

}  // namespace

// The recurrence bodies are written as a function of (stream, first row, row count).  Round 1 ran them as two
// independent half-batch chains on two streams (no kernel of the loop mixes batch rows); bit-identical but SLOWER (52.8
// vs 47.8 us per step: the step kernels are latency-bound, a half-batch launch takes as long as a full one), so since
// round 3 there is one chain on the caller's stream and the machinery is gone (DESIGN.md 6 keeps the numbers).
namespace {

template <class Body>   // body(stream, first_row, max_rows) -> 0 or error code
int run_chains(hipStream_t st, int /*kind*/, int B, Body&& body) { return body(st, 0, B); }

}  // namespace

int seq_workspace(const scnattn_dims* d, const scnattn_pool* pool, size_t* saved_bytes, size_t* scratch_bytes) {
    SCN_TRY(check_dims(d));
    PoolDesc pd;
    SCN_TRY(check_pool(d, pool, pd));
    Saved s;
    FwdScratch f;
    BwdScratch b;
    const size_t sv = carve_saved(*d, pd.Q, nullptr, s);
    const size_t fw = carve_fwd(*d, pd.Q, nullptr, f);
    const size_t bw = carve_bwd(*d, pd.Q, nullptr, b);
    if (saved_bytes) *saved_bytes = sv;
    if (scratch_bytes) *scratch_bytes = fw > bw ? fw : bw;
    return 0;
}

int seq_fwd(hipStream_t st, const scnattn_dims* dp, const scnattn_params* w, const float* enc, const float* tags,
            const int64_t* caps, const int32_t* dl_dev, const int32_t* bt, const float* drop_mask, float* saved,
            float* scratch, float* preds, float* alphas, const scnattn_pool* pool) {
    SCN_TRY(check_dims(dp));
    SCN_TRY(check_bt(dp, bt));
    const scnattn_dims& d = *dp;
    PoolDesc pd;
    SCN_TRY(check_pool(dp, pool, pd));
    const int Q = pd.Q;             // > 0: `enc` is the un-pooled map x [B][Q][E]
    SCN_ARG(w && enc && tags && caps && dl_dev && saved && scratch && preds, "seq_fwd: null argument");
    SCN_ARG(!d.has_att || alphas, "seq_fwd: alphas is NULL");
    const int B = d.B, P = d.P, E = d.E, A = d.A, D = d.D, F = d.F, F4 = 4 * F, M = d.M, T = d.T;
    const int NA = ncatA(d), colph = d.has_att ? A + E : 0;
    Saved s;
    FwdScratch f;
    carve_saved(d, Q, saved, s);
    carve_fwd(d, Q, scratch, f);

    // ---- weight re-layout: every per-step contraction streams a row-major [K][N] matrix ----------
    if (d.has_att) {
        SCN_TRY(transpose2d(st, A, D, w->attention_decoder_att_weight, D, f.WcatA, NA));       // Wd^T
        SCN_TRY(transpose2d(st, E, D, w->f_beta_weight, D, f.WcatA + A, NA));                  // Wbeta^T
    }
    SCN_TRY(copy2d(st, D, F4, w->decode_step_weight_ha, F4, f.WcatA + colph, NA));             // Ha
    for (int g = 0; g < 4; ++g) {
        float* wd = f.WD + (long)g * 2 * F * D;
        SCN_TRY(transpose2d(st, D, F, w->decode_step_weight_ic + g * F, F4, wd, D));           // Wc_g^T
        SCN_TRY(transpose2d(st, D, F, w->decode_step_weight_hc + g * F, F4, wd + (long)F * D, D));  // Hc_g^T
    }

    const bool bf = bf16_mode(d);
    const int bfm = bf ? (g_dec_bf16 >= 2 ? 2 : 1) : 0;     // 2: the skinny products on the bf16 matrix instruction
    if (bf) {
        SCN_TRY(f32_to_bf16(st, sz(D, NA), f.WcatA, f.WcatAh));
        SCN_TRY(f32_to_bf16(st, sz(4, 2 * F, D), f.WD, f.WDh));
        if (d.has_att) SCN_TRY(f32_to_bf16(st, sz(E, F4), w->decode_step_weight_ia + (long)M * F4, f.WaMh));
    }
    const void* WcatA = bf ? (const void*)f.WcatAh : f.WcatA;
    const void* WD = bf ? (const void*)f.WDh : f.WD;
    const void* WaM = bf ? (const void*)f.WaMh : (d.has_att ? w->decode_step_weight_ia + (long)M * F4 : nullptr);

    // ---- time-invariant pieces -----------------------------------------------------------------
    if (d.has_att && Q > 0) {
        // att1 = pool(x) . We^T + be = pool(x . We^T) + be: the projection runs on Q rows per image, not P
        SCN_TRY(sgemm_ws(st, false, true, B * Q, A, E, 1.f, enc, E, w->attention_encoder_att_weight, E, 0.f, f.y, A,
                      nullptr, nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
        SCN_TRY(pool_expand(st, B, P, A, pd, f.y, w->attention_encoder_att_bias, s.att1));
    } else if (d.has_att) {
        SCN_TRY(sgemm_ws(st, false, true, B * P, A, E, 1.f, enc, E, w->attention_encoder_att_weight, E, 0.f, s.att1, A,
                      w->attention_encoder_att_bias, nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
    }
    if (bf && d.has_att) {     // what the step kernels stream: bf16 copies of att1 and of the map the context is summed over
        SCN_TRY(f32_to_bf16(st, sz(B, P, A), s.att1, s.att1h));
        SCN_TRY(f32_to_bf16(st, sz(B, Q > 0 ? Q : P, E), enc, s.ench));
    }
    const void* att1_s = (bf && d.has_att) ? (const void*)s.att1h : s.att1;
    const void* enc_s = (bf && d.has_att) ? (const void*)s.ench : enc;
    SCN_TRY(sgemm_ws(st, false, false, B, F4, d.S, 1.f, tags, d.S, w->decode_step_weight_ib, F4, 0.f, s.qx, F4, nullptr,
                  nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
    SCN_TRY(sgemm_ws(st, false, false, B, F4, d.S, 1.f, tags, d.S, w->decode_step_weight_hb, F4, 0.f, s.qh, F4, nullptr,
                  nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
    SCN_TRY(gather_rows_tm(st, B, T, d.L, M, (const long long*)caps, w->embedding_weight, d.V, s.emb_tm));
    SCN_TRY(sgemm_ws(st, false, false, T * B, F4, M, 1.f, s.emb_tm, M, w->decode_step_weight_ia, F4, 0.f, s.ex, F4,
                  nullptr, nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
    if (Q > 0) SCN_TRY(weighted_rows(st, B, Q, E, enc, pd.col_w, s.mean_enc));
    else SCN_TRY(mean_pixels(st, B, P, E, enc, s.mean_enc));
    SCN_TRY(sgemm_ws(st, false, true, B, D, E, 1.f, s.mean_enc, E, w->init_h_weight, E, 0.f, s.Hs, D, w->init_h_bias,
                  nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
    SCN_TRY(sgemm_ws(st, false, true, B, D, E, 1.f, s.mean_enc, E, w->init_c_weight, E, 0.f, s.Cs, D, w->init_c_bias,
                  nullptr, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));

    // ---- the recurrence --------------------------------------------------------------------------
    const long BD = (long)B * D;
    hipEvent_t ev0 = prof_begin(st);
    SCN_TRY(run_chains(st, 0, B, [&](hipStream_t cs, int r0, int rmax) -> int {
        const void* enc_c = eoff(enc_s, (long)r0 * (Q > 0 ? Q : P) * E, bf && d.has_att);
        const void* att1_c = d.has_att ? eoff(att1_s, (long)r0 * P * A, bf) : nullptr;
        float* slabA = f.slabA + (long)r0 * NA;
        float* slabC = d.has_att ? f.slabC + (long)r0 * F4 : nullptr;
        float* slabD = f.slabD + (long)r0 * D;
        float* e_c = d.has_att ? f.e + (long)r0 * P : nullptr;
        float* xcat = f.xcat + (long)r0 * 8 * F;
        const float* qx = s.qx + (long)r0 * F4;
        const float* qh = s.qh + (long)r0 * F4;
        for (int t = 0; t < T; ++t) {
            const int bt_ = (bt[t] - r0 < rmax) ? bt[t] - r0 : rmax;   // rows of this chain still decoding
            if (bt_ <= 0) break;                                       // bt is non-increasing
            const long rowT = (long)t * B + r0;                        // first row of this chain in [T][B][.] buffers
            const float* h = s.Hs + rowT * D;
            const float* c = s.Cs + rowT * D;
            const int ksA = pick(bt_, NA, D, 1);
            SCN_TRY(skinny_gemm(cs, bt_, NA, D, 1, h, D, 0, WcatA, NA, 0, slabA, NA, 0, (long)B * NA, ksA, bfm));
            Slabs pz{nullptr, 0, 0, 0};
            bool mixed = false;
            if (d.has_att) {
                float* alpha_out = alphas + (long)r0 * T * P + (long)t * P;
                if (Q > 0) {
                    SCN_TRY(attn_scores(cs, bt_, P, A, att1_c, Slabs{slabA, ksA, (long)B * NA, NA},
                                        w->attention_decoder_att_bias, w->attention_full_att_weight,
                                        w->attention_full_att_bias, e_c, s.att2_all + rowT * A, bf));
                    hipEvent_t evc = g_profile >= 2 ? prof_begin(cs) : nullptr;
                    SCN_TRY(attn_context_pooled(cs, bt_, P, E, enc_c, pd, e_c, Slabs{slabA + A, ksA, (long)B * NA, NA},
                                                w->f_beta_bias, alpha_out, (long)T * P, s.alpha_tm + rowT * P,
                                                s.alphaq_tm + rowT * Q, s.awe_all + rowT * E, s.gate_all + rowT * E,
                                                s.z_all + rowT * E, bf));
                    prof_end(cs, evc, 2, 1);
                } else {
                    SCN_TRY(attn_scores(cs, bt_, P, A, att1_c, Slabs{slabA, ksA, (long)B * NA, NA},
                                        w->attention_decoder_att_bias, w->attention_full_att_weight,
                                        w->attention_full_att_bias, e_c, s.att2_all + rowT * A, bf));
                    hipEvent_t evc = g_profile >= 2 ? prof_begin(cs) : nullptr;   // per-launch timing of the dominant kernel
                    SCN_TRY(attn_context(cs, bt_, P, E, enc_c, e_c, Slabs{slabA + A, ksA, (long)B * NA, NA},
                                         w->f_beta_bias, alpha_out, (long)T * P, s.alpha_tm + rowT * P,
                                         s.awe_all + rowT * E, s.gate_all + rowT * E, s.z_all + rowT * E, bf));
                    prof_end(cs, evc, 2, 1);
                }
                // z . Wa[M:], and -- inside the same launch, by the workgroup that arrives last at each 32-column unit --
                // the SCN mix that consumes it (scn_cell.py:73-86); stand-alone kernel when the launch cannot take it
                const int ksC = pick(bt_, F4, E, 1);
                SkinnyTail mix{2, bt_, 0, F4, {s.ex + rowT * F4, qx, qh, nullptr},
                               {s.pa_all + rowT * F4, s.ph_all + rowT * F4, xcat, nullptr}, 0,
                               Slabs{slabA + colph, ksA, (long)B * NA, NA}};
                SCN_TRY(skinny_gemm(cs, bt_, F4, E, 1, s.z_all + rowT * E, E, 0, WaM, F4, 0, slabC, F4, 0, (long)B * F4, ksC,
                                    bfm, &mix, &mixed));
                pz = Slabs{slabC, ksC, (long)B * F4, F4};
            }
            if (!mixed)
                SCN_TRY(scn_mix_fwd(cs, bt_, F4, pz, s.ex + rowT * F4, Slabs{slabA + colph, ksA, (long)B * NA, NA}, qx, qh,
                                    s.pa_all + rowT * F4, s.ph_all + rowT * F4, xcat));
            // [mx | mh] . [Wc; Hc] for the four gates, and the LSTM update of the units each 32-column tile feeds
            const int ksD = pick(bt_, D, 2 * F, 4);
            SkinnyTail cell{1, bt_, 0, D, {w->decode_step_bias_ih, w->decode_step_bias_hh, c, nullptr},
                            {s.gates_all + rowT * 4 * D, s.Cs + (rowT + B) * D, s.Hs + (rowT + B) * D, s.tanhc_all + rowT * D}, 0,
                            Slabs{nullptr, 0, 0, 0}};
            bool celled = false;
            SCN_TRY(skinny_gemm(cs, bt_, D, 2 * F, 4, xcat, 8 * F, 2 * F, WD, D, (long)2 * F * D, slabD, D, BD, 4 * BD,
                                ksD, bfm, &cell, &celled));
            if (!celled)
                SCN_TRY(lstm_fwd(cs, bt_, D, Slabs{slabD, ksD, 4 * BD, D}, BD, w->decode_step_bias_ih,
                                 w->decode_step_bias_hh, c, s.gates_all + rowT * 4 * D, s.Cs + (rowT + B) * D,
                                 s.Hs + (rowT + B) * D, s.tanhc_all + rowT * D));
        }
        return 0;
    }));

    prof_end(st, ev0, 0, T);

    // ---- dropout + fc over all (b,t) rows at once ------------------------------------------------
    SCN_TRY(hidden_to_bm(st, B, T, D, dl_dev, s.Hs + BD, drop_mask, s.Hd_bm, s.rowmask));
    SCN_TRY(sgemm_ws(st, false, true, B * T, d.V, D, 1.f, s.Hd_bm, D, w->fc_weight, D, 0.f, preds, d.V, w->fc_bias,
                  s.rowmask, 1, 0, 0, 0, f.gws, GEMM_WS_FLOATS));
    return 0;
}

// `wst`: optional second stream for the weight gradients.  Nothing downstream of this call needs d loss / d weight
// before the optimizer step, while d encoder_out heads the whole encoder backward pass: with `wst` the fc weight
// gradient runs beside the (latency-bound) reverse recurrence and the post-loop weight-gradient GEMMs beside whatever
// the caller enqueues next on `st`.  Ordering is by events: `wst` waits for what it reads, `st` never waits for `wst`
// -- the CALLER joins `wst` before it reads `g`.  NULL (or == st): everything in order on `st`.
int seq_bwd(hipStream_t st, hipStream_t wst, const scnattn_dims* dp, const scnattn_params* w, const float* enc,
            const float* tags, const int64_t* caps, const int32_t* dl_dev, const int32_t* bt, const float* drop_mask,
            const float* saved, float* scratch, const float* dpreds, const float* dalphas, const scnattn_params* g,
            float* denc, float* dtags, const scnattn_pool* pool) {
    SCN_TRY(check_dims(dp));
    SCN_TRY(check_bt(dp, bt));
    const scnattn_dims& d = *dp;
    PoolDesc pd;
    SCN_TRY(check_pool(dp, pool, pd));
    const int Q = pd.Q;             // > 0: `enc` is x [B][Q][E] and `denc` receives d x [B][Q][E]
    SCN_ARG(w && g && enc && tags && caps && dl_dev && saved && scratch && dpreds, "seq_bwd: null argument");
    const int B = d.B, P = d.P, E = d.E, A = d.A, D = d.D, F = d.F, F4 = 4 * F, M = d.M, T = d.T, V = d.V;
    const int NC = ncatA(d);  // columns of the concatenated [dph | dgpre | datt2] operand
    const int TB = T * B;
    const long BD = (long)B * D;
    Saved s;
    BwdScratch k;
    carve_saved(d, Q, const_cast<float*>(saved), s);
    carve_bwd(d, Q, scratch, k);
    const bool two = wst && wst != st;
    hipStream_t ws = two ? wst : st;            // stream of the weight gradients
    float* wgws = two ? k.gws2 : k.gws;         // ... and its split-K workspace
    // ws_after_main(): everything enqueued on `st` so far happens-before what is enqueued on `ws` next
    auto ws_after_main = [&]() -> int {
        if (!two) return 0;
        hipEvent_t e = nullptr;
        SCN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hipError_t r = hipEventRecord(e, st);
        if (r == hipSuccess) r = hipStreamWaitEvent(ws, e, 0);
        (void)hipEventDestroy(e);               // released once the wait has been satisfied
        SCN_HIP(r);
        return 0;
    };

    // ---- fc / dropout ----------------------------------------------------------------------------
    SCN_TRY(ws_after_main());
    if (g->fc_weight)
        SCN_TRY(sgemm_ws(ws, true, false, V, D, B * T, 1.f, dpreds, V, s.Hd_bm, D, 0.f, g->fc_weight, D, nullptr, nullptr,
                      1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
    if (g->fc_bias)  // only rows that were decoded carry the bias
        SCN_TRY(colsum_masked(ws, B * T, V, dpreds, V, s.rowmask, g->fc_bias, 0.f));
    SCN_TRY(sgemm_ws(st, false, false, B * T, D, V, 1.f, dpreds, V, w->fc_weight, D, 0.f, k.dHd_bm, D, nullptr, nullptr, 1,
                  0, 0, 0, k.gws, GEMM_WS_FLOATS));
    SCN_TRY(hidden_from_bm(st, B, T, D, dl_dev, k.dHd_bm, drop_mask, k.dhfc_tm));

    // ---- transposed weight layouts for the backward contractions ------------------------------------
    for (int gi = 0; gi < 4; ++gi) {
        float* wd = k.WDb + (long)gi * D * 2 * F;
        SCN_TRY(copy2d(st, D, F, w->decode_step_weight_ic + gi * F, F4, wd, 2 * F));
        SCN_TRY(copy2d(st, D, F, w->decode_step_weight_hc + gi * F, F4, wd + F, 2 * F));
    }
    SCN_TRY(transpose2d(st, D, F4, w->decode_step_weight_ha, F4, k.WcatT, D));  // Ha^T : [4F][D]
    if (d.has_att) {
        SCN_TRY(transpose2d(st, E, F4, w->decode_step_weight_ia + (long)M * F4, F4, k.WaTz, E));  // Wa[M:]^T
        SCN_TRY(copy2d(st, E, D, w->f_beta_weight, D, k.WcatT + (long)F4 * D, D));
        SCN_TRY(copy2d(st, A, D, w->attention_decoder_att_weight, D, k.WcatT + (long)(F4 + E) * D, D));
    }
    const bool bf = bf16_mode(d);
    const int bfm = bf ? (g_dec_bf16 >= 2 ? 2 : 1) : 0;     // 2: the skinny products on the bf16 matrix instruction
    if (bf) {
        SCN_TRY(f32_to_bf16(st, sz(4, D, 2 * F), k.WDb, k.WDbh));
        SCN_TRY(f32_to_bf16(st, sz(NC, D), k.WcatT, k.WcatTh));
        if (d.has_att) SCN_TRY(f32_to_bf16(st, sz(F4, E), k.WaTz, k.WaTzh));
    }
    const void* WDb = bf ? (const void*)k.WDbh : k.WDb;
    const void* WcatT = bf ? (const void*)k.WcatTh : k.WcatT;
    const void* WaTz = bf ? (const void*)k.WaTzh : k.WaTz;
    const void* att1_s = (bf && d.has_att) ? (const void*)s.att1h : s.att1;     // the copies the forward pass streamed
    const void* enc_s = (bf && d.has_att) ? (const void*)s.ench : enc;
    SCN_HIP(hipMemsetAsync(k.dc, 0, sizeof(float) * BD, st));
    SCN_HIP(hipMemsetAsync(k.dqx_acc, 0, sizeof(float) * B * F4, st));
    SCN_HIP(hipMemsetAsync(k.dqh_acc, 0, sizeof(float) * B * F4, st));

    // ---- reverse recurrence ----------------------------------------------------------------------
    hipEvent_t ev0 = prof_begin(st);
    SCN_TRY(run_chains(st, 1, B, [&](hipStream_t cs, int r0, int rmax) -> int {
        const void* enc_c = eoff(enc_s, (long)r0 * (Q > 0 ? Q : P) * E, bf && d.has_att);
        const void* att1_c = d.has_att ? eoff(att1_s, (long)r0 * P * A, bf) : nullptr;
        float* sDb = k.sDb + (long)r0 * 2 * F;
        float* sZ = d.has_att ? k.sZ + (long)r0 * E : nullptr;
        float* sH = k.sH + (long)r0 * D;
        float* dalpha = d.has_att ? k.dalpha + (long)r0 * P : nullptr;
        float* dc = k.dc + (long)r0 * D;
        const float* qx = s.qx + (long)r0 * F4;
        const float* qh = s.qh + (long)r0 * F4;
        float* dqx_acc = k.dqx_acc + (long)r0 * F4;
        float* dqh_acc = k.dqh_acc + (long)r0 * F4;
        auto rows_at = [&](int t) { const int n = bt[t] - r0; return n < 0 ? 0 : (n < rmax ? n : rmax); };
        int ksH = 0, tlast = -1;
        bool lstm_done = false;
        for (int t = T - 1; t >= 0; --t) {
            const int bt_ = rows_at(t);
            if (bt_ <= 0) continue;                                    // this chain starts decoding later (shorter rows)
            const int btn = (t + 1 < T) ? rows_at(t + 1) : 0;
            const long rowT = (long)t * B + r0;
            float* dr = k.dr_all + rowT * 4 * D;
            float* dcat = k.dcat_all + rowT * NC;
            float* dpx = k.dpx_all + rowT * F4;
            // the LSTM backward of this step ran inside the previous iteration's last product (below) unless that launch
            // could not take it, or this is the first iteration
            if (!lstm_done)
                SCN_TRY(lstm_bwd(cs, bt_, btn, D, k.dhfc_tm + rowT * D,
                                 btn > 0 ? Slabs{sH, ksH, BD, D} : Slabs{nullptr, 0, 0, 0}, dc, s.gates_all + rowT * 4 * D,
                                 s.Cs + rowT * D, s.tanhc_all + rowT * D, dr));
            lstm_done = false;
            const int ksDb = pick(bt_, 2 * F, D, 4);
            SkinnyTail mixb{3, bt_, 0, F4, {qx, qh, s.pa_all + rowT * F4, s.ph_all + rowT * F4}, {dpx, dcat, dqx_acc, dqh_acc}, NC,
                            Slabs{nullptr, 0, 0, 0}};
            bool mixed = false;
            SCN_TRY(skinny_gemm(cs, bt_, 2 * F, D, 4, dr, 4 * D, D, WDb, 2 * F, (long)D * 2 * F, sDb, 2 * F,
                                (long)B * 2 * F, (long)4 * B * 2 * F, ksDb, bfm, &mixb, &mixed));
            if (!mixed)
                SCN_TRY(scn_mix_bwd(cs, bt_, F4, Slabs{sDb, ksDb, (long)4 * B * 2 * F, 2 * F}, (long)B * 2 * F, qx, qh,
                                    s.pa_all + rowT * F4, s.ph_all + rowT * F4, dpx, dcat, NC, dqx_acc, dqh_acc));
            if (d.has_att) {
                const int ksZ = pick(bt_, E, F4, 1);
                float* dawe = k.dawe_all + rowT * E;
                SkinnyTail gateb{4, bt_, 0, E, {s.awe_all + rowT * E, s.gate_all + rowT * E, nullptr, nullptr},
                                 {dawe, dcat + F4, nullptr, nullptr}, NC, Slabs{nullptr, 0, 0, 0}};
                bool gated = false;
                SCN_TRY(skinny_gemm(cs, bt_, E, F4, 1, dpx, F4, 0, WaTz, E, 0, sZ, E, 0, (long)B * E, ksZ, bfm, &gateb, &gated));
                if (!gated)
                    SCN_TRY(gate_bwd(cs, bt_, E, Slabs{sZ, ksZ, (long)B * E, E}, s.awe_all + rowT * E,
                                     s.gate_all + rowT * E, dawe, dcat + F4, NC));
                const float* din = dalphas ? dalphas + (long)r0 * T * P + (long)t * P : nullptr;
                if (Q > 0) {
                    // Q dot products per image against x, folded onto the P pooled pixels inside softmax_bwd
                    float* dalphaq = k.dalphaq + (long)r0 * Q;
                    SCN_TRY(attn_dalpha(cs, bt_, Q, E, enc_c, dawe, nullptr, 0, dalphaq, bf));
                    SCN_TRY(attn_softmax_bwd_pooled(cs, bt_, P, A, att1_c, s.att2_all + rowT * A,
                                                    w->attention_full_att_weight, s.alpha_tm + rowT * P, pd, dalphaq, din,
                                                    (long)T * P, k.de_all + rowT * P, dcat + F4 + E, NC, bf));
                } else {
                    SCN_TRY(attn_dalpha(cs, bt_, P, E, enc_c, dawe, din, (long)T * P, dalpha, bf));
                    SCN_TRY(attn_softmax_bwd(cs, bt_, P, A, att1_c, s.att2_all + rowT * A, w->attention_full_att_weight,
                                             s.alpha_tm + rowT * P, dalpha, k.de_all + rowT * P, dcat + F4 + E, NC, bf));
                }
            }
            ksH = pick(bt_, D, NC, 1);
            // d cat . Wcat^T = this step's d h, and with it -- same launch -- the LSTM backward of step t-1, whose only
            // missing input it is (scn_cell.py:134-152 transposed)
            SkinnyTail cellb{5, t > 0 ? rows_at(t - 1) : 0, bt_, D,
                             {t > 0 ? k.dhfc_tm + (rowT - B) * D : nullptr, t > 0 ? s.gates_all + (rowT - B) * 4 * D : nullptr,
                              t > 0 ? s.Cs + (rowT - B) * D : nullptr, t > 0 ? s.tanhc_all + (rowT - B) * D : nullptr},
                             {dc, t > 0 ? k.dr_all + (rowT - B) * 4 * D : nullptr, nullptr, nullptr}, 0, Slabs{nullptr, 0, 0, 0}};
            SCN_TRY(skinny_gemm(cs, bt_, D, NC, 1, dcat, NC, 0, WcatT, D, 0, sH, D, 0, BD, ksH, bfm, t > 0 ? &cellb : nullptr,
                                &lstm_done));
            tlast = t;
        }
        // d loss / d h0 for this chain's rows (d/d c0 is k.dc); every row decodes at t = 0
        SCN_ARG(tlast == 0, "internal: chain without a step at t = 0");
        SCN_TRY(reduce_slabs(cs, rmax, D, Slabs{sH, ksH, BD, D}, k.dh0 + (long)r0 * D));
        return 0;
    }));
    prof_end(st, ev0, 1, T);

    // ---- after the loop: two strands ------------------------------------------------------------------
    // (1) weight gradients, one GEMM per weight over the stacked (t,b) rows: stream `ws`, off the critical path;
    // (2) d tags, d att1 -> d encoder_out: stream `st`, heads the encoder's backward pass.
    // The events are recorded where the inputs of (1) come into being; with two streams its launches are enqueued
    // LAST, so that the host reaches strand (2) first.
    hipEvent_t ev_loop = nullptr, ev_att = nullptr;
    auto mark = [&](hipEvent_t* e) -> int {
        if (!two) return 0;
        SCN_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        SCN_HIP(hipEventRecord(*e, st));
        return 0;
    };
    auto ws_wait = [&](hipEvent_t* e) -> int {
        if (!two || !*e) return 0;
        const hipError_t r = hipStreamWaitEvent(ws, *e, 0);
        (void)hipEventDestroy(*e);
        *e = nullptr;
        SCN_HIP(r);
        return 0;
    };
    auto wgrad_loop = [&]() -> int {      // everything whose operands the reverse recurrence produced
        SCN_TRY(ws_wait(&ev_loop));
        if (g->decode_step_weight_ia) {
            SCN_TRY(sgemm_ws(ws, true, false, M, F4, TB, 1.f, s.emb_tm, M, k.dpx_all, F4, 0.f, g->decode_step_weight_ia, F4,
                          nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
            if (d.has_att)
                SCN_TRY(sgemm_ws(ws, true, false, E, F4, TB, 1.f, s.z_all, E, k.dpx_all, F4, 0.f,
                              g->decode_step_weight_ia + (long)M * F4, F4, nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        }
        if (g->embedding_weight) {
            SCN_TRY(sgemm_ws(ws, false, true, TB, M, F4, 1.f, k.dpx_all, F4, w->decode_step_weight_ia, F4, 0.f, k.demb_tm, M,
                          nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
            SCN_TRY(scatter_add_rows_tm(ws, B, T, d.L, M, (const long long*)caps, dl_dev, k.demb_tm, V,
                                        g->embedding_weight, reinterpret_cast<int*>(k.present)));
        }
        if (g->decode_step_weight_ic) {
            SCN_TRY(mul_bcast(ws, T, B, F4, s.pa_all, s.qx, k.mx_all));
            SCN_TRY(sgemm_ws(ws, true, false, D, F, TB, 1.f, k.dr_all, 4 * D, k.mx_all, F4, 0.f, g->decode_step_weight_ic, F4,
                          nullptr, nullptr, 4, D, F, F, wgws, GEMM_WS_FLOATS));
        }
        if (g->decode_step_weight_hc) {
            SCN_TRY(mul_bcast(ws, T, B, F4, s.ph_all, s.qh, k.mx_all));
            SCN_TRY(sgemm_ws(ws, true, false, D, F, TB, 1.f, k.dr_all, 4 * D, k.mx_all, F4, 0.f, g->decode_step_weight_hc, F4,
                          nullptr, nullptr, 4, D, F, F, wgws, GEMM_WS_FLOATS));
        }
        if (g->decode_step_weight_ha)
            SCN_TRY(sgemm_ws(ws, true, false, D, F4, TB, 1.f, s.Hs, D, k.dcat_all, NC, 0.f, g->decode_step_weight_ha, F4,
                          nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        if (g->decode_step_weight_ib)
            SCN_TRY(sgemm_ws(ws, true, false, d.S, F4, B, 1.f, tags, d.S, k.dqx_acc, F4, 0.f, g->decode_step_weight_ib, F4,
                          nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        if (g->decode_step_weight_hb)
            SCN_TRY(sgemm_ws(ws, true, false, d.S, F4, B, 1.f, tags, d.S, k.dqh_acc, F4, 0.f, g->decode_step_weight_hb, F4,
                          nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        if (g->decode_step_bias_ih) SCN_TRY(colsum(ws, TB, 4 * D, k.dr_all, 4 * D, g->decode_step_bias_ih, 0.f));
        if (g->decode_step_bias_hh) SCN_TRY(colsum(ws, TB, 4 * D, k.dr_all, 4 * D, g->decode_step_bias_hh, 0.f));
        if (d.has_att) {
            if (g->f_beta_weight)
                SCN_TRY(sgemm_ws(ws, true, false, E, D, TB, 1.f, k.dcat_all + F4, NC, s.Hs, D, 0.f, g->f_beta_weight, D,
                              nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
            if (g->f_beta_bias) SCN_TRY(colsum(ws, TB, E, k.dcat_all + F4, NC, g->f_beta_bias, 0.f));
            if (g->attention_decoder_att_weight)
                SCN_TRY(sgemm_ws(ws, true, false, A, D, TB, 1.f, k.dcat_all + F4 + E, NC, s.Hs, D, 0.f,
                              g->attention_decoder_att_weight, D, nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
            if (g->attention_decoder_att_bias)
                SCN_TRY(colsum(ws, TB, A, k.dcat_all + F4 + E, NC, g->attention_decoder_att_bias, 0.f));
        }
        if (g->init_h_weight)
            SCN_TRY(sgemm_ws(ws, true, false, D, E, B, 1.f, k.dh0, D, s.mean_enc, E, 0.f, g->init_h_weight, E, nullptr,
                          nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        if (g->init_h_bias) SCN_TRY(colsum(ws, B, D, k.dh0, D, g->init_h_bias, 0.f));
        if (g->init_c_weight)
            SCN_TRY(sgemm_ws(ws, true, false, D, E, B, 1.f, k.dc, D, s.mean_enc, E, 0.f, g->init_c_weight, E, nullptr, nullptr,
                          1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        if (g->init_c_bias) SCN_TRY(colsum(ws, B, D, k.dc, D, g->init_c_bias, 0.f));
        return 0;
    };
    auto wgrad_att = [&]() -> int {       // d encoder_att: needs d att1 (/ d y)
        SCN_TRY(ws_wait(&ev_att));
        if (!d.has_att) return 0;
        if (g->attention_encoder_att_weight) {
            if (Q > 0)
                SCN_TRY(sgemm_ws(ws, true, false, A, E, B * Q, 1.f, k.dy, A, enc, E, 0.f, g->attention_encoder_att_weight,
                              E, nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
            else
                SCN_TRY(sgemm_ws(ws, true, false, A, E, B * P, 1.f, k.datt1, A, enc, E, 0.f,
                              g->attention_encoder_att_weight, E, nullptr, nullptr, 1, 0, 0, 0, wgws, GEMM_WS_FLOATS));
        }
        if (g->attention_encoder_att_bias)
            SCN_TRY(colsum(ws, B * P, A, k.datt1, A, g->attention_encoder_att_bias, 0.f));
        return 0;
    };
    SCN_TRY(mark(&ev_loop));
    if (!two) SCN_TRY(wgrad_loop());
    if (dtags) {
        SCN_TRY(sgemm_ws(st, false, true, B, d.S, F4, 1.f, k.dqx_acc, F4, w->decode_step_weight_ib, F4, 0.f, dtags, d.S,
                      nullptr, nullptr, 1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
        SCN_TRY(sgemm_ws(st, false, true, B, d.S, F4, 1.f, k.dqh_acc, F4, w->decode_step_weight_hb, F4, 1.f, dtags, d.S,
                      nullptr, nullptr, 1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
    }
    if (d.has_att) {
        int nblk = 0;
        SCN_TRY(attn_datt1_post(st, B, P, A, T, dl_dev, att1_s, s.att2_all, k.de_all, w->attention_full_att_weight,
                                k.datt1, k.dwpart, &nblk, bf));
        SCN_TRY(colsum(st, nblk, A + 1, k.dwpart, A + 1, k.dwtmp, 0.f));
        if (g->attention_full_att_weight)
            SCN_TRY(copy2d(st, 1, A, k.dwtmp, A + 1, g->attention_full_att_weight, A));
        if (g->attention_full_att_bias) SCN_TRY(copy2d(st, 1, 1, k.dwtmp + A, 1, g->attention_full_att_bias, 1));
        if (Q > 0 && (g->attention_encoder_att_weight || denc))
            SCN_TRY(pool_transpose(st, B, P, A, pd, k.datt1, k.dy));      // d y = pool^T (d att1): [B*Q][A]
    }
    SCN_TRY(mark(&ev_att));
    if (!two) SCN_TRY(wgrad_att());

    // ---- d loss / d encoder_out (only when the encoder is fine-tuned) -------------------------------
    if (denc && Q > 0) {
        // d x = d y . We  +  sum_t alphaq_t (x) dawe_t  +  col_w (x) d mean      (all on the Q source pixels)
        if (d.has_att) {
            SCN_TRY(sgemm_ws(st, false, false, B * Q, E, A, 1.f, k.dy, A, w->attention_encoder_att_weight, E, 0.f, denc,
                          E, nullptr, nullptr, 1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
            SCN_TRY(sgemm_ws(st, true, false, Q, E, T, 1.f, s.alphaq_tm, (long)B * Q, k.dawe_all, (long)B * E, 1.f, denc,
                          E, nullptr, nullptr, B, Q, E, (long)Q * E, k.gws, GEMM_WS_FLOATS));
        } else {
            SCN_HIP(hipMemsetAsync(denc, 0, sizeof(float) * B * Q * E, st));
        }
        SCN_TRY(sgemm_ws(st, false, false, B, E, D, 1.f, k.dh0, D, w->init_h_weight, E, 0.f, k.dmean, E, nullptr, nullptr,
                      1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
        SCN_TRY(sgemm_ws(st, false, false, B, E, D, 1.f, k.dc, D, w->init_c_weight, E, 1.f, k.dmean, E, nullptr, nullptr,
                      1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
        SCN_TRY(add_bcast_rows_w(st, B, Q, E, pd.col_w, k.dmean, denc));
    } else if (denc) {
        if (d.has_att) {
            SCN_TRY(sgemm_ws(st, false, false, B * P, E, A, 1.f, k.datt1, A, w->attention_encoder_att_weight, E, 0.f,
                          denc, E, nullptr, nullptr, 1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
            // denc[b] += alpha_b^T (P x T) . dawe_b (T x E), batched over b
            SCN_TRY(sgemm_ws(st, true, false, P, E, T, 1.f, s.alpha_tm, (long)B * P, k.dawe_all, (long)B * E, 1.f, denc, E,
                          nullptr, nullptr, B, P, E, (long)P * E, k.gws, GEMM_WS_FLOATS));
        } else {
            SCN_HIP(hipMemsetAsync(denc, 0, sizeof(float) * B * P * E, st));
        }
        SCN_TRY(sgemm_ws(st, false, false, B, E, D, 1.f, k.dh0, D, w->init_h_weight, E, 0.f, k.dmean, E, nullptr, nullptr,
                      1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
        SCN_TRY(sgemm_ws(st, false, false, B, E, D, 1.f, k.dc, D, w->init_c_weight, E, 1.f, k.dmean, E, nullptr, nullptr,
                      1, 0, 0, 0, k.gws, GEMM_WS_FLOATS));
        SCN_TRY(add_bcast_rows(st, B, P, E, k.dmean, 1.f / (float)P, denc));
    }
    if (two) {
        SCN_TRY(wgrad_loop());
        SCN_TRY(wgrad_att());
    }
    return 0;
}

}  // namespace scn
