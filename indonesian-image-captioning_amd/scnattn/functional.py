"""torch.autograd.Function wrappers over the C ABI (libscnattn.so).

Everything numerical happens in the HIP kernels; this file only allocates torch tensors (so the
caching allocator and stream semantics apply), passes raw pointers, and wires autograd.  The math each
wrapper covers, by reference file:
  decoder_sequence -> models/decoders/attention_scn.py:124-156, pure_scn.py:114-138
  scn_cell         -> models/scn_cell.py:52-154
  attention        -> models/attention.py:26-44
  pool_permute     -> models/encoders/caption.py:41-43
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import Pool, Dims, Params, PARAM_FIELDS, call, ptr, stream_of, f32c, require_cuda


# ----------------------------------------------------------------------------------------------
# thin primitive helpers
# ----------------------------------------------------------------------------------------------
def gemm(a, b, ta=False, tb=False, bias=None, out=None, beta=0.0, alpha=1.0, M=None, N=None, K=None,
         lda=None, ldb=None, ldc=None, batch=1, sa=0, sb=0, sc=0, rowmask=None):
    """out = alpha*op(a).op(b) + beta*out + bias; all sizes/strides may be overridden to address
    sub-blocks of larger row-major buffers."""
    if M is None:
        M = a.shape[1] if ta else a.shape[0]
    if K is None:
        K = a.shape[0] if ta else a.shape[1]
    if N is None:
        N = b.shape[0] if tb else b.shape[1]
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    ldc = out.stride(0) if ldc is None else ldc
    ws = _gemm_workspace(a.device)
    call("scnattn_sgemm_ws", stream_of(a), int(ta), int(tb), M, N, K, alpha, ptr(a), lda, ptr(b), ldb, beta,
         ptr(out), ldc, ptr(bias), ptr(rowmask), batch, sa, sb, sc, ptr(ws), ws.numel())
    return out


_gemm_ws = {}


def _gemm_workspace(dev):
    """32 MiB of split-K partial sums per (device, stream): the stand-alone modules' GEMMs have few rows (beam
    search: k <= 5, PureAttention at batch 4), where the tile grid alone leaves most of the chip idle."""
    # per host thread too: a GEMM is two launches (partials, then their reduction) and two threads enqueueing on
    # one stream could interleave them
    key = (dev, torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()),
           threading.get_ident())
    ws = _gemm_ws.get(key)
    if ws is None:
        ws = _gemm_ws[key] = torch.empty(24 << 20, device=dev, dtype=torch.float32)
    return ws


def colsum(x, R=None, N=None, ld=None):
    R = x.shape[0] if R is None else R
    N = x.shape[1] if N is None else N
    ld = x.stride(0) if ld is None else ld
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    call("scnattn_colsum", stream_of(x), R, N, ptr(x), ld, ptr(out), 0.0)
    return out


def set_option(name, value):
    call("scnattn_set_option", name.encode(), int(value))


# ----------------------------------------------------------------------------------------------
# whole-sequence decoder
# ----------------------------------------------------------------------------------------------
def _params_struct(tensors):
    p = Params()
    for name, t in zip(PARAM_FIELDS, tensors):
        setattr(p, name, None if t is None else t.data_ptr())
    return p


class PoolTaps:
    """`encoder_out = AdaptiveAvgPool2d(out)(x)` (models/encoders/caption.py:20,41) as tap tables: pooled pixel p
    averages <= 4 source pixels (the pool up-samples: 8x8 -> 14x14 windows are 1 or 2 wide), `matrix()` is the
    dense (P, Q) pooling matrix they encode.  Device tensors + the scnattn_pool struct the C ABI takes."""

    def __init__(self, in_h, in_w, out_h, out_w, device):
        def windows(n_in, n_out):     # torch's adaptive pooling windows: [floor(i*n/o), ceil((i+1)*n/o))
            return [(i * n_in // n_out, -(-(i + 1) * n_in // n_out)) for i in range(n_out)]
        rows, cols = windows(in_h, out_h), windows(in_w, out_w)
        P, Q = out_h * out_w, in_h * in_w
        tap_idx = torch.zeros(P, 4, dtype=torch.int32)
        tap_w = torch.zeros(P, 4, dtype=torch.float32)
        per_q = [[] for _ in range(Q)]
        for i, (r0, r1) in enumerate(rows):
            for j, (c0, c1) in enumerate(cols):
                p = i * out_w + j
                src = [(r, c) for r in range(r0, r1) for c in range(c0, c1)]
                if len(src) > 4:
                    raise ValueError("pooling window of %d pixels: the pooled path handles up-sampling pools "
                                     "(windows of at most 2x2)" % len(src))
                for k, (r, c) in enumerate(src):
                    q = r * in_w + c
                    tap_idx[p, k] = q
                    tap_w[p, k] = 1.0 / len(src)
                    per_q[q].append((p, 1.0 / len(src)))
        qmax = max(len(l) for l in per_q)
        qtap_idx = torch.full((Q, qmax), -1, dtype=torch.int32)
        qtap_w = torch.zeros(Q, qmax, dtype=torch.float32)
        for q, l in enumerate(per_q):
            for k, (p, wv) in enumerate(l):
                qtap_idx[q, k] = p
                qtap_w[q, k] = wv
        col_w = (qtap_w.double().sum(dim=1) / P).float()
        self.P, self.Q, self.qmax, self.shape = P, Q, qmax, (in_h, in_w, out_h, out_w)
        self.tap_idx, self.tap_w = tap_idx.to(device), tap_w.to(device)
        self.qtap_idx, self.qtap_w, self.col_w = qtap_idx.to(device), qtap_w.to(device), col_w.to(device)
        self.cstruct = Pool(Q, qmax, self.tap_idx.data_ptr(), self.tap_w.data_ptr(), self.qtap_idx.data_ptr(),
                            self.qtap_w.data_ptr(), self.col_w.data_ptr())

    def matrix(self):
        m = torch.zeros(self.P, self.Q, dtype=torch.float32, device=self.tap_w.device)
        m.scatter_add_(1, self.tap_idx.long(), self.tap_w)
        return m


_pool_cache = {}


def pool_taps(in_h, in_w, out_h, out_w, device):
    key = (in_h, in_w, out_h, out_w, str(device))
    if key not in _pool_cache:
        _pool_cache[key] = PoolTaps(in_h, in_w, out_h, out_w, device)
    return _pool_cache[key]


_POISON = False     # tests: NaN-fill workspaces that are claimed to be fully overwritten


def _workspace(nfloats, dev, fully_written):
    if not fully_written:
        return torch.zeros(nfloats, device=dev, dtype=torch.float32)
    ws = torch.empty(nfloats, device=dev, dtype=torch.float32)
    if _POISON:
        ws.fill_(float("nan"))
    return ws


# Weight gradients of the decoder on the side stream (scnattn_seq_bwd_streams).  Bit-identical (tested) but not faster
# on one MI355X: 782-797 images/s with it, 793 without -- the GEMMs it moves off the main stream land beside an encoder
# backward pass that already keeps the chip busy, and d fc.weight beside the reverse recurrence slows that loop by
# ~1 us per step.  Opt-in.
DECODER_SIDE_WGRAD = os.environ.get("SCNATTN_DECODER_SIDE_WGRAD", "0") != "0"


class _DecoderSeq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, enc, tags, caps, dl_dev, drop_mask, *weights):
        dims_t, bt_host, pool = meta       # pool: PoolTaps or None; with it `enc` is the un-pooled map (B, Q, E)
        d = Dims(*dims_t)
        cpool = None if pool is None else C.byref(pool.cstruct)
        require_cuda(enc, tags, caps, dl_dev, *weights)
        dev = enc.device
        enc, tags = f32c(enc), f32c(tags)
        caps = caps.contiguous()
        params_in = weights                  # the caller's tensors (Parameters): backward looks at their .grad
        weights = tuple(None if w is None else f32c(w.detach()) for w in weights)
        drop_mask = f32c(drop_mask)
        sv, sc = C.c_size_t(), C.c_size_t()
        call("scnattn_seq_workspace", C.byref(d), cpool, C.byref(sv), C.byref(sc))
        # Rows (t, b >= b_t) of the time-major buffers are never written by the kernels but are read by the
        # post-loop GEMMs (and returned, for alphas), so they must be zero -- unless every row decodes at every
        # step (fixed-length captions), where each element is written before it is read and the 120 MB fill is
        # skipped.  `_POISON` (tests) fills with NaN instead to prove exactly that.
        full = min(bt_host) == d.B
        saved = _workspace(sv.value // 4, dev, full)
        scratch = torch.empty(sc.value // 4, device=dev, dtype=torch.float32)
        preds = torch.empty((d.B, d.T, d.V), device=dev, dtype=torch.float32)
        alphas = _workspace(d.B * d.T * d.P, dev, full).view(d.B, d.T, d.P) if d.has_att else None
        bt = (C.c_int32 * d.T)(*bt_host)
        w = _params_struct(weights)
        call("scnattn_seq_fwd", stream_of(enc), C.byref(d), C.byref(w), ptr(enc), ptr(tags), ptr(caps), ptr(dl_dev),
             C.cast(bt, C.c_void_p), ptr(drop_mask), ptr(saved), ptr(scratch), ptr(preds), ptr(alphas), cpool)
        ctx.meta = (dims_t, tuple(bt_host), sc.value, pool)
        ctx.params = params_in
        ctx.save_for_backward(enc, tags, caps, dl_dev, drop_mask, saved, *[x for x in weights if x is not None])
        ctx.wmask = tuple(x is not None for x in weights)
        if alphas is None:
            alphas = preds.new_zeros(0)
            ctx.mark_non_differentiable(alphas)
        return preds, alphas

    @staticmethod
    def backward(ctx, dpreds, dalphas):
        dims_t, bt_host, scratch_bytes, pool = ctx.meta
        d = Dims(*dims_t)
        cpool = None if pool is None else C.byref(pool.cstruct)
        enc, tags, caps, dl_dev, drop_mask, saved, *wl = ctx.saved_tensors
        it = iter(wl)
        weights = tuple(next(it) if m else None for m in ctx.wmask)
        dev = enc.device
        dpreds = f32c(dpreds)
        dalphas = f32c(dalphas) if (d.has_att and dalphas is not None) else None
        scratch = _workspace(scratch_bytes // 4, dev, min(bt_host) == d.B)
        need = ctx.needs_input_grad  # (meta, enc, tags, caps, dl, mask, *weights)
        grads = []
        for i, (name, wt) in enumerate(zip(PARAM_FIELDS, weights)):
            if wt is None or not need[6 + i]:
                grads.append(None)
            elif name == "embedding_weight":
                grads.append(torch.zeros_like(wt))
            else:
                grads.append(torch.empty_like(wt))
        denc = torch.empty_like(enc) if need[1] else None
        dtags = torch.empty_like(tags) if need[2] else None
        bt = (C.c_int32 * d.T)(*bt_host)
        w, g = _params_struct(weights), _params_struct(grads)
        # Weight gradients on the side stream (scnattn/conv.py::_Side, shared with the trunk): d fc.weight runs beside
        # the reverse recurrence, the post-loop weight-gradient GEMMs beside the encoder's backward pass that `denc`
        # starts.  Safe only while autograd merely STORES the returned tensors (conv.side_ok); the stream is joined at
        # the end of the autograd sweep (conv._Side.mark) or by whoever reads the gradients first (FlatBuffer.gather).
        side = None
        if DECODER_SIDE_WGRAD and not torch.is_grad_enabled():
            from . import conv as _conv
            if _conv.side_ok(*ctx.params):
                side = _conv._side(dev)
        if side is None:
            call("scnattn_seq_bwd", stream_of(enc), C.byref(d), C.byref(w), ptr(enc), ptr(tags), ptr(caps), ptr(dl_dev),
                 C.cast(bt, C.c_void_p), ptr(drop_mask), ptr(saved), ptr(scratch), ptr(dpreds), ptr(dalphas),
                 C.byref(g), ptr(denc), ptr(dtags), cpool)
        else:
            main = torch.cuda.current_stream(dev)
            side.fork(main, enc, tags, caps, dl_dev, drop_mask, saved, scratch, dpreds, *weights, *grads)
            call("scnattn_seq_bwd_streams", stream_of(enc), C.c_void_p(side.stream.cuda_stream), C.byref(d), C.byref(w),
                 ptr(enc), ptr(tags), ptr(caps), ptr(dl_dev), C.cast(bt, C.c_void_p), ptr(drop_mask), ptr(saved),
                 ptr(scratch), ptr(dpreds), ptr(dalphas), C.byref(g), ptr(denc), ptr(dtags), cpool)
            side.mark()
        return (None, denc, dtags, None, None, None, *grads)


def decoder_sequence(dims, bt_host, enc, tags, caps, dl_dev, drop_mask, weights, pool=None):
    """dims: 12-tuple in scnattn_dims order; weights: tensors in PARAM_FIELDS order (None = absent).
    pool: PoolTaps -- then `enc` is the un-pooled feature map (B, Q, E) and the pooled encoder_out is never
    built.  Returns (predictions (B,T,V), alphas (B,T,P) or None)."""
    if pool is not None and (enc.dim() != 3 or enc.shape[1] != pool.Q or dims[1] != pool.P):
        raise RuntimeError("decoder_sequence: the un-pooled map must be (B, %d, E) for this pool" % pool.Q)
    preds, alphas = _DecoderSeq.apply((tuple(dims), tuple(bt_host), pool), enc, tags, caps, dl_dev, drop_mask, *weights)
    return preds, (alphas if dims[-1] else None)


# ----------------------------------------------------------------------------------------------
# stand-alone SCN cell (any batch size), split exactly where the reference splits it:
#   scn_input      = SCNCell.forward's x side      (models/scn_cell.py:64-91)
#   scn_recurrent  = SCNCell.recurrent_step        (models/scn_cell.py:112-154)
# Dense contractions run on the MFMA sgemm, the rest on mul_bcast / lstm kernels.
# ----------------------------------------------------------------------------------------------
def _mul(x, q):
    out = torch.empty_like(x)
    call("scnattn_mul_bcast", stream_of(x), 1, x.shape[0], x.shape[1], ptr(x), ptr(q), ptr(out))
    return out


def _off(t, nfloats):
    return C.c_void_p(t.data_ptr() + 4 * nfloats)


class _SCNInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, s, Wa, Wb, Wc, bih):
        require_cuda(u, s, Wa, Wb, Wc, bih)
        u, s = f32c(u), f32c(s)
        Wa, Wb, Wc = (f32c(x.detach()) for x in (Wa, Wb, Wc))
        bih = None if bih is None else f32c(bih.detach())
        B, H, F4 = u.shape[0], Wc.shape[0], Wa.shape[1]
        F = F4 // 4
        pa, qx = gemm(u, Wa), gemm(s, Wb)
        mx = _mul(pa, qx)
        x = torch.empty((4, B, H), device=u.device, dtype=torch.float32)
        for g in range(4):  # x_g = mx_g . Wc_g^T + b_ih_g
            call("scnattn_sgemm", stream_of(u), 0, 1, B, H, F, 1.0, _off(mx, g * F), F4, _off(Wc, g * F), F4, 0.0,
                 _off(x, g * B * H), H, None if bih is None else _off(bih, g * H), None, 1, 0, 0, 0)
        ctx.save_for_backward(u, s, Wa, Wb, Wc, pa, qx, mx)
        ctx.has_bias = bih is not None
        return x[0], x[1], x[2], x[3]

    @staticmethod
    def backward(ctx, *dxs):
        u, s, Wa, Wb, Wc, pa, qx, mx = ctx.saved_tensors
        B, H, F4 = u.shape[0], Wc.shape[0], Wa.shape[1]
        F = F4 // 4
        dev = u.device
        dx = torch.stack([torch.zeros((B, H), device=dev) if d is None else f32c(d) for d in dxs])  # [4,B,H]
        dmx = torch.empty((B, F4), device=dev, dtype=torch.float32)
        gemm(dx, Wc, out=dmx, M=B, N=F, K=H, lda=H, ldb=F4, ldc=F4, batch=4, sa=B * H, sb=F, sc=F)
        dpa, dqx = _mul(dmx, qx), _mul(dmx, pa)
        need = ctx.needs_input_grad
        du = gemm(dpa, Wa, tb=True) if need[0] else None
        ds = gemm(dqx, Wb, tb=True) if need[1] else None
        dWa = gemm(u, dpa, ta=True) if need[2] else None
        dWb = gemm(s, dqx, ta=True) if need[3] else None
        dWc = None
        if need[4]:
            dWc = torch.empty_like(Wc)
            gemm(dx, mx, ta=True, out=dWc, M=H, N=F, K=B, lda=H, ldb=F4, ldc=F4, batch=4, sa=B * H, sb=F, sc=F)
        dbih = None
        if ctx.has_bias and need[5]:
            dbih = torch.empty(4 * H, device=dev, dtype=torch.float32)
            for g in range(4):
                call("scnattn_colsum", stream_of(u), B, H, _off(dx, g * B * H), H, _off(dbih, g * H), 0.0)
        return du, ds, dWa, dWb, dWc, dbih


class _SCNRecurrent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_i, x_f, x_o, x_c, s, h, c, Ha, Hb, Hc, bhh):
        require_cuda(x_i, x_f, x_o, x_c, s, h, c, Ha, Hb, Hc, bhh)
        s, h, c = f32c(s), f32c(h), f32c(c)
        Ha, Hb, Hc = (f32c(x.detach()) for x in (Ha, Hb, Hc))
        bhh = None if bhh is None else f32c(bhh.detach())
        B, H, F4 = h.shape[0], Hc.shape[0], Ha.shape[1]
        F = F4 // 4
        st = stream_of(h)
        r = torch.stack([f32c(x_i), f32c(x_f), f32c(x_o), f32c(x_c)])  # [4,B,H]: r starts as x
        ph, qh = gemm(h, Ha), gemm(s, Hb)
        mh = _mul(ph, qh)
        gemm(mh, Hc, tb=True, out=r, beta=1.0, M=B, N=H, K=F, lda=F4, ldb=F4, ldc=H, batch=4, sa=F, sb=F, sc=B * H)
        gates = torch.empty((B, 4 * H), device=h.device, dtype=torch.float32)
        c_new, h_new, tanhc = torch.empty_like(c), torch.empty_like(c), torch.empty_like(c)
        call("scnattn_lstm_fwd", st, B, H, ptr(r), 1, 0, H, B * H, None, ptr(bhh), ptr(c), ptr(gates), ptr(c_new),
             ptr(h_new), ptr(tanhc))
        ctx.save_for_backward(s, h, c, Ha, Hb, Hc, ph, qh, mh, gates, tanhc)
        ctx.has_bias = bhh is not None
        return h_new, c_new

    @staticmethod
    def backward(ctx, dh, dc_in):
        s, h, c, Ha, Hb, Hc, ph, qh, mh, gates, tanhc = ctx.saved_tensors
        B, H, F4 = h.shape[0], Hc.shape[0], Ha.shape[1]
        F = F4 // 4
        st = stream_of(h)
        dev = h.device
        dh = torch.zeros_like(c) if dh is None else f32c(dh)
        dc = torch.zeros_like(c) if dc_in is None else f32c(dc_in).clone()
        dr = torch.empty((B, 4 * H), device=dev, dtype=torch.float32)
        call("scnattn_lstm_bwd", st, B, B, H, ptr(dh), None, 0, 0, H, ptr(dc), ptr(gates), ptr(c), ptr(tanhc), ptr(dr))
        need = ctx.needs_input_grad
        dxs = [dr[:, g * H:(g + 1) * H].contiguous() if need[g] else None for g in range(4)]
        dmh = torch.empty((B, F4), device=dev, dtype=torch.float32)
        gemm(dr, Hc, out=dmh, M=B, N=F, K=H, lda=4 * H, ldb=F4, ldc=F4, batch=4, sa=H, sb=F, sc=F)
        dph, dqh = _mul(dmh, qh), _mul(dmh, ph)
        ds = gemm(dqh, Hb, tb=True) if need[4] else None
        dhp = gemm(dph, Ha, tb=True) if need[5] else None
        dHa = gemm(h, dph, ta=True) if need[7] else None
        dHb = gemm(s, dqh, ta=True) if need[8] else None
        dHc = None
        if need[9]:
            dHc = torch.empty_like(Hc)
            gemm(dr, mh, ta=True, out=dHc, M=H, N=F, K=B, lda=4 * H, ldb=F4, ldc=F4, batch=4, sa=H, sb=F, sc=F)
        dbhh = colsum(dr) if (ctx.has_bias and need[10]) else None
        return (*dxs, ds, dhp, dc if need[6] else None, dHa, dHb, dHc, dbhh)


def scn_input(u, s, Wa, Wb, Wc, bih):
    return _SCNInput.apply(u, s, Wa, Wb, Wc, bih)


def scn_recurrent(x_i, x_f, x_o, x_c, s, h, c, Ha, Hb, Hc, bhh):
    return _SCNRecurrent.apply(x_i, x_f, x_o, x_c, s, h, c, Ha, Hb, Hc, bhh)


# ----------------------------------------------------------------------------------------------
# stand-alone soft attention
# ----------------------------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, h, We, be, Wd, bd, wf, b0):
        require_cuda(enc, h, We, be, Wd, bd, wf, b0)
        enc, h = f32c(enc), f32c(h)
        We, be, Wd, bd, wf, b0 = (f32c(x.detach()) for x in (We, be, Wd, bd, wf, b0))
        B, P, E = enc.shape
        A = We.shape[0]
        st = stream_of(enc)
        dev = enc.device
        att1 = gemm(enc.view(B * P, E), We, tb=True, bias=be)
        att2 = gemm(h, Wd, tb=True, bias=bd)
        e = torch.empty((B, P), device=dev, dtype=torch.float32)
        call("scnattn_attn_scores", st, B, P, A, ptr(att1), ptr(att2), 1, 0, A, None, ptr(wf), ptr(b0), ptr(e), None)
        alpha = torch.empty((B, P), device=dev, dtype=torch.float32)
        awe = torch.empty((B, E), device=dev, dtype=torch.float32)
        call("scnattn_attn_context", st, B, P, E, ptr(enc), ptr(e), None, 0, 0, 0, None, ptr(alpha), P, None,
             ptr(awe), None, None)
        ctx.save_for_backward(enc, h, We, Wd, wf, att1, att2, alpha)
        return awe, alpha

    @staticmethod
    def backward(ctx, dawe, dalpha_in):
        enc, h, We, Wd, wf, att1, att2, alpha = ctx.saved_tensors
        B, P, E = enc.shape
        A = We.shape[0]
        st = stream_of(enc)
        dev = enc.device
        dawe = torch.zeros((B, E), device=dev) if dawe is None else f32c(dawe)
        dalpha_in = None if dalpha_in is None else f32c(dalpha_in)
        dalpha = torch.empty((B, P), device=dev, dtype=torch.float32)
        call("scnattn_attn_dalpha", st, B, P, E, ptr(enc), ptr(dawe), ptr(dalpha_in), P, ptr(dalpha))
        de = torch.empty((B, P), device=dev, dtype=torch.float32)
        datt2 = torch.empty((B, A), device=dev, dtype=torch.float32)
        call("scnattn_attn_softmax_bwd", st, B, P, A, ptr(att1), ptr(att2), ptr(wf), ptr(alpha), ptr(dalpha), ptr(de),
             ptr(datt2), A)
        nblk = _lib.lib().scnattn_attn_datt1_post_blocks(B, P)
        datt1 = torch.empty((B * P, A), device=dev, dtype=torch.float32)
        dwpart = torch.empty((nblk, A + 1), device=dev, dtype=torch.float32)
        ones = torch.ones(B, device=dev, dtype=torch.int32)
        call("scnattn_attn_datt1_post", st, B, P, A, 1, ptr(ones), ptr(att1), ptr(att2), ptr(de), ptr(wf), ptr(datt1),
             ptr(dwpart))
        dwb = colsum(dwpart)
        need = ctx.needs_input_grad
        denc = None
        if need[0]:
            denc = gemm(datt1, We).view(B, P, E)
            # denc[b] += alpha_b (P x 1) . dawe_b (1 x E)
            gemm(alpha, dawe, ta=True, out=denc, beta=1.0, M=P, N=E, K=1, lda=B * P, ldb=B * E, ldc=E, batch=B,
                 sa=P, sb=E, sc=P * E)
        dh = gemm(datt2, Wd) if need[1] else None
        dWe = gemm(datt1, enc.view(B * P, E), ta=True) if need[2] else None
        dbe = colsum(datt1) if need[3] else None
        dWd = gemm(datt2, h, ta=True) if need[4] else None
        dbd = colsum(datt2) if need[5] else None
        dwf = dwb[:A].reshape(1, A).clone() if need[6] else None
        db0 = dwb[A:A + 1].clone() if need[7] else None
        return denc, dh, dWe, dbe, dWd, dbd, dwf, db0


def attention(enc, h, We, be, Wd, bd, wf, b0):
    return _Attention.apply(enc, h, We, be, Wd, bd, wf, b0)


# ----------------------------------------------------------------------------------------------
# nn.Linear on the MFMA sgemm (init_h / init_c / f_beta / fc outside the fused sequence path)
# ----------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        require_cuda(x, W, b)
        x2 = f32c(x).reshape(-1, x.shape[-1])
        W = f32c(W.detach())
        y = gemm(x2, W, tb=True, bias=None if b is None else f32c(b.detach()))
        ctx.save_for_backward(x2, W)
        ctx.xshape = x.shape
        ctx.has_bias = b is not None
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        dy2 = f32c(dy).reshape(-1, W.shape[0])
        need = ctx.needs_input_grad
        dx = gemm(dy2, W).view(ctx.xshape) if need[0] else None
        dW = gemm(dy2, x2, ta=True) if need[1] else None
        db = colsum(dy2) if (ctx.has_bias and need[2]) else None
        return dx, dW, db


def linear(x, W, b=None):
    return _Linear.apply(x, W, b)


# ----------------------------------------------------------------------------------------------
# encoder tail: AdaptiveAvgPool2d + permute(0,2,3,1), fused, any input memory format
# ----------------------------------------------------------------------------------------------
class _PoolPermute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_size):
        require_cuda(x)
        ctx.in_dtype = x.dtype
        if x.dtype != torch.float32:     # bf16 trunk: the decoder side is fp32
            x = x.float()
        B, Cn, Hin, Win = x.shape
        y = torch.empty((B, out_size, out_size, Cn), device=x.device, dtype=torch.float32)
        sb, scs, sh, sw = x.stride()
        call("scnattn_pool_permute_fwd", stream_of(x), B, Cn, Hin, Win, out_size, out_size, ptr(x), sb, scs, sh, sw, ptr(y))
        ctx.shape = (B, Cn, Hin, Win, out_size)
        ctx.channels_last = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Cn, Hin, Win, out_size = ctx.shape
        dy = f32c(dy)
        dx = torch.empty((B, Cn, Hin, Win), device=dy.device, dtype=torch.float32,
                         memory_format=torch.channels_last if ctx.channels_last else torch.contiguous_format)
        sb, scs, sh, sw = dx.stride()
        call("scnattn_pool_permute_bwd", stream_of(dy), B, Cn, Hin, Win, out_size, out_size, ptr(dy), ptr(dx), sb, scs, sh, sw)
        return (dx if ctx.in_dtype == torch.float32 else dx.to(ctx.in_dtype)), None


def pool_permute(x, out_size):
    return _PoolPermute.apply(x, out_size)


# ----------------------------------------------------------------------------------------------
# fused BatchNorm2d (+ residual) (+ ReLU), channels-last  (csrc/batchnorm.hip)
# ----------------------------------------------------------------------------------------------
BN_MASK_FROM_Z = True    # bn -> relu groups without residual: recompute the ReLU mask from z in the backward pass
_bn_ws = {}
_bn_fn = None


def _bn_workspace(dev, C):
    # stream-ordered reuse: one buffer per (device, stream) -- two trunks may run on two streams at once (the frozen
    # tagger beside the caption encoder, trains/harness.py)
    key = (dev, torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))
    ws = _bn_ws.get(key)
    if ws is None or ws.numel() < 512 * C:
        need = max(_lib.lib().scnattn_bn_workspace_floats(C), 1 << 20)
        ws = torch.empty(need, device=dev, dtype=torch.float32)
        _bn_ws[key] = ws
    return ws


def _bn_fns():
    """Pre-bound ctypes entry points: the BN group runs ~300 times per train step, so the Python cost per
    call matters (the GPU kernels take 5-20 us each)."""
    global _bn_fn
    if _bn_fn is None:
        h = _lib.lib()
        _bn_fn = (h.scnattn_bn_stats, h.scnattn_bn_apply, h.scnattn_bn_bwd, torch._C._cuda_getCurrentRawStream)
    return _bn_fn


class _BNAct(torch.autograd.Function):
    """Feature maps may be fp32 or bf16 (trunk under bf16 autocast); parameters/statistics are fp32."""

    @staticmethod
    def forward(ctx, z, res, gamma, beta, run_mean, run_var, training, momentum, eps, relu):
        if not z.is_cuda:
            raise RuntimeError("scnattn: fused BatchNorm needs tensors on an MI355X device; there is no CPU fallback")
        if z.dtype not in (torch.float32, torch.bfloat16):
            z = z.float()
        bf16 = 1 if z.dtype == torch.bfloat16 else 0
        if not z.is_contiguous(memory_format=torch.channels_last):
            z = z.contiguous(memory_format=torch.channels_last)
        if res is not None and (res.dtype != z.dtype or not res.is_contiguous(memory_format=torch.channels_last)):
            res = res.to(z.dtype).contiguous(memory_format=torch.channels_last)
        N, Cn, H, W = z.shape
        R = N * H * W
        f_stats, f_apply, _, raw_stream = _bn_fns()
        st = raw_stream(z.device.index)
        if training:
            stats = torch.empty((2, Cn), device=z.device, dtype=torch.float32)
            mean, invstd = stats[0], stats[1]
            rc = f_stats(st, R, Cn, z.data_ptr(), bf16, eps, momentum, _bn_workspace(z.device, Cn).data_ptr(),
                         mean.data_ptr(), invstd.data_ptr(), run_mean.data_ptr(), run_var.data_ptr())
            if rc:
                _lib.check(rc, "scnattn_bn_stats")
        else:
            mean, invstd = run_mean.float(), torch.rsqrt(run_var.float() + eps)
        y = torch.empty_like(z)   # preserves the channels-last strides
        rc = f_apply(st, R, Cn, z.data_ptr(), None if res is None else res.data_ptr(), bf16, mean.data_ptr(),
                     invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), int(relu), y.data_ptr())
        if rc:
            _lib.check(rc, "scnattn_bn_apply")
        # bn -> relu without a residual in between (bn1/bn2 of a bottleneck), fp32: the backward pass recomputes the
        # ReLU mask from z and never reads y
        mask_from_z = BN_MASK_FROM_Z and bool(relu) and res is None and not bf16
        ctx.save_for_backward(z, y if (relu and not mask_from_z) else None, mean, invstd, gamma,
                              beta if mask_from_z else None)
        ctx.cfg = (bool(training), bool(relu), res is not None, bf16)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, y, mean, invstd, gamma, beta = ctx.saved_tensors
        training, relu, has_res, bf16 = ctx.cfg
        N, Cn, H, W = z.shape
        R = N * H * W
        if dy.dtype != z.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(z.dtype).contiguous(memory_format=torch.channels_last)
        need = ctx.needs_input_grad
        dz = torch.empty_like(z) if need[0] else None
        dres = torch.empty_like(z) if (has_res and need[1]) else None
        dgb = torch.empty((2, Cn), device=z.device, dtype=torch.float32)
        _, _, f_bwd, raw_stream = _bn_fns()
        rc = f_bwd(raw_stream(z.device.index), R, Cn, dy.data_ptr(), None if y is None else y.data_ptr(), z.data_ptr(),
                   bf16, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), None if beta is None else beta.data_ptr(),
                   int(relu), int(training),
                   _bn_workspace(z.device, Cn).data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                   None if dz is None else dz.data_ptr(), None if dres is None else dres.data_ptr())
        if rc:
            _lib.check(rc, "scnattn_bn_bwd")
        return (dz, dres, dgb[1] if need[2] else None, dgb[0] if need[3] else None, None, None, None, None, None, None)


def bn_act(z, res, gamma, beta, run_mean, run_var, training, momentum, eps, relu):
    return _BNAct.apply(z, res, gamma, beta, run_mean, run_var, training, momentum, eps, relu)


# ----------------------------------------------------------------------------------------------
# The train step's loss (trains/attention_scn.py:222-236) on the unpacked (B, T, V) scores
# ----------------------------------------------------------------------------------------------
class _CaptionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, alphas, caps_sorted, dl_dev, n_tokens, alpha_c):
        require_cuda(scores, alphas, caps_sorted, dl_dev)
        scores = f32c(scores)
        alphas = None if alphas is None else f32c(alphas)
        B, T, V = scores.shape
        P = alphas.shape[2] if alphas is not None else 0
        if caps_sorted.dtype != torch.int64 or caps_sorted.dim() != 2 or caps_sorted.stride(1) != 1 \
                or caps_sorted.shape[0] != B or caps_sorted.shape[1] < T + 1:
            raise RuntimeError("caption_loss: caps_sorted must be int64 (B, >= T+1) with unit column stride")
        if dl_dev.dtype != torch.int32 or dl_dev.numel() != B or not dl_dev.is_contiguous():
            raise RuntimeError("caption_loss: decode lengths must be a contiguous int32 vector of B entries")
        if alphas is not None and tuple(alphas.shape[:2]) != (B, T):
            raise RuntimeError("caption_loss: alphas must be (B, T, P)")
        dev = scores.device
        ws = torch.empty(2 * B * T + B * P + B + 1, device=dev, dtype=torch.float32)
        row_lse, row_loss = ws[:B * T], ws[B * T:2 * B * T]
        sm1, reg_part = ws[2 * B * T:2 * B * T + B * P], ws[2 * B * T + B * P:2 * B * T + B * P + B]
        loss = ws[-1:]
        tgt = C.c_void_p(caps_sorted.data_ptr() + 8)            # targets = caps_sorted[:, 1:]
        call("scnattn_caption_loss_fwd", stream_of(scores), B, T, V, P, ptr(scores), tgt, caps_sorted.stride(0),
             ptr(dl_dev), int(n_tokens), ptr(alphas), float(alpha_c), ptr(row_lse), ptr(row_loss),
             ptr(sm1) if P else None, ptr(reg_part) if P else None, ptr(loss))
        ctx.save_for_backward(scores, caps_sorted, dl_dev, ws)
        ctx.meta = (B, T, V, P, int(n_tokens), float(alpha_c), alphas is not None)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        scores, caps_sorted, dl_dev, ws = ctx.saved_tensors
        B, T, V, P, n_tokens, alpha_c, has_alpha = ctx.meta
        g = f32c(g).reshape(1)
        dscores = torch.empty_like(scores) if ctx.needs_input_grad[0] else None
        dalphas = torch.empty((B, T, P), device=scores.device, dtype=torch.float32) \
            if (has_alpha and ctx.needs_input_grad[1]) else None
        if dscores is None:                                       # the kernel pair always produces d scores
            dscores = torch.empty_like(scores)
        row_lse = ws[:B * T]
        sm1 = ws[2 * B * T:2 * B * T + B * P]
        call("scnattn_caption_loss_bwd", stream_of(scores), B, T, V, P, ptr(scores),
             C.c_void_p(caps_sorted.data_ptr() + 8), caps_sorted.stride(0), ptr(dl_dev), n_tokens, ptr(row_lse),
             ptr(sm1) if dalphas is not None else None, alpha_c, ptr(g), ptr(dscores), ptr(dalphas))
        return (dscores if ctx.needs_input_grad[0] else None), dalphas, None, None, None, None


def caption_loss(scores, caps_sorted, decode_lengths, alphas=None, alpha_c=1.0, dl_dev=None):
    """`CrossEntropyLoss()(pack(scores), pack(caps_sorted[:, 1:])) + alpha_c * ((1 - alphas.sum(1))**2).mean()`
    of trains/attention_scn.py:222-236 without building the packed batch.  scores (B, T, V) and alphas
    (B, T, P) as the decoder returns them, decode_lengths the decoder's list; dl_dev optionally the same
    lengths as an int32 device vector (saves one small host-to-device copy)."""
    T = scores.shape[1]
    n_tokens = sum(min(int(l), T) for l in decode_lengths)
    if dl_dev is None:
        dl_dev = torch.tensor(list(decode_lengths), dtype=torch.int32, device=scores.device)
    return _CaptionLoss.apply(scores, alphas, caps_sorted, dl_dev, n_tokens, alpha_c)


# ----------------------------------------------------------------------------------------------
def clamp_adam_(p, g, m, v, lr, step, clip, beta1=0.9, beta2=0.999, eps=1e-8, gscale=1.0):
    """Fused clamp(+-clip) + Adam on flat fp32 buffers (utils/optimizer.py:1-11 + torch.optim.Adam)."""
    require_cuda(p, g, m, v)
    call("scnattn_clamp_adam", stream_of(p), p.numel(), ptr(p), ptr(g), ptr(m), ptr(v), lr, beta1, beta2, eps,
         int(step), float(clip if clip is not None else 0.0), gscale)
