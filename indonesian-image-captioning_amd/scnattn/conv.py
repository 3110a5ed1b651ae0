"""Fused Bottleneck of the ResNet-152 trunk on MI355X (fp32, channels-last), one autograd node per block.

torchvision's Bottleneck behind the reference's encoder (models/encoders/caption.py:17-22) is
    out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity),   identity = x | bn_d(conv_d(x))
with conv1 / conv3 / conv_d 1x1 and conv2 3x3.  On channels-last maps a 1x1 convolution is the GEMM
[R = N*H*W, Cin] x [Cin, Cout] and a 3x3 one an implicit GEMM; every convolution of the block -- forward, d input,
d weight -- runs on the hand-written kernels of csrc/cgemm.hip / csrc/conv3.hip, and the BatchNorm work rides on them:

  forward   conv1  + statistics epilogue (sum, sum^2 of z1 per channel)      -> no bn1 statistics pass
            bn1 apply + relu -> a1 (materialised: the 3x3's zero padding must be zeros of a1, not of z1)
            conv2 3x3 (implicit GEMM, strided for layerN.0) + statistics epilogue for bn2
            conv3 with the bn2+relu PROLOGUE on its input operand (a2 is never written or read)
                  + statistics epilogue for bn3                                 -> no bn2 apply, no bn3 statistics pass
            bn3 apply + identity + relu -> out
  backward  bn3 (two passes) -> dz3, d identity
            conv3 wgrad with the bn2+relu prologue on its activation operand (a2 recomputed on load)
            conv3 dgrad with the MASK epilogue: g2 = d a2 * [a2 > 0] + the two bn2-backward column sums
            bn2 element-wise half -> dz2
            conv2 wgrad: the halo-staged kernel (stride 1) / the gathered form (stride 2), on the side stream
            conv2 dgrad: nine flipped taps over K (stride 1) / four parity classes of d-input pixels (stride 2)
            bn1 backward (two passes) ; conv1 wgrad ; conv1 dgrad ACCUMULATING into d identity (beta = 1)
The strided 1x1 downsample convolution gathers its input rows inside the kernel.  No library (MIOpen / rocBLAS) kernel
runs in a block; `SCNATTN_CONV3=miopen` swaps conv2 back to MIOpen for A/B measurements only.

`Bottleneck.forward` (scnattn/resnet.py) calls `bottleneck()` for fp32 CUDA inputs in training mode and
scnattn/conv16.py's mixed-precision twin for bf16 ones; everything else (eval mode, CPU structure tests) takes the
unfused module path."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvExtra

ENABLED = True          # class-wide switch: tests / A-B runs compare against the unfused path
SIDE_WGRAD = True       # weight gradients on a second HIP stream (see _Side)
CONV3 = os.environ.get("SCNATTN_CONV3", "hip")     # conv2 (3x3): "hip" = the hand-written kernels (the product path);
                        # "miopen" = MIOpen for all three directions (A/B measurements; tools/, tests that ask for it)
if CONV3 not in ("hip", "miopen"):
    raise RuntimeError("SCNATTN_CONV3 must be 'hip' or 'miopen' (the per-shape stopwatch of round 2 is gone)")
W3_SLICES = int(os.environ.get("SCNATTN_W3_SLICES", "0"))   # tuning: K slices of the halo-staged 3x3 weight gradient (0: policy)

_bufs = {}


def _buffers(dev):
    """Scratch per (device, current stream), reused in stream order: split-K slabs of the GEMMs and the statistics
    partials.  Per stream because two trunks may run at once (the frozen tagger beside the caption encoder)."""
    key = (dev, torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))
    b = _bufs.get(key)
    if b is None:
        b = _bufs[key] = (torch.empty(16 << 20, device=dev, dtype=torch.float32),     # 64 MiB split-K slabs
                          torch.empty(2 << 20, device=dev, dtype=torch.float32),      # [64-row blocks][2][C] partials
                          torch.empty(2 << 20, device=dev, dtype=torch.float32))      # BN chunk partials (bn_stats)
    return b


def _concurrent_stream(dev, attempts=8, beside=()):
    """A stream whose kernels really run beside the current stream's.  HIP multiplexes its streams onto a few hardware
    queues (4 by default); two streams that land on the same queue are serialised, and which ones collide depends on
    how many streams the process created before (RCCL's, the allocator's, another module's).  Measured on MI355X: with
    a torch.distributed group initialised first, the side stream shared the main stream's queue and the step lost all
    of its overlap (47.8 instead of 40.6 ms).  So the stream is chosen by experiment: a spin kernel occupies the main
    stream, a tiny kernel is enqueued on the candidate; if the tiny kernel finishes while the spin kernel is still
    running, the two are concurrent.  `beside`: further streams the new one must not collide with either."""
    main = torch.cuda.current_stream(dev)
    probe = torch.zeros(64, device=dev)
    tried = []
    with torch.cuda.device(dev):
        for i in range(attempts):
            cand = torch.cuda.Stream(device=dev)
            tried.append(cand)                 # keep it referenced: the pool hands out a different stream next time
            with torch.cuda.stream(cand):
                probe.add_(1.0)                # code object / allocator warm-up on the candidate
            torch.cuda.synchronize(dev)
            busy = []
            for st in (main,) + tuple(beside):
                with torch.cuda.stream(st):
                    torch.cuda._sleep(40_000_000)      # a few milliseconds of spinning
                    ev = torch.cuda.Event()
                    ev.record(st)
                    busy.append(ev)
            with torch.cuda.stream(cand):
                probe.add_(1.0)
                done_side = torch.cuda.Event()
                done_side.record(cand)
            done_side.synchronize()
            concurrent = not any(ev.query() for ev in busy)
            torch.cuda.synchronize(dev)
            if concurrent:
                return cand, "attempt %d of %d" % (i + 1, attempts)
    return tried[0], "no concurrent stream found in %d attempts" % attempts


class _Side:
    """Weight gradients off the critical path.  In a block's backward pass the three (four) weight-gradient products
    feed nothing until the optimizer step, while the d-input chain is strictly serial and every kernel in it has a
    head (first loads) and a tail (last stores) during which most of the chip idles.  They are enqueued on a second
    HIP stream, forked from the main stream by an event once their inputs exist, and joined lazily -- `join()` is called
    by whoever first READS the gradients (FlatBuffer.gather: optimizer step or a data-parallel bucket).  Tensors the
    side stream touches are tagged with record_stream so the caching allocator keeps them alive."""

    def __init__(self, dev):
        self.stream, self.probe = _concurrent_stream(dev)
        self.ws = torch.empty(16 << 20, device=dev, dtype=torch.float32)
        self.pending = None
        self.join_task = -1

    def fork(self, main, *tensors):
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)
        for t in tensors:
            if t is not None:
                t.record_stream(self.stream)
        return self.stream.cuda_stream

    def mark(self):
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.pending = ev
        # Inside an autograd sweep: join at its end, so that `loss.backward()` keeps its contract -- every .grad is
        # ready in stream order on the caller's stream, whatever optimizer / clipping code reads it next (the
        # reference's own loop reads p.grad right after backward: utils/optimizer.py:1-11, trains/attention_scn.py:244-252).
        task = torch._C._current_graph_task_id()
        if task != -1 and task != self.join_task:
            self.join_task = task
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        self.join_task = -1
        self.join()

    def join(self, main=None):
        if self.pending is not None:
            (main or torch.cuda.current_stream(self.stream.device)).wait_event(self.pending)
            self.pending = None


_sides = {}


def _side(dev):
    sd = _sides.get(dev)
    if sd is None:
        sd = _sides[dev] = _Side(dev)
    return sd


def _sweep_id():
    """Id of the autograd sweep this thread is executing (-1 outside one)."""
    return torch._C._current_graph_task_id()


def side_ok(*params):
    """May the gradients of `params` be produced on the side stream?  Only while autograd merely STORES what backward
    returns for them: a leaf whose .grad is None and that this sweep has not produced a gradient for yet
    (AccumulateGrad keeps the tensor; no kernel touches it before the join at the end of the sweep).  A non-leaf's
    gradient is consumed by the next backward node on the main stream at once; an existing .grad -- or a second use of
    the same weight in one sweep -- is added to on the main stream at once: then everything the side stream still has
    in flight is joined first and the gradients are produced in line."""
    if not SIDE_WGRAD:
        return False
    task = _sweep_id()
    ok = True
    for p in params:
        if p is None:
            continue
        if not p.is_leaf or p.grad is not None or (task != -1 and getattr(p, "_scn_sweep", None) == task):
            ok = False
    if not ok:
        for sd in _sides.values():
            ev = torch.cuda.Event()
            ev.record(sd.stream)
            torch.cuda.current_stream(sd.stream.device).wait_event(ev)
    return ok


def join_side_streams():
    """Make the current stream of every device wait for the weight gradients still running on a side stream."""
    for sd in _sides.values():
        sd.join()


_fn = None


def _fns():
    global _fn
    if _fn is None:
        h = _lib.lib()
        _fn = (h, torch._C._cuda_getCurrentRawStream)
    return _fn


def _chk(rc, what):
    if rc:
        _lib.check(rc, what)


def _as2d(t4):
    """(N, C, H, W) channels-last -> its [N*H*W, C] matrix view (no copy)."""
    n, c, h, w = t4.shape
    return t4.permute(0, 2, 3, 1).reshape(n * h * w, c)


def _as4d(t2, n, h, w):
    """[N*H*W, C] -> the (N, C, H, W) channels-last view MIOpen takes (no copy)."""
    return t2.view(n, h, w, t2.shape[1]).permute(0, 3, 1, 2)


def _conv_fwd(h, st, x2, w2, R, Cin, Cout, ws, ex):
    y = torch.empty((R, Cout), device=x2.device, dtype=torch.float32)
    _chk(h.scnattn_conv1x1_fwd(st, R, Cin, Cout, x2.data_ptr(), w2.data_ptr(), y.data_ptr(), C.byref(ex), ws.data_ptr(),
                               ws.numel()), "scnattn_conv1x1_fwd")
    return y


def _shift(bn):
    """Conditioning shift s for the sums of (z - s), (z - s)^2 a statistics epilogue takes: the PREVIOUS step's batch mean
    of this BatchNorm (the running mean before the first step).  Any vector works mathematically; it must not be the
    running mean itself, because the kernel that finalizes the statistics updates that in place while workgroups that
    started later still read the shift (include/scnattn.h: scnattn_bn_apply_fin)."""
    s = getattr(bn, "_scn_shift", None)
    if s is None or s.device != bn.running_mean.device or s.shape != bn.running_mean.shape:
        s = bn.running_mean.detach().clone()
    return s


def _finalize(h, st, R, Cn, part, bn_mod, shift, gamma, beta, want_ss):
    """Statistics only (the consumer normalises on load): mean / invstd / running statistics / folded {scale, shift}."""
    dev = part.device
    stats = torch.empty((2, Cn), device=dev, dtype=torch.float32)
    ss = torch.empty((Cn, 2), device=dev, dtype=torch.float32) if want_ss else None
    _chk(h.scnattn_bn_finalize(st, R, Cn, part.data_ptr(), h.scnattn_cgemm_stat_ld(R), h.scnattn_cgemm_row_tiles(R),
                               shift.data_ptr(), bn_mod.eps, bn_mod.momentum, stats[0].data_ptr(), stats[1].data_ptr(),
                               bn_mod.running_mean.data_ptr(), bn_mod.running_var.data_ptr(),
                               gamma.data_ptr() if want_ss else None, beta.data_ptr() if want_ss else None,
                               ss.data_ptr() if want_ss else None), "scnattn_bn_finalize")
    bn_mod._scn_shift = stats[0]
    return stats, ss


def _apply_fin(h, st, R, Cn, z, res, part, bn_mod, shift, gamma, beta, relu):
    """y = [relu](bn(z) [+ res]) with the statistics finalized inside the same launch (csrc/batchnorm.hip)."""
    stats = torch.empty((2, Cn), device=z.device, dtype=torch.float32)
    y = torch.empty_like(z)
    _chk(h.scnattn_bn_apply_fin(st, R, Cn, z.data_ptr(), None if res is None else res.data_ptr(),
                                1 if z.dtype == torch.bfloat16 else 0, part.data_ptr(),
                                h.scnattn_cgemm_stat_ld(R), h.scnattn_cgemm_row_tiles(R), shift.data_ptr(), bn_mod.eps,
                                bn_mod.momentum, gamma.data_ptr(), beta.data_ptr(), 1 if relu else 0, y.data_ptr(),
                                stats[0].data_ptr(), stats[1].data_ptr(), bn_mod.running_mean.data_ptr(),
                                bn_mod.running_var.data_ptr(), None), "scnattn_bn_apply_fin")
    bn_mod._scn_shift = stats[0]
    return y, stats


def _bwd_reduce(h, st, R, Cn, dy, y, z, stats, relu, bnpart, want_g):
    """BatchNorm(+ReLU) backward, first half: g = dy * [y > 0] (or dy) and the channel-major partial sums of g, g*xhat."""
    g = torch.empty((R, Cn), device=dy.device, dtype=dy.dtype) if want_g else None
    nch = C.c_int(0)
    _chk(h.scnattn_bn_bwd_reduce(st, R, Cn, dy.data_ptr(), None if y is None else y.data_ptr(), z.data_ptr(),
                                 1 if dy.dtype == torch.bfloat16 else 0, stats[0].data_ptr(), stats[1].data_ptr(), 1 if relu else 0, bnpart.data_ptr(),
                                 bnpart.numel() // (2 * Cn), None if g is None else g.data_ptr(), C.byref(nch)),
         "scnattn_bn_bwd_reduce")
    return g, nch.value


def _bwd_dx_fin(h, st, R, Cn, g, z, stats, gamma, partial, ldp, nchunk, dz):
    """Second half: d beta / d gamma summed from the partials and dz, one launch; dz may alias g."""
    dgb = torch.empty((2, Cn), device=g.device, dtype=torch.float32)
    _chk(h.scnattn_bn_bwd_dx_fin(st, R, Cn, g.data_ptr(), z.data_ptr(), 1 if g.dtype == torch.bfloat16 else 0,
                                 stats[0].data_ptr(), stats[1].data_ptr(),
                                 gamma.data_ptr(), partial.data_ptr(), ldp, nchunk, dgb[0].data_ptr(), dgb[1].data_ptr(),
                                 dz.data_ptr()), "scnattn_bn_bwd_dx_fin")
    return dgb


def _grad_out(w):
    """Where a weight gradient is written: the parameter's slice of the flat gradient buffer when it has one
    (scnattn/flat.py; saves the optimizer's gather copy), a fresh tensor otherwise.

    The flat slice is handed out as a FRESH alias (`detach()`): AccumulateGrad steals a gradient only when nobody else
    holds the tensor object, and the persistent view is held by the FlatBuffer -- returning that object made autograd
    clone it on the main stream while the side stream could still be writing it.  And only for the first gradient of a
    sweep into an empty .grad: a kernel writing into the slice a second time would overwrite what autograd is about to
    add to (an existing .grad after `FlatBuffer.gather` without `zero_grad`, or a weight used twice in one graph)."""
    gv = getattr(w, "_scn_flat_grad", None)
    task = _sweep_id()
    first = task == -1 or getattr(w, "_scn_sweep", None) != task
    w._scn_sweep = task
    if gv is None or w.grad is not None or not first:
        return torch.empty_like(w)
    return gv.detach()


def _wt(h, st, w, cout, cin):
    """[Cout][Cin] 1x1 weight -> [Cin][Cout] (<= 4 MB, one small kernel): conv1's d input then runs with both operands
    on the k-contiguous LDS image (a [K][N] weight with N = Cin large is the slow layout: 56 vs 40 us on layer3)."""
    wt = torch.empty((cin, cout), device=w.device, dtype=torch.float32)
    _chk(h.scnattn_transpose2d(st, cout, cin, w.data_ptr(), cin, wt.data_ptr(), cout), "scnattn_transpose2d")
    return wt


class _BottleneckFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd):
        h, raw_stream = _fns()
        dev = x.device
        st = raw_stream(dev.index)
        ws, part, bnpart = _buffers(dev)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        N, Cin, Hi, Wi = x.shape
        p = w1.shape[0]
        C4 = w3.shape[0]
        s = mod.stride
        Ho, Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
        Rin, Rout = N * Hi * Wi, N * Ho * Wo
        x2 = _as2d(x)
        bn1, bn2, bn3 = mod.bn1, mod.bn2, mod.bn3
        # conv1 (+ bn1 statistics) -> bn1 apply + relu with the finalize inside
        sh1 = _shift(bn1)
        ex = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=sh1.data_ptr())
        z1 = _conv_fwd(h, st, x2, w1.view(p, Cin), Rin, Cin, p, ws, ex)
        a1, st1 = _apply_fin(h, st, Rin, p, z1, None, part, bn1, sh1, g1, b1, True)
        # conv2 (3x3, strided for layerN.0): the implicit-GEMM mode of the same kernel with the bn2 statistics epilogue
        sh2 = _shift(bn2)
        if CONV3 == "hip":
            z2 = torch.empty((Rout, p), device=dev, dtype=torch.float32)
            ex3 = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=sh2.data_ptr())
            _chk(h.scnattn_conv3x3_fwd(st, N, Hi, Wi, p, p, s, a1.data_ptr(), w2.data_ptr(), z2.data_ptr(), C.byref(ex3),
                                       ws.data_ptr(), ws.numel()), "scnattn_conv3x3_fwd")
            st2, ss2 = _finalize(h, st, Rout, p, part, bn2, sh2, g2, b2, True)
        else:       # A/B only: MIOpen + a statistics pass over z2
            z4 = torch.ops.aten.convolution(_as4d(a1, N, Hi, Wi), w2, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1)
            if not z4.is_contiguous(memory_format=torch.channels_last):
                z4 = z4.contiguous(memory_format=torch.channels_last)
            z2 = _as2d(z4)
            st2 = torch.empty((2, p), device=dev, dtype=torch.float32)
            ss2 = torch.empty((p, 2), device=dev, dtype=torch.float32)
            _chk(h.scnattn_bn_stats_fold(st, Rout, p, z2.data_ptr(), bn2.eps, bn2.momentum, bnpart.data_ptr(),
                                         st2[0].data_ptr(), st2[1].data_ptr(), bn2.running_mean.data_ptr(),
                                         bn2.running_var.data_ptr(), g2.data_ptr(), b2.data_ptr(), ss2.data_ptr()),
                 "scnattn_bn_stats_fold")
        # conv3 with the bn2+relu prologue (+ bn3 statistics into `part`)
        sh3 = _shift(bn3)
        ex = ConvExtra(pro=1, epi=1, pro_ss=ss2.data_ptr(), stat_partial=part.data_ptr(), stat_shift=sh3.data_ptr())
        z3 = _conv_fwd(h, st, z2, w3.view(C4, p), Rout, p, C4, ws, ex)
        # identity (the downsample branch keeps its statistics in the second partial buffer: `part` is still pending)
        zd = std = None
        if wd is not None:
            bnd = mod.downsample[1]
            shd = _shift(bnd)
            ex = ConvExtra(epi=1, stat_partial=bnpart.data_ptr(), stat_shift=shd.data_ptr(),
                           stride=s, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo)
            zd = _conv_fwd(h, st, x2, wd.view(C4, Cin), Rout, Cin, C4, ws, ex)
            idn, std = _apply_fin(h, st, Rout, C4, zd, None, bnpart, bnd, shd, gd, bd, False)
        else:
            idn = x2
        out, st3 = _apply_fin(h, st, Rout, C4, z3, idn, part, bn3, sh3, g3, b3, True)
        # conv1's d input wants w1 transposed ([Cin][p]: both operands k-contiguous); the weights do not change between this
        # forward pass and its backward pass, so the transpose runs NOW on the idle side stream instead of on the backward
        # pass's critical path
        wt = wt_ev = None
        if ctx.needs_input_grad[1] and SIDE_WGRAD:
            sd = _side(dev)
            wt = torch.empty((Cin, p), device=dev, dtype=torch.float32)
            sw = sd.fork(torch.cuda.current_stream(dev), w1, wt)
            _chk(h.scnattn_transpose2d(sw, p, Cin, w1.data_ptr(), Cin, wt.data_ptr(), p), "scnattn_transpose2d")
            wt_ev = torch.cuda.Event()
            wt_ev.record(sd.stream)
        ctx.wt, ctx.wt_ev = wt, wt_ev
        ctx.geom = (N, Cin, Hi, Wi, p, C4, s, Ho, Wo)
        ctx.has_down = wd is not None
        ctx.save_for_backward(x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd, z1, a1, z2, z3, out, zd, st1, st2, ss2,
                              st3, std)
        return _as4d(out, N, Ho, Wo)

    @staticmethod
    def backward(ctx, dout):
        h, raw_stream = _fns()
        (x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd, z1, a1, z2, z3, out, zd, st1, st2, ss2, st3,
         std) = ctx.saved_tensors
        N, Cin, Hi, Wi, p, C4, s, Ho, Wo = ctx.geom
        dev = x.device
        st = raw_stream(dev.index)
        ws, part, bnpart = _buffers(dev)
        Rin, Rout = N * Hi * Wi, N * Ho * Wo
        need = ctx.needs_input_grad      # (mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd)
        if dout.dtype != torch.float32 or not dout.is_contiguous(memory_format=torch.channels_last):
            dout = dout.float().contiguous(memory_format=torch.channels_last)
        dout2 = _as2d(dout)
        x2 = _as2d(x)
        f32 = dict(device=dev, dtype=torch.float32)
        ld_out, nc_out = h.scnattn_cgemm_stat_ld(Rout), h.scnattn_cgemm_row_tiles(Rout)
        ld_in, nc_in = h.scnattn_cgemm_stat_ld(Rin), h.scnattn_cgemm_row_tiles(Rin)
        # ---- bn3 (+ identity + relu) backward: g = dout * [out > 0] = d identity; dz3 ---------------------------------
        dres, nch = _bwd_reduce(h, st, Rout, C4, dout2, out, z3, st3, True, bnpart, True)
        dz3 = torch.empty((Rout, C4), **f32)
        dgb3 = _bwd_dx_fin(h, st, Rout, C4, dres, z3, st3, g3, bnpart, (nch + 3) & ~3, nch, dz3)
        # ---- conv3: wgrad with a2 recomputed on load, dgrad with the bn2 mask / reduction pass -------------------------
        main = torch.cuda.current_stream(dev)
        side = _side(dev) if side_ok(w1, w2, w3, wd) else None
        dw3 = None
        if need[8]:
            dw3 = _grad_out(w3)
            ex = ConvExtra(pro=2, pro_ss=ss2.data_ptr())
            sw, wsw = (side.fork(main, dz3, z2, ss2, dw3), side.ws) if side else (st, ws)
            _chk(h.scnattn_conv1x1_wgrad(sw, Rout, p, C4, dz3.data_ptr(), z2.data_ptr(), dw3.data_ptr(), C.byref(ex),
                                         wsw.data_ptr(), wsw.numel()), "scnattn_conv1x1_wgrad")
        dz2 = torch.empty((Rout, p), **f32)        # first the masked d a2, then (in place) dz2
        ex = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z2.data_ptr(), emean=st2[0].data_ptr(),
                       einvstd=st2[1].data_ptr(), egamma=g2.data_ptr(), ebeta=b2.data_ptr(), ldz=p,
                       pro_ss=ss2.data_ptr())      # mask = [fma(z2, scale, shift) > 0]: what conv3's prologue evaluated
        _chk(h.scnattn_conv1x1_dgrad(st, Rout, p, C4, dz3.data_ptr(), w3.data_ptr(), 0, 0.0, dz2.data_ptr(), C.byref(ex),
                                     ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
        dgb2 = _bwd_dx_fin(h, st, Rout, p, dz2, z2, st2, g2, part, ld_out, nc_out, dz2)
        # ---- conv2: weight gradient (side stream) and d input; bn1's mask / reduction rides on the d input ---------------
        dw2 = None
        dz1 = torch.empty((Rin, p), **f32)         # first d a1 (masked), then (in place) dz1
        if CONV3 == "hip":
            if need[5]:
                dw2 = _grad_out(w2)
                sw, wsw = (side.fork(main, dz2, a1, dw2), side.ws) if side else (st, ws)
                _chk(h.scnattn_conv3x3_wgrad(sw, N, Hi, Wi, p, p, s, dz2.data_ptr(), a1.data_ptr(), dw2.data_ptr(),
                                             wsw.data_ptr(), wsw.numel(), W3_SLICES), "scnattn_conv3x3_wgrad")
            if s == 1:      # mask = [fma((z1-mean)*invstd, gamma, beta) > 0]: the expression bn1's apply evaluated
                ex = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z1.data_ptr(), emean=st1[0].data_ptr(),
                               einvstd=st1[1].data_ptr(), egamma=g1.data_ptr(), ebeta=b1.data_ptr(), ldz=p)
                _chk(h.scnattn_conv3x3_dgrad(st, N, Hi, Wi, p, p, dz2.data_ptr(), w2.data_ptr(), dz1.data_ptr(), C.byref(ex),
                                             ws.data_ptr(), ws.numel()), "scnattn_conv3x3_dgrad")
                dgb1 = _bwd_dx_fin(h, st, Rin, p, dz1, z1, st1, g1, part, ld_in, nc_in, dz1)
            else:
                da1 = torch.empty((Rin, p), **f32)
                _chk(h.scnattn_conv3x3_dgrad_strided(st, N, Hi, Wi, p, p, s, dz2.data_ptr(), w2.data_ptr(), da1.data_ptr(),
                                                     ws.data_ptr(), ws.numel()), "scnattn_conv3x3_dgrad_strided")
        else:       # A/B only: MIOpen
            dz2_4, a1_4 = _as4d(dz2, N, Ho, Wo), _as4d(a1, N, Hi, Wi)
            if need[5]:
                if side:
                    side.fork(main, dz2, a1)
                with torch.cuda.stream(side.stream if side else main):
                    dw2 = torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                              [False, True, False])[1]
            da1_4 = torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [True, False, False])[0]
            if not da1_4.is_contiguous(memory_format=torch.channels_last):
                da1_4 = da1_4.contiguous(memory_format=torch.channels_last)
            da1 = _as2d(da1_4)
        if not (CONV3 == "hip" and s == 1):     # the mask comes from a1 itself: a1 = relu(bn1(z1)) > 0
            g1m, nch = _bwd_reduce(h, st, Rin, p, da1, a1, z1, st1, True, bnpart, True)
            dgb1 = _bwd_dx_fin(h, st, Rin, p, g1m, z1, st1, g1, bnpart, (nch + 3) & ~3, nch, dz1)
            del g1m, da1
        need_dx = need[1]
        dw1 = None
        if need[2]:
            dw1 = _grad_out(w1)
            sw, wsw = (side.fork(main, dz1, x, dw1), side.ws) if side else (st, ws)
            ex1 = ConvExtra()
            _chk(h.scnattn_conv1x1_wgrad(sw, Rin, Cin, p, dz1.data_ptr(), x2.data_ptr(), dw1.data_ptr(), C.byref(ex1),
                                         wsw.data_ptr(), wsw.numel()), "scnattn_conv1x1_wgrad")
        # ---- identity branch and d x --------------------------------------------------------------------------------
        wt = ctx.wt
        if need_dx:
            if wt is None:
                wt = _wt(h, st, w1, p, Cin)
            else:
                main.wait_event(ctx.wt_ev)
        dwd = dgbd = None
        dx = None
        if ctx.has_down:
            _, nch = _bwd_reduce(h, st, Rout, C4, dres, None, zd, std, False, bnpart, False)
            dzd = torch.empty((Rout, C4), **f32)
            dgbd = _bwd_dx_fin(h, st, Rout, C4, dres, zd, std, gd, bnpart, (nch + 3) & ~3, nch, dzd)
            if need[11]:
                dwd = _grad_out(wd)
                ex = ConvExtra(stride=s, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo)
                sw, wsw = (side.fork(main, dzd, x, dwd), side.ws) if side else (st, ws)
                _chk(h.scnattn_conv1x1_wgrad(sw, Rout, Cin, C4, dzd.data_ptr(), x2.data_ptr(), dwd.data_ptr(),
                                             C.byref(ex), wsw.data_ptr(), wsw.numel()), "scnattn_conv1x1_wgrad")
            if need_dx:
                dx = torch.empty((Rin, Cin), **f32)
                _chk(h.scnattn_conv1x1_dgrad(st, Rin, Cin, p, dz1.data_ptr(), wt.data_ptr(), 1, 0.0,
                                             dx.data_ptr(), None, ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
                dxd = torch.empty((Rout, Cin), **f32)
                _chk(h.scnattn_conv1x1_dgrad(st, Rout, Cin, C4, dzd.data_ptr(), wd.data_ptr(), 0, 0.0, dxd.data_ptr(), None,
                                             ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
                # scatter the strided rows back (2 of the 50 blocks need this: layer3.0, layer4.0)
                dx.view(N, Hi, Wi, Cin)[:, ::s, ::s].add_(dxd.view(N, Ho, Wo, Cin))
        elif need_dx:
            # d x = d identity + dz1 . W1, accumulated in place (beta = 1): no residual-gradient add kernel
            dx = dres
            _chk(h.scnattn_conv1x1_dgrad(st, Rin, Cin, p, dz1.data_ptr(), wt.data_ptr(), 1, 1.0,
                                         dx.data_ptr(), None, ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
        if side:
            side.mark()      # joined by the first reader of the weight gradients (FlatBuffer.gather)
        dx4 = _as4d(dx, N, Hi, Wi) if dx is not None else None
        return (None, dx4, dw1, dgb1[1] if need[3] else None, dgb1[0] if need[4] else None,
                dw2 if need[5] else None, dgb2[1] if need[6] else None, dgb2[0] if need[7] else None,
                dw3, dgb3[1] if need[9] else None, dgb3[0] if need[10] else None,
                dwd, (dgbd[1] if need[12] else None) if dgbd is not None else None,
                (dgbd[0] if need[13] else None) if dgbd is not None else None)


def usable(mod, x):
    """The fused path covers what the train step runs: fp32 CUDA maps, BatchNorm in training mode with running
    statistics and affine parameters, widths that the 16-byte LDS-DMA granules can address."""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and mod.training and not torch.is_autocast_enabled()):
        return False
    for bn in (mod.bn1, mod.bn2, mod.bn3):
        if bn.weight is None or not bn.track_running_stats or bn.momentum is None or bn.weight.dtype != torch.float32:
            return False
    p, cin = mod.conv1.weight.shape[0], mod.conv1.weight.shape[1]
    if p % 16 or cin % 16 or mod.conv2.groups != 1 or mod.conv2.dilation != (1, 1):
        return False
    if mod.conv2.kernel_size != (3, 3) or mod.conv2.padding != (1, 1) or mod.stride not in (1, 2) \
            or not mod.conv2.weight.is_contiguous(memory_format=torch.channels_last):
        return False
    if CONV3 == "hip":      # what the 3x3 kernels address: 32-channel blocks (stride 1), 128-channel
        H, W = x.shape[2], x.shape[3]      # column tiles and even maps (stride 2)
        if mod.stride == 1 and p % 32:
            return False
        if mod.stride == 2 and (p % 128 or H % 2 or W % 2):
            return False
    if mod.downsample is not None:
        d0, d1 = mod.downsample[0], mod.downsample[1]
        if d0.kernel_size != (1, 1) or d0.stride != (mod.stride, mod.stride) or d1.weight is None or d1.momentum is None:
            return False
    return True


def bottleneck(mod, x):
    """One fused forward of `mod` (a scnattn.resnet.Bottleneck); autograd gets a single node."""
    for bn in (mod.bn1, mod.bn2, mod.bn3) + ((mod.downsample[1],) if mod.downsample is not None else ()):
        if not bn.counter_managed and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
    if mod.downsample is not None:
        wd, gd, bd = mod.downsample[0].weight, mod.downsample[1].weight, mod.downsample[1].bias
    else:
        wd = gd = bd = None
    return _BottleneckFn.apply(mod, x, mod.conv1.weight, mod.bn1.weight, mod.bn1.bias, mod.conv2.weight, mod.bn2.weight,
                               mod.bn2.bias, mod.conv3.weight, mod.bn3.weight, mod.bn3.bias, wd, gd, bd)
