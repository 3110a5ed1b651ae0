"""Fused Bottleneck of the ResNet-152 trunk on MI355X (fp32, channels-last), one autograd node per block.

torchvision's Bottleneck behind the reference's encoder (models/encoders/caption.py:17-22) is
    out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity),   identity = x | bn_d(conv_d(x))
with conv1 / conv3 / conv_d 1x1 and conv2 3x3.  On channels-last maps a 1x1 convolution is the GEMM
[R = N*H*W, Cin] x [Cin, Cout]; this module runs every 1x1 convolution -- forward, d input, d weight -- on the
hand-written LDS-DMA pipelined MFMA kernel of csrc/cgemm.hip and lets the BatchNorm work ride on it:

  forward   conv1  + statistics epilogue (sum, sum^2 of z1 per channel)      -> no bn1 statistics pass
            bn1 apply + relu -> a1 (materialised: the 3x3 conv2 is MIOpen's and needs it)
            conv2 (MIOpen) -> z2 ; bn2 statistics pass (also writes the folded scale/shift)
            conv3 with the bn2+relu PROLOGUE on its input operand (a2 is never written or read)
                  + statistics epilogue for bn3                                 -> no bn2 apply, no bn3 statistics pass
            bn3 apply + identity + relu -> out
  backward  bn3 (two passes, as before) -> dz3, d identity
            conv3 wgrad with the bn2+relu prologue on its activation operand (a2 recomputed on load)
            conv3 dgrad with the MASK epilogue: g2 = d a2 * [a2 > 0] + the two bn2-backward column sums
                                                                                -> no bn2 reduction pass
            bn2 element-wise half -> dz2 ; conv2 backward (MIOpen) ; bn1 backward (two passes)
            conv1 wgrad ; conv1 dgrad ACCUMULATING into d identity (beta = 1)   -> no residual-gradient add kernel
What stays a separate pass is what a 3x3 MIOpen convolution forces (it needs materialised, normalised inputs).
The strided 1x1 downsample convolution gathers its input rows inside the kernel.

`Bottleneck.forward` (scnattn/resnet.py) calls `bottleneck()` for fp32 CUDA inputs in training mode; everything else
(eval mode, bf16 autocast, CPU structure tests) takes the unfused module path."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvExtra

ENABLED = True          # class-wide switch: tests / A-B runs compare against the unfused path
SIDE_WGRAD = True       # weight gradients on a second HIP stream (see _Side)
CONV3 = os.environ.get("SCNATTN_CONV3", "auto")   # conv2 (3x3) forward / d input: "hip" = the implicit-GEMM mode of csrc/cgemm.hip, "miopen", or
                        # "auto" = time both once per shape on first use and keep the faster (what MIOpen's own find
                        # step does among its solvers); the 3x3 weight gradient stays on MIOpen (side stream)

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


_c3_choice = {}


def _conv3_use_hip(kind, key, run_hip, run_miopen):
    """-> True when the hand-written 3x3 path is to be used for this (kind, shape).  "auto": both are timed once
    (3 launches each after one warm-up, HIP events) the first time a shape is seen."""
    if CONV3 == "hip":
        return True
    if CONV3 == "miopen":
        return False
    k = (kind,) + key
    c = _c3_choice.get(k)
    if c is None:
        t = []
        for fn in (run_hip, run_miopen):
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn()
            b.record()
            b.synchronize()
            t.append(a.elapsed_time(b))
        c = _c3_choice[k] = t[0] <= t[1] * 1.02
    return c


CONV3_WGRAD = os.environ.get("SCNATTN_CONV3_WGRAD", "miopen")   # 3x3 weight gradient: "miopen" (on the side stream; measured 793 vs 771 images/s) or the implicit-GEMM mode ("hip")


def conv3_choices():
    """{(kind, N, H, W, C[, stride]): "hip" | "miopen"} as decided so far by the per-shape autotune."""
    return {k: ("hip" if v else "miopen") for k, v in _c3_choice.items()}


SIDE_PRIORITY = os.environ.get("SCNATTN_SIDE_PRIORITY", "default")     # "low": lowest stream priority the device offers


def _new_stream(dev, low):
    """PyTorch's stream pool knows two priorities (normal, high); HIP has a third, lower one.  "low": a raw HIP stream of
    the lowest priority (include/scnattn.h scnattn_stream_create) wrapped as an ExternalStream."""
    if low:
        h = _lib.lib()
        lo, hi = C.c_int(0), C.c_int(0)
        _chk(h.scnattn_stream_priority_range(C.byref(lo), C.byref(hi)), "scnattn_stream_priority_range")
        with torch.cuda.device(dev):
            raw = C.c_void_p()
            _chk(h.scnattn_stream_create(lo.value, C.byref(raw)), "scnattn_stream_create")
        st = torch.cuda.ExternalStream(raw.value, device=dev)
        st._scn_priority = (lo.value, hi.value)
        return st
    return torch.cuda.Stream(device=dev)


def _concurrent_stream(dev, attempts=8, beside=(), low=None):
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
            cand = _new_stream(dev, SIDE_PRIORITY == "low" if low is None else low)
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


def side_ok(*params):
    """May the gradients of `params` be produced on the side stream?  Only while autograd merely STORES what backward
    returns for them: a leaf whose .grad is None (AccumulateGrad keeps the tensor; no kernel touches it before the join
    at the end of the sweep).  A non-leaf's gradient is consumed by the next backward node on the main stream at once,
    an existing .grad is added to on the main stream at once."""
    if not SIDE_WGRAD:
        return False
    for p in params:
        if p is None:
            continue
        if not p.is_leaf or p.grad is not None:
            return False
    return True


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


class _Ctx:
    pass


SIDE_WGRAD_TARGET = int(os.environ.get("SCNATTN_SIDE_WGRAD_TARGET", "0"))   # experiment: workgroups a side-stream wgrad aims for


def _wgrad_split(on_side, M, N, K):
    """force_split for a weight-gradient product dW [M][N] over K rows when it runs on the side stream and
    SIDE_WGRAD_TARGET is set: fewer, longer workgroups leave the main stream's convolutions more residency slots."""
    if not (on_side and SIDE_WGRAD_TARGET > 0):
        return 0
    tiles = -(-M // 64) * -(-N // 128)
    return max(1, min(SIDE_WGRAD_TARGET // max(tiles, 1), max(1, K // 128), 128))


def _conv_fwd(h, st, x2, w2, R, Cin, Cout, ws, ex):
    y = torch.empty((R, Cout), device=x2.device, dtype=torch.float32)
    _chk(h.scnattn_conv1x1_fwd(st, R, Cin, Cout, x2.data_ptr(), w2.data_ptr(), y.data_ptr(), C.byref(ex), ws.data_ptr(),
                               ws.numel()), "scnattn_conv1x1_fwd")
    return y


def _finalize(h, st, R, Cn, part, bn_mod, training, gamma, beta, want_ss):
    dev = part.device
    stats = torch.empty((2, Cn), device=dev, dtype=torch.float32)
    ss = torch.empty((Cn, 2), device=dev, dtype=torch.float32) if want_ss else None
    mt = h.scnattn_cgemm_row_tiles(R)
    _chk(h.scnattn_bn_finalize(st, R, Cn, mt, part.data_ptr(), bn_mod.running_mean.data_ptr(), bn_mod.eps,
                               bn_mod.momentum, stats[0].data_ptr(), stats[1].data_ptr(),
                               bn_mod.running_mean.data_ptr(), bn_mod.running_var.data_ptr(),
                               gamma.data_ptr() if want_ss else None, beta.data_ptr() if want_ss else None,
                               ss.data_ptr() if want_ss else None), "scnattn_bn_finalize")
    return stats, ss


def _grad_out(w):
    """Where a weight gradient is written: the parameter's slice of the flat gradient buffer when it has one
    (scnattn/flat.py; saves the optimizer's gather copy), a fresh tensor otherwise."""
    gv = getattr(w, "_scn_flat_grad", None)
    return gv if gv is not None else torch.empty_like(w)


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
        # conv1 (+ bn1 statistics)
        ex = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=bn1.running_mean.data_ptr())
        z1 = _conv_fwd(h, st, x2, w1.view(p, Cin), Rin, Cin, p, ws, ex)
        st1, _ = _finalize(h, st, Rin, p, part, bn1, True, g1, b1, False)
        a1 = torch.empty_like(z1)
        _chk(h.scnattn_bn_apply(st, Rin, p, z1.data_ptr(), None, 0, st1[0].data_ptr(), st1[1].data_ptr(), g1.data_ptr(),
                                b1.data_ptr(), 1, a1.data_ptr()), "scnattn_bn_apply")
        # conv2 (3x3): the implicit-GEMM mode of the same kernel with the bn2 statistics epilogue, or MIOpen + a statistics
        # pass -- whichever is faster for this shape (CONV3)
        a1_4 = _as4d(a1, N, Hi, Wi)
        c3ok = p % 16 == 0 and w2.is_contiguous(memory_format=torch.channels_last)

        def hip_fwd():
            z = torch.empty((Rout, p), device=dev, dtype=torch.float32)
            ex3 = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=bn2.running_mean.data_ptr())
            _chk(h.scnattn_conv3x3_fwd(st, N, Hi, Wi, p, p, s, a1.data_ptr(), w2.data_ptr(), z.data_ptr(), C.byref(ex3),
                                       ws.data_ptr(), ws.numel()), "scnattn_conv3x3_fwd")
            return z

        def miopen_fwd():
            z4 = torch.ops.aten.convolution(a1_4, w2, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1)
            if not z4.is_contiguous(memory_format=torch.channels_last):
                z4 = z4.contiguous(memory_format=torch.channels_last)
            return _as2d(z4)

        if c3ok and _conv3_use_hip("fwd", (N, Hi, Wi, p, s), hip_fwd, miopen_fwd):
            z2 = hip_fwd()
            st2, ss2 = _finalize(h, st, Rout, p, part, bn2, True, g2, b2, True)
        else:
            z2 = miopen_fwd()
            st2 = torch.empty((2, p), device=dev, dtype=torch.float32)
            ss2 = torch.empty((p, 2), device=dev, dtype=torch.float32)
            _chk(h.scnattn_bn_stats_fold(st, Rout, p, z2.data_ptr(), bn2.eps, bn2.momentum, bnpart.data_ptr(),
                                         st2[0].data_ptr(), st2[1].data_ptr(), bn2.running_mean.data_ptr(),
                                         bn2.running_var.data_ptr(), g2.data_ptr(), b2.data_ptr(), ss2.data_ptr()),
                 "scnattn_bn_stats_fold")
        # conv3 with the bn2+relu prologue (+ bn3 statistics)
        ex = ConvExtra(pro=1, epi=1, pro_ss=ss2.data_ptr(), stat_partial=part.data_ptr(),
                       stat_shift=bn3.running_mean.data_ptr())
        z3 = _conv_fwd(h, st, z2, w3.view(C4, p), Rout, p, C4, ws, ex)
        st3, _ = _finalize(h, st, Rout, C4, part, bn3, True, g3, b3, False)
        # identity
        zd = std = None
        if wd is not None:
            bnd = mod.downsample[1]
            ex = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=bnd.running_mean.data_ptr(),
                           stride=s, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo)
            zd = _conv_fwd(h, st, x2, wd.view(C4, Cin), Rout, Cin, C4, ws, ex)
            std, _ = _finalize(h, st, Rout, C4, part, bnd, True, gd, bd, False)
            idn = torch.empty_like(zd)
            _chk(h.scnattn_bn_apply(st, Rout, C4, zd.data_ptr(), None, 0, std[0].data_ptr(), std[1].data_ptr(),
                                    gd.data_ptr(), bd.data_ptr(), 0, idn.data_ptr()), "scnattn_bn_apply")
        else:
            idn = x2
        out = torch.empty((Rout, C4), device=dev, dtype=torch.float32)
        _chk(h.scnattn_bn_apply(st, Rout, C4, z3.data_ptr(), idn.data_ptr(), 0, st3[0].data_ptr(), st3[1].data_ptr(),
                                g3.data_ptr(), b3.data_ptr(), 1, out.data_ptr()), "scnattn_bn_apply")
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
        # ---- bn3 (+ identity + relu) backward: dz3, d identity ------------------------------------------------
        dz3 = torch.empty((Rout, C4), **f32)
        need_res = need[1] or ctx.has_down
        dres = torch.empty((Rout, C4), **f32) if need_res else None
        dgb3 = torch.empty((2, C4), **f32)
        _chk(h.scnattn_bn_bwd(st, Rout, C4, dout2.data_ptr(), out.data_ptr(), z3.data_ptr(), 0, st3[0].data_ptr(),
                              st3[1].data_ptr(), g3.data_ptr(), None, 1, 1, bnpart.data_ptr(), dgb3[0].data_ptr(),
                              dgb3[1].data_ptr(), dz3.data_ptr(), None if dres is None else dres.data_ptr()),
             "scnattn_bn_bwd")
        # ---- conv3: wgrad with a2 recomputed on load, dgrad with the bn2 mask / reduction epilogue ----------------
        main = torch.cuda.current_stream(dev)
        side = _side(dev) if side_ok(w1, w2, w3, wd) else None
        dw3 = None
        if need[8]:
            dw3 = _grad_out(w3)
            ex = ConvExtra(pro=2, pro_ss=ss2.data_ptr(), force_split=_wgrad_split(side is not None, C4, p, Rout))
            sw, wsw = (side.fork(main, dz3, z2, ss2, dw3), side.ws) if side else (st, ws)
            _chk(h.scnattn_conv1x1_wgrad(sw, Rout, p, C4, dz3.data_ptr(), z2.data_ptr(), dw3.data_ptr(), C.byref(ex),
                                         wsw.data_ptr(), wsw.numel()), "scnattn_conv1x1_wgrad")
        g2m = torch.empty((Rout, p), **f32)
        ex = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z2.data_ptr(), emean=st2[0].data_ptr(),
                       einvstd=st2[1].data_ptr(), egamma=g2.data_ptr(), ebeta=b2.data_ptr(), ldz=p,
                       pro_ss=ss2.data_ptr())      # mask = [fma(z2, scale, shift) > 0]: what conv3's prologue evaluated
        _chk(h.scnattn_conv1x1_dgrad(st, Rout, p, C4, dz3.data_ptr(), w3.data_ptr(), 0, 0.0, g2m.data_ptr(), C.byref(ex),
                                     ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
        dgb2 = torch.empty((2, p), **f32)
        _chk(h.scnattn_bn_bwd_finalize(st, p, h.scnattn_cgemm_row_tiles(Rout), part.data_ptr(), dgb2[0].data_ptr(),
                                       dgb2[1].data_ptr()), "scnattn_bn_bwd_finalize")
        dz2 = torch.empty((Rout, p), **f32)
        _chk(h.scnattn_bn_bwd_dx(st, Rout, p, g2m.data_ptr(), z2.data_ptr(), st2[0].data_ptr(), st2[1].data_ptr(),
                                 g2.data_ptr(), dgb2[0].data_ptr(), dgb2[1].data_ptr(), dz2.data_ptr()), "scnattn_bn_bwd_dx")
        del g2m
        # ---- conv2 (MIOpen) -------------------------------------------------------------------------------------
        dz2_4, a1_4 = _as4d(dz2, N, Ho, Wo), _as4d(a1, N, Hi, Wi)
        dw2 = None
        if need[5]:       # weight gradient: side stream when there is one, the same kernel on the main stream otherwise
            if CONV3_WGRAD == "hip" and p % 128 == 0 and w2.is_contiguous(memory_format=torch.channels_last):
                dw2 = torch.empty_like(w2)
                sw, wsw = (side.fork(main, dz2, a1, dw2), side.ws) if side else (st, ws)
                _chk(h.scnattn_conv3x3_wgrad(sw, N, Hi, Wi, p, p, s, dz2.data_ptr(), a1.data_ptr(), dw2.data_ptr(),
                                             wsw.data_ptr(), wsw.numel()), "scnattn_conv3x3_wgrad")
            elif side:
                side.fork(main, dz2, a1)
                with torch.cuda.stream(side.stream):
                    _, dw2, _ = torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [s, s], [1, 1], [1, 1], False,
                                                                    [0, 0], 1, [False, True, False])
            else:
                _, dw2, _ = torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [s, s], [1, 1], [1, 1], False,
                                                                [0, 0], 1, [False, True, False])
        c3ok = s == 1 and p % 16 == 0 and w2.is_contiguous(memory_format=torch.channels_last)

        def hip_dgrad():
            d = torch.empty((Rin, p), **f32)
            _chk(h.scnattn_conv3x3_dgrad(st, N, Hi, Wi, p, p, dz2.data_ptr(), w2.data_ptr(), d.data_ptr(), None,
                                         ws.data_ptr(), ws.numel()), "scnattn_conv3x3_dgrad")
            return _as4d(d, N, Hi, Wi)

        def miopen_dgrad():
            return torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [True, False, False])[0]

        da1_4 = hip_dgrad() if (c3ok and _conv3_use_hip("dgrad", (N, Hi, Wi, p), hip_dgrad, miopen_dgrad)) \
            else miopen_dgrad()
        if not da1_4.is_contiguous(memory_format=torch.channels_last):
            da1_4 = da1_4.contiguous(memory_format=torch.channels_last)
        da1 = _as2d(da1_4)
        # ---- bn1 (+ relu, mask recomputed from z1) ----------------------------------------------------------------
        need_dx = need[1]
        dz1 = torch.empty((Rin, p), **f32)
        dgb1 = torch.empty((2, p), **f32)
        _chk(h.scnattn_bn_bwd(st, Rin, p, da1.data_ptr(), None, z1.data_ptr(), 0, st1[0].data_ptr(), st1[1].data_ptr(),
                              g1.data_ptr(), b1.data_ptr(), 1, 1, bnpart.data_ptr(), dgb1[0].data_ptr(), dgb1[1].data_ptr(),
                              dz1.data_ptr(), None), "scnattn_bn_bwd")
        dw1 = None
        if need[2]:
            dw1 = _grad_out(w1)
            sw, wsw = (side.fork(main, dz1, x, dw1), side.ws) if side else (st, ws)
            ex1 = ConvExtra(force_split=_wgrad_split(side is not None, p, Cin, Rin))
            _chk(h.scnattn_conv1x1_wgrad(sw, Rin, Cin, p, dz1.data_ptr(), x2.data_ptr(), dw1.data_ptr(), C.byref(ex1),
                                         wsw.data_ptr(), wsw.numel()), "scnattn_conv1x1_wgrad")
        # ---- identity branch and d x --------------------------------------------------------------------------------
        dwd = dgbd = None
        dx = None
        if ctx.has_down:
            dzd = torch.empty((Rout, C4), **f32)
            dgbd = torch.empty((2, C4), **f32)
            _chk(h.scnattn_bn_bwd(st, Rout, C4, dres.data_ptr(), None, zd.data_ptr(), 0, std[0].data_ptr(),
                                  std[1].data_ptr(), gd.data_ptr(), None, 0, 1, bnpart.data_ptr(), dgbd[0].data_ptr(),
                                  dgbd[1].data_ptr(), dzd.data_ptr(), None), "scnattn_bn_bwd")
            if need[11]:
                dwd = _grad_out(wd)
                ex = ConvExtra(stride=s, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo)
                sw, wsw = (side.fork(main, dzd, x, dwd), side.ws) if side else (st, ws)
                _chk(h.scnattn_conv1x1_wgrad(sw, Rout, Cin, C4, dzd.data_ptr(), x2.data_ptr(), dwd.data_ptr(),
                                             C.byref(ex), wsw.data_ptr(), wsw.numel()), "scnattn_conv1x1_wgrad")
            if need_dx:
                dx = torch.empty((Rin, Cin), **f32)
                _chk(h.scnattn_conv1x1_dgrad(st, Rin, Cin, p, dz1.data_ptr(), _wt(h, st, w1, p, Cin).data_ptr(), 1, 0.0,
                                             dx.data_ptr(), None, ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
                dxd = torch.empty((Rout, Cin), **f32)
                _chk(h.scnattn_conv1x1_dgrad(st, Rout, Cin, C4, dzd.data_ptr(), wd.data_ptr(), 0, 0.0, dxd.data_ptr(), None,
                                             ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
                # scatter the strided rows back (2 of the 50 blocks need this: layer3.0, layer4.0)
                dx.view(N, Hi, Wi, Cin)[:, ::s, ::s].add_(dxd.view(N, Ho, Wo, Cin))
        elif need_dx:
            # d x = d identity + dz1 . W1, accumulated in place (beta = 1): no residual-gradient add kernel
            dx = dres
            _chk(h.scnattn_conv1x1_dgrad(st, Rin, Cin, p, dz1.data_ptr(), _wt(h, st, w1, p, Cin).data_ptr(), 1, 1.0,
                                         dx.data_ptr(), None, ws.data_ptr(), ws.numel()), "scnattn_conv1x1_dgrad")
        if side:
            side.mark()      # joined by the first reader of the weight gradients (FlatBuffer.gather)
        dx4 = _as4d(dx, N, Hi, Wi) if dx is not None else None
        return (None, dx4, dw1, dgb1[1] if need[3] else None, dgb1[0] if need[4] else None,
                dw2 if need[5] else None, dgb2[1] if need[6] else None, dgb2[0] if need[7] else None,
                dw3, dgb3[1] if need[9] else None, dgb3[0] if need[10] else None,
                dwd, (dgbd[1] if need[12] else None) if dgbd is not None else None,
                (dgbd[0] if need[13] else None) if dgbd is not None else None)


# Identity blocks through the whole-block C drivers (one call forward, one backward): SCNATTN_BLOCK_DRIVER=1.  Measured on
# one MI355X: the step is GPU-bound either way; the per-call path with its per-shape conv2 autotune is 1.5 % faster on a
# single GPU (789-800 vs 777-782 images/s) and equal within noise under the data-parallel hooks (--force-dist, one rank:
# 760 / 770 vs 766 / 769 images/s with the torch / C-ABI RCCL back end).  So the drivers are an entry point for hosts that
# cannot afford ~40 Python-level calls per block (include/scnattn.h: scnattn_block_*), not the default of this one.
_drv = os.environ.get("SCNATTN_BLOCK_DRIVER", "0")
C_DRIVER = None if _drv == "auto" else (_drv != "0")


def _use_c_driver():
    if C_DRIVER is not None:
        return C_DRIVER
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

_block_sizes = {}
_scratch_rings = {}


def _scratch(dev, nfloats, main):
    """Backward scratch of the block driver from a ring of three buffers per size.  A fresh torch.empty per call would
    have to be record_stream()-ed for the side stream, and with the host now far ahead of the GPU the caching allocator
    could not reuse such blocks in time (it falls back to hipMalloc, which stalls the device).  A ring slot is reused two
    blocks later; the main stream first waits for the event the side stream recorded when it last used the slot."""
    key = (dev, nfloats)
    ring = _scratch_rings.get(key)
    if ring is None:
        ring = _scratch_rings[key] = {"i": 0, "slots": [[torch.empty(nfloats, device=dev, dtype=torch.float32), None]
                                                         for _ in range(3)]}
    slot = ring["slots"][ring["i"]]
    ring["i"] = (ring["i"] + 1) % 3
    if slot[1] is not None:
        main.wait_event(slot[1])
    return slot


def _block_struct(mod, N, Hi, Wi):
    from ._lib import Block
    bn1, bn2, bn3 = mod.bn1, mod.bn2, mod.bn3
    p, cin = mod.conv1.weight.shape[0], mod.conv1.weight.shape[1]
    b = Block()
    b.N, b.Hi, b.Wi, b.Cin, b.P, b.stride, b.has_down = N, Hi, Wi, cin, p, mod.stride, 0
    b.eps1, b.mom1, b.eps2, b.mom2, b.eps3, b.mom3 = bn1.eps, bn1.momentum, bn2.eps, bn2.momentum, bn3.eps, bn3.momentum
    b.w1, b.g1, b.b1 = mod.conv1.weight.data_ptr(), bn1.weight.data_ptr(), bn1.bias.data_ptr()
    b.w2, b.g2, b.b2 = mod.conv2.weight.data_ptr(), bn2.weight.data_ptr(), bn2.bias.data_ptr()
    b.w3, b.g3, b.b3 = mod.conv3.weight.data_ptr(), bn3.weight.data_ptr(), bn3.bias.data_ptr()
    b.rm1, b.rv1 = bn1.running_mean.data_ptr(), bn1.running_var.data_ptr()
    b.rm2, b.rv2 = bn2.running_mean.data_ptr(), bn2.running_var.data_ptr()
    b.rm3, b.rv3 = bn3.running_mean.data_ptr(), bn3.running_var.data_ptr()
    key = (N, Hi, Wi, cin, p, mod.stride)
    sz_ = _block_sizes.get(key)
    if sz_ is None:
        h, _ = _fns()
        sv, sc = C.c_size_t(), C.c_size_t()
        offs = (C.c_long * 8)()
        _chk(h.scnattn_block_sizes(C.byref(b), C.byref(sv), C.byref(sc), offs), "scnattn_block_sizes")
        sz_ = _block_sizes[key] = (sv.value, sc.value, tuple(offs))
    return b, sz_


class _BlockFnC(torch.autograd.Function):
    """An identity Bottleneck (no downsample, stride 1) through csrc/bottleneck.cpp: ONE C call enqueues the ~15 forward
    kernels, ONE the ~25 backward kernels (3x3 conv2 forward / d input as implicit GEMMs of the same kernel family);
    only conv2's weight gradient is left to MIOpen, on the side stream."""

    @staticmethod
    def forward(ctx, mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3):
        h, raw_stream = _fns()
        dev = x.device
        st = raw_stream(dev.index)
        ws, part, bnpart = _buffers(dev)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        N, Cin, Hi, Wi = x.shape
        blk, (svf, scf, offs) = _block_struct(mod, N, Hi, Wi)
        saved = torch.empty(svf, device=dev, dtype=torch.float32)
        out = torch.empty((N * Hi * Wi, Cin), device=dev, dtype=torch.float32)
        _chk(h.scnattn_block_fwd(st, C.byref(blk), x.data_ptr(), saved.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                 ws.numel(), part.data_ptr(), bnpart.data_ptr()), "scnattn_block_fwd")
        ctx.mod = mod
        ctx.geom = (N, Cin, Hi, Wi)
        ctx.save_for_backward(x, saved, out, w1, g1, b1, w2, g2, b2, w3, g3, b3)
        return _as4d(out, N, Hi, Wi)

    @staticmethod
    def backward(ctx, dout):
        from ._lib import BlockGrads
        h, raw_stream = _fns()
        x, saved, out, w1, g1, b1, w2, g2, b2, w3, g3, b3 = ctx.saved_tensors
        mod = ctx.mod
        N, Cin, Hi, Wi = ctx.geom
        dev = x.device
        st = raw_stream(dev.index)
        ws, part, bnpart = _buffers(dev)
        need = ctx.needs_input_grad      # (mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3)
        if dout.dtype != torch.float32 or not dout.is_contiguous(memory_format=torch.channels_last):
            dout = dout.float().contiguous(memory_format=torch.channels_last)
        blk, (svf, scf, offs) = _block_struct(mod, N, Hi, Wi)
        p = blk.P
        R = N * Hi * Wi
        main = torch.cuda.current_stream(dev)
        slot = _scratch(dev, scf, main)
        scratch = slot[0]
        dx = torch.empty((R, Cin), device=dev, dtype=torch.float32)      # always: it first receives d identity
        dw1 = _grad_out(w1) if need[2] else None
        dw3 = _grad_out(w3) if need[8] else None
        gr = BlockGrads(None if dw1 is None else dw1.data_ptr(), None if dw3 is None else dw3.data_ptr())
        side = _side(dev) if side_ok(w1, w2, w3) else None
        if side is not None:
            for t_ in (saved, x):
                t_.record_stream(side.stream)
        def run(phase):
            _chk(h.scnattn_block_bwd(st, None if side is None else side.stream.cuda_stream, C.byref(blk), x.data_ptr(),
                                     saved.data_ptr(), out.data_ptr(), dout.data_ptr(), scratch.data_ptr(), dx.data_ptr(),
                                     C.byref(gr), ws.data_ptr(), None if side is None else side.ws.data_ptr(), ws.numel(),
                                     part.data_ptr(), bnpart.data_ptr(), phase), "scnattn_block_bwd")

        dw2 = None
        if need[5]:      # conv2's weight gradient: MIOpen, on the side stream, as soon as dz2 (scratch) exists
            run(1)
            a1_4 = _as4d(saved[offs[0]:offs[0] + R * p].view(R, p), N, Hi, Wi)
            dz2_4 = _as4d(scratch[offs[1]:offs[1] + R * p].view(R, p), N, Hi, Wi)
            if side is not None:
                side.fork(main)
                with torch.cuda.stream(side.stream):
                    dw2 = torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                              [False, True, False])[1]
            else:
                dw2 = torch.ops.aten.convolution_backward(dz2_4, a1_4, w2, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                          [False, True, False])[1]
            run(2)
        else:
            run(0)
        if side is not None:
            side.mark()
            slot[1] = side.pending        # the ring slot may be rewritten once the side stream got here
        # the BatchNorm gradients leave the ring slot (it is rewritten two blocks later): one small copy
        gb = scratch[offs[5]:offs[5] + 12 * p].clone()
        db1, dg1, db2, dg2 = gb[0:p], gb[p:2 * p], gb[2 * p:3 * p], gb[3 * p:4 * p]
        db3, dg3 = gb[4 * p:8 * p], gb[8 * p:12 * p]
        return (None, _as4d(dx, N, Hi, Wi) if need[1] else None, dw1, dg1 if need[3] else None, db1 if need[4] else None,
                dw2, dg2 if need[6] else None, db2 if need[7] else None, dw3, dg3 if need[9] else None,
                db3 if need[10] else None)


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
        # (64-channel 3x3 convolutions -- layer1 -- are the one shape where MIOpen's kernel wins clearly: those blocks
        #  keep the per-call path with its autotuned conv2)
        if _use_c_driver() and mod.stride == 1 and mod.conv2.weight.is_contiguous(memory_format=torch.channels_last) \
                and mod.conv1.weight.shape[1] == 4 * mod.conv1.weight.shape[0] and mod.conv1.weight.shape[0] >= 128:
            return _BlockFnC.apply(mod, x, mod.conv1.weight, mod.bn1.weight, mod.bn1.bias, mod.conv2.weight,
                                   mod.bn2.weight, mod.bn2.bias, mod.conv3.weight, mod.bn3.weight, mod.bn3.bias)
    return _BottleneckFn.apply(mod, x, mod.conv1.weight, mod.bn1.weight, mod.bn1.bias, mod.conv2.weight, mod.bn2.weight,
                               mod.bn2.bias, mod.conv3.weight, mod.bn3.weight, mod.bn3.bias, wd, gd, bd)
