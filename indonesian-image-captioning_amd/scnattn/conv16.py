"""Mixed-precision (bf16) Bottleneck of the ResNet-152 trunk on MI355X -- BASELINE configs[4] -- one autograd node per block.

Same block as scnattn/conv.py (torchvision's Bottleneck behind models/encoders/caption.py:17-22), same split of the work,
but every feature map and every gradient map is bf16 in HBM and every convolution runs on v_mfma_f32_32x32x16_bf16:

  * operands: bf16 maps as they are; bf16 COPIES of the fp32 master weights, made once per step for the whole trunk by one
    launch (`refresh_weights` -> scnattn_bf16_weights): the plain copy [Cout][taps][Cin] for the forward products and a
    transposed copy [Cin][taps][Cout] for the d-input products, so both operands of every forward / d-input product are
    k-contiguous (csrc/cgemm16.hip);
  * fp32 where it matters: accumulation, BatchNorm statistics (taken from the fp32 accumulators in the GEMM epilogue,
    finalized on load by the BatchNorm kernels), BatchNorm parameters and their gradients, the weight gradients (written
    straight into the fp32 flat gradient buffer by csrc/wgrad16.hip) and the optimizer state;
  * a2 = relu(bn2(z2)) is materialised here (in fp32 it is recomputed on load by conv3's prologue): at 2 bytes per element
    the extra map costs less than a prologue pass in a kernel whose matrix work is 8x shorter.

`Bottleneck.forward` (scnattn/resnet.py) calls `bottleneck()` when its input is a bf16 CUDA map in training mode; the
trunk produces one when it runs under `torch.autocast("cuda", dtype=torch.bfloat16)` (scnattn/stem.py)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import conv as _conv
from ._lib import ConvExtra

BF = torch.bfloat16


class _WeightDesc(C.Structure):      # scnattn_weight_desc
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_t", C.c_void_p), ("cout", C.c_int), ("taps", C.c_int),
                ("cin", C.c_int), ("pad", C.c_int)]


class _Weights:
    """bf16 operand copies of every 1x1 / 3x3 convolution weight of a trunk: two flat bf16 buffers (plain, transposed) and
    the device-side descriptor table of the conversion kernel."""

    def __init__(self, trunk, dev):
        self.convs = [m for m in trunk.modules()
                      if isinstance(m, torch.nn.Conv2d) and m.kernel_size in ((1, 1), (3, 3)) and m.groups == 1
                      and m.weight.shape[0] % 32 == 0 and m.weight.shape[1] % 32 == 0]
        n = sum(m.weight.numel() for m in self.convs)
        self.buf = torch.empty(n, device=dev, dtype=BF)
        self.buf_t = torch.empty(n, device=dev, dtype=BF)
        self.ptrs = None
        off = 0
        for m in self.convs:
            co, ci, kh, kw = m.weight.shape
            k = m.weight.numel()
            m._w16 = self.buf[off:off + k].view(co, kh * kw * ci)           # [Cout][taps][Cin]
            m._w16t = self.buf_t[off:off + k].view(ci, kh * kw * co)        # [Cin][taps][Cout]
            off += k

    def _table(self, dev):
        descs = (_WeightDesc * len(self.convs))()
        prefix = np.zeros(len(self.convs) + 1, dtype=np.int32)
        for i, m in enumerate(self.convs):
            w = m.weight
            co, ci, kh, kw = w.shape
            taps = kh * kw
            if taps > 1 and not w.is_contiguous(memory_format=torch.channels_last):
                raise RuntimeError("conv16: 3x3 weights must be channels-last ([Cout][3][3][Cin] in memory)")
            descs[i] = _WeightDesc(w.data_ptr(), m._w16.data_ptr(), m._w16t.data_ptr(), co, taps, ci, 0)
            prefix[i + 1] = prefix[i] + taps * (co // 32) * (ci // 32)
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        self.desc = torch.from_numpy(raw).to(dev)
        self.prefix = torch.from_numpy(prefix).to(dev)
        self.total = int(prefix[-1])
        self.ptrs = [m.weight.data_ptr() for m in self.convs]

    def refresh(self, dev):
        # the master weights may have been re-homed (FlatBuffer makes p.data a view of its flat buffer): rebuild the table then
        if self.ptrs is None or any(m.weight.data_ptr() != p for m, p in zip(self.convs, self.ptrs)):
            self._table(dev)
        st = torch._C._cuda_getCurrentRawStream(dev.index)
        _conv._chk(_lib.lib().scnattn_bf16_weights(st, len(self.convs), self.desc.data_ptr(), self.prefix.data_ptr(),
                                                   self.total), "scnattn_bf16_weights")


def refresh_weights(trunk):
    """fp32 master weights -> bf16 operand copies for every convolution of `trunk`: ONE launch; call once per forward pass
    (the optimizer has moved the masters since the last one)."""
    dev = next(trunk.parameters()).device
    w = getattr(trunk, "_scn_w16", None)
    if w is None or w.buf.device != dev:
        w = _Weights(trunk, dev)
        object.__setattr__(trunk, "_scn_w16", w)
    w.refresh(dev)


def usable(mod, x):
    if not (_conv.ENABLED and x.is_cuda and x.dtype == BF and mod.training and hasattr(mod.conv1, "_w16")):
        return False
    for bn in (mod.bn1, mod.bn2, mod.bn3):
        if bn.weight is None or not bn.track_running_stats or bn.momentum is None or bn.weight.dtype != torch.float32:
            return False
    p, cin = mod.conv1.weight.shape[0], mod.conv1.weight.shape[1]
    if p % 64 or cin % 64 or mod.conv2.groups != 1 or mod.conv2.dilation != (1, 1) or mod.conv2.kernel_size != (3, 3) \
            or mod.conv2.padding != (1, 1) or mod.stride not in (1, 2):
        return False
    if mod.stride == 2 and (x.shape[2] % 2 or x.shape[3] % 2):
        return False
    if mod.downsample is not None:
        d0, d1 = mod.downsample[0], mod.downsample[1]
        if d0.kernel_size != (1, 1) or d0.stride != (mod.stride, mod.stride) or d1.weight is None or d1.momentum is None \
                or not hasattr(d0, "_w16"):
            return False
    return True


def _mm(h, st, M, N, K, a, lda, b, ldb, out, ldc, ws, ex=None, beta=0.0):
    _chk = _conv._chk
    _chk(h.scnattn_cgemm16(st, M, N, K, a.data_ptr(), lda, b.data_ptr(), ldb, beta, out.data_ptr(), ldc,
                           1 if out.dtype == BF else 0, ws.data_ptr(), ws.numel(), None if ex is None else C.byref(ex)),
         "scnattn_cgemm16")


class _Bottleneck16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd):
        h, raw_stream = _conv._fns()
        dev = x.device
        st = raw_stream(dev.index)
        ws, part, bnpart = _conv._buffers(dev)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        N, Cin, Hi, Wi = x.shape
        p, C4, s = w1.shape[0], w3.shape[0], mod.stride
        Ho, Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
        Rin, Rout = N * Hi * Wi, N * Ho * Wo
        x2 = _conv._as2d(x)
        bn1, bn2, bn3 = mod.bn1, mod.bn2, mod.bn3
        bf = dict(device=dev, dtype=BF)
        # conv1 (+ bn1 statistics from the fp32 accumulators) -> bn1 apply + relu, finalize inside
        sh1 = _conv._shift(bn1)
        z1 = torch.empty((Rin, p), **bf)
        _mm(h, st, Rin, p, Cin, x2, Cin, mod.conv1._w16, Cin, z1, p, ws,
            ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=sh1.data_ptr()))
        a1, st1 = _conv._apply_fin(h, st, Rin, p, z1, None, part, bn1, sh1, g1, b1, True)
        # conv2 3x3 (strided for layerN.0)
        sh2 = _conv._shift(bn2)
        z2 = torch.empty((Rout, p), **bf)
        ex = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=sh2.data_ptr())
        _conv._chk(h.scnattn_conv3x3_fwd16(st, N, Hi, Wi, p, p, s, a1.data_ptr(), mod.conv2._w16.data_ptr(), z2.data_ptr(),
                                           C.byref(ex), ws.data_ptr(), ws.numel()), "scnattn_conv3x3_fwd16")
        a2, st2 = _conv._apply_fin(h, st, Rout, p, z2, None, part, bn2, sh2, g2, b2, True)
        # conv3
        sh3 = _conv._shift(bn3)
        z3 = torch.empty((Rout, C4), **bf)
        _mm(h, st, Rout, C4, p, a2, p, mod.conv3._w16, p, z3, C4, ws,
            ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=sh3.data_ptr()))
        zd = std = None
        if wd is not None:
            bnd = mod.downsample[1]
            shd = _conv._shift(bnd)
            zd = torch.empty((Rout, C4), **bf)
            _mm(h, st, Rout, C4, Cin, x2, Cin, mod.downsample[0]._w16, Cin, zd, C4, ws,
                ConvExtra(epi=1, stat_partial=bnpart.data_ptr(), stat_shift=shd.data_ptr(), stride=s, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo))
            idn, std = _conv._apply_fin(h, st, Rout, C4, zd, None, bnpart, bnd, shd, gd, bd, False)
        else:
            idn = x2
        out, st3 = _conv._apply_fin(h, st, Rout, C4, z3, idn, part, bn3, sh3, g3, b3, True)
        ctx.mod = mod
        ctx.geom = (N, Cin, Hi, Wi, p, C4, s, Ho, Wo)
        ctx.has_down = wd is not None
        ctx.save_for_backward(x, w1, g1, w2, g2, w3, g3, wd, gd, z1, a1, z2, a2, z3, out, zd, st1, st2, st3, std)
        return _conv._as4d(out, N, Ho, Wo)

    @staticmethod
    def backward(ctx, dout):
        h, raw_stream = _conv._fns()
        (x, w1, g1, w2, g2, w3, g3, wd, gd, z1, a1, z2, a2, z3, out, zd, st1, st2, st3, std) = ctx.saved_tensors
        mod = ctx.mod
        N, Cin, Hi, Wi, p, C4, s, Ho, Wo = ctx.geom
        dev = x.device
        st = raw_stream(dev.index)
        ws, part, bnpart = _conv._buffers(dev)
        Rin, Rout = N * Hi * Wi, N * Ho * Wo
        need = ctx.needs_input_grad      # (mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd)
        if dout.dtype != BF or not dout.is_contiguous(memory_format=torch.channels_last):
            dout = dout.to(BF).contiguous(memory_format=torch.channels_last)
        dout2, x2 = _conv._as2d(dout), _conv._as2d(x)
        bf = dict(device=dev, dtype=BF)
        _chk, _red, _dx = _conv._chk, _conv._bwd_reduce, _conv._bwd_dx_fin
        main = torch.cuda.current_stream(dev)
        side = _conv._side(dev) if _conv.side_ok(w1, w2, w3, wd) else None

        def wstream(*tensors):
            return (side.fork(main, *tensors), side.ws) if side else (st, ws)

        # ---- bn3 (+ identity + relu): g = dout * [out > 0] = d identity; dz3 -------------------------------------------
        dres, nch = _red(h, st, Rout, C4, dout2, out, z3, st3, True, bnpart, True)
        dz3 = torch.empty((Rout, C4), **bf)
        dgb3 = _dx(h, st, Rout, C4, dres, z3, st3, g3, bnpart, (nch + 3) & ~3, nch, dz3)
        dw3 = None
        if need[8]:
            dw3 = _conv._grad_out(w3)
            sw, wsw = wstream(dz3, a2, dw3)
            _chk(h.scnattn_wgrad16_rows(sw, Rout, p, C4, dz3.data_ptr(), a2.data_ptr(), Rout, dw3.data_ptr(), p, 0, 0, 0, 0, 0,
                                        0, 0, wsw.data_ptr(), wsw.numel(), 0), "scnattn_wgrad16_rows")
        # ---- conv3 d input, bn2 ---------------------------------------------------------------------------------------------
        # the d-input product writes g2 = d a2 * [a2 > 0] and the two sums of bn2's backward itself (mask epilogue)
        def mask_ex(z, stats, bn_mod, Cn):
            return ConvExtra(epi=2, stat_partial=bnpart.data_ptr(), ez=z.data_ptr(), ldz=Cn, emean=stats[0].data_ptr(),
                             einvstd=stats[1].data_ptr(), egamma=bn_mod.weight.data_ptr(), ebeta=bn_mod.bias.data_ptr())
        dz2 = torch.empty((Rout, p), **bf)
        _mm(h, st, Rout, p, C4, dz3, C4, mod.conv3._w16t, C4, dz2, p, ws, mask_ex(z2, st2, mod.bn2, p))
        dgb2 = _dx(h, st, Rout, p, dz2, z2, st2, g2, bnpart, h.scnattn_cgemm_stat_ld(Rout), h.scnattn_cgemm_row_tiles(Rout), dz2)
        # ---- conv2: weight gradient (side stream), d input, bn1 ------------------------------------------------------------
        dw2 = None
        if need[5]:
            dw2 = _conv._grad_out(w2)
            sw, wsw = wstream(dz2, a1, dw2)
            if s == 1:
                _chk(h.scnattn_wgrad16_3x3(sw, N, Hi, Wi, p, p, dz2.data_ptr(), a1.data_ptr(), dw2.data_ptr(), wsw.data_ptr(),
                                           wsw.numel(), 0), "scnattn_wgrad16_3x3")
            else:       # a stride-2 3x3 tap by tap: source pixel (2 ho + dh - 1, 2 wo + dw - 1) gathered per row
                for tap in range(9):
                    _chk(h.scnattn_wgrad16_rows(sw, Rout, p, p, dz2.data_ptr(), a1.data_ptr(), Rin,
                                                dw2.data_ptr() + 4 * tap * p, 9 * p, s, Hi, Wi, Ho, Wo, tap // 3 - 1, tap % 3 - 1,
                                                wsw.data_ptr(), wsw.numel(), 0), "scnattn_wgrad16_rows")
        dz1 = torch.empty((Rin, p), **bf)
        if s == 1:      # mask epilogue with bn1
            ex1 = mask_ex(z1, st1, mod.bn1, p)
            _chk(h.scnattn_conv3x3_dgrad16(st, N, Hi, Wi, p, p, 1, dz2.data_ptr(), mod.conv2._w16t.data_ptr(), dz1.data_ptr(),
                                           C.byref(ex1), ws.data_ptr(), ws.numel()), "scnattn_conv3x3_dgrad16")
            dgb1 = _dx(h, st, Rin, p, dz1, z1, st1, g1, bnpart, h.scnattn_cgemm_stat_ld(Rin), h.scnattn_cgemm_row_tiles(Rin), dz1)
        else:           # parity classes write scattered rows: the reduce pass stays
            _chk(h.scnattn_conv3x3_dgrad16(st, N, Hi, Wi, p, p, s, dz2.data_ptr(), mod.conv2._w16t.data_ptr(), dz1.data_ptr(),
                                           None, ws.data_ptr(), ws.numel()), "scnattn_conv3x3_dgrad16")
            dz1, nch = _red(h, st, Rin, p, dz1, a1, z1, st1, True, bnpart, True)
            dgb1 = _dx(h, st, Rin, p, dz1, z1, st1, g1, bnpart, (nch + 3) & ~3, nch, dz1)
        dw1 = None
        if need[2]:
            dw1 = _conv._grad_out(w1)
            sw, wsw = wstream(dz1, x, dw1)
            _chk(h.scnattn_wgrad16_rows(sw, Rin, Cin, p, dz1.data_ptr(), x2.data_ptr(), Rin, dw1.data_ptr(), Cin, 0, 0, 0, 0, 0,
                                        0, 0, wsw.data_ptr(), wsw.numel(), 0), "scnattn_wgrad16_rows")
        # ---- identity branch and d x -----------------------------------------------------------------------------------------
        need_dx = need[1]
        dwd = dgbd = dx = None
        if ctx.has_down:
            _, nch = _red(h, st, Rout, C4, dres, None, zd, std, False, bnpart, False)
            dzd = torch.empty((Rout, C4), **bf)
            dgbd = _dx(h, st, Rout, C4, dres, zd, std, gd, bnpart, (nch + 3) & ~3, nch, dzd)
            if need[11]:
                dwd = _conv._grad_out(wd)
                sw, wsw = wstream(dzd, x, dwd)
                _chk(h.scnattn_wgrad16_rows(sw, Rout, Cin, C4, dzd.data_ptr(), x2.data_ptr(), Rin, dwd.data_ptr(), Cin, s, Hi, Wi,
                                            Ho, Wo, 0, 0, wsw.data_ptr(), wsw.numel(), 0), "scnattn_wgrad16_rows")
            if need_dx:
                dx = torch.empty((Rin, Cin), **bf)
                _mm(h, st, Rin, Cin, p, dz1, p, mod.conv1._w16t, p, dx, Cin, ws)
                dxd = torch.empty((Rout, Cin), **bf)
                _mm(h, st, Rout, Cin, C4, dzd, C4, mod.downsample[0]._w16t, C4, dxd, Cin, ws)
                dx.view(N, Hi, Wi, Cin)[:, ::s, ::s].add_(dxd.view(N, Ho, Wo, Cin))
        elif need_dx:
            dx = dres       # d x = d identity + dz1 . W1, accumulated in place (beta = 1)
            _mm(h, st, Rin, Cin, p, dz1, p, mod.conv1._w16t, p, dx, Cin, ws, beta=1.0)
        if side:
            side.mark()
        dx4 = _conv._as4d(dx, N, Hi, Wi) if dx is not None else None
        return (None, dx4, dw1, dgb1[1] if need[3] else None, dgb1[0] if need[4] else None,
                dw2 if need[5] else None, dgb2[1] if need[6] else None, dgb2[0] if need[7] else None,
                dw3, dgb3[1] if need[9] else None, dgb3[0] if need[10] else None,
                dwd, (dgbd[1] if need[12] else None) if dgbd is not None else None,
                (dgbd[0] if need[13] else None) if dgbd is not None else None)


class _Block16(C.Structure):      # scnattn_block16 (include/scnattn.h)
    _fields_ = [("N", C.c_int), ("Cin", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("p", C.c_int), ("stride", C.c_int),
                ("has_down", C.c_int), ("pad", C.c_int),
                ("gamma", C.c_void_p * 4), ("beta", C.c_void_p * 4), ("run_mean", C.c_void_p * 4), ("run_var", C.c_void_p * 4),
                ("shift", C.c_void_p * 4), ("eps", C.c_float * 4), ("momentum", C.c_float * 4),
                ("w", C.c_void_p * 4), ("wt", C.c_void_p * 4),
                ("ws", C.c_void_p), ("ws_floats", C.c_long), ("part", C.c_void_p), ("bnpart", C.c_void_p), ("bnpart_floats", C.c_long),
                ("x", C.c_void_p), ("save", C.c_void_p), ("stats", C.c_void_p),
                ("dout", C.c_void_p), ("tmp", C.c_void_p), ("dgb", C.c_void_p), ("dw", C.c_void_p * 4), ("need_dx", C.c_int),
                ("pad2", C.c_int), ("side_stream", C.c_void_p), ("side_ws", C.c_void_p), ("side_ws_floats", C.c_long)]


class _Plan:
    """Per (module, input geometry): the filled scnattn_block16 and the sizes of the three buffers the calls need."""

    def __init__(self, mod, x, h):
        N, Cin, Hi, Wi = x.shape
        b = _Block16()
        b.N, b.Cin, b.Hi, b.Wi, b.p, b.stride = N, Cin, Hi, Wi, mod.conv1.weight.shape[0], mod.stride
        b.has_down = 1 if mod.downsample is not None else 0
        sz = [C.c_long(0) for _ in range(5)]
        _conv._chk(h.scnattn_block16_sizes(C.byref(b), *[C.byref(v) for v in sz]), "scnattn_block16_sizes")
        self.b = b
        self.save_elems, self.out_off, self.stats_floats, self.tmp_elems, self.dgb_floats = (v.value for v in sz)
        self.key = (N, Cin, Hi, Wi, x.device)
        self.bns = (mod.bn1, mod.bn2, mod.bn3) + ((mod.downsample[1],) if mod.downsample is not None else ())
        self.convs = (mod.conv1, mod.conv2, mod.conv3) + ((mod.downsample[0],) if mod.downsample is not None else ())
        self.C4 = 4 * b.p
        self.Ho, self.Wo = (Hi - 1) // mod.stride + 1, (Wi - 1) // mod.stride + 1


def _plan(mod, x, h):
    pl = getattr(mod, "_scn_plan16", None)
    if pl is None or pl.key != (x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.device):
        pl = _Plan(mod, x, h)
        object.__setattr__(mod, "_scn_plan16", pl)
    return pl


class _Bottleneck16DriverFn(torch.autograd.Function):
    """The same block as _Bottleneck16Fn with ONE library call per direction (csrc/block16.cpp): identical kernels on
    identical operands in the identical order (bit-identical by test); what changes is the host's cost, ~105 -> ~35 us
    per block forward and ~300 -> ~70 us backward, which is what bounds the bf16 step."""

    @staticmethod
    def forward(ctx, mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd):
        h, raw_stream = _conv._fns()
        dev = x.device
        st = raw_stream(dev.index)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        pl = _plan(mod, x, h)
        b = pl.b
        ws, part, bnpart = _conv._buffers(dev)
        save = torch.empty(pl.save_elems, device=dev, dtype=BF)
        stats = torch.empty(pl.stats_floats, device=dev, dtype=torch.float32)
        for i, bn in enumerate(pl.bns):
            b.gamma[i], b.beta[i] = bn.weight.data_ptr(), bn.bias.data_ptr()
            b.run_mean[i], b.run_var[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            b.shift[i] = _conv._shift(bn).data_ptr()
            b.eps[i], b.momentum[i] = bn.eps, bn.momentum
        for i, cv in enumerate(pl.convs):
            b.w[i], b.wt[i] = cv._w16.data_ptr(), cv._w16t.data_ptr()
        b.ws, b.ws_floats, b.part, b.bnpart, b.bnpart_floats = ws.data_ptr(), ws.numel(), part.data_ptr(), bnpart.data_ptr(), bnpart.numel()
        b.x, b.save, b.stats = x.data_ptr(), save.data_ptr(), stats.data_ptr()
        _conv._chk(h.scnattn_block16_fwd(st, C.byref(b)), "scnattn_block16_fwd")
        C4 = pl.C4
        for i, bn in enumerate(pl.bns):       # next step's conditioning shift: this step's batch mean
            bn._scn_shift = stats[2 * i * C4:2 * i * C4 + bn.num_features]
        Rout = x.shape[0] * pl.Ho * pl.Wo
        out = save[pl.out_off:pl.out_off + Rout * C4].view(Rout, C4)
        ctx.mod, ctx.pl = mod, pl
        ctx.save_for_backward(x, save, stats, w1, w2, w3, wd)
        return _conv._as4d(out, x.shape[0], pl.Ho, pl.Wo)

    @staticmethod
    def backward(ctx, dout):
        h, raw_stream = _conv._fns()
        x, save, stats, w1, w2, w3, wd = ctx.saved_tensors
        mod, pl = ctx.mod, ctx.pl
        b = pl.b
        dev = x.device
        st = raw_stream(dev.index)
        need = ctx.needs_input_grad      # (mod, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd)
        if dout.dtype != BF or not dout.is_contiguous(memory_format=torch.channels_last):
            dout = dout.to(BF).contiguous(memory_format=torch.channels_last)
        ws, part, bnpart = _conv._buffers(dev)
        tmp = torch.empty(pl.tmp_elems, device=dev, dtype=BF)
        dgb = torch.empty(pl.dgb_floats, device=dev, dtype=torch.float32)
        main = torch.cuda.current_stream(dev)
        side = _conv._side(dev) if _conv.side_ok(w1, w2, w3, wd) else None
        dws = [(_conv._grad_out(w) if (w is not None and need[k]) else None) for w, k in ((w1, 2), (w2, 5), (w3, 8), (wd, 11))]
        for i, bn in enumerate(pl.bns):       # the forward call of ANOTHER input may have re-pointed the plan since
            b.gamma[i] = bn.weight.data_ptr()
        for i, cv in enumerate(pl.convs):
            b.w[i], b.wt[i] = cv._w16.data_ptr(), cv._w16t.data_ptr()
        b.ws, b.ws_floats, b.part, b.bnpart, b.bnpart_floats = ws.data_ptr(), ws.numel(), part.data_ptr(), bnpart.data_ptr(), bnpart.numel()
        b.x, b.save, b.stats = x.data_ptr(), save.data_ptr(), stats.data_ptr()
        b.dout, b.tmp, b.dgb = dout.data_ptr(), tmp.data_ptr(), dgb.data_ptr()
        for i in range(4):
            b.dw[i] = dws[i].data_ptr() if dws[i] is not None else None
        b.need_dx = 1 if need[1] else 0
        if side is not None:
            b.side_stream, b.side_ws, b.side_ws_floats = side.stream.cuda_stream, side.ws.data_ptr(), side.ws.numel()
            for t in (x, save, tmp) + tuple(d for d in dws if d is not None):
                t.record_stream(side.stream)
        else:
            b.side_stream, b.side_ws, b.side_ws_floats = None, None, 0
        dxp, dxdp = C.c_void_p(0), C.c_void_p(0)
        _conv._chk(h.scnattn_block16_bwd(st, C.byref(b), C.byref(dxp), C.byref(dxdp)), "scnattn_block16_bwd")
        if side is not None:
            side.mark()
        N, Cin, Hi, Wi = x.shape
        dx4 = None
        if need[1]:
            off = (dxp.value - tmp.data_ptr()) // 2
            dx = tmp[off:off + N * Hi * Wi * Cin].view(N * Hi * Wi, Cin)
            if dxdp.value:
                offd = (dxdp.value - tmp.data_ptr()) // 2
                s = mod.stride
                dxd = tmp[offd:offd + N * pl.Ho * pl.Wo * Cin]
                dx.view(N, Hi, Wi, Cin)[:, ::s, ::s].add_(dxd.view(N, pl.Ho, pl.Wo, Cin))
            dx4 = _conv._as4d(dx, N, Hi, Wi)
        C4 = pl.C4

        def gb(i, which, k, n):          # which: 0 d beta, 1 d gamma
            return dgb[(2 * i + which) * C4:(2 * i + which) * C4 + n] if need[k] else None
        p = b.p
        has_down = wd is not None
        return (None, dx4, dws[0], gb(0, 1, 3, p), gb(0, 0, 4, p), dws[1], gb(1, 1, 6, p), gb(1, 0, 7, p),
                dws[2], gb(2, 1, 9, C4), gb(2, 0, 10, C4),
                dws[3], gb(3, 1, 12, C4) if has_down else None, gb(3, 0, 13, C4) if has_down else None)


BLOCK16 = os.environ.get("SCNATTN_BLOCK16", "c")     # "c": one library call per block and direction; "py": the per-launch path



def bottleneck(mod, x):
    for bn in (mod.bn1, mod.bn2, mod.bn3) + ((mod.downsample[1],) if mod.downsample is not None else ()):
        if not bn.counter_managed and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
    if mod.downsample is not None:
        wd, gd, bd = mod.downsample[0].weight, mod.downsample[1].weight, mod.downsample[1].bias
    else:
        wd = gd = bd = None
    fn = _Bottleneck16DriverFn if BLOCK16 == "c" else _Bottleneck16Fn
    return fn.apply(mod, x, mod.conv1.weight, mod.bn1.weight, mod.bn1.bias, mod.conv2.weight, mod.bn2.weight,
                    mod.bn2.bias, mod.conv3.weight, mod.bn3.weight, mod.bn3.bias, wd, gd, bd)
