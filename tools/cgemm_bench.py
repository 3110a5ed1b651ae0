"""csrc/cgemm.hip on the ResNet-152 1x1-convolution shapes (B=32, 256x256 input) and the decoder's dense products:
correctness against fp64 torch on every variant (plain / prologue / statistics epilogue / mask epilogue / split-K /
strided gather), then time per launch against the round-1 sgemm kernel and MIOpen (torch conv, channels-last,
cudnn.benchmark + FAST find).  Usage: python tools/cgemm_bench.py [check|time|all]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "2"); os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
import torch
import torch.nn.functional as F
from scnattn import functional as SF
from scnattn._lib import call, ptr, stream_of, ConvExtra, lib

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
WS = torch.empty(48 << 20, device=dev)


_BLK = None


def t_us(fn, it=20):
    """GPU time per call.  The calls are enqueued BEHIND a few milliseconds of other work, so the stream never waits for
    the host (a ctypes call costs 20-45 us of Python here -- more than many of the kernels measured)."""
    global _BLK
    if _BLK is None:
        _BLK = (torch.randn(8192, 8192, device=dev), torch.randn(8192, 8192, device=dev), torch.empty(8192, 8192, device=dev))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(1 + it // 40):
        torch.mm(_BLK[0], _BLK[1], out=_BLK[2])        # ~10 ms of fp32 work each
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


def cgemm(a, b, ta, tb, out, M, N, K, ex=None, beta=0.0, ws=True):
    call("scnattn_cgemm", stream_of(a), int(ta), int(tb), M, N, K, 1.0, ptr(a), a.stride(0), ptr(b), b.stride(0), beta,
         ptr(out), out.stride(0), None, None, 1, 0, 0, 0, ptr(WS) if ws else None, WS.numel() if ws else 0,
         None if ex is None else C.byref(ex))
    return out


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def check():
    g = torch.Generator(device="cpu").manual_seed(0)
    worst = 0.0
    for (R, Cin, Cout) in [(2048, 512, 2048), (8192, 1024, 256), (2048, 64, 256), (300, 48, 72), (4096, 256, 64)]:
        x = torch.randn(R, Cin, generator=g).to(dev); w = (torch.randn(Cout, Cin, generator=g) * 0.1).to(dev)
        dy = torch.randn(R, Cout, generator=g).to(dev)
        sc = (1 + 0.5 * torch.randn(Cin, generator=g)).to(dev); sh = (0.2 * torch.randn(Cin, generator=g)).to(dev)
        ss = torch.stack([sc, sh], dim=1).contiguous()
        xd, wd, dyd = x.double(), w.double(), dy.double()
        for split, mi in ((0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (4, 0)):
            if split > 1 and Cin // split < 16:
                continue
            SF.set_option("cgemm_mi", mi)
            # fwd plain
            ex = ConvExtra(force_split=split)
            y = cgemm(x, w, False, True, torch.empty(R, Cout, device=dev), R, Cout, Cin, ex)
            e = rel(y, xd @ wd.t()); worst = max(worst, e); assert e < 3e-6, ("fwd", R, Cin, Cout, split, e)
            # fwd prologue + stats epilogue
            mt = lib().scnattn_cgemm_row_tiles(R)
            part = torch.full((2, Cout, lib().scnattn_cgemm_stat_ld(R)), float("nan"), device=dev)
            sft = (0.1 * torch.randn(Cout, generator=g)).to(dev)
            ex = ConvExtra(pro=1, epi=1, pro_ss=ss.data_ptr(), stat_partial=part.data_ptr(),
                           stat_shift=sft.data_ptr(), force_split=split)
            y = cgemm(x, w, False, True, torch.empty(R, Cout, device=dev), R, Cout, Cin, ex)
            a_ref = torch.relu(xd * sc.double() + sh.double())
            y_ref = a_ref @ wd.t()
            e = rel(y, y_ref); worst = max(worst, e); assert e < 3e-6, ("fwd pro", R, Cin, Cout, split, e)
            d = y_ref - sft.double()
            e1 = rel(part[0, :, :mt].double().sum(1), d.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (d * d).sum(0))
            assert e1 < 2e-5 and e2 < 2e-5, ("stats", R, Cin, Cout, split, e1, e2)
            # dgrad plain + beta
            dx0 = torch.randn(R, Cin, generator=g).to(dev)
            dx = cgemm(dy, w, False, False, dx0.clone(), R, Cin, Cout, ConvExtra(force_split=split), beta=1.0)
            e = rel(dx, dyd @ wd + dx0.double()); worst = max(worst, e); assert e < 3e-6, ("dgrad", R, Cin, Cout, split, e)
            # dgrad mask epilogue
            if Cin % 4 == 0:
                z = torch.randn(R, Cin, generator=g).to(dev)
                mu = (0.1 * torch.randn(Cin, generator=g)).to(dev); isd = (1 + 0.2 * torch.rand(Cin, generator=g)).to(dev)
                ga = (1 + 0.3 * torch.randn(Cin, generator=g)).to(dev); be = (0.2 * torch.randn(Cin, generator=g)).to(dev)
                part = torch.full((2, Cin, lib().scnattn_cgemm_stat_ld(R)), float("nan"), device=dev)
                ex = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z.data_ptr(), emean=mu.data_ptr(), einvstd=isd.data_ptr(),
                               egamma=ga.data_ptr(), ebeta=be.data_ptr(), ldz=Cin, force_split=split)
                gk = cgemm(dy, w, False, False, torch.empty(R, Cin, device=dev), R, Cin, Cout, ex)
                xh = (z - mu) * isd
                mask = (torch.addcmul(be, xh, ga) > 0)
                g_ref = (dyd @ wd) * mask.double()
                e = rel(gk, g_ref); assert e < 3e-6, ("dgrad mask", R, Cin, Cout, split, e)
                e1 = rel(part[0, :, :mt].double().sum(1), g_ref.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (g_ref * xh.double()).sum(0))
                assert e1 < 2e-5 and e2 < 2e-5, ("mask stats", R, Cin, Cout, split, e1, e2)
            # wgrad plain and with the B prologue
            if Cout % 4 == 0 and Cin % 4 == 0 and (split <= 1 or R // split >= 16):
                dw = cgemm(dy, x, True, False, torch.empty(Cout, Cin, device=dev), Cout, Cin, R, ConvExtra(force_split=split))
                e = rel(dw, dyd.t() @ xd); worst = max(worst, e); assert e < 1e-5, ("wgrad", R, Cin, Cout, split, e)
                ex = ConvExtra(pro=2, pro_ss=ss.data_ptr(), force_split=split)
                dw = cgemm(dy, x, True, False, torch.empty(Cout, Cin, device=dev), Cout, Cin, R, ex)
                e = rel(dw, dyd.t() @ a_ref); worst = max(worst, e); assert e < 1e-5, ("wgrad pro", R, Cin, Cout, split, e)
    SF.set_option("cgemm_mi", 0)
    # strided gather (downsample.0: 1x1, stride 2)
    Bn, Hi, Cin, Cout = 3, 8, 64, 128
    xm = torch.randn(Bn, Hi, Hi, Cin, generator=g).to(dev); w = (0.1 * torch.randn(Cout, Cin, generator=g)).to(dev)
    xs = xm[:, ::2, ::2].reshape(-1, Cin).contiguous()
    R = xs.shape[0]
    ex = ConvExtra(stride=2, Hi=Hi, Wi=Hi, Ho=Hi // 2, Wo=Hi // 2)
    y = cgemm(xm.view(-1, Cin), w, False, True, torch.empty(R, Cout, device=dev), R, Cout, Cin, ex)
    e = rel(y, xs.double() @ w.double().t()); assert e < 3e-6, ("gather fwd", e)
    dy = torch.randn(R, Cout, generator=g).to(dev)
    dw = cgemm(dy, xm.view(-1, Cin), True, False, torch.empty(Cout, Cin, device=dev), Cout, Cin, R, ex)
    e = rel(dw, dy.double().t() @ xs.double()); assert e < 1e-5, ("gather wgrad", e)
    # the 4 x 1 wave layout (128 x 64 tiles) forced on wider outputs: forward with both fusions, dgrad, split-K
    for (R, Cin, Cout) in [(1000, 128, 256), (4096, 256, 192)]:
        x = torch.randn(R, Cin, generator=g).to(dev); w = (torch.randn(Cout, Cin, generator=g) * 0.1).to(dev)
        dy = torch.randn(R, Cout, generator=g).to(dev)
        sc = (1 + 0.5 * torch.randn(Cin, generator=g)).to(dev); sh = (0.2 * torch.randn(Cin, generator=g)).to(dev)
        ss = torch.stack([sc, sh], dim=1).contiguous()
        for split in (0, 2):
            part = torch.full((2, Cout, lib().scnattn_cgemm_stat_ld(R)), float("nan"), device=dev); mt = lib().scnattn_cgemm_row_tiles(R)
            ex = ConvExtra(pro=1, epi=1, pro_ss=ss.data_ptr(), stat_partial=part.data_ptr(), force_mi=4, force_split=split)
            y = cgemm(x, w, False, True, torch.empty(R, Cout, device=dev), R, Cout, Cin, ex)
            y_ref = torch.relu(x.double() * sc.double() + sh.double()) @ w.double().t()
            e = rel(y, y_ref); assert e < 3e-6, ("w41 fwd", R, Cin, Cout, split, e)
            e1 = rel(part[0, :, :mt].double().sum(1), y_ref.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (y_ref * y_ref).sum(0))
            assert e1 < 2e-5 and e2 < 2e-5, ("w41 stats", R, Cin, Cout, split, e1, e2)
            dx = cgemm(dy, w, False, False, torch.empty(R, Cin, device=dev), R, Cin, Cout, ConvExtra(force_mi=4, force_split=split))
            e = rel(dx, dy.double() @ w.double()); assert e < 3e-6, ("w41 dgrad", R, Cin, Cout, split, e)
    # TT and odd shapes through the generic entry
    a = torch.randn(96, 200, generator=g).to(dev); b = torch.randn(60, 96, generator=g).to(dev)
    o = cgemm(a, b, True, True, torch.empty(200, 60, device=dev), 200, 60, 96)
    e = rel(o, a.double().t() @ b.double().t()); assert e < 3e-6, ("TT", e)
    print("check ok, worst rel err %.2e" % worst, flush=True)


def comb(timing=True):
    """In-launch split-K combine (cgemm_combine 1: write-through slabs, 2: plain slabs + agent release) against the two-launch
    protocol (0): the outputs must be BIT-identical (same slab order), also while a second stream keeps the chip unevenly
    busy and the consumer's caches warm; then the time per launch of the trunk's split products under each protocol."""
    g = torch.Generator(device="cpu").manual_seed(5)
    side = torch.cuda.Stream()
    junk_a = torch.randn(3000, 1536, device=dev); junk_b = torch.randn(1536, 1000, device=dev); junk_o = torch.empty(3000, 1000, device=dev)
    nbad = 0
    for (R, Cin, Cout) in [(8192, 1024, 256), (2048, 2048, 512), (8192, 256, 1024), (1000, 512, 132), (32768, 512, 128)]:
        x = torch.randn(R, Cin, generator=g).to(dev); w = (torch.randn(Cout, Cin, generator=g) * 0.1).to(dev)
        dy = torch.randn(R, Cout, generator=g).to(dev); z = torch.randn(R, Cin, generator=g).to(dev)
        mu = (0.1 * torch.randn(Cin, generator=g)).to(dev); isd = (1 + 0.2 * torch.rand(Cin, generator=g)).to(dev)
        ga = (1 + 0.3 * torch.randn(Cin, generator=g)).to(dev); be = (0.2 * torch.randn(Cin, generator=g)).to(dev)
        sc = (1 + 0.5 * torch.randn(Cin, generator=g)).to(dev); sh = (0.2 * torch.randn(Cin, generator=g)).to(dev)
        ss = torch.stack([sc, sh], dim=1).contiguous()
        mt = lib().scnattn_cgemm_row_tiles(R)
        for split in (2, 3, 4, 8):
            if Cin // split < 16 or Cout // split < 16 or split * R * max(Cin, Cout) > WS.numel():
                continue
            outs = {}
            for mode in (0, 1, 2):
                SF.set_option("cgemm_combine", mode)
                for rep in range(6 if mode else 1):
                    with torch.cuda.stream(side):            # uneven background load, different every repetition
                        for _ in range(1 + rep):
                            torch.mm(junk_a, junk_b, out=junk_o)
                    pf = torch.full((2, Cout, lib().scnattn_cgemm_stat_ld(R)), float("nan"), device=dev)
                    y = cgemm(x, w, False, True, torch.empty(R, Cout, device=dev), R, Cout, Cin,
                              ConvExtra(pro=1, epi=1, pro_ss=ss.data_ptr(), stat_partial=pf.data_ptr(), force_split=split))
                    pm = torch.full((2, Cin, lib().scnattn_cgemm_stat_ld(R)), float("nan"), device=dev)
                    gk = cgemm(dy, w, False, False, torch.empty(R, Cin, device=dev), R, Cin, Cout,
                               ConvExtra(epi=2, stat_partial=pm.data_ptr(), ez=z.data_ptr(), emean=mu.data_ptr(), einvstd=isd.data_ptr(),
                                         egamma=ga.data_ptr(), ebeta=be.data_ptr(), ldz=Cin, force_split=split))
                    dx0 = torch.randn(R, Cin, generator=g).to(dev)
                    dxb = cgemm(dy, w, False, False, dx0.clone(), R, Cin, Cout, ConvExtra(force_split=split), beta=1.0)
                    dw = cgemm(dy, x, True, False, torch.empty(Cout, Cin, device=dev), Cout, Cin, R, ConvExtra(force_split=split))
                    torch.cuda.synchronize()
                    cur = dict(y=y, gk=gk, dw=dw, sf=pf[:, :, :mt].sum(2), sm=pm[:, :, :mt].sum(2))
                    if mode == 0:
                        outs = cur
                        outs["dxb0"] = dxb - dx0
                        yr = torch.relu(x.double() * sc.double() + sh.double()) @ w.double().t()
                        assert rel(y, yr) < 3e-6
                    else:
                        for k in ("y", "gk", "dw"):
                            if not torch.equal(cur[k], outs[k]):
                                nbad += 1
                                print("MISMATCH", (R, Cin, Cout), "split", split, "mode", mode, "rep", rep, k,
                                      int((cur[k] != outs[k]).sum()), "elements", flush=True)
                        for k in ("sf", "sm"):
                            e = rel(cur[k], outs[k])
                            if not e < 2e-5:
                                nbad += 1
                                print("STATS", (R, Cin, Cout), split, mode, rep, k, e, flush=True)
                        e = rel(dxb - dx0, outs["dxb0"])
                        if not e < 1e-5:
                            nbad += 1
                            print("BETA", (R, Cin, Cout), split, mode, rep, e, flush=True)
    assert nbad == 0, "in-launch combine differs from the two-launch protocol"
    print("combine: bit-identical to the two-launch protocol on every shape / split / repetition", flush=True)
    if not timing:
        SF.set_option("cgemm_combine", 1)
        return
    shapes = [("l1.conv1", 64, 64, 256, 64), ("l1.conv3", 64, 64, 64, 256), ("l2.conv1", 32, 32, 512, 128),
              ("l2.conv3", 32, 32, 128, 512), ("l3.conv1", 16, 16, 1024, 256), ("l3.conv3", 16, 16, 256, 1024),
              ("l4.conv1", 8, 8, 2048, 512), ("l4.conv3", 8, 8, 512, 2048)]
    B = 32
    print("policy split, us per product under cgemm_combine 0 / 1 / 2:  fwd+pro+stats | dgrad+mask | dgrad beta=1 | wgrad", flush=True)
    for name, H, W, Cin, Cout in shapes:
        R = B * H * W
        x2 = torch.randn(R, Cin, device=dev); w2 = 0.1 * torch.randn(Cout, Cin, device=dev); dy2 = torch.randn(R, Cout, device=dev)
        y2 = torch.empty(R, Cout, device=dev); dx2 = torch.empty(R, Cin, device=dev); dw2 = torch.empty(Cout, Cin, device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
        ss = torch.stack([sc, sh], dim=1).contiguous()
        part = torch.empty(2, max(Cin, Cout), lib().scnattn_cgemm_stat_ld(R), device=dev)
        z = torch.randn(R, Cin, device=dev); v = torch.rand(Cin, device=dev) + 0.5
        exf = ConvExtra(pro=1, epi=1, pro_ss=ss.data_ptr(), stat_partial=part.data_ptr())
        exd = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z.data_ptr(), emean=sh.data_ptr(), einvstd=v.data_ptr(),
                        egamma=v.data_ptr(), ebeta=sh.data_ptr(), ldz=Cin)
        row = []
        for mode in (0, 1, 2):
            SF.set_option("cgemm_combine", mode)
            row.append((t_us(lambda: cgemm(x2, w2, False, True, y2, R, Cout, Cin, exf)),
                        t_us(lambda: cgemm(dy2, w2, False, False, dx2, R, Cin, Cout, exd)),
                        t_us(lambda: cgemm(dy2, w2, False, False, dx2, R, Cin, Cout, None, beta=1.0)),
                        t_us(lambda: cgemm(dy2, x2, True, False, dw2, Cout, Cin, R))))
        print("%-9s %6d %5d %5d | %s" % (name, R, Cin, Cout, " | ".join(" ".join("%6.1f" % row[m][k] for m in range(3)) for k in range(4))), flush=True)
    SF.set_option("cgemm_combine", 1)


def timeit():
    shapes = [("l1.conv1", 64, 64, 256, 64), ("l1.conv3", 64, 64, 64, 256), ("l2.conv1", 32, 32, 512, 128),
              ("l2.conv3", 32, 32, 128, 512), ("l3.conv1", 16, 16, 1024, 256), ("l3.conv3", 16, 16, 256, 1024),
              ("l4.conv1", 8, 8, 2048, 512), ("l4.conv3", 8, 8, 512, 2048)]
    B = 32
    print("%-9s %6s %5s %5s | fwd: new  +pro+epi  old  miopen (TF new) | dgrad: new +mask old miopen | wgrad: new +pro old miopen"
          % ("layer", "R", "Cin", "Cout"), flush=True)
    for name, H, W, Cin, Cout in shapes:
        R = B * H * W
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (0.1 * torch.randn(Cout, Cin, 1, 1, device=dev)).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        x2 = x.permute(0, 2, 3, 1).reshape(R, Cin); w2 = w.view(Cout, Cin); dy2 = dy.permute(0, 2, 3, 1).reshape(R, Cout)
        y2 = torch.empty(R, Cout, device=dev); dx2 = torch.empty(R, Cin, device=dev); dw2 = torch.empty(Cout, Cin, device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
        ss = torch.stack([sc, sh], dim=1).contiguous()
        mt = lib().scnattn_cgemm_row_tiles(R)
        part = torch.empty(2, max(Cin, Cout), lib().scnattn_cgemm_stat_ld(R), device=dev)
        z = torch.randn(R, Cin, device=dev); v = torch.rand(Cin, device=dev) + 0.5
        exf = ConvExtra(pro=1, epi=1, pro_ss=ss.data_ptr(), stat_partial=part.data_ptr())
        exd = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z.data_ptr(), emean=sh.data_ptr(), einvstd=v.data_ptr(),
                        egamma=v.data_ptr(), ebeta=sh.data_ptr(), ldz=Cin)
        exw = ConvExtra(pro=2, pro_ss=ss.data_ptr())
        f0 = t_us(lambda: cgemm(x2, w2, False, True, y2, R, Cout, Cin))
        f1 = t_us(lambda: cgemm(x2, w2, False, True, y2, R, Cout, Cin, exf))
        SF.set_option("use_cgemm", 0)
        f2 = t_us(lambda: SF.gemm(x2, w2, tb=True, out=y2))
        g2 = t_us(lambda: SF.gemm(dy2, w2, out=dx2))
        h2 = t_us(lambda: SF.gemm(dy2, x2, ta=True, out=dw2))
        SF.set_option("use_cgemm", 1)
        f3 = t_us(lambda: F.conv2d(x, w))
        g0 = t_us(lambda: cgemm(dy2, w2, False, False, dx2, R, Cin, Cout))
        g1 = t_us(lambda: cgemm(dy2, w2, False, False, dx2, R, Cin, Cout, exd))
        g3 = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
        h0 = t_us(lambda: cgemm(dy2, x2, True, False, dw2, Cout, Cin, R))
        h1 = t_us(lambda: cgemm(dy2, x2, True, False, dw2, Cout, Cin, R, exw))
        h3 = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
        fl = 2.0 * R * Cin * Cout
        print("%-9s %6d %5d %5d | %6.1f %6.1f %6.1f %6.1f (%5.1f) | %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f %7.1f %6.1f"
              % (name, R, Cin, Cout, f0, f1, f2, f3, fl / f0 / 1e6, g0, g1, g2, g3, h0, h1, h2, h3), flush=True)
    print("decoder / square shapes: new vs old (TF)")
    for (M, N, K, ta, tb) in [(4096, 4096, 4096, 0, 1), (4096, 4096, 4096, 0, 0), (4096, 4096, 4096, 1, 0), (1632, 10000, 512, 0, 1),
                              (10000, 512, 1632, 1, 0), (1632, 512, 10000, 0, 0), (2048, 512, 2048, 0, 1), (1632, 2048, 512, 0, 0),
                              (32, 2048, 1000, 0, 0), (32, 512, 2048, 0, 1), (512, 2048, 1632, 1, 0), (2048, 2048, 1632, 1, 0)]:
        a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
        o = torch.empty(M, N, device=dev)
        t0 = t_us(lambda: cgemm(a, b, ta, tb, o, M, N, K))
        SF.set_option("use_cgemm", 0)
        t1 = t_us(lambda: SF.gemm(a, b, ta=bool(ta), tb=bool(tb), out=o))
        SF.set_option("use_cgemm", 1)
        t2 = t_us(lambda: torch.mm(a.t() if ta else a, b.t() if tb else b, out=o))
        fl = 2.0 * M * N * K
        print("M=%5d N=%5d K=%5d ta=%d tb=%d | new %7.1f us %6.1f TF | old %7.1f us %6.1f TF | torch.mm %7.1f us %6.1f TF"
              % (M, N, K, ta, tb, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6), flush=True)


def conv3(x4, w4, stride):
    """x4 (N,Cin,H,W) channels-last, w4 (Cout,Cin,3,3) channels-last -> y [N*Ho*Wo, Cout] via scnattn_conv3x3_fwd"""
    N, Cin, H, W = x4.shape
    Cout = w4.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(N * Ho * Wo, Cout, device=dev)
    call("scnattn_conv3x3_fwd", stream_of(x4), N, H, W, Cin, Cout, stride, ptr(x4), ptr(w4), ptr(y), None, ptr(WS), WS.numel())
    return y


def check3():
    g = torch.Generator(device="cpu").manual_seed(1)
    for (N, H, Cin, Cout, s) in [(2, 8, 128, 128, 1), (3, 10, 128, 256, 2), (2, 7, 256, 128, 1), (2, 16, 128, 128, 2), (1, 5, 512, 512, 1),
                                 (2, 12, 64, 64, 1), (3, 9, 128, 64, 2), (2, 11, 64, 128, 1)]:    # <= 64 wide: the 128 x 64 tile
        x = torch.randn(N, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        w = (0.1 * torch.randn(Cout, Cin, 3, 3, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
        Ho = (H - 1) // s + 1
        ref = F.conv2d(x.double(), w.double(), stride=s, padding=1)
        y = conv3(x, w, s)
        e = rel(y, ref.permute(0, 2, 3, 1).reshape(-1, Cout)); assert e < 3e-6, ("conv3 fwd", N, H, Cin, Cout, s, e)
        dy = torch.randn(N, Cout, Ho, Ho, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        F.conv2d(xd, wd, stride=s, padding=1).backward(dy.double())
        if Cin % 128 == 0:      # the gathered form (k_slices < 0 forces it for stride 1 too)
            dw = torch.empty_like(w)
            call("scnattn_conv3x3_wgrad", stream_of(x), N, H, H, Cin, Cout, s, ptr(dy), ptr(x), ptr(dw), ptr(WS), WS.numel(), -1)
            e = rel(dw, wd.grad); assert e < 1e-5, ("conv3 wgrad gather", N, H, Cin, Cout, s, e)
        if s == 1:
            dx = torch.empty_like(x)
            call("scnattn_conv3x3_dgrad", stream_of(x), N, H, H, Cin, Cout, ptr(dy), ptr(w), ptr(dx), None, ptr(WS), WS.numel())
            e = rel(dx, xd.grad); assert e < 3e-6, ("conv3 dgrad", N, H, Cin, Cout, e)
    # halo-staged weight gradient (csrc/conv3.hip): strips of 16 and of 8 columns, non-square maps, every K split incl.
    # wave ranges that start in the middle of a strip, Cin != Cout
    for (N, H, W, Cin, Cout) in [(2, 8, 8, 128, 128), (3, 16, 16, 64, 96), (2, 32, 32, 32, 64), (2, 6, 24, 64, 32), (5, 7, 16, 96, 32),
                                 (1, 3, 40, 32, 32), (4, 8, 8, 512, 512), (32, 16, 16, 256, 256),
                                 (2, 4, 4, 256, 256), (2, 2, 2, 512, 512), (3, 5, 12, 64, 64), (2, 9, 20, 32, 96)]:      # ragged last strips
        x = torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, Cout, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        w = torch.zeros(Cout, Cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                  [False, True, False])[1]
        for ksl in (0, 1, 2, 3, 7):
            nseg_lines = N * (W // 16 if W % 16 == 0 else (W + 7) // 8) * H
            if ksl > max(1, nseg_lines // 16) or ksl * Cout * 9 * Cin > WS.numel():
                continue
            dw = torch.full_like(w, float("nan"))
            call("scnattn_conv3x3_wgrad", stream_of(x), N, H, W, Cin, Cout, 1, ptr(dy), ptr(x), ptr(dw), ptr(WS), WS.numel(), ksl)
            e = rel(dw, ref); assert e < 1e-5, ("conv3 wgrad halo", N, H, W, Cin, Cout, ksl, e)
    # d input of the stride-2 convolutions: four parity classes in one launch
    for (N, Hi, Wi, Cin, Cout) in [(2, 8, 8, 128, 128), (3, 16, 16, 64, 32), (2, 12, 20, 48, 64), (1, 4, 6, 256, 16), (32, 16, 16, 512, 512)]:
        w = (0.1 * torch.randn(Cout, Cin, 3, 3, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, Cout, Hi // 2, Wi // 2, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        x0 = torch.zeros(N, Cin, Hi, Wi, device=dev, dtype=torch.float64)
        ref = torch.ops.aten.convolution_backward(dy.double(), x0, w.double(), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                  [True, False, False])[0]
        dx = torch.full((N, Cin, Hi, Wi), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
        call("scnattn_conv3x3_dgrad_strided", stream_of(dy), N, Hi, Wi, Cin, Cout, 2, ptr(dy), ptr(w), ptr(dx), ptr(WS), WS.numel())
        e = rel(dx, ref); assert e < 3e-6, ("conv3 dgrad strided", N, Hi, Wi, Cin, Cout, e)
    print("check3 ok", flush=True)


def checkbn():
    """Finalize-on-load BatchNorm kernels (csrc/batchnorm.hip) against fp64: apply (+res, +relu) with statistics from
    channel-major partials, backward reduce (mask from y) + dx with d beta / d gamma summed inside."""
    g = torch.Generator(device="cpu").manual_seed(3)
    for (R, Cn) in [(2048, 512), (8192, 256), (300, 48), (131072, 64), (70, 2048), (32768, 128)]:
        z = (torch.randn(R, Cn, generator=g) * (0.5 + torch.rand(Cn, generator=g)) + torch.randn(Cn, generator=g)).to(dev)
        res = torch.randn(R, Cn, generator=g).to(dev)
        ga = (1 + 0.3 * torch.randn(Cn, generator=g)).to(dev); be = (0.2 * torch.randn(Cn, generator=g)).to(dev)
        shift = (z.mean(0) + 0.1 * torch.randn(Cn, device=dev)).contiguous()
        mt = lib().scnattn_cgemm_row_tiles(R); ld = lib().scnattn_cgemm_stat_ld(R)
        part = torch.full((2, Cn, ld), float("nan"), device=dev)
        d = (z - shift).double()
        nfull = (R // 64) * 64
        p1 = d[:nfull].view(-1, 64, Cn).sum(1); p2 = (d[:nfull] ** 2).view(-1, 64, Cn).sum(1)
        if nfull < R:
            p1 = torch.cat([p1, d[nfull:].sum(0, keepdim=True)]); p2 = torch.cat([p2, (d[nfull:] ** 2).sum(0, keepdim=True)])
        part[0, :, :mt] = p1.t().float(); part[1, :, :mt] = p2.t().float()
        zd = z.double(); mu = zd.mean(0); var = zd.var(0, unbiased=False); eps, mom = 1e-5, 0.1
        for relu in (0, 1):
            for use_res in (False, True):
                y = torch.full((R, Cn), float("nan"), device=dev); st = torch.empty(2, Cn, device=dev)
                rm = torch.zeros(Cn, device=dev); rv = torch.ones(Cn, device=dev); ss = torch.empty(Cn, 2, device=dev)
                call("scnattn_bn_apply_fin", stream_of(z), R, Cn, ptr(z), ptr(res) if use_res else None, 0, ptr(part), ld, mt, ptr(shift),
                     eps, mom, ptr(ga), ptr(be), relu, ptr(y), ptr(st[0]), ptr(st[1]), ptr(rm), ptr(rv), ptr(ss))
                ref = (zd - mu) / torch.sqrt(var + eps) * ga.double() + be.double()
                if use_res: ref = ref + res.double()
                if relu: ref = torch.relu(ref)
                e = rel(y, ref); assert e < 1e-5, ("bn_apply_fin", R, Cn, relu, use_res, e)
                assert rel(st[0], mu) < 1e-5 and rel(st[1], 1 / torch.sqrt(var + eps)) < 1e-5
                assert rel(rm, mom * mu) < 1e-5 and rel(rv, 0.9 + mom * zd.var(0, unbiased=True)) < 1e-5
                sc = ga.double() / torch.sqrt(var + eps)
                assert rel(ss[:, 0], sc) < 1e-5 and (ss[:, 1].double() - (be.double() - mu * sc)).abs().max() < 1e-4
        st2 = torch.empty(2, Cn, device=dev)
        call("scnattn_bn_finalize", stream_of(z), R, Cn, ptr(part), ld, mt, ptr(shift), eps, mom, ptr(st2[0]), ptr(st2[1]), None, None, None, None, None)
        assert torch.equal(st2, st), "finalize and finalize-on-load must give the same bits"
        # backward
        dy = torch.randn(R, Cn, generator=g).to(dev)
        yv = torch.relu(torch.randn(R, Cn, generator=g)).to(dev)
        cap = 260
        bpart = torch.full((2, Cn, cap), float("nan"), device=dev)
        gout = torch.full((R, Cn), float("nan"), device=dev)
        nch = C.c_int(0)
        call("scnattn_bn_bwd_reduce", stream_of(z), R, Cn, ptr(dy), ptr(yv), ptr(z), 0, ptr(st[0]), ptr(st[1]), 1, ptr(bpart), cap, ptr(gout), C.byref(nch))
        gref = dy.double() * (yv > 0).double()
        assert torch.equal(gout.double(), gref)
        xh = (zd - st[0].double()) * st[1].double()
        ldb = (nch.value + 3) & ~3          # the kernel packs the partial with this leading dimension
        bp = bpart.view(-1)[:2 * Cn * ldb].view(2, Cn, ldb)
        e1 = rel(bp[0, :, :nch.value].double().sum(1), gref.sum(0)); e2 = rel(bp[1, :, :nch.value].double().sum(1), (gref * xh).sum(0))
        assert e1 < 2e-5 and e2 < 2e-5, ("bn_bwd_reduce", R, Cn, e1, e2)
        dz = torch.full((R, Cn), float("nan"), device=dev); dgb = torch.empty(2, Cn, device=dev)
        call("scnattn_bn_bwd_dx_fin", stream_of(z), R, Cn, ptr(gout), ptr(z), 0, ptr(st[0]), ptr(st[1]), ptr(ga), ptr(bp), ldb, nch.value, ptr(dgb[0]), ptr(dgb[1]), ptr(dz))
        db, dg = gref.sum(0), (gref * xh).sum(0)
        dzr = ga.double() * st[1].double() * (gref - db / R - xh * dg / R)
        assert rel(dgb[0], db) < 2e-5 and rel(dgb[1], dg) < 2e-5
        e = rel(dz, dzr); assert e < 2e-5, ("bn_bwd_dx_fin", R, Cn, e)
    print("checkbn ok", flush=True)


def check16():
    """bf16 convolution path (csrc/cgemm16.hip, csrc/wgrad16.hip, scnattn_bf16_weights) against fp64 on the bf16-rounded
    operands: fp32 outputs (weight gradients, fp32-output products) 2e-5 -- only the fp32 accumulation order differs --,
    bf16 outputs 4e-3 of the largest element (one rounding to 8 significant bits), statistics 2e-5."""
    from scnattn import conv16 as C16
    BF = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *sh, sc=1.0: (sc * torch.randn(*sh, generator=g)).to(dev).to(BF)
    # ---- plain products: forward / d input of 1x1 convolutions, both row tiles, split-K, statistics, beta, gather -------
    for (R, Cin, Cout) in [(2048, 512, 2048), (8192, 1024, 256), (300, 64, 72), (4096, 256, 64), (640, 2048, 512)]:
        x = rnd(R, Cin); w = rnd(Cout, Cin, sc=0.1); c0 = rnd(R, Cout)
        ref = x.double() @ w.double().t()
        for split, mi, obf in ((0, 0, 1), (1, 1, 1), (1, 2, 0), (2, 0, 1), (4, 2, 0)):
            if split > 1 and Cin // split < 32:
                continue
            mt = lib().scnattn_cgemm_row_tiles(R); ld = lib().scnattn_cgemm_stat_ld(R)
            part = torch.full((2, Cout, ld), float("nan"), device=dev)
            sft = (0.1 * torch.randn(Cout, generator=g)).to(dev)
            ex = ConvExtra(epi=1, stat_partial=part.data_ptr(), stat_shift=sft.data_ptr(), force_split=split, force_mi=mi)
            y = torch.full((R, Cout), float("nan"), device=dev, dtype=BF if obf else torch.float32)
            call("scnattn_cgemm16", stream_of(x), R, Cout, Cin, ptr(x), Cin, ptr(w), Cin, 0.0, ptr(y), Cout, obf, ptr(WS), WS.numel(), C.byref(ex))
            e = rel(y, ref); assert e < (4e-3 if obf else 2e-5), ("cgemm16", R, Cin, Cout, split, mi, obf, e)
            d = ref - sft.double()
            e1 = rel(part[0, :, :mt].double().sum(1), d.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (d * d).sum(0))
            assert e1 < 2e-5 and e2 < 2e-5, ("cgemm16 stats", R, Cin, Cout, split, e1, e2)
            # beta = 1 accumulate (conv1's d input into the residual gradient)
            yb = c0.clone() if obf else c0.float().clone()
            call("scnattn_cgemm16", stream_of(x), R, Cout, Cin, ptr(x), Cin, ptr(w), Cin, 1.0, ptr(yb), Cout, obf, ptr(WS), WS.numel(),
                 C.byref(ConvExtra(force_split=split, force_mi=mi)))
            e = rel(yb, ref + c0.double()); assert e < (4e-3 if obf else 2e-5), ("cgemm16 beta", R, Cin, Cout, split, mi, obf, e)
    # ---- mask epilogue (EPI 2): g = (dy . W) * [fma((z - mean) * invstd, gamma, beta) > 0] rounded to bf16, and the column
    #      sums of g and g * xhat of THOSE rounded values, for both row tiles; 1x1 d input and the stride-1 3x3 d input -----
    for (R, K, Nn, mi) in [(8192, 1024, 256, 0), (2048, 2048, 512, 0), (32768, 512, 128, 0), (1000, 256, 64, 1), (1000, 256, 64, 2)]:
        dy = rnd(R, K); wt = rnd(Nn, K, sc=0.1); z = rnd(R, Nn)
        mu = (0.1 * torch.randn(Nn, generator=g)).to(dev); isd = (1 + 0.2 * torch.rand(Nn, generator=g)).to(dev)
        ga = (1 + 0.3 * torch.randn(Nn, generator=g)).to(dev); be = (0.2 * torch.randn(Nn, generator=g)).to(dev)
        mt = lib().scnattn_cgemm_row_tiles(R); ld = lib().scnattn_cgemm_stat_ld(R)
        part = torch.full((2, Nn, ld), float("nan"), device=dev)
        ex = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z.data_ptr(), ldz=Nn, emean=mu.data_ptr(), einvstd=isd.data_ptr(),
                       egamma=ga.data_ptr(), ebeta=be.data_ptr(), force_mi=mi)
        gk = torch.full((R, Nn), float("nan"), device=dev, dtype=BF)
        call("scnattn_cgemm16", stream_of(dy), R, Nn, K, ptr(dy), K, ptr(wt), K, 0.0, ptr(gk), Nn, 1, ptr(WS), WS.numel(), C.byref(ex))
        xh = (z.float() - mu) * isd
        mask = torch.addcmul(be, xh, ga) > 0
        ref = (dy.double() @ wt.double().t()) * mask.double()
        e = rel(gk, ref); assert e < 4e-3, ("cgemm16 mask", R, K, Nn, mi, e)
        assert bool(((gk == 0) | mask).all()), "a masked element is not zero"
        gq = gk.double()                                  # the sums are those of the stored (rounded) values
        e1 = rel(part[0, :, :mt].double().sum(1), gq.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (gq * xh.double()).sum(0))
        assert e1 < 2e-5 and e2 < 2e-5, ("cgemm16 mask sums", R, K, Nn, mi, e1, e2)
    for (N, H, W, Cc) in [(32, 16, 16, 256), (4, 9, 20, 64), (8, 32, 32, 128)]:
        conv = torch.nn.Conv2d(Cc, Cc, 3, stride=1, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
        C16.refresh_weights(torch.nn.Sequential(conv))
        R = N * H * W
        dy = rnd(N, H, W, Cc); z = rnd(R, Cc)
        mu = (0.1 * torch.randn(Cc, generator=g)).to(dev); isd = (1 + 0.2 * torch.rand(Cc, generator=g)).to(dev)
        ga = (1 + 0.3 * torch.randn(Cc, generator=g)).to(dev); be = (0.2 * torch.randn(Cc, generator=g)).to(dev)
        mt = lib().scnattn_cgemm_row_tiles(R); ld = lib().scnattn_cgemm_stat_ld(R)
        part = torch.full((2, Cc, ld), float("nan"), device=dev)
        ex = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z.data_ptr(), ldz=Cc, emean=mu.data_ptr(), einvstd=isd.data_ptr(),
                       egamma=ga.data_ptr(), ebeta=be.data_ptr())
        gk = torch.full((R, Cc), float("nan"), device=dev, dtype=BF)
        call("scnattn_conv3x3_dgrad16", stream_of(dy), N, H, W, Cc, Cc, 1, ptr(dy), ptr(conv._w16t), ptr(gk), C.byref(ex), ptr(WS), WS.numel())
        wq = conv.weight.detach().to(BF).double()
        dxr = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), wq, stride=1, padding=1).permute(0, 2, 3, 1).reshape(R, Cc)
        xh = (z.float() - mu) * isd
        mask = torch.addcmul(be, xh, ga) > 0
        e = rel(gk, dxr * mask.double()); assert e < 4e-3, ("conv3 dgrad16 mask", N, H, W, Cc, e)
        gq = gk.double()
        e1 = rel(part[0, :, :mt].double().sum(1), gq.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (gq * xh.double()).sum(0))
        assert e1 < 2e-5 and e2 < 2e-5, ("conv3 dgrad16 mask sums", N, H, W, Cc, e1, e2)
    Bn, Hi, Cin, Cout = 3, 8, 64, 128
    xm = rnd(Bn, Hi, Hi, Cin); w = rnd(Cout, Cin, sc=0.1)
    xs = xm[:, ::2, ::2].reshape(-1, Cin).contiguous()
    y = torch.empty(xs.shape[0], Cout, device=dev, dtype=BF)
    ex = ConvExtra(stride=2, Hi=Hi, Wi=Hi, Ho=Hi // 2, Wo=Hi // 2)
    call("scnattn_cgemm16", stream_of(xm), xs.shape[0], Cout, Cin, ptr(xm), Cin, ptr(w), Cin, 0.0, ptr(y), Cout, 1, ptr(WS), WS.numel(), C.byref(ex))
    e = rel(y, xs.double() @ w.double().t()); assert e < 4e-3, ("cgemm16 gather", e)
    # ---- 3x3: forward, d input (stride 1 = forward with flipped taps on the transposed copy; stride 2 = parity classes),
    #      weight gradients (halo-staged stride 1; stride 2 tap by tap), and the weight conversion kernel ------------------
    for (N, H, W, Cin, Cout, s) in [(2, 8, 8, 128, 128, 1), (3, 10, 12, 64, 256, 2), (2, 7, 16, 256, 64, 1), (2, 16, 16, 128, 128, 2),
                                    (1, 5, 40, 512, 512, 1), (32, 16, 16, 256, 256, 1), (4, 32, 32, 128, 128, 2), (2, 3, 4, 64, 64, 1)]:
        conv = torch.nn.Conv2d(Cin, Cout, 3, stride=s, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
        trunk = torch.nn.Sequential(conv)
        C16.refresh_weights(trunk)
        wq = conv.weight.detach().to(BF)
        assert torch.equal(conv._w16.view(Cout, 3, 3, Cin), wq.permute(0, 2, 3, 1)), "bf16 weight copy"
        assert torch.equal(conv._w16t.view(Cin, 3, 3, Cout), wq.permute(1, 2, 3, 0)), "transposed bf16 weight copy"
        x = rnd(N, H, W, Cin)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        xd = x.double().permute(0, 3, 1, 2).requires_grad_(True); wd_ = wq.double().requires_grad_(True)
        ref = F.conv2d(xd, wd_, stride=s, padding=1)
        y = torch.full((N * Ho * Wo, Cout), float("nan"), device=dev, dtype=BF)
        mt = lib().scnattn_cgemm_row_tiles(N * Ho * Wo); ld = lib().scnattn_cgemm_stat_ld(N * Ho * Wo)
        part = torch.full((2, Cout, ld), float("nan"), device=dev)
        ex = ConvExtra(epi=1, stat_partial=part.data_ptr())
        call("scnattn_conv3x3_fwd16", stream_of(x), N, H, W, Cin, Cout, s, ptr(x), ptr(conv._w16), ptr(y), C.byref(ex), ptr(WS), WS.numel())
        r2 = ref.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
        e = rel(y, r2); assert e < 4e-3, ("conv3 fwd16", N, H, W, Cin, Cout, s, e)
        e1 = rel(part[0, :, :mt].double().sum(1), r2.sum(0)); assert e1 < 2e-5 or r2.sum(0).abs().max() < 1e-3, ("conv3 fwd16 stats", e1)
        dy = rnd(N, Ho, Wo, Cout)
        ref.backward(dy.double().permute(0, 3, 1, 2))
        if s == 1 or (H % 2 == 0 and W % 2 == 0):
            dx = torch.full((N * H * W, Cin), float("nan"), device=dev, dtype=BF)
            call("scnattn_conv3x3_dgrad16", stream_of(x), N, H, W, Cin, Cout, s, ptr(dy), ptr(conv._w16t), ptr(dx), None, ptr(WS), WS.numel())
            e = rel(dx, xd.grad.permute(0, 2, 3, 1).reshape(-1, Cin)); assert e < 4e-3, ("conv3 dgrad16", N, H, W, Cin, Cout, s, e)
        dwr = wd_.grad.permute(0, 2, 3, 1)                 # [Cout][3][3][Cin]
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=dev)
        if s == 1:
            for ksl in (0, 1, 3):
                if ksl > max(1, (N * ((W + 15) // 16) * H) // 16) or ksl * dw.numel() > WS.numel():
                    continue
                dw.fill_(float("nan"))
                call("scnattn_wgrad16_3x3", stream_of(x), N, H, W, Cin, Cout, ptr(dy), ptr(x), ptr(dw), ptr(WS), WS.numel(), ksl)
                e = rel(dw, dwr); assert e < 2e-5, ("wgrad16 3x3", N, H, W, Cin, Cout, ksl, e)
        elif Cin % 64 == 0 and Cout % 64 == 0:
            for tap in range(9):
                call("scnattn_wgrad16_rows", stream_of(x), N * Ho * Wo, Cin, Cout, ptr(dy), ptr(x), N * H * W, dw.data_ptr() + 4 * tap * Cin, 9 * Cin,
                     s, H, W, Ho, Wo, tap // 3 - 1, tap % 3 - 1, ptr(WS), WS.numel(), 0)
            e = rel(dw, dwr); assert e < 2e-5, ("wgrad16 strided 3x3 by taps", N, H, W, Cin, Cout, e)
    # ---- 1x1 weight gradients: plain, forced splits, strided downsample ------------------------------------------------------
    for (R, Cin, Cout) in [(2048, 512, 2048), (8192, 1024, 256), (300, 64, 128), (131072, 64, 64), (37, 128, 64)]:
        x = rnd(R, Cin); dy = rnd(R, Cout)
        ref = dy.double().t() @ x.double()
        for ksl in (0, 1, 2, 5):
            if ksl > max(1, ((R + 15) // 16) // 16):
                continue
            dw = torch.full((Cout, Cin), float("nan"), device=dev)
            call("scnattn_wgrad16_rows", stream_of(x), R, Cin, Cout, ptr(dy), ptr(x), R, ptr(dw), Cin, 0, 0, 0, 0, 0, 0, 0, ptr(WS), WS.numel(), ksl)
            e = rel(dw, ref); assert e < 2e-5, ("wgrad16 rows", R, Cin, Cout, ksl, e)
    xm = rnd(3, 8, 12, 64); dy = rnd(3 * 4 * 6, 128)
    dw = torch.full((128, 64), float("nan"), device=dev)
    call("scnattn_wgrad16_rows", stream_of(xm), 72, 64, 128, ptr(dy), ptr(xm), 3 * 8 * 12, ptr(dw), 64, 2, 8, 12, 4, 6, 0, 0, ptr(WS), WS.numel(), 0)
    e = rel(dw, dy.double().t() @ xm[:, ::2, ::2].reshape(-1, 64).double()); assert e < 2e-5, ("wgrad16 strided 1x1", e)
    # ---- BatchNorm finalize-on-load kernels on bf16 maps -----------------------------------------------------------------------
    for (R, Cn) in [(2048, 512), (300, 64), (8192, 256)]:
        z = rnd(R, Cn); res = rnd(R, Cn)
        ga = (1 + 0.3 * torch.randn(Cn, generator=g)).to(dev); be = (0.2 * torch.randn(Cn, generator=g)).to(dev)
        zd = z.double(); shift = zd.mean(0).float().contiguous()
        mt = lib().scnattn_cgemm_row_tiles(R); ld = lib().scnattn_cgemm_stat_ld(R)
        part = torch.zeros((2, Cn, ld), device=dev)
        d = zd - shift.double()
        part[0, :, 0] = d.sum(0).float(); part[1, :, 0] = (d * d).sum(0).float()
        y = torch.empty(R, Cn, device=dev, dtype=BF); stt = torch.empty(2, Cn, device=dev)
        call("scnattn_bn_apply_fin", stream_of(z), R, Cn, ptr(z), ptr(res), 1, ptr(part), ld, mt, ptr(shift), 1e-5, 0.1, ptr(ga), ptr(be), 1, ptr(y),
             ptr(stt[0]), ptr(stt[1]), None, None, None)
        mu, var = zd.mean(0), zd.var(0, unbiased=False)
        ref = torch.relu((zd - mu) / torch.sqrt(var + 1e-5) * ga.double() + be.double() + res.double())
        e = rel(y, ref); assert e < 4e-3, ("bn_apply_fin bf16", R, Cn, e)
        dy = rnd(R, Cn); yv = torch.relu(rnd(R, Cn))
        bpart = torch.full((2 * Cn * 260,), float("nan"), device=dev); gout = torch.empty(R, Cn, device=dev, dtype=BF); nch = C.c_int(0)
        call("scnattn_bn_bwd_reduce", stream_of(z), R, Cn, ptr(dy), ptr(yv), ptr(z), 1, ptr(stt[0]), ptr(stt[1]), 1, ptr(bpart), 260, ptr(gout), C.byref(nch))
        gref = dy.double() * (yv > 0).double()
        assert torch.equal(gout.double(), gref)
        xh = (zd - stt[0].double()) * stt[1].double()
        ldb = (nch.value + 3) & ~3
        dz = torch.empty(R, Cn, device=dev, dtype=BF); dgb = torch.empty(2, Cn, device=dev)
        call("scnattn_bn_bwd_dx_fin", stream_of(z), R, Cn, ptr(gout), ptr(z), 1, ptr(stt[0]), ptr(stt[1]), ptr(ga), ptr(bpart), ldb, nch.value, ptr(dgb[0]), ptr(dgb[1]), ptr(dz))
        db, dg = gref.sum(0), (gref * xh).sum(0)
        assert rel(dgb[0], db) < 2e-5 and rel(dgb[1], dg) < 2e-5
        e = rel(dz, ga.double() * stt[1].double() * (gref - db / R - xh * dg / R)); assert e < 4e-3, ("bn_bwd_dx_fin bf16", R, Cn, e)
    print("check16 ok", flush=True)


def time16():
    """bf16 kernels on the trunk's shapes (B = 32): forward / d input / weight gradient per layer, us."""
    from scnattn import conv16 as C16
    BF = torch.bfloat16
    print("bf16: layer | 1x1 conv1 fwd, conv3 fwd, conv1 dgrad(+beta), wgrad conv1, wgrad conv3 | 3x3 fwd, dgrad, wgrad")
    for name, H, Cin, p, s in [("l1", 64, 256, 64, 1), ("l2", 32, 512, 128, 1), ("l3", 16, 1024, 256, 1), ("l4", 8, 2048, 512, 1),
                               ("l3.0", 32, 512, 256, 2)]:
        N = 32
        Ho = H // s
        Rin, Rout = N * H * H, N * Ho * Ho
        x = torch.randn(Rin, Cin, device=dev).to(BF); a1 = torch.randn(Rin, p, device=dev).to(BF); a2 = torch.randn(Rout, p, device=dev).to(BF)
        w1 = torch.randn(p, Cin, device=dev).to(BF); w1t = torch.randn(Cin, p, device=dev).to(BF); w3 = torch.randn(4 * p, p, device=dev).to(BF)
        w2 = torch.randn(p, 9 * p, device=dev).to(BF)
        z1 = torch.empty(Rin, p, device=dev, dtype=BF); z3 = torch.empty(Rout, 4 * p, device=dev, dtype=BF); z2 = torch.empty(Rout, p, device=dev, dtype=BF)
        dx = torch.empty(Rin, Cin, device=dev, dtype=BF); dz1 = torch.randn(Rin, p, device=dev).to(BF); dz3 = torch.randn(Rout, 4 * p, device=dev).to(BF)
        dz2 = torch.randn(Rout, p, device=dev).to(BF); da1 = torch.empty(Rin, p, device=dev, dtype=BF)
        dw1 = torch.empty(p, Cin, device=dev); dw3 = torch.empty(4 * p, p, device=dev); dw2 = torch.empty(p, 9 * p, device=dev)
        part = torch.empty(2, 4 * p, lib().scnattn_cgemm_stat_ld(Rin), device=dev)
        ex = ConvExtra(epi=1, stat_partial=part.data_ptr())
        mm = lambda M, Nn, K, a, b, o, beta=0.0, e=None: t_us(lambda: call("scnattn_cgemm16", stream_of(a), M, Nn, K, ptr(a), K, ptr(b), K, beta, ptr(o), Nn, 1, ptr(WS), WS.numel(), None if e is None else C.byref(e)))
        t = [mm(Rin, p, Cin, x, w1, z1, 0.0, ex), mm(Rout, 4 * p, p, a2, w3, z3, 0.0, ex), mm(Rin, Cin, p, dz1, w1t, dx, 1.0),
             t_us(lambda: call("scnattn_wgrad16_rows", stream_of(x), Rin, Cin, p, ptr(dz1), ptr(x), Rin, ptr(dw1), Cin, 0, 0, 0, 0, 0, 0, 0, ptr(WS), WS.numel(), 0)),
             t_us(lambda: call("scnattn_wgrad16_rows", stream_of(x), Rout, p, 4 * p, ptr(dz3), ptr(a2), Rout, ptr(dw3), p, 0, 0, 0, 0, 0, 0, 0, ptr(WS), WS.numel(), 0)),
             t_us(lambda: call("scnattn_conv3x3_fwd16", stream_of(x), N, H, H, p, p, s, ptr(a1), ptr(w2), ptr(z2), C.byref(ex), ptr(WS), WS.numel())),
             t_us(lambda: call("scnattn_conv3x3_dgrad16", stream_of(x), N, H, H, p, p, s, ptr(dz2), ptr(w2), ptr(da1), None, ptr(WS), WS.numel())),
             t_us(lambda: call("scnattn_wgrad16_3x3", stream_of(x), N, H, H, p, p, ptr(dz2), ptr(a1), ptr(dw2), ptr(WS), WS.numel(), 0)) if s == 1 else float("nan")]
        print("%-5s | %s" % (name, " ".join("%7.1f" % v for v in t)), flush=True)


def timebn():
    """Finalize-on-load BatchNorm kernels on the trunk's map shapes (B = 32, fp32 and bf16 maps): us and achieved TB/s on the
    algorithmic bytes (maps read + written)."""
    BF = torch.bfloat16
    print("BN kernels: shape | apply_fin(relu) apply_fin(res+relu) bwd_reduce(+g) bwd_dx_fin | TB/s of each")
    for dt in (torch.float32, BF):
        for name, R, Cn in [("l2 p", 32768, 128), ("l2 4p", 32768, 512), ("l3 p", 8192, 256), ("l3 4p", 8192, 1024), ("l4 p", 2048, 512),
                            ("l4 4p", 2048, 2048), ("l1 4p", 131072, 256)]:
            z = torch.randn(R, Cn, device=dev).to(dt); res = torch.randn(R, Cn, device=dev).to(dt); y = torch.empty_like(z)
            ga = torch.rand(Cn, device=dev) + 0.5; be = torch.randn(Cn, device=dev) * 0.1
            mt = lib().scnattn_cgemm_row_tiles(R); ld = lib().scnattn_cgemm_stat_ld(R)
            part = torch.rand(2, Cn, ld, device=dev); shift = torch.zeros(Cn, device=dev); stt = torch.rand(2, Cn, device=dev) + 0.5
            bf = 1 if dt == BF else 0
            es = 2 if bf else 4
            f1 = lambda r: t_us(lambda: call("scnattn_bn_apply_fin", stream_of(z), R, Cn, ptr(z), ptr(res) if r else None, bf, ptr(part), ld, mt, ptr(shift), 1e-5, 0.1,
                                            ptr(ga), ptr(be), 1, ptr(y), ptr(stt[0]), ptr(stt[1]), None, None, None))
            bpart = torch.empty(2 * Cn * 260, device=dev); gout = torch.empty_like(z); nch = C.c_int(0); dgb = torch.empty(2, Cn, device=dev)
            t_r = t_us(lambda: call("scnattn_bn_bwd_reduce", stream_of(z), R, Cn, ptr(res), ptr(y), ptr(z), bf, ptr(stt[0]), ptr(stt[1]), 1, ptr(bpart), 260, ptr(gout), C.byref(nch)))
            ldb = (nch.value + 3) & ~3
            t_d = t_us(lambda: call("scnattn_bn_bwd_dx_fin", stream_of(z), R, Cn, ptr(gout), ptr(z), bf, ptr(stt[0]), ptr(stt[1]), ptr(ga), ptr(bpart), ldb, nch.value,
                                    ptr(dgb[0]), ptr(dgb[1]), ptr(y)))
            ts = [f1(False), f1(True), t_r, t_d]
            maps = [2, 3, 4, 3]
            print("%-5s %-6s (%6d x %4d) | %s | %s" % ("bf16" if bf else "fp32", name, R, Cn, " ".join("%6.1f" % t for t in ts),
                                                        " ".join("%5.2f" % (m * R * Cn * es / t / 1e6) for m, t in zip(maps, ts))), flush=True)


def checkstem():
    g = torch.Generator(device="cpu").manual_seed(2)
    for (N, H, W, cl_x, cl_w) in [(2, 64, 64, False, True), (3, 50, 70, True, True), (1, 33, 17, False, False), (32, 256, 256, True, True)]:
        x = torch.randn(N, 3, H, W, generator=g).to(dev)
        w = (0.1 * torch.randn(64, 3, 7, 7, generator=g)).to(dev)
        if cl_x: x = x.contiguous(memory_format=torch.channels_last)
        if cl_w: w = w.contiguous(memory_format=torch.channels_last)
        Hz, Wz = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        nt = lib().scnattn_stem_tiles(N, H, W)
        z = torch.full((N * Hz * Wz, 64), float("nan"), device=dev)
        mt = nt; part = torch.full((2, 64, (nt + 3) & ~3), float("nan"), device=dev)
        sft = (0.1 * torch.randn(64, generator=g)).to(dev)
        call("scnattn_stem_conv7", stream_of(x), N, H, W, ptr(x), *x.stride(), ptr(w), *w.stride(), ptr(z), ptr(part), ptr(sft))
        ref = F.conv2d(x.double(), w.double(), stride=2, padding=3).permute(0, 2, 3, 1).reshape(-1, 64)
        e = rel(z, ref); assert e < 3e-6, ("stem conv7", N, H, W, e)
        d = ref - sft.double()
        e1 = rel(part[0, :, :mt].double().sum(1), d.sum(0)); e2 = rel(part[1, :, :mt].double().sum(1), (d * d).sum(0))
        assert e1 < 2e-5 and e2 < 2e-5, ("stem stats", N, H, W, e1, e2)
        ss = torch.stack([1 + 0.5 * torch.randn(64, generator=g), 0.3 * torch.randn(64, generator=g)], dim=1).to(dev).contiguous()
        Hp, Wp = (Hz - 1) // 2 + 1, (Wz - 1) // 2 + 1
        out = torch.full((N * Hp * Wp, 64), float("nan"), device=dev)
        call("scnattn_stem_bn_relu_maxpool", stream_of(x), N, Hz, Wz, 64, ptr(z), ptr(ss), ptr(out), 0)
        a = torch.relu(z.view(N, Hz, Wz, 64) * ss[:, 0] + ss[:, 1]).permute(0, 3, 1, 2)
        pref = F.max_pool2d(a, 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, 64)
        assert torch.allclose(out, pref, rtol=1e-6, atol=1e-6), ("stem pool", N, H, W, (out - pref).abs().max().item())
    print("checkstem ok", flush=True)


def time3():
    print("3x3: layer (N,H,Cin,Cout,s) | fwd new / miopen (TF new) | dgrad new / miopen | wgrad halo (policy) / gather / miopen | halo at k_slices 1 2 4 8 16 32")
    for name, H, Cin, s in [("l1", 64, 64, 1), ("l2.0", 64, 128, 2), ("l2", 32, 128, 1), ("l3.0", 32, 256, 2), ("l3", 16, 256, 1),
                            ("l4.0", 16, 512, 2), ("l4", 8, 512, 1)]:
        N, Cout = 32, Cin
        x = torch.randn(N, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = (0.1 * torch.randn(Cout, Cin, 3, 3, device=dev)).contiguous(memory_format=torch.channels_last)
        Ho = (H - 1) // s + 1
        dy = torch.randn(N, Cout, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
        y = torch.empty(N * Ho * Ho, Cout, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
        f0 = t_us(lambda: call("scnattn_conv3x3_fwd", stream_of(x), N, H, H, Cin, Cout, s, ptr(x), ptr(w), ptr(y), None, ptr(WS), WS.numel()))
        f1 = t_us(lambda: F.conv2d(x, w, stride=s, padding=1))
        if s == 1:
            g0 = t_us(lambda: call("scnattn_conv3x3_dgrad", stream_of(x), N, H, H, Cin, Cout, ptr(dy), ptr(w), ptr(dx), None, ptr(WS), WS.numel()))
        else:
            g0 = t_us(lambda: call("scnattn_conv3x3_dgrad_strided", stream_of(x), N, H, H, Cin, Cout, s, ptr(dy), ptr(w), ptr(dx), ptr(WS), WS.numel()))
        g1 = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
        wg = lambda ksl: t_us(lambda: call("scnattn_conv3x3_wgrad", stream_of(x), N, H, H, Cin, Cout, s, ptr(dy), ptr(x), ptr(dw), ptr(WS), WS.numel(), ksl))
        h0 = wg(0) if s == 1 else float("nan")
        h2 = wg(-1) if Cin % 128 == 0 else float("nan")
        h1 = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        sw = []
        if s == 1:
            for ksl in (1, 2, 4, 8, 16, 32):
                lines = N * (H // (16 if H % 16 == 0 else 8)) * H
                sw.append(wg(ksl) if (ksl <= max(1, lines // 16) and ksl * Cout * 9 * Cin <= WS.numel()) else float("nan"))
        fl = 2.0 * N * Ho * Ho * Cout * 9 * Cin
        print("%-5s (%d,%d,%d,%d,%d) | %7.1f / %7.1f (%5.1f TF) | %7.1f / %7.1f | %7.1f / %7.1f / %7.1f | %s"
              % (name, N, H, Cin, Cout, s, f0, f1, fl / f0 / 1e6, g0, g1, h0, h2, h1, " ".join("%6.1f" % t for t in sw)), flush=True)
    # the stem
    N, H = 32, 256
    x = torch.randn(N, 3, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = (0.1 * torch.randn(64, 3, 7, 7, device=dev)).contiguous(memory_format=torch.channels_last)
    z = torch.empty(N * 128 * 128, 64, device=dev); out = torch.empty(N * 64 * 64, 64, device=dev)
    nt = lib().scnattn_stem_tiles(N, H, H)
    part = torch.empty(2, 64, (nt + 3) & ~3, device=dev); ss = torch.rand(64, 2, device=dev)
    c0 = t_us(lambda: call("scnattn_stem_conv7", stream_of(x), N, H, H, ptr(x), *x.stride(), ptr(w), *w.stride(), ptr(z), ptr(part), None))
    c1 = t_us(lambda: F.conv2d(x, w, stride=2, padding=3))
    p0 = t_us(lambda: call("scnattn_stem_bn_relu_maxpool", stream_of(x), N, 128, 128, 64, ptr(z), ptr(ss), ptr(out), 0))
    z4 = z.view(N, 128, 128, 64).permute(0, 3, 1, 2)
    p1 = t_us(lambda: F.max_pool2d(z4, 3, 2, 1))
    print("stem: conv7 + stats %7.1f us (miopen conv %7.1f) (%5.1f TF) | bn+relu+maxpool %7.1f us (aten maxpool alone %7.1f)"
          % (c0, c1, 2.0 * N * 128 * 128 * 64 * 147 / c0 / 1e6, p0, p1), flush=True)


def ab():
    """Per-feature cost on the 1x1 shapes: forward plain / prologue / statistics / both, dgrad plain / mask epilogue,
    wgrad plain / prologue, and the split-K target (workgroups aimed for)."""
    shapes = [("l1.conv1", 131072, 256, 64), ("l1.conv3", 131072, 64, 256), ("l2.conv1", 32768, 512, 128),
              ("l2.conv3", 32768, 128, 512), ("l3.conv1", 8192, 1024, 256), ("l3.conv3", 8192, 256, 1024),
              ("l4.conv1", 2048, 2048, 512), ("l4.conv3", 2048, 512, 2048), ("sq4096", 4096, 4096, 4096)]
    print("%-9s | fwd: plain   pro   epi  both | dgrad: plain  mask | wgrad: plain   pro | mi=1: fwd dgrad, mi=2: fwd dgrad | wgrad mi=1 mi=2" % "layer")
    for name, R, Cin, Cout in shapes:
        x = torch.randn(R, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.1; y = torch.empty(R, Cout, device=dev)
        dy = torch.randn(R, Cout, device=dev); dx = torch.empty(R, Cin, device=dev); dw = torch.empty(Cout, Cin, device=dev)
        ss = torch.rand(Cin, 2, device=dev); z = torch.randn(R, Cin, device=dev); v = torch.rand(Cin, device=dev) + 0.5
        part = torch.empty(2, max(Cin, Cout), lib().scnattn_cgemm_stat_ld(R), device=dev)
        ex_p = ConvExtra(pro=1, pro_ss=ss.data_ptr())
        ex_e = ConvExtra(epi=1, stat_partial=part.data_ptr())
        ex_b = ConvExtra(pro=1, epi=1, pro_ss=ss.data_ptr(), stat_partial=part.data_ptr())
        ex_m = ConvExtra(epi=2, stat_partial=part.data_ptr(), ez=z.data_ptr(), emean=v.data_ptr(), einvstd=v.data_ptr(),
                         egamma=v.data_ptr(), ebeta=v.data_ptr(), ldz=Cin)
        ex_w = ConvExtra(pro=2, pro_ss=ss.data_ptr())
        f = [t_us(lambda: cgemm(x, w, False, True, y, R, Cout, Cin, e)) for e in (None, ex_p, ex_e, ex_b)]
        d = [t_us(lambda: cgemm(dy, w, False, False, dx, R, Cin, Cout, e)) for e in (None, ex_m)]
        g = [t_us(lambda: cgemm(dy, x, True, False, dw, Cout, Cin, R, e)) for e in (None, ex_w)]
        tf, tw = [], []
        for mi in (1, 2):
            SF.set_option("cgemm_mi", mi)
            tf.append(t_us(lambda: cgemm(x, w, False, True, y, R, Cout, Cin)))
            tf.append(t_us(lambda: cgemm(dy, w, False, False, dx, R, Cin, Cout)))
            tw.append(t_us(lambda: cgemm(dy, x, True, False, dw, Cout, Cin, R)))
        SF.set_option("cgemm_mi", 0)
        fmt = lambda l: " ".join("%6.1f" % q for q in l)
        print("%-9s | %s | %s | %s | %s | %s" % (name, fmt(f), fmt(d), fmt(g), fmt(tf), fmt(tw)), flush=True)


def sweep3():
    """3x3 forward (with the statistics epilogue, as the block issues it) and dgrad: row tile x split-K factor."""
    combos = [(2, 4), (2, 2), (1, 2), (1, 1), (4, 1), (4, 2), (4, 3), (4, 4)]
    print("3x3 (mi,S): " + " ".join("%9s" % (c,) for c in combos) + " | policy")
    for name, H, Cin in [("l2", 32, 128), ("l3", 16, 256), ("l4", 8, 512)]:
        N, Cout = 32, Cin
        x = torch.randn(N, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = (0.1 * torch.randn(Cout, Cin, 3, 3, device=dev)).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, Cout, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        y = torch.empty(N * H * H, Cout, device=dev); dx = torch.empty_like(x)
        part = torch.empty(2, Cout, lib().scnattn_cgemm_stat_ld(N * H * H), device=dev)
        for kind in ("fwd+stats", "dgrad"):
            ts = []
            for mi, S in combos + [(0, 0)]:
                ex = ConvExtra(force_mi=mi, force_split=S, epi=1 if kind != "dgrad" else 0, stat_partial=part.data_ptr())
                if kind == "dgrad":
                    fn = lambda: call("scnattn_conv3x3_dgrad", stream_of(x), N, H, H, Cin, Cout, ptr(dy), ptr(w), ptr(dx), C.byref(ex), ptr(WS), WS.numel())
                else:
                    fn = lambda: call("scnattn_conv3x3_fwd", stream_of(x), N, H, H, Cin, Cout, 1, ptr(x), ptr(w), ptr(y), C.byref(ex), ptr(WS), WS.numel())
                ts.append(t_us(fn))
            print("%-3s %-9s " % (name, kind) + " ".join("%9.1f" % t for t in ts[:-1]) + " | %6.1f" % ts[-1], flush=True)


def sweepw():
    """3x3 weight gradient: row tile x split-K target (workgroups aimed for), against MIOpen."""
    combos = [(2, tgt) for tgt in (512, 768, 1024, 1536, 2048, 3072, 4096)]
    print("3x3 wgrad (mi,target): " + " ".join("%10s" % (c,) for c in combos) + " |   miopen")
    for name, H, Cin in [("l2", 32, 128), ("l3", 16, 256), ("l4", 8, 512)]:
        N, Cout = 32, Cin
        x = torch.randn(N, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        w = (0.1 * torch.randn(Cout, Cin, 3, 3, device=dev)).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, Cout, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        dw = torch.empty_like(w)
        ts = []
        for mi, tgt in combos:
            SF.set_option("cgemm_mi", mi); SF.set_option("cgemm_target", tgt)
            ts.append(t_us(lambda: call("scnattn_conv3x3_wgrad", stream_of(x), N, H, H, Cin, Cout, 1, ptr(dy), ptr(x), ptr(dw), ptr(WS), WS.numel(), -1)))
        SF.set_option("cgemm_mi", 0); SF.set_option("cgemm_target", 512)
        m = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        print("%-3s " % name + " " * 19 + " ".join("%10.1f" % t for t in ts) + " | %8.1f" % m, flush=True)
    print("1x1 wgrad (same targets), plain / with the BatchNorm prologue on the activation operand")
    for name, R, Cin, Cout in [("l1.conv3", 131072, 64, 256), ("l2.conv1", 32768, 512, 128), ("l2.conv3", 32768, 128, 512),
                               ("l3.conv1", 8192, 1024, 256), ("l3.conv3", 8192, 256, 1024), ("l4.conv1", 2048, 2048, 512),
                               ("l4.conv3", 2048, 512, 2048)]:
        x = torch.randn(R, Cin, device=dev); dy = torch.randn(R, Cout, device=dev); dw = torch.empty(Cout, Cin, device=dev)
        ss = torch.rand(Cin, 2, device=dev)
        ex_w = ConvExtra(pro=2, pro_ss=ss.data_ptr())
        ts, tp = [], []
        for mi, tgt in combos:
            SF.set_option("cgemm_target", tgt)
            ts.append(t_us(lambda: cgemm(dy, x, True, False, dw, Cout, Cin, R)))
            tp.append(t_us(lambda: cgemm(dy, x, True, False, dw, Cout, Cin, R, ex_w)))
        SF.set_option("cgemm_target", 512)
        print("%-9s " % name + " ".join("%6.1f/%-6.1f" % (a, b) for a, b in zip(ts, tp)), flush=True)


def gemmsweep():
    """The decoder's dense products (BASELINE dims): row tile x split-K factor against the policy's own pick."""
    shapes = [("att1 y", 2048, 512, 2048, False, True), ("att1 dense", 6272, 512, 2048, False, True),
              ("fc", 1632, 10000, 512, False, True), ("dHd", 1632, 512, 10000, False, False),
              ("dWfc", 10000, 512, 1632, True, False), ("dWe", 512, 2048, 6272, True, False),
              ("denc", 6272, 2048, 512, False, False), ("dWa", 2048, 2048, 1632, True, False),
              ("ex", 1632, 2048, 512, False, False), ("dWaM", 512, 2048, 1632, True, False),
              ("dWbeta", 2048, 512, 1632, True, False), ("dWd", 512, 512, 1632, True, False),
              ("demb", 1632, 512, 2048, False, True), ("dWc_g", 512, 512, 1632, True, False)]
    combos = [(mi, S) for mi in (2, 1) for S in (1, 2, 3, 4, 6, 8, 12, 16)]
    print("%-11s %-18s | policy us (TF) | best (mi,S) us (TF) | all: %s" % ("name", "MxNxK", " ".join("%d/%d" % c for c in combos)))
    for name, M, N, K, ta, tb in shapes:
        a = torch.randn(K, M, device=dev) if ta else torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
        out = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        t0 = t_us(lambda: cgemm(a, b, ta, tb, out, M, N, K))
        ts = []
        for mi, S in combos:
            if S > 1 and K // S < 128:
                ts.append(float("nan")); continue
            ex = ConvExtra(force_mi=mi, force_split=S)
            try:
                ts.append(t_us(lambda: cgemm(a, b, ta, tb, out, M, N, K, ex)))
            except Exception:
                ts.append(float("nan"))
        best = min((t, c) for t, c in zip(ts, combos) if t == t)
        print("%-11s %-18s | %6.1f (%5.1f) | %s %6.1f (%5.1f) | %s" % (name, "%dx%dx%d" % (M, N, K), t0, fl / t0 / 1e6, best[1], best[0], fl / best[0] / 1e6,
                                                                      " ".join("%.0f" % t for t in ts)), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "sweepw":
        sweepw()
    if what == "gemmsweep":
        globals()["gemmsweep"]()
    if what == "sweep3":
        sweep3()
    if what in ("check", "all"):
        check()
    if what in ("time", "all"):
        timeit()
    if what == "comb":
        comb()
    if what == "ab":
        ab()
    if what in ("check3", "all3"):
        check3()
    if what in ("checkstem", "all3"):
        checkstem()
    if what in ("checkbn", "all3"):
        checkbn()
    if what == "timebn":
        timebn()
    if what in ("check16", "all16"):
        check16()
    if what in ("time16", "all16"):
        time16()
    if what in ("time3", "all3"):
        time3()
