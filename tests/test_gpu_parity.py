"""GPU parity tests (run on the MI355X box with ``-m gpu``): every HIP kernel / driver of libscnattn
against the CPU oracle (oracle/scnattn_ref.py) and against the committed golden vectors that were
produced by the reference's own modules.  All calls go through the C ABI (ctypes).

Tolerance: north_star asks for outputs within 1e-4 rel-err of the CPU reference in fp32; we use
rel_err = max|a-b| / max|b| <= 1e-4 for outputs and <= 2e-4 for gradients (sums over up to 51 steps
with a different, but fixed, reduction order)."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden, params_from, t, rel_err, rel_l2

pytestmark = pytest.mark.gpu

TOL_OUT = 1e-4
TOL_GRAD = 2e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scnattn import _lib
    _lib.lib()  # must load: there is no fallback
    return torch.device("cuda:0")


def _ok(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, "%s rel_err %.3e > %.1e" % (what, e, tol)
    return e


def _check_grads(named, ref_of, tol, floors=None, report=None):
    """Compare every gradient; softmax shift-invariance makes d/d(full_att.bias) exactly 0 in exact
    arithmetic, so that one is checked absolutely (both sides are rounding noise)."""
    bad = []
    for k, p in named:
        r = ref_of(k)
        if r is None:
            continue
        if k.endswith("full_att.bias"):
            err = (p.grad.detach().double().cpu() - torch.as_tensor(r).double()).abs().max().item()
            lim = 1e-4   # exact value is 0; both sides are accumulated rounding noise
        else:
            err, lim = rel_err(p.grad, r), tol
            if floors is not None:
                lim = max(lim, floors.get(k, 0.0))
        if report is not None:
            report.append("%-40s err %.3e lim %.1e" % (k, err, lim))
        if err > lim:
            bad.append("%s err %.3e > %.1e" % (k, err, lim))
    assert not bad, "; ".join(bad)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (37, 53, 29), (300, 130, 257), (1, 700, 96), (196, 512, 2048)])
def test_sgemm(dev, ta, tb, M, N, K):
    from scnattn import functional as SF
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    tol = 2e-6 * max(1.0, (K / 256.0) ** 0.5)   # fp32 accumulation error grows ~ sqrt(K)
    out = SF.gemm(a.to(dev), b.to(dev), ta=bool(ta), tb=bool(tb))
    _ok(out, ref, tol, "gemm")
    out2 = c0.to(dev).clone()
    SF.gemm(a.to(dev), b.to(dev), ta=bool(ta), tb=bool(tb), bias=bias.to(dev), out=out2, beta=0.5, alpha=2.0)
    _ok(out2, 2.0 * ref + 0.5 * c0.double() + bias.double(), tol, "gemm epilogue")
    mask = (torch.arange(M) % 3 != 0).float()
    out3 = SF.gemm(a.to(dev), b.to(dev), ta=bool(ta), tb=bool(tb), bias=bias.to(dev), rowmask=mask.to(dev))
    _ok(out3, (ref + bias.double()) * mask.double()[:, None], tol, "gemm rowmask")
    assert out3[0].abs().max().item() == 0.0


def test_sgemm_batched_strided(dev):
    from scnattn import functional as SF
    g = torch.Generator().manual_seed(5)
    B, Pn, T, E = 5, 9, 7, 12
    alpha = torch.randn(T, B, Pn, generator=g)
    dawe = torch.randn(T, B, E, generator=g)
    out = torch.zeros(B, Pn, E).to(dev)
    SF.gemm(alpha.to(dev), dawe.to(dev), ta=True, out=out, M=Pn, N=E, K=T, lda=B * Pn, ldb=B * E, ldc=E,
            batch=B, sa=Pn, sb=E, sc=Pn * E)
    ref = torch.einsum("tbp,tbe->bpe", alpha.double(), dawe.double())
    _ok(out, ref, 2e-6)


@pytest.mark.parametrize("rows,N,K,groups,ks", [(32, 128, 512, 1, 0), (32, 2048, 2048, 1, 0), (7, 45, 77, 1, 3),
                                                (32, 512, 1024, 4, 0), (5, 36, 40, 4, 2), (1, 33, 9, 1, 1),
                                                (40, 64, 128, 1, 2), (32, 4608, 512, 1, 4), (32, 512, 4608, 1, 16)])
def test_skinny_gemm(dev, rows, N, K, groups, ks):
    import ctypes as C
    from scnattn._lib import call, ptr, stream_of
    g = torch.Generator().manual_seed(rows + N + K)
    X = torch.randn(rows, groups * K, generator=g)
    W = torch.randn(groups, K, N, generator=g)
    Xd, Wd = X.to(dev), W.to(dev)
    Y = torch.full((16, groups, rows, N), float("nan"), device=dev)
    used = C.c_int(0)
    call("scnattn_skinny_gemm", stream_of(Xd), rows, N, K, groups, ptr(Xd), groups * K, K, ptr(Wd), N, K * N,
         ptr(Y), N, rows * N, groups * rows * N, ks, C.byref(used))
    out = Y[:used.value].sum(0)
    ref = torch.einsum("rgk,gkn->grn", X.view(rows, groups, K).double(), W.double())
    _ok(out, ref, 3e-6, "skinny ks=%d" % used.value)


# ------------------------------------------------------------------------------------------------
def test_attention_module_golden(dev):
    from models.attention import Attention
    d = load_golden("attention")
    E, A = d["p.encoder_att.weight"].shape[1], d["p.encoder_att.weight"].shape[0]
    D = d["p.decoder_att.weight"].shape[1]
    m = Attention(E, D, A).to(dev)
    m.load_state_dict(params_from(d))
    enc, h = t(d["enc"]).to(dev).requires_grad_(True), t(d["h"]).to(dev).requires_grad_(True)
    awe, alpha = m(enc, h)
    _ok(awe, d["awe"], TOL_OUT, "awe"); _ok(alpha, d["alpha"], TOL_OUT, "alpha")
    ((awe * t(d["w_awe"]).to(dev)).sum() + (alpha * t(d["w_alpha"]).to(dev)).sum()).backward()
    _ok(enc.grad, d["denc"], TOL_GRAD, "denc"); _ok(h.grad, d["dh"], TOL_GRAD, "dh")
    _check_grads(m.named_parameters(), lambda k: d["g." + k], TOL_GRAD)


def _report(lines, title):
    """Append a per-tensor error table to gpurun_out/parity_report.txt (kept with the run's output)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_report.txt"), "a") as f:
        f.write("== %s\n" % title)
        for ln in lines:
            f.write(ln + "\n")


def _floors(g32, g64, mult=3.0):
    """Conditioning floor per tensor: how far the reference's own fp32 CPU arithmetic is from fp64.
    ReLU'(x) is discontinuous at 0: a pre-activation within rounding of 0 flips its mask bit between
    two equally valid fp32 evaluations and moves a whole gradient row by O(1/sqrt(#terms)).  A HIP
    gradient is accepted when it is within max(TOL_GRAD, 3 x that floor) of the fp64 oracle."""
    return {k: mult * rel_err(g32[k], g64[k]) for k in g32}


def test_attention_module_full_size(dev):
    from models.attention import Attention
    from oracle import scnattn_ref as R
    torch.manual_seed(3)
    B, Pn, E, D, A = 8, 196, 2048, 512, 512
    m = Attention(E, D, A)
    enc, h = torch.rand(B, Pn, E), torch.randn(B, D) * 0.5
    wa, wl = torch.randn(B, E), torch.randn(B, Pn)
    res = {}
    for dt in (torch.float32, torch.float64):
        P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        e1, h1 = enc.clone().to(dt).requires_grad_(True), h.clone().to(dt).requires_grad_(True)
        awe_r, al_r = R.attention_forward(P, "", e1, h1)
        ((awe_r * wa.to(dt)).sum() + (al_r * wl.to(dt)).sum()).backward()
        g = {k: v.grad for k, v in P.items()}
        g["__enc"], g["__h"] = e1.grad, h1.grad
        res[dt] = (awe_r.detach(), al_r.detach(), g)
    floors = _floors(res[torch.float32][2], res[torch.float64][2])
    awe64, al64, g64 = res[torch.float64]
    m = m.to(dev)
    e2, h2 = enc.to(dev).requires_grad_(True), h.to(dev).requires_grad_(True)
    awe, al = m(e2, h2)
    _ok(awe, awe64, TOL_OUT, "awe"); _ok(al, al64, TOL_OUT, "alpha")
    assert abs(al.sum(1).max().item() - 1.0) < 1e-5
    ((awe * wa.to(dev)).sum() + (al * wl.to(dev)).sum()).backward()
    rep = []
    _ok(e2.grad, g64["__enc"], max(TOL_GRAD, floors["__enc"]), "denc")
    _ok(h2.grad, g64["__h"], max(TOL_GRAD, floors["__h"]), "dh")
    try:
        _check_grads(m.named_parameters(), lambda k: g64[k], TOL_GRAD, floors, rep)
    finally:
        _report(rep, "attention full size")


def test_scn_cell_module_golden(dev):
    from models.scn_cell import SCNCell
    d = load_golden("scn_cell")
    I, F4 = d["p.weight_ia"].shape
    S, H = d["p.weight_ib"].shape[0], d["p.weight_ic"].shape[0]
    m = SCNCell(I, H, S, F4 // 4).to(dev)
    m.load_state_dict(params_from(d))
    assert repr(m) == str(d["repr"])
    u, s = t(d["u"]).to(dev).requires_grad_(True), t(d["s"]).to(dev).requires_grad_(True)
    h0, c0 = t(d["h0"]).to(dev).requires_grad_(True), t(d["c0"]).to(dev).requires_grad_(True)
    h, c = m(u, s, (h0, c0))
    _ok(h, d["h"], TOL_OUT, "h"); _ok(c, d["c"], TOL_OUT, "c")
    ((h * t(d["wh"]).to(dev)).sum() + (c * t(d["wc"]).to(dev)).sum()).backward()
    _ok(u.grad, d["du"], TOL_GRAD, "du"); _ok(s.grad, d["ds"], TOL_GRAD, "ds")
    _ok(h0.grad, d["dh0"], TOL_GRAD, "dh0"); _ok(c0.grad, d["dc0"], TOL_GRAD, "dc0")
    for k, p in m.named_parameters():
        _ok(p.grad, d["g." + k], TOL_GRAD, k)
    h2, c2 = m(u.detach(), s.detach())  # hx=None -> zeros
    _ok(h2, d["h_none"], TOL_OUT); _ok(c2, d["c_none"], TOL_OUT)
    msgs = [str(x) for x in d["errors"]]
    B = u.shape[0]
    with pytest.raises(RuntimeError) as ei:
        m(torch.randn(B, I + 1, device=dev), s.detach())
    assert str(ei.value) == msgs[0]
    with pytest.raises(RuntimeError) as ei:
        m(u.detach(), s.detach(), (torch.randn(B + 1, H, device=dev), torch.randn(B + 1, H, device=dev)))
    assert str(ei.value) == msgs[1]
    with pytest.raises(RuntimeError) as ei:
        m(u.detach(), s.detach(), (torch.randn(B, H + 1, device=dev), torch.randn(B, H + 1, device=dev)))
    assert str(ei.value) == msgs[2]


# ------------------------------------------------------------------------------------------------
def _build_decoder(kind, d, dev):
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from models.decoders.pure_attention import PureAttention
    V, M = d["p.embedding.weight"].shape
    D, E = d["p.init_h.weight"].shape
    if kind == "pure_attention":
        A = d["p.attention.encoder_att.weight"].shape[0]
        m = PureAttention(A, M, D, V, encoder_dim=E, dropout=0.0)
    else:
        S, F4 = d["p.decode_step.weight_ib"].shape
        if kind == "attention_scn":
            A = d["p.attention.encoder_att.weight"].shape[0]
            m = AttentionSCN(A, M, D, F4 // 4, S, V, encoder_dim=E, dropout=0.0)
        else:
            m = PureSCN(M, D, F4 // 4, S, V, encoder_dim=E, dropout=0.0)
    m.load_state_dict(params_from(d))
    return m.to(dev).train()


@pytest.mark.parametrize("name,kind", [
    ("attention_scn_distinct", "attention_scn"), ("attention_scn_tied", "attention_scn"),
    ("attention_scn_full", "attention_scn"), ("attention_scn_odd", "attention_scn"),
    ("pure_scn_distinct", "pure_scn"), ("pure_attention_distinct", "pure_attention")])
def test_decoder_golden(dev, name, kind):
    from oracle import scnattn_ref as R
    d = load_golden(name)
    m = _build_decoder(kind, d, dev)
    enc = t(d["enc"]).to(dev).requires_grad_(True)
    tags, caps, caplens = t(d["tags"]).to(dev), t(d["caps"]).to(dev), t(d["caplens"]).to(dev)
    si = t(d["sort_ind"]).to(dev)
    if kind == "pure_attention":
        preds, caps_s, dl, alphas, sort_ind = m(enc, caps, caplens, sort_ind=si)
    elif kind == "pure_scn":
        preds, caps_s, dl, sort_ind = m(enc, tags, caps, caplens, sort_ind=si)
        alphas = None
    else:
        preds, caps_s, dl, alphas, sort_ind = m(enc, tags, caps, caplens, sort_ind=si)
    _ok(preds, d["preds"], TOL_OUT, "preds")
    assert np.array_equal(caps_s.cpu().numpy(), d["caps_sorted"])
    assert list(dl) == list(d["decode_lengths"])
    if alphas is not None:
        _ok(alphas, d["alphas"], TOL_OUT, "alphas")
    for b, l in enumerate(dl):  # B8: cells past the decode length are exactly zero
        if l < preds.size(1):
            assert preds[b, l:].abs().max().item() == 0.0
            if alphas is not None:
                assert alphas[b, l:].abs().max().item() == 0.0
    loss, sc, tg = R.caption_loss(preds, caps_s, dl, alphas, 1.0)
    _ok(loss, d["loss"], TOL_OUT, "loss")
    loss.backward()
    _check_grads(m.named_parameters(), lambda k: d.get("g_raw." + k), TOL_GRAD)
    _ok(enc.grad, d["denc"], TOL_GRAD, "denc")


def test_decoder_sort_is_done_on_device(dev):
    """Without an injected permutation the module's own stable sort must reproduce the reference's
    permutation for distinct lengths."""
    d = load_golden("attention_scn_distinct")
    m = _build_decoder("attention_scn", d, dev)
    out = m(t(d["enc"]).to(dev), t(d["tags"]).to(dev), t(d["caps"]).to(dev), t(d["caplens"]).to(dev))
    assert np.array_equal(out[4].cpu().numpy(), d["sort_ind"])
    _ok(out[0], d["preds"], TOL_OUT)


@pytest.mark.parametrize("kind,ragged", [("attention_scn", False), ("attention_scn", True), ("pure_scn", True)])
def test_decoder_full_size_vs_oracle(dev, kind, ragged):
    """BASELINE dims (B=32, P=196, E=2048, A=D=F=M=512, S=1000) at a reduced vocabulary / length so that
    the CPU oracle finishes in seconds; forward, loss and every gradient, anchored on the fp64 oracle."""
    from oracle import scnattn_ref as R
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    torch.manual_seed(7)
    B, V, L = 32, 1000, 14
    if kind == "attention_scn":
        m = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5)
    else:
        m = PureSCN(512, 512, 512, 1000, V, dropout=0.5)
    g = torch.Generator().manual_seed(11)
    enc = torch.rand(B, 14, 14, 2048, generator=g)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.randint(5, L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        caps[b, 0] = V - 2
        caps[b, 1:n - 1] = torch.randint(1, V - 3, (n - 2,), generator=g)
        caps[b, n - 1] = V - 1
    caplens = lens.unsqueeze(1)
    T = int(lens.max()) - 1
    mask = (torch.rand(B, T, 512, generator=g) > 0.5).float() * 2.0
    si = torch.sort(lens, descending=True, stable=True)[1]
    res = {}
    for dt in (torch.float32, torch.float64):
        P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        e1 = enc.clone().to(dt).requires_grad_(True)
        if kind == "attention_scn":
            pr, cs, dl, al, _ = R.attention_scn_forward(P, e1, tags.to(dt), caps, caplens, drop_mask=mask.to(dt),
                                                        sort_ind=si, hoist=True)
        else:
            pr, cs, dl, _ = R.pure_scn_forward(P, e1, tags.to(dt), caps, caplens, drop_mask=mask.to(dt), sort_ind=si)
            al = None
        loss_r, _, _ = R.caption_loss(pr, cs, dl, al, 1.0)
        loss_r.backward()
        gr = {k: v.grad for k, v in P.items()}
        gr["__enc"] = e1.grad
        res[dt] = (pr.detach(), None if al is None else al.detach(), loss_r.detach(), gr)
    floors = _floors(res[torch.float32][3], res[torch.float64][3])
    pr64, al64, loss64, g64 = res[torch.float64]
    m = m.to(dev).train()
    m.drop_mask_override = mask.to(dev)
    e2 = enc.to(dev).requires_grad_(True)
    out = m(e2, tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev))
    preds, alphas = out[0], (out[3] if kind == "attention_scn" else None)
    rep = ["preds  err %.3e (cpu fp32 vs fp64: %.3e)" % (rel_err(preds, pr64), rel_err(res[torch.float32][0], pr64))]
    _ok(preds, pr64, TOL_OUT, "preds")
    if alphas is not None:
        rep.append("alphas err %.3e" % rel_err(alphas, al64))
        _ok(alphas, al64, TOL_OUT, "alphas")
    loss, _, _ = R.caption_loss(preds, out[1], out[2], alphas, 1.0)
    _ok(loss, loss64, TOL_OUT, "loss")
    loss.backward()
    try:
        _check_grads(m.named_parameters(), lambda k: g64[k], TOL_GRAD, floors, rep)
        rep.append("%-40s err %.3e lim %.1e" % ("d/d encoder_out", rel_err(e2.grad, g64["__enc"]),
                                                 max(TOL_GRAD, floors["__enc"])))
        _ok(e2.grad, g64["__enc"], max(TOL_GRAD, floors["__enc"]), "denc")
    finally:
        _report(rep, "decoder full size %s ragged=%s" % (kind, ragged))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cl", [False, True])
def test_pool_permute(dev, cl):
    from scnattn import functional as SF
    from oracle import scnattn_ref as R
    torch.manual_seed(0)
    x = torch.randn(3, 70, 8, 8)
    xr = x.clone().requires_grad_(True)
    yr = R.pool_permute(xr, 14)
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    xd = x.to(dev)
    if cl:
        xd = xd.contiguous(memory_format=torch.channels_last)
    xd.requires_grad_(True)
    y = SF.pool_permute(xd, 14)
    _ok(y, yr, 1e-6)
    (y * w.to(dev)).sum().backward()
    _ok(xd.grad, xr.grad, 1e-6)


def test_clamp_adam_matches_torch(dev):
    from scnattn import functional as SF
    torch.manual_seed(1)
    n = 10007
    p0 = torch.randn(n)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=4e-4)
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        g = torch.randn(n) * 4.0
        pr.grad = g.clone().clamp_(-5.0, 5.0)   # utils/optimizer.py clip_gradient
        opt.step()
        SF.clamp_adam_(p, g.to(dev), m, v, 4e-4, step, 5.0)
        _ok(p, pr.detach(), 1e-6, "adam step %d" % step)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("training,relu,with_res", [(True, True, True), (True, True, False), (True, False, False),
                                                     (False, True, True), (False, False, False)])
@pytest.mark.parametrize("shape", [(4, 64, 9, 7), (32, 256, 16, 16), (3, 2048, 8, 8)])
def test_fused_batchnorm_act(dev, training, relu, with_res, shape):
    """csrc/batchnorm.hip vs torch's BatchNorm2d (+ add + relu) in fp64 on CPU: outputs, running
    statistics and every gradient."""
    from scnattn.resnet import FusedBatchNorm2d
    torch.manual_seed(sum(shape))
    N, C, H, W = shape
    x = torch.randn(N, C, H, W) * 1.7 + 0.6
    res = torch.randn(N, C, H, W) if with_res else None
    wgt = torch.randn(N, C, H, W)
    ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.normal_(0, 0.3)
        ref.running_mean.normal_(0, 0.2); ref.running_var.uniform_(0.5, 2.0)
    m = FusedBatchNorm2d(C)
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    m = m.to(dev)
    ref.train(training); m.train(training)
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    yr = ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    (yr * wgt.double()).sum().backward()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = res.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    y = m(xd, residual=rd, relu=relu)
    _ok(y, yr, 1e-5, "y")
    (y * wgt.to(dev)).sum().backward()
    _ok(xd.grad, xr.grad, 5e-5, "dx")
    if with_res:
        _ok(rd.grad, rr.grad, 1e-6, "dres")
    _ok(m.weight.grad, ref.weight.grad, 5e-5, "dgamma"); _ok(m.bias.grad, ref.bias.grad, 5e-5, "dbeta")
    _ok(m.running_mean, ref.running_mean, 1e-5, "running_mean"); _ok(m.running_var, ref.running_var, 1e-5, "running_var")
    assert int(m.num_batches_tracked) == int(ref.num_batches_tracked)


def test_encoder_forward_backward_vs_cpu(dev):
    """Whole EncoderCaption (ResNet-152 trunk on MIOpen + fused BN kernels + fused pool/permute) against
    the same weights run by plain torch ops on CPU.  The reference's own encoder (torchvision) is absent:
    this pins our GPU path to our CPU definition, not to the reference (parity unpinned)."""
    from models.encoders.caption import EncoderCaption
    from oracle import scnattn_ref as R
    torch.manual_seed(0)
    enc = EncoderCaption(channels_last=True)
    enc.fine_tune(True)
    enc.train()
    x = torch.randn(4, 3, 128, 128)
    import copy
    cpu = copy.deepcopy(enc)
    feat = cpu.resnet(x)
    yc = R.pool_permute(feat, 14)
    w = torch.randn_like(yc)
    (yc * w).sum().backward()
    g = enc.to(dev)
    y = g(x.to(dev))
    assert y.shape == (4, 14, 14, 2048) and y.is_contiguous()
    # 152 layers of fp32 convs + batch statistics over few samples amplify summation-order differences to
    # ~1e-3; a structural mistake (wrong residual, stride, statistics) shows up as O(1)
    _ok(y, yc, 5e-3, "encoder_out")
    (y * w.to(dev)).sum().backward()
    # gradients: only the last bottleneck is compared here -- after ~150 more batch-normalised layers the
    # fp32 gradient of the early layers is dominated by ReLU-mask flips (see test_shallow_trunk_gradients
    # for an all-parameter check on a short stack)
    last_block = []
    for (k, p), (_, pc) in zip(g.named_parameters(), cpu.named_parameters()):
        if pc.grad is None:
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.shape == pc.grad.shape, k
            if k.startswith("resnet.7.2."):      # the last bottleneck: backward reaches it first, nothing has compounded yet
                last_block.append((k, rel_l2(p.grad, pc.grad)))
    # value check (round 3): the last block's gradients against the CPU's fp32 ones; the whole-trunk value checks are
    # test_gpu_parity_r3.py::test_well_conditioned_trunk_gradients_vs_fp64 (fixed 1e-3 against fp64) and
    # test_gpu_parity_r2.py::test_encoder_gradients_are_as_close_to_fp64_as_cpu_fp32_is
    assert len(last_block) == 9
    for k, e in last_block:
        # a sign / scale / missing-term error is O(1); two fp32 evaluations of this badly conditioned, randomly initialised
        # trunk differ by ~5e-2 here already (ReLU-mask flips: measured 4.8e-2 on conv1.weight of this block)
        assert e <= 1.5e-1, "%s: rel-l2 %.3e vs CPU fp32" % (k, e)
    for (k, b), (_, bc) in zip(g.named_buffers(), cpu.named_buffers()):
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(b, bc) < 1e-3, k
        elif k.endswith("num_batches_tracked"):
            assert int(b) == int(bc) == 1, k


def test_shallow_trunk_gradients(dev):
    """A 1-1-1-1 bottleneck stack (same block code as ResNet-152) on the GPU: the fused BatchNorm kernels
    against torch's own batch_norm/add/relu ops around the SAME MIOpen convolutions (tight), and against
    CPU fp64 (loose: fp32 convolutions + ReLU-mask flips under batch statistics)."""
    from scnattn.resnet import resnet152_trunk, FusedBatchNorm2d
    import copy
    torch.manual_seed(1)
    cpu = resnet152_trunk(depths=(1, 1, 1, 1)).double().train()
    x = torch.randn(8, 3, 96, 96)
    yc = cpu(x.double())
    w = torch.randn_like(yc)
    (yc * w).sum().backward()
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for fused in (True, False):
        g = copy.deepcopy(cpu).float().to(dev).to(memory_format=torch.channels_last).train()
        g.zero_grad()
        FusedBatchNorm2d.use_fused = fused
        try:
            y = g(xg)
            (y * w.float().to(dev)).sum().backward()
        finally:
            FusedBatchNorm2d.use_fused = True
        res[fused] = (y.detach(), {k: p.grad.clone() for k, p in g.named_parameters()},
                      {k: b.clone() for k, b in g.named_buffers()})
    _ok(res[True][0], res[False][0], 2e-5, "fused vs torch BN: output")
    for k in res[True][1]:
        assert rel_l2(res[True][1][k], res[False][1][k]) < 5e-3, (k, rel_l2(res[True][1][k], res[False][1][k]))
    for k in res[True][2]:
        if not k.endswith("num_batches_tracked"):
            assert rel_err(res[True][2][k], res[False][2][k]) < 1e-5, k
    _ok(res[True][0], yc, 1e-4, "trunk out vs fp64")
    worst_f = max(rel_l2(res[True][1][k], p.grad) for k, p in cpu.named_parameters())
    worst_t = max(rel_l2(res[False][1][k], p.grad) for k, p in cpu.named_parameters())
    assert worst_f < max(1e-2, 3 * worst_t), (worst_f, worst_t)   # no worse than torch's own fp32 path


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,lens", [(1, [4]), (2, [2, 2]), (40, None), (33, None)])
def test_decoder_edge_batches(dev, B, lens):
    """Edge shapes of the sequence drivers: a single row, a single decode step (caption = <start><end>),
    and batches larger than one 32-row MFMA tile (ragged, so the second tile empties first)."""
    from oracle import scnattn_ref as R
    from models.decoders.attention_scn import AttentionSCN
    torch.manual_seed(B)
    V, L = 50, 9
    m = AttentionSCN(24, 20, 28, 36, 14, V, encoder_dim=40, dropout=0.0)
    g = torch.Generator().manual_seed(B + 1)
    enc = torch.rand(B, 3, 3, 40, generator=g)
    tags = torch.rand(B, 14, generator=g)
    ln = torch.tensor(lens) if lens is not None else torch.randint(2, L + 1, (B,), generator=g)
    caps = torch.randint(1, V - 3, (B, L), generator=g)
    caplens = ln.unsqueeze(1)
    si = torch.sort(ln, descending=True, stable=True)[1]
    P = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    e1 = enc.double().requires_grad_(True)
    pr, cs, dl, al, _ = R.attention_scn_forward(P, e1, tags.double(), caps, caplens, sort_ind=si)
    loss_r, _, _ = R.caption_loss(pr, cs, dl, al, 1.0)
    loss_r.backward()
    m = m.to(dev).train()
    e2 = enc.to(dev).requires_grad_(True)
    preds, caps_s, dl2, alphas, _ = m(e2, tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev))
    assert list(dl2) == list(dl)
    _ok(preds, pr, TOL_OUT, "preds"); _ok(alphas, al, TOL_OUT, "alphas")
    loss, _, _ = R.caption_loss(preds, caps_s, dl2, alphas, 1.0)
    loss.backward()
    _check_grads(m.named_parameters(), lambda k: P[k].grad, TOL_GRAD)
    _ok(e2.grad, e1.grad, TOL_GRAD, "denc")


def test_sample_beam_search_runs_and_is_consistent(dev):
    """sample() (attention_scn.py:160-296 with the floor-division fix): beam 1 must equal greedy decoding
    done with the stand-alone modules; beam 3 returns a sequence that starts with <start>."""
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from scnattn import functional as SF
    torch.manual_seed(4)
    V = 30
    word_map = {"<pad>": 0, "<unk>": V - 3, "<start>": V - 2, "<end>": V - 1}
    for i in range(1, V - 3):
        word_map["w%d" % i] = i
    m = AttentionSCN(24, 20, 28, 36, 14, V, encoder_dim=40, dropout=0.5).to(dev).eval()
    enc = torch.rand(1, 4, 4, 40, device=dev)
    tags = torch.rand(1, 14, device=dev)
    with torch.no_grad():
        seq, alphas = m.sample(1, word_map, enc, tags)
        assert seq[0] == V - 2 and len(alphas) == len(seq)
        # greedy reference with the same modules
        e = enc.view(1, -1, 40)
        h, c = m.init_hidden_state(e)
        w = torch.tensor([V - 2], device=dev)
        out = [V - 2]
        for _ in range(len(seq) - 1):
            awe, _ = m.attention(e, h)
            gate = torch.sigmoid(SF.linear(h, m.f_beta.weight, m.f_beta.bias))
            h, c = m.decode_step(torch.cat([m.embedding(w), gate * awe], dim=1), tags, (h, c))
            w = SF.linear(h, m.fc.weight, m.fc.bias).argmax(dim=1)
            out.append(int(w))
        assert out == seq
        seq3, _ = m.sample(3, word_map, enc, tags)
        assert seq3[0] == V - 2
        ps = PureSCN(20, 28, 36, 14, V, encoder_dim=40, dropout=0.5).to(dev).eval()
        s2 = ps.sample(2, word_map, enc, tags)
        assert s2[0] == V - 2


def test_fused_optimizer_step_matches_reference_clip_adam(dev):
    """FusedClampAdam on a real decoder == utils.optimizer.clip_gradient + torch.optim.Adam (the
    reference's trains/attention_scn.py:244-252), two steps."""
    from models.decoders.pure_scn import PureSCN
    from utils.optimizer import FusedClampAdam, clip_gradient
    import copy
    torch.manual_seed(2)
    a = PureSCN(12, 16, 20, 10, 23, encoder_dim=32, dropout=0.0).to(dev)
    b = copy.deepcopy(a)
    oa = FusedClampAdam(a.parameters(), lr=4e-4, grad_clip=5.0)
    ob = torch.optim.Adam(b.parameters(), lr=4e-4)
    for step in range(2):
        gs = [torch.randn_like(p) * 8 for p in b.parameters()]
        oa.zero_grad(); ob.zero_grad()
        for p, q, g_ in zip(a.parameters(), b.parameters(), gs):
            p.grad = g_.clone(); q.grad = g_.clone()
        clip_gradient(ob, 5.0); ob.step()
        oa.step()
        for (k, p), q in zip(a.named_parameters(), b.parameters()):
            _ok(p, q, 1e-5, k)


def test_dp_reducer_on_rccl_single_rank(dev):
    """The data-parallel code path on the real backend (nccl == RCCL) with a 1-rank group and the
    reducers forced on: hooks fire during backward, buckets are gathered + all-reduced on the GPU, and
    the parameters after one fused clamp+Adam step equal the non-distributed run's."""
    import os
    import socket
    import torch.distributed as dist
    from trains.harness import TrainStep, synthetic_batch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        kw = dict(kind="attention_scn", device=dev, encoder=False, batch_size=6, max_len=7, vocab_size=60,
                  emb_dim=32, attention_dim=32, decoder_dim=32, factored_dim=32, semantic_dim=20, dropout=0.0,
                  bucket_mb=0)
        a = TrainStep(**kw)
        b = TrainStep(**kw)
        for r in b.reducers:
            r.enabled = True     # world == 1 would normally bypass the collective
        assert len(b.reducers[0].buckets) > 3
        imgs, tags, caps, caplens = synthetic_batch(6, 60, 7, 8, 20, dev, 3, ragged=True)
        enc = torch.rand(6, 14, 14, 2048, device=dev)
        la = a.step(imgs, tags, caps, caplens, enc)
        lb = b.step(imgs, tags, caps, caplens, enc)
        torch.cuda.synchronize()
        assert abs(la.item() - lb.item()) < 1e-6
        assert all(b.reducers[0].launched)
        assert torch.equal(a.decoder_optimizer.flat.flat_p, b.decoder_optimizer.flat.flat_p)   # bitwise: no atomics anywhere
    finally:
        dist.destroy_process_group()


def test_embedding_gradient_is_deterministic_and_handles_repeated_tokens(dev):
    """All captions use the same few tokens (every vocabulary row that occurs is hit by hundreds of
    cells, one of them more than the kernel's in-LDS list holds): the embedding gradient must equal the
    oracle's and be bitwise identical across runs (no float atomics)."""
    from oracle import scnattn_ref as R
    from models.decoders.pure_scn import PureSCN
    torch.manual_seed(0)
    B, V, L = 48, 12, 40
    m = PureSCN(16, 16, 16, 6, V, encoder_dim=24, dropout=0.0)
    enc = torch.rand(B, 2, 2, 24)
    tags = torch.rand(B, 6)
    caps = torch.randint(1, 4, (B, L))
    caps[:, ::2] = 5                     # token 5 fills > 1024 of the 48*39 cells
    caplens = torch.full((B, 1), L)
    si = torch.arange(B)     # all lengths tie: pin the permutation (quirk Q2: the sort is not stable on CPU)
    P = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    pr, cs, dl, _ = R.pure_scn_forward(P, enc.double(), tags.double(), caps, caplens, sort_ind=si)
    loss_r, _, _ = R.caption_loss(pr, cs, dl, None)
    loss_r.backward()
    m = m.to(dev).train()
    grads = []
    for _ in range(2):
        m.zero_grad()
        preds, caps_s, dl2, _ = m(enc.to(dev), tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev))
        loss, _, _ = R.caption_loss(preds, caps_s, dl2, None)
        loss.backward()
        grads.append(m.embedding.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])
    _ok(grads[0], P["embedding.weight"].grad, TOL_GRAD, "embedding grad")


def test_validate_path(dev):
    """validate() (trains/attention_scn.py:274-385) on the HIP decoder vs the same bookkeeping done with
    the CPU oracle: loss, top-5 accuracy and corpus BLEU-4 (NLTK-free)."""
    from oracle import scnattn_ref as R
    from models.decoders.attention_scn import AttentionSCN
    from trains.harness import validate
    from utils.metric import corpus_bleu
    torch.manual_seed(3)
    B, V, L = 6, 40, 9
    wm = {"<pad>": 0, "<unk>": V - 3, "<start>": V - 2, "<end>": V - 1}
    m = AttentionSCN(24, 20, 28, 36, 14, V, encoder_dim=40, dropout=0.5)
    g = torch.Generator().manual_seed(5)
    enc = torch.rand(B, 3, 3, 40, generator=g)
    tags = torch.rand(B, 14, generator=g)
    lens = torch.tensor([9, 7, 8, 5, 6, 4])
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(lens[b]); caps[b, 0] = V - 2; caps[b, 1:n - 1] = torch.randint(1, V - 3, (n - 2,), generator=g); caps[b, n - 1] = V - 1
    allcaps = torch.stack([caps, caps.roll(1, 0)], dim=1)           # 2 references per image
    caplens = lens.unsqueeze(1)
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    pr, cs, dl, al, si = R.attention_scn_forward(P, enc, tags, caps, caplens)
    loss_r, sc, tg = R.caption_loss(pr, cs, dl, al, 1.0)
    skip = {wm["<start>"], wm["<pad>"]}
    refs = [[[w for w in c if w not in skip] for c in allcaps[si][j].tolist()] for j in range(B)]
    hyps = [p[:dl[j]] for j, p in enumerate(pr.argmax(dim=2).tolist())]
    bleu_r = corpus_bleu(refs, hyps)
    m = m.to(dev)
    crit = torch.nn.CrossEntropyLoss()
    bleu, loss, top5 = validate([(None, caps.to(dev), caplens.to(dev), allcaps.to(dev))], type("E", (), {"eval": lambda s: None, "__call__": lambda s, x: enc.to(dev)})(),
                                lambda x: tags.to(dev), m, crit, wm)
    assert not m.training
    assert abs(loss - loss_r.item()) < 1e-4 * abs(loss_r.item())
    assert abs(top5 - R.topk_accuracy(sc, tg, 5)) < 1e-6
    assert abs(bleu - bleu_r) < 1e-12


def test_tags_gradient_and_eval_mode(dev):
    """d loss / d semantic_input (only formed when the tags require grad, e.g. a trainable tagger head) and
    eval()-mode forward (dropout off even with p = 0.5)."""
    from oracle import scnattn_ref as R
    from models.decoders.attention_scn import AttentionSCN
    torch.manual_seed(9)
    B, V, L = 5, 30, 8
    m = AttentionSCN(24, 20, 28, 36, 14, V, encoder_dim=40, dropout=0.5)
    g = torch.Generator().manual_seed(2)
    enc = torch.rand(B, 3, 3, 40, generator=g)
    tags = torch.rand(B, 14, generator=g)
    lens = torch.tensor([8, 6, 7, 3, 5])
    caps = torch.randint(1, V - 3, (B, L), generator=g)
    caplens = lens.unsqueeze(1)
    P = {k: v.detach().clone().double() for k, v in m.state_dict().items()}
    t1 = tags.double().requires_grad_(True)
    pr, cs, dl, al, si = R.attention_scn_forward(P, enc.double(), t1, caps, caplens)
    loss_r, _, _ = R.caption_loss(pr, cs, dl, al, 1.0)
    loss_r.backward()
    m = m.to(dev).eval()
    t2 = tags.to(dev).requires_grad_(True)
    preds, caps_s, dl2, alphas, _ = m(enc.to(dev), t2, caps.to(dev), caplens.to(dev))
    _ok(preds, pr, TOL_OUT, "eval-mode preds")
    loss, _, _ = R.caption_loss(preds, caps_s, dl2, alphas, 1.0)
    loss.backward()
    _ok(t2.grad, t1.grad, TOL_GRAD, "dtags")


def test_pure_attention_full_width(dev):
    """BASELINE config 1 shape (PureAttention, B=4, max_len 20 -> T=21) at the real widths, V reduced."""
    from oracle import scnattn_ref as R
    from models.decoders.pure_attention import PureAttention
    torch.manual_seed(5)
    B, V, L = 4, 300, 22
    m = PureAttention(512, 512, 512, V, dropout=0.0)
    g = torch.Generator().manual_seed(6)
    enc = torch.rand(B, 14, 14, 2048, generator=g)
    lens = torch.tensor([22, 15, 19, 9])
    caps = torch.randint(1, V - 3, (B, L), generator=g)
    caplens = lens.unsqueeze(1)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    pr, cs, dl, al, si = R.pure_attention_forward(P, enc, caps, caplens)
    loss_r, _, _ = R.caption_loss(pr, cs, dl, al, 1.0)
    loss_r.backward()
    m = m.to(dev).train()
    preds, caps_s, dl2, alphas, _ = m(enc.to(dev), caps.to(dev), caplens.to(dev))
    _ok(preds, pr, TOL_OUT, "preds"); _ok(alphas, al, TOL_OUT, "alphas")
    loss, _, _ = R.caption_loss(preds, caps_s, dl2, alphas, 1.0)
    loss.backward()
    floors = {k: 1e-2 for k in P if k.startswith("attention.encoder_att") or k.startswith("attention.decoder_att")}
    _check_grads(m.named_parameters(), lambda k: P[k].grad, 1e-3, floors)


def test_argument_errors_surface_as_runtime_errors(dev):
    """Limits of the LDS staging are reported through the C ABI's error channel, not by a fault."""
    from scnattn._lib import call, ptr, stream_of
    x = torch.zeros(8, device=dev)
    with pytest.raises(RuntimeError, match="attention_dim too large"):
        call("scnattn_attn_scores", stream_of(x), 1, 4, 1 << 16, ptr(x), ptr(x), 1, 0, 1 << 16, None, ptr(x), None, ptr(x), None)
    with pytest.raises(RuntimeError, match="K must be >= 1"):
        call("scnattn_sgemm", stream_of(x), 0, 0, 2, 2, 0, 1.0, ptr(x), 2, ptr(x), 2, 0.0, ptr(x), 2, None, None, 1, 0, 0, 0)
    from models.decoders.attention_scn import AttentionSCN
    m = AttentionSCN(8, 8, 8, 8, 4, 12, encoder_dim=8, dropout=0.0).to(dev)
    with pytest.raises(RuntimeError):   # a caption of length 1 would decode 0 steps
        m(torch.rand(2, 2, 2, 8, device=dev), torch.rand(2, 4, device=dev), torch.zeros(2, 4, dtype=torch.long, device=dev),
          torch.tensor([[3], [1]], device=dev))


@pytest.mark.parametrize("training,with_res", [(True, True), (True, False), (False, True)])
def test_fused_batchnorm_bf16_storage(dev, training, with_res):
    """bf16 feature maps through the fused BatchNorm kernels (trunk under bf16 autocast, BASELINE config 5):
    fp32 statistics/arithmetic, so the only error is the bf16 rounding of inputs/outputs (2^-8 relative)."""
    from scnattn.resnet import FusedBatchNorm2d
    torch.manual_seed(11)
    N, C, H, W = 8, 128, 12, 12
    x = (torch.randn(N, C, H, W) * 1.5 + 0.4).bfloat16()
    res = torch.randn(N, C, H, W).bfloat16() if with_res else None
    wgt = torch.randn(N, C, H, W).bfloat16()
    ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.normal_(0, 0.3)
        ref.running_mean.normal_(0, 0.2); ref.running_var.uniform_(0.5, 2.0)
    m = FusedBatchNorm2d(C)
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    m = m.to(dev)
    ref.train(training); m.train(training)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    if with_res:
        yr = yr + res.double()
    yr = torch.relu(yr)
    (yr * wgt.double()).sum().backward()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = res.to(dev).contiguous(memory_format=torch.channels_last) if with_res else None
    y = m(xd, residual=rd, relu=True)
    assert y.dtype == torch.bfloat16
    _ok(y.float(), yr, 1e-2, "y")
    (y.float() * wgt.to(dev).float()).sum().backward()
    assert xd.grad.dtype == torch.bfloat16
    _ok(xd.grad.float(), xr.grad, 2e-2, "dx")
    _ok(m.weight.grad, ref.weight.grad, 1e-2, "dgamma"); _ok(m.bias.grad, ref.bias.grad, 1e-2, "dbeta")
    _ok(m.running_mean, ref.running_mean, 1e-4, "running_mean"); _ok(m.running_var, ref.running_var, 1e-4, "running_var")


@pytest.mark.gpu
def test_two_process_data_parallel_decoder_step(dev, tmp_path):
    """SURVEY 8e cross-check on the HIP path: two processes (gloo collectives, both on this GPU) each run the
    decoder forward/backward on half of a batch with the bucketed hook-driven all-reduce; their scaled
    gradient must equal the single-process gradient on the whole batch (fixed-length captions => equal
    token counts), every bucket must fire during backward, and both ranks must start from rank 0's weights
    and end the optimizer step with identical parameters."""
    import socket
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "dp_gpu_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    env = dict(os.environ)
    single = subprocess.run([sys.executable, worker, "0", "1", port, str(tmp_path)], env=env, capture_output=True,
                            text=True, timeout=300)
    assert single.returncode == 0, single.stderr[-2000:]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for pr in procs:
        out, err = pr.communicate(timeout=300)
        assert pr.returncode == 0, err[-2000:]
    one = torch.load(str(tmp_path / "w1_r0.pt"))
    r0 = torch.load(str(tmp_path / "w2_r0.pt"))
    r1 = torch.load(str(tmp_path / "w2_r1.pt"))
    assert torch.equal(r0["p0"], r1["p0"]), "broadcast did not align the ranks"
    assert torch.equal(r0["grad"], r1["grad"]) and torch.equal(r0["p1"], r1["p1"])
    assert r0["buckets"] >= 5 and r0["fired"] == r0["buckets"] == r1["fired"]
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - one["loss"]) <= 1e-5 * abs(one["loss"])
    # the single process used seed 77 like rank 0, so its weights are the broadcast ones
    assert torch.equal(one["p0"], r0["p0"])
    for name, off, n in zip(one["names"], one["offsets"], one["numels"]):
        a, b = r0["grad"][off:off + n], one["grad"][off:off + n]
        if name.endswith("full_att.bias"):
            assert (a - b).abs().max().item() <= 1e-5
            continue
        assert rel_l2(a, b) <= 2e-5, "%s: 2-rank gradient differs from the whole-batch gradient" % name


@pytest.mark.gpu
def test_two_process_data_parallel_encoder_buckets_on_the_side_stream(dev, tmp_path):
    """ADVICE r02 (dp.py:117): the CUDA branch of GradReducer with MORE than one rank and the fine-tuned trunk in the
    step -- in-place flat weight gradients produced on the side stream, bucket gather + all-reduce enqueued on that
    same stream from post-accumulate hooks, join only in finish().  Two processes on this GPU (gloo collectives) run
    the SAME batch, so their local gradients are identical and the reduced, scaled flat gradient buffers of decoder and
    encoder must equal the single-process ones bit for bit; every bucket fires during backward."""
    import socket
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "dp_gpu_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    env = dict(os.environ)
    single = subprocess.run([sys.executable, worker, "0", "1", port, str(tmp_path), "enc"], env=env, capture_output=True,
                            text=True, timeout=600)
    assert single.returncode == 0, single.stderr[-2000:]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, str(tmp_path), "enc"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for pr in procs:
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, err[-2000:]
    one = torch.load(str(tmp_path / "enc_w1_r0.pt"))
    r0 = torch.load(str(tmp_path / "enc_w2_r0.pt"))
    r1 = torch.load(str(tmp_path / "enc_w2_r1.pt"))
    assert r0["buckets"][1] >= 10 and r0["fired"] == r0["buckets"] == r1["fired"], (r0["buckets"], r0["fired"])
    assert abs(r0["loss"] - one["loss"]) <= 1e-6 * abs(one["loss"])
    for which in (0, 1):
        assert float(one["grads"][which].abs().max()) > 0
        assert torch.equal(r0["grads"][which], r1["grads"][which]), "ranks disagree after the all-reduce"
        assert torch.equal(r0["grads"][which], one["grads"][which]), \
            "2-rank flat gradient (%s) differs from the 1-rank one: max abs %.3e" % \
            ("decoder" if which == 0 else "encoder", (r0["grads"][which] - one["grads"][which]).abs().max().item())


@pytest.mark.parametrize("B,T,V,P,lens,with_alpha", [
    (6, 5, 40, 9, [5, 5, 4, 3, 3, 1], True),        # ragged, V % 4 == 0
    (4, 7, 37, 16, [7, 6, 2, 1], True),             # V % 4 != 0: scalar path
    (5, 4, 1000, 0, [4, 4, 4, 4, 4], False),        # PureSCN: no alphas; full lengths
    (3, 6, 10000, 196, [6, 4, 3], True),            # BASELINE vocabulary / pixel count
])
def test_fused_caption_loss_matches_oracle(dev, B, T, V, P, lens, with_alpha):
    """csrc/loss.hip against the oracle's restatement of trains/attention_scn.py:222-236
    (pack_padded_sequence x2, CrossEntropyLoss, doubly-stochastic term): value, d scores, d alphas; rows that
    were not decoded get exactly zero gradient; the loss is scaled by an upstream factor to exercise g."""
    from oracle import scnattn_ref as R
    from scnattn import functional as SF
    g = torch.Generator().manual_seed(B * 100 + V)
    scores = (3 * torch.randn(B, T, V, generator=g)).requires_grad_(True)
    caps = torch.randint(0, V, (B, T + 2), generator=g)
    alphas = None
    if with_alpha:
        alphas = torch.softmax(torch.randn(B, T, P, generator=g), dim=2)
        for b, l in enumerate(lens):
            alphas[b, l:] = 0                      # the decoder leaves rows beyond decode_length at zero
        alphas.requires_grad_(True)
    ref, _, _ = R.caption_loss(scores, caps, lens, alphas, alpha_c=0.7)
    (2.5 * ref).backward()
    sd = scores.detach().to(dev).requires_grad_(True)
    ad = alphas.detach().to(dev).requires_grad_(True) if with_alpha else None
    got = SF.caption_loss(sd, caps.to(dev), lens, ad, alpha_c=0.7)
    assert got.dim() == 0
    (2.5 * got).backward()
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    _ok(sd.grad, scores.grad, 1e-5, "d scores")
    for b, l in enumerate(lens):
        assert float(sd.grad[b, l:].abs().max()) == 0.0 if l < T else True
        assert float(sd.grad[b, :l].sum(dim=1).abs().max()) <= 1e-6          # softmax - onehot sums to 0 per row
    if with_alpha:
        _ok(ad.grad, alphas.grad, 1e-5, "d alphas")
    dl_dev = torch.tensor(lens, dtype=torch.int32, device=dev)
    again = SF.caption_loss(sd.detach(), caps.to(dev), lens, None if ad is None else ad.detach(), 0.7, dl_dev)
    assert torch.equal(again, got.detach())                                    # deterministic, dl_dev path identical
    bad = caps.clone()
    bad[0, 1] = V + 5
    assert torch.isnan(SF.caption_loss(sd.detach(), bad.to(dev), lens, None, 0.7))   # bad label: NaN, no fault
    with pytest.raises(RuntimeError):
        SF.caption_loss(sd.detach().cpu(), caps, lens)


def test_train_step_fused_loss_equals_reference_op_sequence(dev):
    """Whole decoder train step at BASELINE dims (B=32, T=51, V=10000), once with the fused loss and once
    with the reference's op sequence (pack_padded_sequence + CrossEntropyLoss + alpha term in torch ops):
    same loss and the same parameters after the update (dropout off, identical seeds)."""
    from trains.harness import TrainStep, synthetic_batch
    outs = []
    for fused in (True, False):
        ts = TrainStep(kind="attention_scn", fine_tune_encoder=False, device=dev, encoder=False, dropout=0.0,
                       fused_loss=fused, seed=4)
        cfg = ts.cfg
        _, tags, caps, caplens = synthetic_batch(32, cfg["vocab_size"], cfg["max_len"], 8, cfg["semantic_dim"], dev, 6,
                                                 ragged=True)
        enc = torch.rand(32, 14, 14, 2048, generator=torch.Generator().manual_seed(8)).to(dev)
        loss = ts.step(None, tags, caps, caplens, enc)
        ts.decoder_optimizer.flat.gather()
        outs.append((loss.item(), ts.decoder_optimizer.flat.flat_g.clone(), ts.decoder_optimizer.flat.flat_p.clone()))
        del ts
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * abs(outs[1][0])
    assert rel_l2(outs[0][1], outs[1][1]) <= 1e-5
    assert rel_l2(outs[0][2], outs[1][2]) <= 1e-5


@pytest.mark.parametrize("kind", ["attention_scn", "pure_scn"])
def test_full_length_batch_never_reads_unwritten_workspace(dev, kind):
    """With fixed-length captions the 120 MB zero fills of the sequence workspaces are skipped because every
    element is written before it is read.  Proof: NaN-poison those workspaces; outputs and every gradient must
    stay finite and bit-identical to the unpoisoned run.  Also covers the skipped identity permutation (the
    batch is already in length order) against an explicitly permuted run."""
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from scnattn import functional as SF
    torch.manual_seed(31)
    B, V, L = 8, 50, 9
    if kind == "attention_scn":
        m = AttentionSCN(32, 24, 32, 40, 12, V, encoder_dim=64, dropout=0.0).to(dev).train()
    else:
        m = PureSCN(24, 32, 40, 12, V, encoder_dim=64, dropout=0.0).to(dev).train()
    enc = torch.rand(B, 4, 4, 64, device=dev)
    tags = torch.rand(B, 12, device=dev)
    caps = torch.randint(1, V - 3, (B, L), device=dev)
    caplens = torch.full((B, 1), L, device=dev)
    runs = []
    for poison in (False, True):
        SF._POISON = poison
        try:
            m.zero_grad(set_to_none=True)
            e = enc.clone().requires_grad_(True)
            out = m(e, tags, caps, caplens)
            loss = out[0].square().sum() + (out[3].square().sum() if kind == "attention_scn" else 0)
            loss.backward()
            runs.append([out[0].detach().clone(), e.grad.clone()] + [p.grad.clone() for p in m.parameters()] +
                        ([out[3].detach().clone()] if kind == "attention_scn" else []))
        finally:
            SF._POISON = False
    for a, b in zip(*runs):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    # identity permutation skipped == explicit permutation applied: reverse the batch and un-reverse the results
    rev = torch.arange(B - 1, -1, -1, device=dev)
    e2 = enc[rev].clone().requires_grad_(True)
    out2 = m(e2, tags, caps[rev], caplens, sort_ind=rev)           # Q1: tags are indexed un-permuted by sorted position
    assert torch.equal(out2[1], caps) and out2[-1] is rev
    _ok(out2[0], runs[0][0], 1e-6, "predictions under an explicit permutation")


@pytest.mark.parametrize("B,lens,hin", [(6, [8, 8, 6, 5, 3, 2], 8), (4, [5, 5, 5, 5], 8), (3, [7, 4, 4], 7)])
def test_pooled_attention_path_equals_dense_path(dev, B, lens, hin):
    """Attention on the trunk's un-pooled map (scnattn_pool: att1 = pool(x.We^T)+be, context / d alpha over the Q
    source pixels, d x produced directly) against the dense path fed with the materialised AdaptiveAvgPool2d(14)
    output: predictions, alphas, every parameter gradient, and d x against the dense d encoder_out pushed back
    through the pool.  Ragged lengths, an explicit permutation (unsorted batch) and a non-square-friendly 7x7 map."""
    from models.decoders.attention_scn import AttentionSCN
    torch.manual_seed(5 + B)
    V, L, E = 50, 10, 64
    m = AttentionSCN(32, 24, 32, 40, 12, V, encoder_dim=E, dropout=0.0).to(dev).train()
    x = torch.rand(B, hin, hin, E, device=dev)
    tags = torch.rand(B, 12, device=dev)
    caps = torch.randint(1, V - 3, (B, L), device=dev)
    order = torch.randperm(B)                     # lengths not sorted: the decoder must permute x itself
    caplens = (torch.tensor(lens)[order] + 1).unsqueeze(1).to(dev)
    w = torch.rand(B, L - 1, V, device=dev)
    results = []
    for pooled in (False, True):
        m.zero_grad(set_to_none=True)
        xx = x.clone().requires_grad_(True)
        enc = torch.nn.functional.adaptive_avg_pool2d(xx.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1)
        if pooled:
            out = m(None, tags, caps, caplens, prepool=xx, pool_size=14)
        else:
            out = m(enc, tags, caps, caplens)
        T = out[0].shape[1]
        loss = (out[0] * w[:, :T]).sum() + (out[3] ** 2).sum()
        loss.backward()
        results.append((out[0].detach(), out[3].detach(), xx.grad.clone(),
                        {k: p.grad.clone() for k, p in m.named_parameters()}, out[4]))
    (p0, a0, dx0, g0, s0), (p1, a1, dx1, g1, s1) = results
    assert torch.equal(s0, s1)
    _ok(p1, p0, 1e-5, "predictions")
    _ok(a1, a0, 1e-5, "alphas")
    assert rel_l2(dx1, dx0) <= 2e-5, "d x: %.3e" % rel_l2(dx1, dx0)
    for k in g0:
        if k.endswith("full_att.bias"):
            assert (g1[k] - g0[k]).abs().max().item() <= 1e-4
            continue
        assert rel_l2(g1[k], g0[k]) <= 5e-5, "%s: %.3e" % (k, rel_l2(g1[k], g0[k]))


def test_encoder_attaches_prepool_and_decoder_uses_it(dev):
    """EncoderCaption tags the pooled tensor it returns with the trunk map; the decoder then takes the pooled
    path on its own (the reference's train loop passes the tensor through untouched), falls back to the dense
    path when the tensor was modified in place, and both give the same numbers."""
    from models.decoders import _common
    from models.decoders.attention_scn import AttentionSCN
    from models.encoders.caption import EncoderCaption
    from scnattn.resnet import resnet152_trunk
    torch.manual_seed(2)
    encm = EncoderCaption(channels_last=True)
    encm.resnet = resnet152_trunk(depths=(1, 1, 1, 1))
    encm = encm.to(dev).eval()
    dec = AttentionSCN(32, 24, 32, 40, 12, 40, dropout=0.0).to(dev).train()
    imgs = torch.randn(2, 3, 64, 64, device=dev)
    tags = torch.rand(2, 12, device=dev)
    caps = torch.randint(1, 36, (2, 7), device=dev)
    caplens = torch.tensor([[7], [5]], device=dev)
    with torch.no_grad():
        y = encm(imgs)
        assert y.shape == (2, 14, 14, 2048) and _common.attached_prepool(y) is not None
        assert _common.attached_prepool(y).shape == (2, 2, 2, 2048)
        assert tuple(encm(imgs, pooled=False).shape) == (2, 2, 2, 2048)
        auto = dec(y, tags, caps, caplens)
        _common.USE_PREPOOL = False
        try:
            dense = dec(y, tags, caps, caplens)
        finally:
            _common.USE_PREPOOL = True
        _ok(auto[0], dense[0], 1e-5, "auto-pooled vs dense")
        y.mul_(1.0)                                   # in-place touch: version counter moves, tag is void
        assert _common.attached_prepool(y) is None
        assert _common.attached_prepool(y * 1.0) is None


def test_pure_scn_pooled_path_equals_dense_path(dev):
    """PureSCN reads the encoder output only through its pixel mean (pure_scn.py:76-85): with the trunk map that
    mean is a weighted mean of the 64 source pixels; predictions, gradients and d x must match the dense path."""
    from models.decoders.pure_scn import PureSCN
    torch.manual_seed(9)
    B, V, L, E = 5, 40, 8, 64
    m = PureSCN(24, 32, 40, 12, V, encoder_dim=E, dropout=0.0).to(dev).train()
    x = torch.rand(B, 8, 8, E, device=dev)
    tags = torch.rand(B, 12, device=dev)
    caps = torch.randint(1, V - 3, (B, L), device=dev)
    caplens = torch.tensor([[8], [8], [6], [4], [3]], device=dev)
    res = []
    for pooled in (False, True):
        m.zero_grad(set_to_none=True)
        xx = x.clone().requires_grad_(True)
        enc = torch.nn.functional.adaptive_avg_pool2d(xx.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1)
        out = m(None, tags, caps, caplens, prepool=xx) if pooled else m(enc, tags, caps, caplens)
        out[0].square().sum().backward()
        res.append((out[0].detach(), xx.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    _ok(res[1][0], res[0][0], 1e-5, "predictions")
    assert rel_l2(res[1][1], res[0][1]) <= 2e-5
    for k in res[0][2]:
        assert rel_l2(res[1][2][k], res[0][2][k]) <= 5e-5, k


def test_pure_attention_sequence_driver_equals_stepwise_loop(dev):
    """PureAttention.forward (LSTMCell expressed as an SCN cell with unit tag factors and identity third factors,
    one C call each way) against its literal step-by-step loop with torch's own LSTMCell: outputs and every
    gradient, including decode_step.weight_ih / weight_hh / biases through the gate re-ordering, ragged lengths."""
    from models.decoders.pure_attention import PureAttention
    torch.manual_seed(17)
    B, V, L = 6, 45, 9
    m = PureAttention(24, 20, 32, V, encoder_dim=64, dropout=0.0).to(dev).train()
    enc = torch.rand(B, 4, 4, 64, device=dev)
    caps = torch.randint(1, V - 3, (B, L), device=dev)
    caplens = torch.tensor([[9], [9], [7], [5], [4], [2]], device=dev)
    res = []
    for fn in (m.forward_stepwise, m.forward):
        m.zero_grad(set_to_none=True)
        e = enc.clone().requires_grad_(True)
        p, _, dl, a, _ = fn(e, caps, caplens)
        (p.square().sum() + (a * a).sum()).backward()
        res.append((p.detach(), a.detach(), e.grad.clone(), {k: q.grad.clone() for k, q in m.named_parameters()}))
    _ok(res[1][0], res[0][0], 1e-5, "predictions")
    _ok(res[1][1], res[0][1], 1e-5, "alphas")
    assert rel_l2(res[1][2], res[0][2]) <= 1e-4, rel_l2(res[1][2], res[0][2])
    for k in res[0][3]:
        if k.endswith("full_att.bias"):
            assert (res[1][3][k] - res[0][3][k]).abs().max().item() <= 1e-4
            continue
        assert rel_l2(res[1][3][k], res[0][3][k]) <= 1e-4, "%s: %.3e" % (k, rel_l2(res[1][3][k], res[0][3][k]))
