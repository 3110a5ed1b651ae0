"""GPU parity tests added in round 2 (``-m gpu``), all through the C ABI.

What they close ("parity gaps"):
  * the path ``bench.py`` times -- attention on the encoder trunk's 8x8 map (``prepool=``) -- against the fp64
    CPU oracle fed with ``AdaptiveAvgPool2d(14)`` of the same map, at BASELINE widths (B=32, Q=64 -> P=196,
    E=2048, A=D=F=M=512, S=1000), ragged and fixed lengths, ``attention_scn`` and ``pure_scn``, and once at the
    exact BASELINE config-3 sizes T=51, V=10 000;
  * a *mask-unambiguous* variant: inputs / ``attention.encoder_att`` rigged (``_make_unambiguous``) so that every
    ReLU pre-activation stays > 1e-3 from 0 (asserted on the fp64 side) while the mask still varies over pixels and
    units, so no mask bit can flip between two fp32 evaluations and
    the gradients of ``attention.{encoder_att,decoder_att}.*`` and ``d x`` must meet 2e-4 with NO floors;
  * the HIP train step (fused loss -> FusedClampAdam, two steps) against the reference-generated
    ``g_clamped.* / p_after.* / p_after2.*`` fixtures (trains/attention_scn.py:238-252);
  * ``EncoderTagger.forward`` (N1) and ``sample()`` (N2) against oracles.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, params_from, t, rel_err, rel_l2

pytestmark = pytest.mark.gpu

TOL_OUT = 1e-4     # north_star: outputs within 1e-4 rel-err of the CPU reference
TOL_GRAD = 2e-4    # gradients (sums over up to 51 steps in a different, fixed reduction order)


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


def _report(lines, title):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "parity_report_r02.txt"), "a") as f:
        f.write("== %s\n" % title)
        for ln in lines:
            f.write(ln + "\n")


def _synthetic_caps(B, V, L, lens, g):
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        caps[b, 0] = V - 2
        caps[b, 1:n - 1] = torch.randint(1, V - 3, (n - 2,), generator=g)
        caps[b, n - 1] = V - 1
    return caps


def _make_unambiguous(sd, x, g):
    """Rig the attention so that NO ReLU pre-activation att1[p,a] + att2[a] comes near 0, while the mask still
    varies over pixels p and units a (a mask that is constant over p would make d att2 exactly zero by the softmax's
    shift invariance, and the test would say nothing about decoder_att):
      * channel 0 of the trunk map x is a random 0 / 3 pattern per source pixel, so its AdaptiveAvgPool2d(14)
        (1-, 2- or 4-tap averages) takes the values {0, .75, 1.5, 2.25, 3}: never closer than 0.25 to 1;
      * encoder_att.weight[:, 0] = +-8 (alternating per unit), encoder_att.bias = -(+-8), the other columns x 0.25:
        pre = +-8 * (pool(x0)[p] - 1) + (small random part + att2) -- at least 2 away from 0 before the random part.
    The margin that actually results is measured on the fp64 oracle (RELU_PROBE) and asserted by the caller."""
    A = sd["attention.encoder_att.bias"].numel()
    sign = torch.where(torch.arange(A) % 2 == 0, 1.0, -1.0)
    W = sd["attention.encoder_att.weight"].clone() * 0.25
    W[:, 0] = 8.0 * sign
    sd["attention.encoder_att.weight"] = W
    sd["attention.encoder_att.bias"] = -8.0 * sign
    x = x.clone()
    x[..., 0] = 3.0 * (torch.rand(x.shape[:-1], generator=g) > 0.5).float()
    return x


def _oracle_run(kind, sd, x, tags, caps, caplens, mask, si, dt, probe=False):
    """CPU oracle on AdaptiveAvgPool2d(14)(x); gradients for every parameter and for x (through the pool)."""
    from oracle import scnattn_ref as R
    P = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in sd.items()}
    x1 = x.clone().to(dt).requires_grad_(True)
    enc = F.adaptive_avg_pool2d(x1.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1)
    R.RELU_PROBE = [] if probe else None
    try:
        if kind == "attention_scn":
            pr, cs, dl, al, _ = R.attention_scn_forward(P, enc, tags.to(dt), caps, caplens,
                                                        drop_mask=None if mask is None else mask.to(dt),
                                                        sort_ind=si, hoist=True)
        elif kind == "pure_scn":
            pr, cs, dl, _ = R.pure_scn_forward(P, enc, tags.to(dt), caps, caplens,
                                               drop_mask=None if mask is None else mask.to(dt), sort_ind=si)
            al = None
        else:
            pr, cs, dl, al, _ = R.pure_attention_forward(P, enc, caps, caplens,
                                                         drop_mask=None if mask is None else mask.to(dt), sort_ind=si)
        margin = min(R.RELU_PROBE) if probe and R.RELU_PROBE else None
    finally:
        R.RELU_PROBE = None
    loss, _, _ = R.caption_loss(pr, cs, dl, al, 1.0)
    loss.backward()
    g = {k: v.grad for k, v in P.items()}
    g["__x"] = x1.grad
    return pr.detach(), (None if al is None else al.detach()), loss.detach(), g, margin


def _hip_run(kind, m, x, tags, caps, caplens, mask, si, dev):
    from oracle import scnattn_ref as R
    m = m.to(dev).train()
    m.drop_mask_override = None if mask is None else mask.to(dev)
    x2 = x.to(dev).requires_grad_(True)
    if kind == "pure_attention":
        out = m(None, caps.to(dev), caplens.to(dev), sort_ind=si.to(dev), prepool=x2, pool_size=14)
        alphas = out[3]
    elif kind == "pure_scn":
        out = m(None, tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev), prepool=x2, pool_size=14)
        alphas = None
    else:
        out = m(None, tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev), prepool=x2, pool_size=14)
        alphas = out[3]
    loss, _, _ = R.caption_loss(out[0], out[1], out[2], alphas, 1.0)
    loss.backward()
    return out[0], alphas, loss, x2.grad, m


def _compare(kind, m, hip, ref64, floors, title, tol_out=TOL_OUT, tol_grad=TOL_GRAD):
    preds, alphas, loss, dx, mg = hip
    pr64, al64, loss64, g64, _ = ref64
    rep = ["preds  err %.3e" % rel_err(preds, pr64)]
    bad = []
    try:
        _ok(preds, pr64, tol_out, "preds")
        if alphas is not None:
            rep.append("alphas err %.3e" % rel_err(alphas, al64))
            _ok(alphas, al64, tol_out, "alphas")
        _ok(loss, loss64, tol_out, "loss")
        named = list(mg.named_parameters()) + [("__x", None)]
        for k, p in named:
            got = dx if k == "__x" else p.grad
            r = g64[k]
            if k.endswith("full_att.bias"):   # exactly 0 in exact arithmetic (softmax shift invariance)
                err, lim = (got.detach().double().cpu() - r.double()).abs().max().item(), 1e-4
            else:
                err, lim = rel_err(got, r), max(tol_grad, floors.get(k, 0.0) if floors else 0.0)
            rep.append("%-40s err %.3e lim %.1e" % ("d x (trunk map)" if k == "__x" else k, err, lim))
            if err > lim:
                bad.append("%s err %.3e > %.1e" % (k, err, lim))
        assert not bad, "; ".join(bad)
    finally:
        _report(rep, title)


# ------------------------------------------------------------------------------------------------
# 1a: the timed (pooled) path against the fp64 oracle at BASELINE widths
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,ragged", [("attention_scn", False), ("attention_scn", True), ("pure_scn", True)])
def test_pooled_path_full_width_vs_oracle(dev, kind, ragged):
    """prepool = x (32,8,8,2048) to the HIP decoder, AdaptiveAvgPool2d(14)(x) to the fp64 oracle: outputs 1e-4, every
    gradient 2e-4 except the five tensors downstream of the ReLU mask (fixed 2e-3, see below).  The mask-unambiguous
    tests further down hold those five to 2e-4 as well."""
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    torch.manual_seed(7)
    B, V, L = 32, 1000, 14
    m = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5) if kind == "attention_scn" \
        else PureSCN(512, 512, 512, 1000, V, dropout=0.5)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 8, 8, 2048, generator=g)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.randint(5, L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    caps = _synthetic_caps(B, V, L, lens, g)
    caplens = lens.unsqueeze(1)
    T = int(lens.max()) - 1
    mask = (torch.rand(B, T, 512, generator=g) > 0.5).float() * 2.0
    si = torch.sort(lens, descending=True, stable=True)[1]
    sd = m.state_dict()
    r64 = _oracle_run(kind, sd, x, tags, caps, caplens, mask, si, torch.float64)
    # Random weights put some of the 41 M ReLU pre-activations (b, t, p, a) within fp32 rounding of 0; each such mask
    # bit is decided by the rounding of one particular evaluation order, and one flipped bit moves a row of the five
    # tensors downstream of the mask by O(1/sqrt(#terms)) (round 1 measured the reference's OWN fp32 arithmetic at
    # 1e-3 from fp64 on them).  Those five get 2e-3 here; everything else 2e-4.  That the HIP gradients DO meet 2e-4 on
    # these very tensors when no pre-activation is ambiguous is what the mask-unambiguous tests below establish.
    floors = {k: 2e-3 for k in ("attention.encoder_att.weight", "attention.encoder_att.bias",
                                "attention.decoder_att.weight", "attention.decoder_att.bias", "__x")}
    hip = _hip_run(kind, m, x, tags, caps, caplens, mask, si, dev)
    _compare(kind, m, hip, r64, floors, "pooled path, full width, %s ragged=%s" % (kind, ragged))


# ------------------------------------------------------------------------------------------------
# 1b: mask-unambiguous -> 2e-4 with no floors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pooled", [True, False])
def test_attention_gradients_meet_2e4_when_relu_mask_is_unambiguous(dev, pooled):
    """Attention rigged by _make_unambiguous: min |att1 + att2| over every (t, b, p, a) is asserted > 1e-3 on the
    fp64 side, so the ReLU mask is the same bit pattern in any fp32 evaluation.  Then EVERY gradient -- including
    attention.{encoder_att,decoder_att}.{weight,bias} and d x / d encoder_out -- must be within 2e-4, no floors."""
    from models.decoders.attention_scn import AttentionSCN
    torch.manual_seed(21)
    B, V, L = 32, 1000, 14
    m = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = _make_unambiguous(sd, torch.rand(B, 8, 8, 2048, generator=g), g)
    m.load_state_dict(sd)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.randint(5, L + 1, (B,), generator=g)
    caps = _synthetic_caps(B, V, L, lens, g)
    caplens = lens.unsqueeze(1)
    T = int(lens.max()) - 1
    mask = (torch.rand(B, T, 512, generator=g) > 0.5).float() * 2.0
    si = torch.sort(lens, descending=True, stable=True)[1]
    r64 = _oracle_run("attention_scn", sd, x, tags, caps, caplens, mask, si, torch.float64, probe=True)
    assert r64[4] is not None and r64[4] > 1e-3, "ReLU margin %.3e" % r64[4]
    if pooled:
        hip = _hip_run("attention_scn", m, x, tags, caps, caplens, mask, si, dev)
    else:   # dense path: the materialised (B,14,14,E) map goes in, d encoder_out is pulled back through the pool
        from oracle import scnattn_ref as R
        mg = m.to(dev).train()
        mg.drop_mask_override = mask.to(dev)
        x2 = x.to(dev).requires_grad_(True)
        enc = F.adaptive_avg_pool2d(x2.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1).contiguous()
        out = mg(enc, tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev))
        loss, _, _ = R.caption_loss(out[0], out[1], out[2], out[3], 1.0)
        loss.backward()
        hip = (out[0], out[3], loss, x2.grad, mg)
    _compare("attention_scn", m, hip, r64, None, "mask-unambiguous (margin %.3f) pooled=%s" % (r64[4], pooled))


def test_baseline_config3_exact_sizes_pooled_path_vs_oracle(dev):
    """BASELINE configs[2] at its exact sizes -- B=32, T=51 (52-wide captions), V=10 000, dropout mask injected --
    on the pooled path against the fp64 oracle, mask-unambiguous so that 1e-4 / 2e-4 hold with no floors."""
    from models.decoders.attention_scn import AttentionSCN
    torch.manual_seed(31)
    B, V, L = 32, 10000, 52
    m = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(13)
    x = _make_unambiguous(sd, torch.rand(B, 8, 8, 2048, generator=g), g)
    m.load_state_dict(sd)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.full((B,), L)
    caps = _synthetic_caps(B, V, L, lens, g)
    caplens = lens.unsqueeze(1)
    mask = (torch.rand(B, L - 1, 512, generator=g) > 0.5).float() * 2.0
    si = torch.arange(B)
    r64 = _oracle_run("attention_scn", sd, x, tags, caps, caplens, mask, si, torch.float64, probe=True)
    assert r64[4] > 1e-3, "ReLU margin %.3e" % r64[4]
    hip = _hip_run("attention_scn", m, x, tags, caps, caplens, mask, si, dev)
    assert hip[0].shape == (B, 51, V)
    _compare("attention_scn", m, hip, r64, None, "BASELINE config 3 exact sizes (T=51, V=10000), margin %.3f" % r64[4])


# ------------------------------------------------------------------------------------------------
# 1d: PureAttention at full width, same scheme (replaces the 1e-3 / 1e-2-floor test)
# ------------------------------------------------------------------------------------------------
def test_pure_attention_full_width_mask_unambiguous(dev):
    """BASELINE config 1 shape (PureAttention, B=4, max_len 20 -> T=21) at the real widths: fp64 oracle, ReLU margin
    asserted, outputs 1e-4 and every gradient 2e-4; both the dense and the pooled path.  One stated exception: in this
    construction d att2 survives the softmax's shift invariance only through the variation of the ReLU mask over the
    pixels, a difference of near-equal sums, and the reference's OWN fp32 evaluation on the CPU is 1.6-2.0e-4 from fp64
    for attention.decoder_att.{weight,bias} and attention.encoder_att.bias (computed below, not assumed).  These three
    are allowed 3x the largest of their CPU-fp32 distances; the measured GPU errors moved between 5.6e-5 and 2.7e-4
    over four builds that differ only in summation order -- the spread of that yardstick."""
    from models.decoders.pure_attention import PureAttention
    from oracle import scnattn_ref as R
    torch.manual_seed(5)
    B, V, L = 4, 300, 22
    m = PureAttention(512, 512, 512, V, dropout=0.0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    x = _make_unambiguous(sd, torch.rand(B, 8, 8, 2048, generator=g), g)
    m.load_state_dict(sd)
    lens = torch.tensor([22, 15, 19, 9])
    caps = torch.randint(1, V - 3, (B, L), generator=g)
    caplens = lens.unsqueeze(1)
    si = torch.sort(lens, descending=True, stable=True)[1]
    r64 = _oracle_run("pure_attention", sd, x, None, caps, caplens, None, si, torch.float64, probe=True)
    assert r64[4] > 1e-3, "ReLU margin %.3e" % r64[4]
    r32 = _oracle_run("pure_attention", sd, x, None, caps, caplens, None, si, torch.float32)
    family = ("attention.decoder_att.weight", "attention.decoder_att.bias", "attention.encoder_att.bias")   # all fed by d att2
    e32 = max(rel_err(r32[3][k], r64[3][k]) for k in family)
    assert 3e-5 < e32 < 1e-3, e32          # the yardstick itself: 1.6-2.0e-4 on the build box's CPU
    floors = {k: 3.0 * e32 for k in family}
    hip = _hip_run("pure_attention", copy.deepcopy(m), x, None, caps, caplens, None, si, dev)
    _compare("pure_attention", m, hip, r64, floors, "pure_attention full width pooled, margin %.3f" % r64[4])
    mg = copy.deepcopy(m).to(dev).train()
    x2 = x.to(dev).requires_grad_(True)
    enc = F.adaptive_avg_pool2d(x2.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1).contiguous()
    out = mg(enc, caps.to(dev), caplens.to(dev), sort_ind=si.to(dev))
    loss, _, _ = R.caption_loss(out[0], out[1], out[2], out[3], 1.0)
    loss.backward()
    _compare("pure_attention", m, (out[0], out[3], loss, x2.grad, mg), r64, floors, "pure_attention full width dense")


# ------------------------------------------------------------------------------------------------
# 1c: HIP train step against the reference-generated train-step fixtures
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kind", [("attention_scn_distinct", "attention_scn"), ("pure_scn_distinct", "pure_scn"),
                                       ("pure_attention_distinct", "pure_attention")])
def test_hip_train_step_matches_reference_fixtures(dev, name, kind):
    """forward -> fused loss kernel -> backward -> FusedClampAdam (clamp +-5, Adam lr 4e-4), two steps, against
    what the REFERENCE's own modules + clip_gradient + torch.optim.Adam produced (oracle/gen_golden.py, from
    trains/attention_scn.py:238-252): loss, clamped gradients, parameters after step 1 and after step 2."""
    from test_gpu_parity import _build_decoder
    from scnattn import functional as SF
    from utils.optimizer import FusedClampAdam
    d = load_golden(name)
    m = _build_decoder(kind, d, dev)
    opt = FusedClampAdam([p for p in m.parameters() if p.requires_grad], lr=4e-4, grad_clip=5.0)
    enc = t(d["enc"]).to(dev)
    tags = t(d["tags"]).to(dev) if "tags" in d else None
    caps, caplens, si = t(d["caps"]).to(dev), t(d["caplens"]).to(dev), t(d["sort_ind"]).to(dev)
    rep = []
    for step, (lkey, pkey) in enumerate((("loss", "p_after."), ("loss2", "p_after2.")), start=1):
        if kind == "pure_attention":
            preds, caps_s, dl, alphas, _ = m(enc, caps, caplens, sort_ind=si)
        elif kind == "pure_scn":
            preds, caps_s, dl, _ = m(enc, tags, caps, caplens, sort_ind=si)
            alphas = None
        else:
            preds, caps_s, dl, alphas, _ = m(enc, tags, caps, caplens, sort_ind=si)
        loss = SF.caption_loss(preds, caps_s, dl, alphas, 1.0)
        rep.append("step %d loss err %.3e" % (step, rel_err(loss, d[lkey])))
        _ok(loss, d[lkey], TOL_OUT, lkey)
        opt.zero_grad()
        loss.backward()
        if step == 1:
            for k, p in m.named_parameters():
                key = "g_clamped." + k
                if key in d:
                    gc = p.grad.detach().clamp(-5.0, 5.0)
                    if k.endswith("full_att.bias"):
                        assert (gc.cpu().double() - torch.as_tensor(d[key]).double()).abs().max().item() <= 1e-4
                    else:
                        rep.append("%-40s clamped-grad err %.3e" % (k, rel_err(gc, d[key])))
                        _ok(gc, d[key], TOL_GRAD, key)
        opt.step()
        for k, p in m.named_parameters():
            # Adam's first steps move every weight by ~lr whatever the gradient's size, so a parameter-level
            # comparison is tight in absolute terms: 1e-6 on values of O(0.1)
            err = (p.detach().cpu().double() - torch.as_tensor(d[pkey + k]).double()).abs().max().item()
            # d loss / d full_att.bias is exactly 0 (softmax shift invariance): both sides hold rounding noise, and Adam
            # turns ANY non-zero gradient into a step of ~lr -- the fixture's own step there is such a noise step
            lim = 2 * step * 4e-4 if k.endswith("full_att.bias") else 2e-6
            assert err <= lim, "%s after step %d: abs err %.3e" % (k, step, err)
    _report(rep, "HIP train step vs reference fixtures: " + name)


# ------------------------------------------------------------------------------------------------
# N1: EncoderTagger.forward
# ------------------------------------------------------------------------------------------------
class _FixedMask(torch.nn.Module):
    """Stands in for nn.Dropout with a pinned, pre-scaled mask so both sides drop the same features."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask.to(x.device)


@pytest.mark.parametrize("train", [True, False])
def test_encoder_tagger_forward_vs_cpu(dev, train):
    """EncoderTagger.forward (reference models/encoders/tagger.py:34-47, called at trains/attention_scn.py:214):
    ResNet-152 trunk + global average pool -> Dropout(0.15) -> Linear(2048, 1000) -> Sigmoid, on the GPU (this
    build's conv / fused-BatchNorm kernels, Linear on the MFMA sgemm) against the SAME weights run by plain torch
    ops on the CPU, train mode (batch statistics, injected dropout mask) and eval mode (running statistics).
    PARITY UNPINNED against the reference itself: its trunk is third-party torchvision, absent from this image."""
    from models.encoders.tagger import EncoderTagger
    torch.manual_seed(3)
    m = EncoderTagger(semantic_size=1000, dropout=0.15, channels_last=True)
    # give the head a usable dynamic range (default init + 2048 near-identical pooled features -> sigmoid ~ 0.5)
    with torch.no_grad():
        m.linear.weight.mul_(4.0)
        for mod in m.resnet.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.05)
                mod.running_var.uniform_(0.8, 1.2)
    x = torch.randn(4, 3, 96, 96)
    mask = (torch.rand(4, 2048) > 0.15).float() / 0.85
    cpu = copy.deepcopy(m)
    cpu.train(train)
    with torch.no_grad():
        feat = cpu.resnet(x).reshape(4, -1)
        if train:
            feat = feat * mask
        ref = torch.sigmoid(F.linear(feat, cpu.linear.weight, cpu.linear.bias))
    g = m.to(dev)
    g.train(train)
    if train:
        g.dropout = _FixedMask(mask)
    with torch.no_grad():
        y = g(x.to(dev))
    assert y.shape == (4, 1000) and float(y.min()) >= 0.0 and float(y.max()) <= 1.0
    assert float(ref.max() - ref.min()) > 0.2, "degenerate test: tag probabilities do not vary"
    # 152 fp32 conv layers with batch statistics over 4 images amplify summation-order differences (the caption
    # encoder's own test accepts 5e-3 on the trunk output); the head is a contraction + sigmoid, which shrinks them
    e = _ok(y, ref, 2e-3, "tag probabilities (train=%s)" % train)
    _report(["tagger train=%s rel_err %.3e" % (train, e)], "EncoderTagger.forward vs CPU torch ops")
    if train:
        for (k, b), (_, bc) in zip(g.named_buffers(), cpu.named_buffers()):
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert rel_err(b, bc) < 1e-3, k


# ------------------------------------------------------------------------------------------------
# N2: sample() against the oracle restatement of the reference's beam search
# ------------------------------------------------------------------------------------------------
def _word_map(V):
    wm = {"<pad>": 0, "<unk>": V - 3, "<start>": V - 2, "<end>": V - 1}
    for i in range(1, V - 3):
        wm["w%d" % i] = i
    return wm


@pytest.mark.parametrize("name,kind", [("attention_scn_odd", "attention_scn"), ("attention_scn_distinct", "attention_scn"),
                                       ("pure_scn_distinct", "pure_scn"), ("pure_attention_distinct", "pure_attention")])
def test_sample_beam_search_vs_oracle(dev, name, kind):
    """sample() on the golden decoder weights (logits sharpened and <end> raised so that beams complete) against
    oracle/beam_ref.py -- the reference's attention_scn.py:160-296 / pure_scn.py:142-249 / pure_attention.py:153-281
    with `//` -- in fp64, for beam sizes 1, 3 and 5 and every image of the fixture: the chosen sequence, every
    completed sequence in completion order, their scores (1e-4) and the attention maps (1e-4).  PARITY UNPINNED
    against the reference's own sample(), which raises IndexError on this torch."""
    from test_gpu_parity import _build_decoder
    from models.decoders import _common
    from oracle import beam_ref as BR
    d = load_golden(name)
    V = d["p.embedding.weight"].shape[0]
    wm = _word_map(V)
    d = dict(d)
    # random-init logits are nearly uniform and stationary: sharpen them (fc.weight x 10) and raise <end> a little so
    # that beams complete at different steps (lengths 2-7 on these fixtures, fp32 == fp64 on the CPU oracle) while
    # some (image, beam) cases still never complete and exercise the documented fallback
    d["p.fc.weight"] = d["p.fc.weight"] * 10.0
    fb = d["p.fc.bias"].copy()
    fb[V - 1] += 0.2
    d["p.fc.bias"] = fb
    m = _build_decoder(kind, d, dev).eval()
    P64 = params_from(d, dtype=torch.float64)
    use_att, use_tags = kind != "pure_scn", kind != "pure_attention"
    compared = 0
    rep = []
    for b in range(d["enc"].shape[0]):
        enc = t(d["enc"])[b:b + 1]
        tags = t(d["tags"])[b:b + 1] if use_tags else None
        for k in (1, 3, 5):
            try:
                ref, ref_all = BR.beam_search(kind, P64, k, wm, enc.double(), None if tags is None else tags.double(),
                                              return_all=True)
            except ValueError:           # no beam completed within 50 steps: the reference would raise here
                with torch.no_grad():
                    out = m.sample(k, wm, enc.to(dev), tags.to(dev)) if use_tags else m.sample(k, wm, enc.to(dev))
                seq = out[0] if use_att else out
                assert seq[0] == V - 2 and len(seq) >= 51     # this build returns the best open beam instead
                continue
            with torch.no_grad():
                got, got_all = _common.beam_search(m, k, wm, enc.to(dev), None if tags is None else tags.to(dev),
                                                   use_attention=use_att, use_tags=use_tags, return_all=True)
            seq_r = ref[0] if use_att else ref
            seq_g = got[0] if use_att else got
            assert seq_g == seq_r, (b, k, seq_g, seq_r)
            assert [s for s, _ in got_all] == [s for s, _ in ref_all], (b, k)
            for (_, sg), (_, sr) in zip(got_all, ref_all):
                assert abs(sg - sr) <= 1e-4 * max(1.0, abs(sr)), (b, k, sg, sr)
            if use_att:
                e = _ok(torch.tensor(got[1]), torch.tensor(ref[1]), TOL_OUT, "alphas of the chosen beam")
                rep.append("image %d beam %d len %d alphas err %.2e" % (b, k, len(seq_r), e))
            # the public entry point returns the same thing
            with torch.no_grad():
                pub = m.sample(k, wm, enc.to(dev), tags.to(dev)) if use_tags else m.sample(k, wm, enc.to(dev))
            assert (pub[0] if use_att else pub) == seq_r
            compared += 1
    assert compared >= 4, "too few completing beams to call this a test (%d)" % compared
    _report(rep + ["%d (image, beam) cases compared" % compared], "sample() vs oracle beam search: " + name)


# ------------------------------------------------------------------------------------------------
# E1: the fused Bottleneck (hand-written 1x1 convolutions + BatchNorm prologues / epilogues)
# ------------------------------------------------------------------------------------------------
_BLOCKS = [  # (name, inplanes, planes, stride, H) -- every distinct 1x1-convolution shape of ResNet-152 at 256x256 input
    ("layer1.0", 64, 64, 1, 64), ("layer1.1", 256, 64, 1, 64), ("layer2.0", 256, 128, 2, 64), ("layer2.1", 512, 128, 1, 32),
    ("layer3.0", 512, 256, 2, 32), ("layer3.1", 1024, 256, 1, 16), ("layer4.0", 1024, 512, 2, 16), ("layer4.1", 2048, 512, 1, 8)]


@pytest.mark.parametrize("conv3", ["hip", "miopen"])
@pytest.mark.parametrize("small", [True, False])
@pytest.mark.parametrize("name,inplanes,planes,stride,H", _BLOCKS)
def test_fused_bottleneck_vs_fp64(dev, name, inplanes, planes, stride, H, small, conv3):
    """One Bottleneck of the trunk behind models/encoders/caption.py:17-22 through scnattn/conv.py -- conv1 / conv3 /
    downsample.0 forward, d input and d weight (and, with conv3 = "hip", the 3x3 conv2 forward and stride-1 d input as
    implicit GEMMs) on csrc/cgemm.hip with the BatchNorm statistics epilogue, the
    normalise-on-load prologue, the mask + reduction epilogue and the in-place residual-gradient accumulation --
    against the SAME module in fp64 on the CPU (plain torch conv / batch_norm / relu): output, d x, every parameter
    gradient, running statistics.  Also against the unfused GPU path (MIOpen convolutions + round-1 BN kernels).
    PARITY UNPINNED against the reference's torchvision (absent from the image): this pins the kernels to the
    public definition of the block."""
    from scnattn.resnet import Bottleneck, FusedBatchNorm2d
    from scnattn import conv as SC
    from torch import nn
    torch.manual_seed(1000 + [b[0] for b in _BLOCKS].index(name))
    if small:          # 2 images on a quarter-size map: ~30x fewer BatchNorm outputs, so a ReLU pre-activation within
        H = max(H // 4, 2 * stride)    # fp32 rounding of 0 is unlikely and the gradients can be held to 1e-4 (below)
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                             FusedBatchNorm2d(planes * 4))
    m = Bottleneck(inplanes, planes, stride, down)
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(mod, nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.8, 1.2)
    if not small:
        # Full size (round 3): with ~10^6 activations in front of each ReLU a handful lie within fp32
        # rounding of 0, and ONE flipped mask bit moves every row of the upstream weight gradients by ~1e-3 (through conv2's d
        # input it reaches all channels of nine pixels) -- the 1e-2 bar of round 2.  The BatchNorms in front of a ReLU get
        # beta += 3.5: the mask stays active (0.02 % of the activations are still cut, and they vary over pixels and
        # channels) but ~500x fewer activations sit within rounding of the threshold, so the fp32 bar (2e-4) holds at full size.
        with torch.no_grad():
            m.bn1.bias.add_(3.5)
            m.bn2.bias.add_(3.5)
            m.bn3.bias.add_(6.0)      # + identity of either sign (raw input / downsample BatchNorm output): further out
    m.train()
    N = 2 if small else 4
    x = torch.relu(torch.randn(N, inplanes, H, H)) + 0.1 * torch.randn(N, inplanes, H, H)
    ref = copy.deepcopy(m).double()
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    wgt = torch.randn_like(yr)
    (yr * wgt).sum().backward()
    res = {}
    for fused in (True, False):
        g = copy.deepcopy(m).to(dev).to(memory_format=torch.channels_last).train()
        xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        SC.ENABLED = fused
        # conv2: "hip" = the product path (implicit-GEMM forward, d input by flipped taps / parity classes, halo-staged or
        # gathered d weight: no library kernel in the block); "miopen" = the A/B mode with MIOpen's conv2
        saved_conv3 = SC.CONV3
        SC.CONV3 = conv3
        try:
            assert SC.usable(g, xg) == fused
            y = g(xg)
            (y * wgt.float().to(dev)).sum().backward()
            torch.cuda.synchronize()
        finally:
            SC.ENABLED = True
            SC.CONV3 = saved_conv3
        res[fused] = (y.detach(), xg.grad.detach(), {k: p.grad.detach() for k, p in g.named_parameters()},
                      {k: b.detach().clone() for k, b in g.named_buffers()})
    rep = []
    for fused in (True, False):
        y, dx, gr, bufs = res[fused]
        tag = "fused" if fused else "unfused"
        rep.append("%s %s: out %.2e  dx %.2e (l2 %.2e)" % (name, tag, rel_err(y, yr), rel_err(dx, xr.grad), rel_l2(dx, xr.grad)))
        for k, p in ref.named_parameters():
            rep.append("   %-24s %s grad l2 %.2e max %.2e" % (k, tag, rel_l2(gr[k], p.grad), rel_err(gr[k], p.grad)))
    _report(rep, "fused bottleneck " + name)
    y, dx, gr, bufs = res[True]
    _ok(y, yr, 2e-5, "output")
    # Gradients: of the ~10^6 BatchNorm outputs in front of a ReLU, a handful lie within fp32 rounding of 0; their mask
    # bit differs between two equally valid evaluations (fp64 here, fp32 there, MIOpen vs this kernel), and one flipped
    # element moves that channel's d beta / d gamma by O(1/sqrt(rows)) and, through the batch statistics, every row of
    # the channel a little (measured: either path shows 5e-4 .. 1e-3 in l2 against fp64 on some blocks, 1e-6 on others,
    # with the roles swapping between blocks).  Round 3: at full size the test makes the masks unambiguous (beta + 3.5
    # above) and holds every gradient to 2e-4; at the reduced size, where no ambiguous bit is expected anyway, 1e-4
    # (measured 4e-7).  The kernels themselves are held to 3e-6 with masks given (test_cgemm_variants_vs_fp64).
    def close(got, ref, what):
        if small:      # no ambiguous mask bit expected at this size (deterministic seeds and kernels): fp32 tolerance
            assert rel_l2(got, ref) <= 1e-4, "%s l2 %.3e" % (what, rel_l2(got, ref))
            return
        assert rel_l2(got, ref) <= 2e-4, "%s l2 %.3e" % (what, rel_l2(got, ref))
    close(dx, xr.grad, "d x")
    for k, p in ref.named_parameters():
        close(gr[k], p.grad, k)
    for k, b in ref.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(bufs[k], b) <= 2e-5, k
        elif k.endswith("num_batches_tracked"):
            assert int(bufs[k]) == int(b) == 1, k
    # and the fused path is no worse than the unfused one (MIOpen convolutions + separate BN passes)
    yu, dxu, gru, _ = res[False]
    assert rel_err(y, yr) <= max(1e-5, 3 * rel_err(yu, yr))
    assert rel_l2(dx, xr.grad) <= 1e-3 + 3 * rel_l2(dxu, xr.grad)


def test_cgemm_variants_vs_fp64(dev):
    """csrc/cgemm.hip through the C ABI (scnattn_cgemm): every layout (NT forward, NN dgrad, TN wgrad, TT), both row
    tiles, split-K, the BatchNorm prologues (A per k, B per n), the statistics epilogue, the mask + reduction pass,
    beta accumulation and the strided row gather, on ResNet-152 1x1 shapes and on an odd shape, against fp64 torch:
    products 3e-6 (1e-5 for the K = rows weight gradients), column statistics 2e-5."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cgemm_bench", os.path.join(root, "tools", "cgemm_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.check()
    mod.check3()      # the 3x3 implicit-GEMM modes (forward, d input, d weight) against fp64 torch conv2d


def test_dp_comm_through_the_c_abi_single_rank(dev):
    """include/scnattn.h scnattn_dp_comm_*: RCCL bound at run time, a communicator of one rank on the library's own
    communication stream, bucketed in-place SUM all-reduce ordered by events against the compute stream, and
    GradReducer driving it (backend="cabi") from post-accumulate hooks: gradients must come out unchanged (world 1),
    bucket after bucket, with the compute stream waiting only in finish().  A 1-GPU box cannot host two RCCL ranks
    (one device per rank); the N-rank arithmetic is covered by the gloo tests, the RCCL path with N ranks by the
    driver's scaling run."""
    import ctypes as C
    from scnattn import _lib
    from scnattn.dp import CabiComm, GradReducer
    from scnattn.flat import FlatBuffer
    comm = CabiComm.get(dev)
    assert comm.world == 1 and _lib.lib().scnattn_dp_comm_world(comm.handle) == 1
    t = torch.randn(1 << 20, device=dev)
    ref = t.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):         # a non-default compute stream: the event ordering must hold there too
        t.mul_(2.0)
        comm.allreduce(t)
        comm.allreduce(t[:1000])
        comm.finish()
        out = t + 0.0
    side.synchronize()
    assert torch.equal(out, ref * 2.0)
    # GradReducer with the C-ABI backend on a small model
    lin = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Linear(256, 8)).to(dev)
    flat = FlatBuffer(lin.parameters())
    red = GradReducer(flat, bucket_bytes=4096, backend="cabi")
    red.enabled = True
    x = torch.randn(16, 64, device=dev)
    red.reset()
    lin(x).square().sum().backward()
    scale = red.finish()
    flat.gather()
    torch.cuda.synchronize()
    assert scale == 1.0 and len(red.buckets) > 1
    lin2 = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Linear(256, 8)).to(dev)
    lin2.load_state_dict(lin.state_dict())
    lin2(x).square().sum().backward()
    for p, q in zip(lin.parameters(), lin2.parameters()):
        assert torch.equal(p.grad, q.grad)
    comm.close()


# ------------------------------------------------------------------------------------------------
# weight gradients of the decoder on the side stream (scnattn_seq_bwd_streams)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,ragged", [("attention_scn", True), ("attention_scn", False), ("pure_scn", True)])
def test_decoder_weight_gradients_on_the_side_stream_are_bit_identical(dev, kind, ragged):
    """include/scnattn.h scnattn_seq_bwd_streams: same kernels, same operands, two streams ordered by events -- every
    gradient must equal the single-stream call bit for bit.  The gradients are read on the main stream right after
    `backward()` WITHOUT a device-wide synchronize: that the numbers are right also shows that the stream is joined at
    the end of the autograd sweep (what the reference's clip_gradient / optimizer.step rely on).  A second backward into
    existing .grad tensors must fall back to one stream (autograd adds on the main stream at once)."""
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from oracle import scnattn_ref as R
    from scnattn import functional as SF
    torch.manual_seed(3)
    B, V, L = 32, 1000, 12
    m = (AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5) if kind == "attention_scn"
         else PureSCN(512, 512, 512, 1000, V, dropout=0.5)).to(dev).train()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 8, 8, 2048, generator=g).to(dev)
    tags = torch.rand(B, 1000, generator=g).to(dev)
    lens = torch.randint(5, L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    caps = _synthetic_caps(B, V, L, lens, g).to(dev)
    caplens = lens.unsqueeze(1).to(dev)
    T = int(lens.max()) - 1
    m.drop_mask_override = ((torch.rand(B, T, 512, generator=g) > 0.5).float() * 2.0).to(dev)

    def run(side, keep_grads=False):
        saved = SF.DECODER_SIDE_WGRAD
        SF.DECODER_SIDE_WGRAD = side
        try:
            if not keep_grads:
                for p in m.parameters():
                    p.grad = None
            x2 = x.clone().requires_grad_(True)
            out = m(None, tags, caps, caplens, prepool=x2, pool_size=14)
            alphas = out[3] if kind == "attention_scn" else None
            loss, _, _ = R.caption_loss(out[0], out[1], out[2], alphas, 1.0)
            loss.backward()
            return {k: p.grad.clone() for k, p in m.named_parameters()}, x2.grad.clone()     # main stream, no sync
        finally:
            SF.DECODER_SIDE_WGRAD = saved

    for _ in range(2):           # second round: allocator blocks of the first are being reused
        one, dx1 = run(False)
        two, dx2 = run(True)
        assert torch.equal(dx1, dx2)
        for k in one:
            assert torch.equal(one[k], two[k]), k
    acc, _ = run(True, keep_grads=True)      # accumulates into the gradients of the last run
    for k in one:
        _ok(acc[k], 2.0 * two[k], 1e-6, k)


# ------------------------------------------------------------------------------------------------
# bf16 storage mode of the decode step (BASELINE configs[4] flavour)
# ------------------------------------------------------------------------------------------------
def test_f32_to_bf16_and_bf16_weight_skinny_gemm(dev):
    """include/scnattn.h scnattn_f32_to_bf16 (round to nearest even, bit-exact against torch's conversion) and
    scnattn_skinny_gemm_bf16w (weights stored as bf16, fp32 MFMA and accumulation): against fp64 on the SAME rounded
    weights the product is as exact as the fp32 one (1e-5)."""
    import ctypes as C
    from scnattn._lib import call, ptr, stream_of
    torch.manual_seed(0)
    for rows, N, K, groups in ((32, 4608, 512, 1), (32, 2048, 2048, 1), (32, 512, 1024, 4), (7, 96, 200, 1)):
        X = torch.randn(rows, groups * K, device=dev)
        W = torch.randn(groups, K, N, device=dev)
        special = torch.tensor([0.0, -0.0, 1.0, 1.00390625, 1.005859375, 3.0e38, -3.0e38, 1e-40, float("inf"), float("nan")],
                               device=dev)
        W.view(-1)[:special.numel()] = special if (rows, N) == (7, 96) else W.view(-1)[:special.numel()]
        Wh = torch.empty(W.shape, device=dev, dtype=torch.bfloat16)
        call("scnattn_f32_to_bf16", stream_of(W), W.numel(), ptr(W), ptr(Wh))
        ref_h = W.to(torch.bfloat16)
        assert torch.equal(Wh.view(torch.int16), ref_h.view(torch.int16)), "f32 -> bf16 is not round-to-nearest-even"
        if (rows, N) == (7, 96):
            Wh = torch.randn(groups, K, N, device=dev).to(torch.bfloat16)
        Y = torch.empty(16, groups, rows, N, device=dev)
        used = C.c_int(0)
        call("scnattn_skinny_gemm_bf16w", stream_of(X), rows, N, K, groups, ptr(X), groups * K, K, ptr(Wh), N, K * N, ptr(Y),
             N, rows * N, groups * rows * N, 0, C.byref(used))
        got = Y[:used.value].double().sum(0)                                     # (groups, rows, N)
        want = torch.einsum("rgk,gkn->grn", X.view(rows, groups, K).double(), Wh.double())
        _ok(got, want, 1e-5, "skinny bf16w %s" % ((rows, N, K, groups),))
        # both operands bf16 on v_mfma_f32_32x32x16_bf16: against fp64 on the SAME rounded operands (the kernel rounds X
        # to nearest even in registers, as torch's conversion does); bf16 x bf16 products are exact in fp32
        Y.zero_()
        call("scnattn_skinny_gemm_bf16", stream_of(X), rows, N, K, groups, ptr(X), groups * K, K, ptr(Wh), N, K * N, ptr(Y),
             N, rows * N, groups * rows * N, 0, C.byref(used))
        got = Y[:used.value].double().sum(0)
        want = torch.einsum("rgk,gkn->grn", X.to(torch.bfloat16).view(rows, groups, K).double(), Wh.double())
        _ok(got, want, 1e-5, "skinny bf16 mfma %s" % ((rows, N, K, groups),))


@pytest.mark.parametrize("mode", [1, 2], ids=["storage", "mfma"])
@pytest.mark.parametrize("kind,ragged,pooled", [("attention_scn", True, True), ("attention_scn", False, True),
                                                ("attention_scn", True, False), ("pure_scn", True, True)])
def test_bf16_storage_decoder_vs_oracle(dev, kind, ragged, pooled, mode):
    """Option "decoder_bf16": the operands the recurrence streams (recurrent weights, att1, the trunk map) are bf16
    copies, everything else fp32.  Against the fp64 oracle at full width: a bf16 element carries 8 significant bits
    (relative rounding 2^-9 = 2e-3); outputs and gradients are held to 5e-3 (measured 1e-3 / 2-3e-3: see
    profiles/r02_parity_report.txt) -- THE bf16 tolerance of this repository -- with one documented exception below.
    Also: switching the option off again must give the fp32 numbers back (no state left behind)."""
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from models.decoders import _common as DC
    from scnattn import functional as SF
    torch.manual_seed(7)
    B, V, L = 32, 1000, 14
    m = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5) if kind == "attention_scn" \
        else PureSCN(512, 512, 512, 1000, V, dropout=0.5)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 8, 8, 2048, generator=g)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.randint(5, L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    caps = _synthetic_caps(B, V, L, lens, g)
    caplens = lens.unsqueeze(1)
    T = int(lens.max()) - 1
    mask = (torch.rand(B, T, 512, generator=g) > 0.5).float() * 2.0
    si = torch.sort(lens, descending=True, stable=True)[1]
    sd = m.state_dict()
    r64 = _oracle_run(kind, sd, x, tags, caps, caplens, mask, si, torch.float64)
    saved_pool = DC.USE_PREPOOL
    try:
        DC.USE_PREPOOL = pooled
        if pooled:
            run = lambda: _hip_run(kind, m, x, tags, caps, caplens, mask, si, dev)
        else:      # dense path: the HIP decoder gets the materialised 14x14 map, as the oracle does
            def run():
                import torch.nn.functional as F
                from oracle import scnattn_ref as R
                mm = m.to(dev).train()
                mm.drop_mask_override = mask.to(dev)
                for p in mm.parameters():
                    p.grad = None
                x2 = x.to(dev).requires_grad_(True)
                enc = F.adaptive_avg_pool2d(x2.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1)
                out = mm(enc, tags.to(dev), caps.to(dev), caplens.to(dev), sort_ind=si.to(dev))
                alphas = out[3] if kind == "attention_scn" else None
                loss, _, _ = R.caption_loss(out[0], out[1], out[2], alphas, 1.0)
                loss.backward()
                return out[0], alphas, loss, x2.grad, mm
        # mode 2 ("mfma"): the activation rows of the per-step products (h, z, the mixed factors and their gradients) are
        # rounded to bf16 as well and multiplied on the bf16 matrix instruction -- both operands bf16, the usual mixed-
        # precision contract; same stated tolerance, 1.5x the allowance on the mask-downstream tensors
        SF.set_option("decoder_bf16", mode)
        scale = 1.0 if mode == 1 else 1.5       # measured: the same 1.5e-3 / 2-3e-3 as the storage mode, 5.8e-2 on d decoder_att.weight
        for p in m.parameters():
            p.grad = None
        hip = run()
        # measured: 2-3e-3 on every gradient (the bf16 level) except the four tensors downstream of the attention's ReLU
        # mask: rounding att1 to bf16 moves it by up to 2^-9 of its size, which flips the mask bit of every
        # pre-activation closer to zero than that -- far more of them than the fp32 rounding flips (2e-3 there) -- and
        # d decoder_att.weight is the most cancellation-heavy of the four.  The test below holds those four to the
        # 5e-3 of the others with an unambiguous mask.
        floors = {k: 6e-2 * scale for k in ("attention.encoder_att.weight", "attention.encoder_att.bias",
                                            "attention.decoder_att.weight", "attention.decoder_att.bias")}
        floors["attention.full_att.weight"] = 2e-2 * scale      # sum of de * relu(att1 + att2): the same mask, and att1 itself
        _compare(kind, m, hip, r64, floors, "bf16 decode step (%s), %s ragged=%s pooled=%s"
                 % ("storage" if mode == 1 else "bf16 MFMA", kind, ragged, pooled), tol_out=5e-3, tol_grad=5e-3)
        bf_preds = hip[0].detach().clone()
        SF.set_option("decoder_bf16", 0)
        for p in m.parameters():
            p.grad = None
        hip32 = run()
        _ok(hip32[0], r64[0], TOL_OUT, "fp32 preds after switching the option off")
        assert not torch.equal(bf_preds, hip32[0].detach()), "the bf16 mode did not change a single bit: not engaged?"
    finally:
        SF.set_option("decoder_bf16", 0)
        DC.USE_PREPOOL = saved_pool


def test_bf16_storage_attention_gradients_with_unambiguous_relu_mask(dev):
    """The bf16 mode with the attention rigged by _make_unambiguous: every ReLU pre-activation is further from zero
    than the bf16 rounding of att1 can move it (margin asserted > 0.1 on the fp64 side against |att1| of order 10), so
    the mask is the same bit pattern as in fp64.  The construction makes att1 an order of magnitude larger than trained
    or randomly initialised weights do, and a bf16 element's ABSOLUTE rounding error grows with it (2^-9 * 16 = 0.03
    per term of a score), so this test states its own bound, 5e-2 for outputs and gradients alike, no floors: what it
    shows is that the tensors downstream of the mask are then no worse than everything else."""
    from models.decoders.attention_scn import AttentionSCN
    from scnattn import functional as SF
    torch.manual_seed(21)
    B, V, L = 32, 1000, 14
    m = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = _make_unambiguous(sd, torch.rand(B, 8, 8, 2048, generator=g), g)
    m.load_state_dict(sd)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.randint(5, L + 1, (B,), generator=g)
    caps = _synthetic_caps(B, V, L, lens, g)
    caplens = lens.unsqueeze(1)
    T = int(lens.max()) - 1
    mask = (torch.rand(B, T, 512, generator=g) > 0.5).float() * 2.0
    si = torch.sort(lens, descending=True, stable=True)[1]
    r64 = _oracle_run("attention_scn", sd, x, tags, caps, caplens, mask, si, torch.float64, probe=True)
    assert r64[4] is not None and r64[4] > 0.1, "ReLU margin %.3e" % r64[4]
    try:
        SF.set_option("decoder_bf16", 1)
        hip = _hip_run("attention_scn", m, x, tags, caps, caplens, mask, si, dev)
        _compare("attention_scn", m, hip, r64, None, "bf16-storage, mask-unambiguous (margin %.3f)" % r64[4],
                 tol_out=5e-2, tol_grad=5e-2)
    finally:
        SF.set_option("decoder_bf16", 0)


# ------------------------------------------------------------------------------------------------
# E1: gradients of the WHOLE ResNet-152 trunk (fused Bottleneck path) against fp64
# ------------------------------------------------------------------------------------------------
def test_encoder_gradients_are_as_close_to_fp64_as_cpu_fp32_is(dev):
    """EncoderCaption forward + backward through all 50 fused Bottleneck blocks on the GPU, against the same definition
    run in fp64 on the CPU.  A randomly initialised 152-layer trunk in training mode (batch statistics) is badly
    conditioned: torch's OWN fp32 CPU gradients are ~1e-1 from fp64 (ReLU-mask flips compounded through ~150
    BatchNorm layers; measured 9.8e-2 over all fine-tuned parameters).  So fp64 is the anchor and CPU fp32 the
    yardstick: over all fine-tuned parameters the GPU's distance to fp64 must not exceed the CPU fp32's by more than 25 %
    (measured 9.7e-2 vs 9.8e-2), per stage (layer2, layer3, layer4: medians of 75 / 327 / 30 tensors) by more than 50 %.
    Both numbers are one draw each of a chaotic amplification of last-bit differences -- the per-stage ratio moved between
    1.13 and 1.29 over three builds of this round whose kernels differ only in summation order -- so this test can only
    show that NOTHING GROSS is lost in composition.  What pins the trunk's precision are the tests that remove the
    chaos: tests/test_gpu_parity_r3.py::test_well_conditioned_trunk_gradients_vs_fp64 (mask-unambiguous construction, all
    432 gradients <= 1e-3, median 3e-5) and the block-level test_fused_bottleneck_vs_fp64 (2e-4 at full size).
    Parity against the reference's torchvision stays unpinned."""
    import copy
    import statistics
    from models.encoders.caption import EncoderCaption
    from oracle import scnattn_ref as R
    torch.manual_seed(0)
    enc = EncoderCaption(channels_last=True)
    enc.fine_tune(True)
    enc.train()
    x = torch.randn(4, 3, 160, 160)

    def run_cpu(dt):
        m = copy.deepcopy(enc).to(dt)
        y = R.pool_permute(m.resnet(x.to(dt)), 14)
        torch.manual_seed(1)
        w = torch.randn(y.shape).to(dt)
        (y * w).sum().backward()
        return y.detach(), {k: p.grad.detach() for k, p in m.named_parameters() if p.grad is not None}

    y64, g64 = run_cpu(torch.float64)
    y32, g32 = run_cpu(torch.float32)
    g = copy.deepcopy(enc).to(dev)
    yg = g(x.to(dev))
    torch.manual_seed(1)
    w = torch.randn(y64.shape)
    (yg * w.to(dev)).sum().backward()
    gg = {k: p.grad.detach().cpu() for k, p in g.named_parameters() if p.grad is not None}
    assert set(gg) == set(g64)

    def rl2(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()

    ey_gpu, ey_cpu = rl2(yg.cpu(), y64), rl2(y32, y64)
    assert ey_gpu <= max(5e-3, 1.5 * ey_cpu), (ey_gpu, ey_cpu)
    eg = {k: rl2(gg[k], g64[k]) for k in g64}
    ec = {k: rl2(g32[k], g64[k]) for k in g64}
    lines = ["encoder_out: gpu %.3e cpu-fp32 %.3e (vs fp64)" % (ey_gpu, ey_cpu)]
    for stage in ("resnet.5", "resnet.6", "resnet.7"):
        ks = [k for k in g64 if k.startswith(stage)]
        mg, mc = statistics.median(eg[k] for k in ks), statistics.median(ec[k] for k in ks)
        lines.append("%s (%d tensors): median rel-l2 to fp64  gpu %.3e  cpu-fp32 %.3e" % (stage, len(ks), mg, mc))
        assert mg <= 1.5 * mc + 1e-3, (stage, mg, mc)
    cat = lambda d: torch.cat([d[k].flatten().double() for k in g64])
    a64 = cat(g64)
    tg, tc = ((cat(gg) - a64).norm() / a64.norm()).item(), ((cat(g32) - a64).norm() / a64.norm()).item()
    lines.append("all fine-tuned parameters: gpu %.3e  cpu-fp32 %.3e" % (tg, tc))
    _report(lines, "whole ResNet-152 trunk, gradients vs fp64 (CPU fp32 as the yardstick)")
    assert tg <= 1.25 * tc + 1e-3, (tg, tc)


# ------------------------------------------------------------------------------------------------
# L1/L3: the reference's literal train-loop body with STOCK torch.optim.Adam on the drop-in modules
# ------------------------------------------------------------------------------------------------
def test_reference_loop_body_with_stock_adam(dev):
    """trains/attention_scn.py:213-252 verbatim on the drop-in modules -- encoder(imgs), decoder(encoder_out, ...),
    pack_padded_sequence + CrossEntropyLoss + the alpha regulariser, zero_grad, backward, clip_gradient, step of two
    STOCK torch.optim.Adam optimizers, two steps.
    (1) Run twice, with the weight gradients on the side stream (default) and with everything on the main stream:
        identical kernels on identical operands, so every gradient (read on the main stream right after backward(), no
        synchronize) and every parameter after two steps must agree -- bit for bit in the decoder, to the 1e-7-level
        noise of MIOpen's atomic kernels in the trunk.  A gradient read before the side stream finished, or written to
        after autograd stored it, would be an O(1) error.
    (2) Against the harness path (prepool hand-over, fused loss, FusedClampAdam over flat buffers): same second-step
        loss to 2e-3 and parameters within the 2 * steps * lr an Adam trajectory can differ by (Adam normalises the
        step, so near-zero gradients of either sign move a weight by +-lr; the randomly initialised trunk is badly
        conditioned, see test_encoder_gradients_are_as_close_to_fp64_as_cpu_fp32_is)."""
    import copy
    from torch.nn.utils.rnn import pack_padded_sequence
    from models.encoders.caption import EncoderCaption
    from models.decoders.attention_scn import AttentionSCN
    from utils.optimizer import clip_gradient, FusedClampAdam
    from scnattn import functional as SF
    from scnattn import conv as SC
    torch.manual_seed(5)
    B, V, L, S = 6, 200, 10, 64
    enc0 = EncoderCaption(channels_last=True).to(dev)
    enc0.fine_tune(True)
    dec0 = AttentionSCN(64, 64, 64, 64, S, V, dropout=0.5).to(dev)
    g = torch.Generator().manual_seed(9)
    imgs = torch.randn(B, 3, 96, 96, generator=g).to(dev)
    tags = torch.rand(B, S, generator=g).to(dev)
    lens = torch.randint(4, L + 1, (B,), generator=g)
    caps = _synthetic_caps(B, V, L, lens, g).to(dev)
    caplens = lens.unsqueeze(1).to(dev)
    T = int(lens.max()) - 1
    masks = [((torch.rand(B, T, 64, generator=g) > 0.5).float() * 2.0).to(dev) for _ in range(2)]
    criterion = torch.nn.CrossEntropyLoss().to(dev)

    def reference_loop(side):
        saved, saved3 = SC.SIDE_WGRAD, SC.CONV3
        SC.SIDE_WGRAD = side
        SC.CONV3 = "hip"          # stride-1 3x3 forward / d-input on this repository's (deterministic) kernel
        try:
            enc, dec = copy.deepcopy(enc0).train(), copy.deepcopy(dec0).train()
            dec_opt = torch.optim.Adam(params=filter(lambda p: p.requires_grad, dec.parameters()), lr=4e-4)
            enc_opt = torch.optim.Adam(params=filter(lambda p: p.requires_grad, enc.parameters()), lr=1e-4)
            first = None
            for step in range(2):
                dec.drop_mask_override = masks[step]
                encoder_out = enc(imgs)
                scores, caps_sorted, decode_lengths, alphas, sort_ind = dec(encoder_out, tags, caps, caplens)
                targets = caps_sorted[:, 1:]
                scores = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
                targets = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
                loss = criterion(scores, targets)
                loss = loss + 1.0 * ((1. - alphas.sum(dim=1)) ** 2).mean()
                dec_opt.zero_grad()
                enc_opt.zero_grad()
                loss.backward()
                if first is None:        # main stream, right after backward(), no synchronize
                    first = {k: p.grad.clone() for k, p in list(dec.named_parameters()) + list(enc.named_parameters())
                             if p.grad is not None}
                clip_gradient(dec_opt, 5.)
                clip_gradient(enc_opt, 5.)
                dec_opt.step()
                enc_opt.step()
            params = {k: p.detach().clone() for k, p in list(dec.named_parameters()) + list(enc.named_parameters())}
            return first, params, loss.detach()
        finally:
            SC.SIDE_WGRAD, SC.CONV3 = saved, saved3

    g_side, p_side, loss_side = reference_loop(True)
    g_main, p_main, loss_main = reference_loop(False)
    g_again, _, _ = reference_loop(False)      # the library's own run-to-run noise, with no second stream anywhere
    assert len(g_side) == len(g_main) and len(g_side) > 300
    # Not torch.equal: MIOpen's kernels (3x3 weight gradients, strided 3x3 d-input, and whichever forward solver its
    # find step picked in this process) sum with atomics, so two runs differ in the last bits (1e-7 .. 1e-5 relative)
    # wherever such a kernel is upstream.  An incomplete or overwritten gradient is an O(1) error.
    worst_g = worst_noise = 0.0
    for k in g_main:
        e = rel_err(g_side[k], g_main[k]) if float(g_main[k].abs().max()) > 0 else 0.0
        worst_g = max(worst_g, e)
        assert e <= 1e-4, "gradient of %s differs between side-stream and main-stream runs: %.3e" % (k, e)
        noise = rel_err(g_again[k], g_main[k]) if float(g_main[k].abs().max()) > 0 else 0.0
        worst_noise = max(worst_noise, noise)
    # Which tensors may differ at all?  Library kernels that sum with atomics give a different last bit whenever the
    # order in which their workgroups run changes, and a second stream changes it: MIOpen's 3x3 weight gradients
    # (conv2.weight), and its strided 3x3 d-input in layer4.0 / layer3.0 / layer2.0 -- everything upstream of those in
    # the backward sweep inherits the noise.  What this repository's kernels alone produce must be bit-identical
    # whichever stream ran it: the whole decoder and the last two blocks of layer4 (backward reaches them first).
    # Round 3: no library (atomic) kernel is left in the step -- 3x3 weight gradients, strided 3x3 d input and the stem are
    # this repository's fixed-order kernels now -- so EVERY gradient must be bit-identical whichever stream produced it.
    differing = sorted(k for k in g_main if not torch.equal(g_side[k], g_main[k]))
    strict = list(g_main)
    assert not differing, "differs between side-stream and main-stream runs: %s" % differing[:6]
    for k in p_main:      # Adam normalises its step: elements whose gradient is at noise level move by +-lr in either run
        err = (p_side[k] - p_main[k]).abs().max().item()
        assert err <= 2 * 2 * 4e-4 * 1.01, "parameter %s differs after two steps: %.3e" % (k, err)
    _ok(loss_side, loss_main, 1e-4, "second-step loss of the two runs")
    # ---- the harness path -------------------------------------------------------------------------------------------
    enc_b, dec_b = copy.deepcopy(enc0).train(), copy.deepcopy(dec0).train()
    dec_fo = FusedClampAdam(filter(lambda p: p.requires_grad, dec_b.parameters()), lr=4e-4, grad_clip=5.)
    enc_fo = FusedClampAdam(filter(lambda p: p.requires_grad, enc_b.parameters()), lr=1e-4, grad_clip=5.)
    worst_g1 = 0.0
    for step in range(2):
        dec_b.drop_mask_override = masks[step]
        prepool = enc_b(imgs, pooled=False)
        scores, caps_sorted, decode_lengths, alphas, sort_ind = dec_b(None, tags, caps, caplens, prepool=prepool,
                                                                        pool_size=enc_b.enc_image_size)
        dl_dev = (caplens.reshape(-1)[sort_ind] - 1).to(torch.int32)
        loss_b = SF.caption_loss(scores, caps_sorted, decode_lengths, alphas, 1.0, dl_dev)
        dec_fo.zero_grad()
        enc_fo.zero_grad()
        loss_b.backward()
        if step == 0:
            # post-Adam parameters within 2*steps*lr hold whatever the gradient is; the FIRST-STEP GRADIENTS of
            # the harness path (flat views filled in place, fused loss, pooled hand-over) against those of the stock
            # loop are the real check: rel-l2 1e-3 per tensor (different but equivalent formulations: pooled vs dense
            # attention path, fused vs packed loss; measured <= 1e-4).
            dec_fo.flat.gather(); enc_fo.flat.gather()
            for k, pb in list(dec_b.named_parameters()) + list(enc_b.named_parameters()):
                if k not in g_side:
                    continue
                if k.endswith("full_att.bias"):       # exactly 0 in exact arithmetic (softmax shift invariance)
                    continue
                e1 = rel_l2(pb.grad, g_side[k])
                worst_g1 = max(worst_g1, e1)
                assert e1 <= 1e-3, "first-step gradient of %s: harness path vs stock loop rel-l2 %.3e" % (k, e1)
        dec_fo.step()
        enc_fo.step()
    _ok(loss_b, loss_side, 2e-3, "second-step loss, harness path vs reference loop body")
    worst = 0.0
    for k, pb in list(dec_b.named_parameters()) + list(enc_b.named_parameters()):
        err = (p_side[k] - pb.detach()).abs().max().item()
        worst = max(worst, err)
        lr = 4e-4 if any(k == kd for kd, _ in dec_b.named_parameters()) else 1e-4
        assert err <= 2 * 2 * lr * 1.01, "%s: abs err %.3e after two steps" % (k, err)
    _report(["side-stream vs main-stream runs of the reference loop body: %d gradient tensors, worst rel err %.3e "
             "(two main-stream-only runs: %.3e); %d tensors differ at all, none of the %d that have no library (atomic) "
             "kernel upstream" % (len(g_main), worst_g, worst_noise, len(differing), len(strict)), "reference loop body vs harness path: second-step loss rel err %.3e, max abs parameter "
             "difference %.3e; first-step gradients harness vs stock loop: worst rel-l2 %.3e" % (rel_err(loss_b, loss_side), worst, worst_g1)],
            "reference loop body with stock torch.optim.Adam")


def test_tagger_beside_the_encoder_gives_the_same_step(dev):
    """trains/harness.py: with a tagger in the step (trains/attention_scn.py:194,214) its forward pass runs on the side
    stream beside the caption encoder's (per-stream scratch buffers, event-ordered hand-over of the tags).  Same
    weights, same batch, against the in-line order: bit-identical since round 3 (see below) -- a tag tensor read before
    it was written, or scratch shared between the two streams, would be an O(1) error in the loss."""
    from trains.harness import TrainStep, synthetic_batch
    res = {}
    for overlap in (False, False, True, True):
        ts = TrainStep(device=dev, tagger=True, tagger_overlap=overlap, seed=77, batch_size=4, max_len=8, vocab_size=300,
                       image_size=96)
        cfg = ts.cfg
        imgs, tags, caps, caplens = synthetic_batch(4, cfg["vocab_size"], cfg["max_len"], cfg["image_size"],
                                                    cfg["semantic_dim"], torch.device(dev), 3, ragged=True)
        torch.manual_seed(11)          # dropout masks of the tagger / decoder
        losses = [ts.step(imgs, tags, caps, caplens).detach().clone() for _ in range(2)]
        key = "overlap" if overlap else "inline"
        cur = (losses, {k: p.detach().clone() for k, p in ts.decoder.named_parameters()},
               {k: b.detach().clone() for k, b in ts.tagger.named_buffers() if k.endswith("running_mean")})
        if key in res:       # second run of a mode: the side stream is warm, buffers are being reused
            key = key + "2"
        res[key] = cur
    rep = ["%s: losses %s" % (k, ["%.6f" % l.item() for l in v[0]]) for k, v in res.items()]
    _report(rep, "tagger beside the encoder: raw losses")
    # round 3: no library kernel is left in either trunk, so two runs of the SAME mode are bit-identical
    for a_, b_ in zip(res["inline"][0], res["inline2"][0]):
        assert torch.equal(a_, b_), "two in-line runs differ: %r vs %r" % (a_.item(), b_.item())
    rep = []
    # ... and, every kernel of both trunks being this repository's (fixed summation orders, per-stream scratch), the
    # overlapped runs reproduce the in-line ones BIT FOR BIT: losses of both steps, every decoder parameter after two Adam
    # steps, the tagger's running statistics.  (Rounds 1-2 could only ask for 5e-3: MIOpen's per-stream handles picked
    # different solvers for the two streams.)
    for key in ("overlap", "overlap2"):
        for i, (a_, b_) in enumerate(zip(res["inline"][0], res[key][0])):
            rep.append("%s: loss of step %d  %.6f vs in-line %.6f" % (key, i + 1, b_.item(), a_.item()))
            assert torch.equal(a_, b_), "loss of step %d (%s): %r vs in-line %r" % (i + 1, key, b_.item(), a_.item())
        for k in res["inline"][1]:
            assert torch.equal(res["inline"][1][k], res[key][1][k]), "%s (%s) differs from the in-line run" % (k, key)
        for k in res["inline"][2]:
            assert torch.equal(res[key][2][k], res["inline"][2][k]), "tagger statistics %s differ (%s)" % (k, key)
    _report(rep, "tagger forward on the side stream beside the caption encoder vs in line")
