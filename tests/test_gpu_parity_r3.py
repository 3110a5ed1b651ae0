"""GPU parity tests added in round 3 (``-m gpu``), all through the C ABI.

What they close (VERDICT r02 "What's weak" 1a-1e):
  * the kernels that replaced the last library convolutions -- halo-staged 3x3 weight gradient, stride-2 3x3 d input by
    parity classes, the 7x7 stem + BatchNorm/ReLU/max-pool -- and the finalize-on-load BatchNorm kernels against fp64
    (tools/cgemm_bench.py check3 / checkstem / checkbn), at every ResNet-152 shape class and at ragged / odd sizes;
  * the flat-gradient alias contract on the FusedClampAdam path with the weight gradients on and off the side stream
    (``p.grad`` IS the flat view after backward; the flat gradient buffers of the two runs are bit-equal);
  * a WELL-CONDITIONED whole-trunk gradient test (residual branches scaled down) that holds layer2-4 to a fixed 1e-3
    instead of "no worse than CPU fp32";
  * BASELINE configs[1] (pure_scn) once at its exact sizes B=32, T=51, V=10 000 against the fp64 oracle;
  * EncoderTagger.forward at the train step's real size 32 x 3 x 256 x 256.
"""
import copy
import importlib.util
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from scnattn import _lib
    _lib.lib()  # must load: there is no fallback
    return torch.device("cuda:0")


def _report(lines, title):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report_r03.txt"), "a") as f:
        f.write("== %s\n" % title)
        for ln in lines:
            f.write(ln + "\n")


def _tool():
    spec = importlib.util.spec_from_file_location("cgemm_bench", os.path.join(ROOT, "tools", "cgemm_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stem_and_finalize_on_load_kernels_vs_fp64(dev):
    """csrc/stem.hip (7x7 / 2 convolution of 3-channel images in NCHW and channels-last, odd sizes, its statistics
    partials, BatchNorm + ReLU + 3x3 / 2 max-pool in one pass) and the finalize-on-load BatchNorm kernels of
    csrc/batchnorm.hip (apply with the statistics summed from channel-major partials inside, backward reduce, backward dx
    with d beta / d gamma summed inside) against fp64 torch: 3e-6 on products, 1e-5 .. 2e-5 on statistics and BatchNorm
    outputs; a stand-alone finalize and the finalize inside `apply` must give the SAME BITS.  (check3 -- the halo-staged
    3x3 weight gradient incl. ragged strips and every K split, the stride-2 d input -- runs in
    test_gpu_parity_r2.py::test_cgemm_variants_vs_fp64.)"""
    mod = _tool()
    mod.checkstem()
    mod.checkbn()


def test_in_launch_split_k_combine_is_bit_identical_under_uneven_load(dev):
    """csrc/cgemm.hip finishes a split product inside its launch: the workgroup that draws the last ticket of a tile sums the
    slabs in slab order and runs the epilogue (plain, beta = 1, statistics, mask).  Against the two-launch protocol
    (option cgemm_combine = 0) the products must be BIT-identical and the statistics partials equal to 2e-5, for both
    publish forms (write-through slabs; plain slabs + agent release), five shapes x splits 2 / 3 / 4 / 8 x six repetitions,
    while a second stream keeps the chip unevenly busy and the last arriver's caches warm (tools/cgemm_bench.py comb;
    MI355X_MICROARCH.md: "test every hand-off under uneven load, consumer L1-warm, checking every word")."""
    _tool().comb(timing=False)


def test_stem_module_path_matches_torch_ops(dev):
    """scnattn/stem.py through EncoderCaption's trunk: the fused stem (training mode: batch statistics + running-stat
    update; eval mode: running statistics) against the same four nn modules run by torch on the CPU in fp64."""
    from scnattn.resnet import resnet152_trunk
    from scnattn import stem as ST
    torch.manual_seed(7)
    trunk = resnet152_trunk(depths=(1, 1, 1, 1))
    with torch.no_grad():
        trunk[1].weight.uniform_(0.5, 1.5); trunk[1].bias.normal_(0, 0.2)
        trunk[1].running_mean.normal_(0, 0.1); trunk[1].running_var.uniform_(0.8, 1.2)
    for p in trunk.parameters():
        p.requires_grad = False
    x = torch.randn(6, 3, 96, 80)
    for train in (True, False):
        ref = copy.deepcopy(trunk).double().train(train)
        g = copy.deepcopy(trunk).to(dev).to(memory_format=torch.channels_last).train(train)
        xr = x.double()
        for i in range(4):
            xr = ref[i](xr)
        xg = x.to(dev)
        assert ST.usable(g, xg)
        y = ST.stem(g, xg)
        assert y.shape == xr.shape and y.is_contiguous(memory_format=torch.channels_last)
        e = rel_err(y, xr)
        assert e <= 2e-5, "stem (train=%s) %.3e" % (train, e)
        assert rel_err(g[1].running_mean, ref[1].running_mean) <= 2e-5 and rel_err(g[1].running_var, ref[1].running_var) <= 2e-5
        assert int(g[1].num_batches_tracked) == int(ref[1].num_batches_tracked)
    # a stem that must produce gradients is NOT taken by the forward-only kernels
    trunk[0].weight.requires_grad = True
    assert not ST.usable(trunk.to(dev), x.to(dev))


def test_flat_gradient_views_with_and_without_the_side_stream(dev):
    """ADVICE r02 (conv.py:285): on the FusedClampAdam path every fused-Bottleneck weight gradient is written straight
    into the parameter's slice of the flat gradient buffer.  After backward() `p.grad` must BE that slice (autograd stole
    the alias instead of cloning it on the main stream while the side stream was still writing), and the flat gradient
    buffer must be bit-equal whether the weight gradients ran on the side stream or in line: every kernel in the block
    is this repository's and sums in a fixed order, so there is no tolerance to grant."""
    from models.encoders.caption import EncoderCaption
    from scnattn.resnet import resnet152_trunk
    from scnattn import conv as SC
    from utils.optimizer import FusedClampAdam
    torch.manual_seed(11)
    enc0 = EncoderCaption(channels_last=True)
    enc0.resnet = resnet152_trunk(depths=(1, 2, 2, 1))
    enc0 = enc0.to(dev).train()
    enc0.fine_tune(True)
    x = torch.randn(8, 3, 128, 128, device=dev)
    w = torch.randn(8, 4, 4, 2048, device=dev)
    flats = {}
    saved = SC.SIDE_WGRAD
    try:
        for side in (True, False):
            SC.SIDE_WGRAD = side
            enc = copy.deepcopy(enc0)
            opt = FusedClampAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-4, grad_clip=5.0)
            for rep in range(2):          # second round: the allocator reuses blocks, the side stream is warm
                opt.zero_grad()
                (enc(x, pooled=False) * w).sum().backward()
                n_conv = 0
                for p, gv in zip(opt.flat.params, opt.flat.gviews):
                    assert p.grad is not None
                    if p.dim() == 4:       # convolution weights: produced in place by the wgrad kernels
                        n_conv += 1
                        assert p.grad.data_ptr() == gv.data_ptr(), "autograd cloned a flat gradient view"
                assert n_conv >= 18
                opt.flat.gather()
            torch.cuda.synchronize()
            flats[side] = opt.flat.flat_g.clone()
    finally:
        SC.SIDE_WGRAD = saved
    assert torch.isfinite(flats[True]).all() and float(flats[True].abs().max()) > 0
    assert torch.equal(flats[True], flats[False]), \
        "flat gradients differ between side-stream and in-line weight gradients: max abs %.3e" % \
        (flats[True] - flats[False]).abs().max().item()


def test_well_conditioned_trunk_gradients_vs_fp64(dev):
    """the whole ResNet-152 trunk (all 50 fused Bottlenecks, training-mode BatchNorm, the stem on
    csrc/stem.hip) forward + backward against the same definition in fp64 on the CPU, made WELL-CONDITIONED so that a
    fixed bar can be held.  Two things make the randomly initialised trunk hard: (1) perturbations are amplified
    through 150 layers -- the last BatchNorm of every block gets gamma = 0.2, so the residual branch is a small
    correction; (2) ReLU-mask ambiguity: an activation within fp32 rounding of 0 has its gradient passed by one
    evaluation and blocked by the other, which moves a weight-gradient row by O(1/sqrt(rows)); with ~10^-6 of ~10^6
    activations per ReLU ambiguous that is ~1e-3 per tensor and block whatever the kernels do (measured with beta = 0:
    median 7e-3 after 36 blocks of layer3) -- every BatchNorm in front of a ReLU gets beta = 3.5, which leaves the mask
    active (0.02 % of the activations are still cut) but puts ~500x fewer of them within rounding of the threshold
    (the construction test_gpu_parity_r2.py uses for the attention's ReLU).  Output 1e-4; every fine-tuned parameter's
    gradient within rel-l2 1e-3 of fp64 at the median tensor and 3e-3 at the worst, and 1e-3 over all of layer2-4 taken
    as one vector.  (The randomly initialised, badly conditioned variant stays in
    test_gpu_parity_r2.py::test_encoder_gradients_are_as_close_to_fp64_as_cpu_fp32_is.)  Parity UNPINNED against the
    reference's torchvision (absent from the image): this pins the kernels' composition to the public definition."""
    import statistics
    from models.encoders.caption import EncoderCaption
    torch.manual_seed(21)
    enc = EncoderCaption(channels_last=True)
    enc.fine_tune(True)
    enc.train()
    with torch.no_grad():
        for name, mod in enc.resnet.named_modules():
            if name.endswith("bn3"):
                mod.weight.fill_(0.2)
                mod.bias.fill_(0.5)          # + identity >= 0: the block's last ReLU is cut only where |xhat| > 2.5 below
            elif name.endswith(("bn1", "bn2", "downsample.1")) or name == "1":
                mod.bias.fill_(3.5)      # (downsample.1: it is the identity of a stage's first block, added in front of a ReLU)
    x = torch.randn(8, 3, 128, 128)
    m64 = copy.deepcopy(enc).double()
    y64 = m64.resnet(x.double())
    torch.manual_seed(1)
    wgt = torch.randn(y64.shape, dtype=torch.float64)
    (y64 * wgt).sum().backward()
    g64 = {k: p.grad.detach() for k, p in m64.named_parameters() if p.grad is not None}
    g = copy.deepcopy(enc).to(dev)
    yg = g(x.to(dev), pooled=False).permute(0, 3, 1, 2)
    (yg * wgt.float().to(dev)).sum().backward()
    gg = {k: p.grad.detach().cpu() for k, p in g.named_parameters() if p.grad is not None}
    assert set(gg) == set(g64) and len(gg) > 400
    ey = rel_l2(yg.cpu(), y64)
    # Some gradients are EXACTLY zero in exact arithmetic: the shift of the last BatchNorm of a stage's last block (and, with
    # the ReLUs almost never cutting, of most bn3 shifts) is removed again by the batch statistics of every BatchNorm
    # downstream (BatchNorm is invariant to a per-channel constant added to its input); their fp64 gradients are 1e-13 of
    # their neighbours' and a relative error against them means nothing.  The denominator is therefore floored at 1e-4
    # (matrices) / 1e-2 (vectors: the bn3 shifts are non-zero only through the 0.02 % of activations the ReLUs still cut) of
    # the median gradient norm of the tensors of the same shape class.
    import statistics as _st
    norm_med = {d: _st.median(g64[k].norm().item() for k in g64 if (g64[k].dim() == 1) == d) for d in (True, False)}
    errs = {k: ((gg[k].double() - g64[k]).norm() / max(g64[k].norm().item(), (1e-2 if g64[k].dim() == 1 else 1e-4) *
                                                           norm_med[g64[k].dim() == 1])).item() for k in g64}
    cat = lambda d: torch.cat([d[k].flatten().double() for k in g64])
    e_all = ((cat(gg) - cat(g64)).norm() / cat(g64).norm()).item()
    worst = max(errs, key=errs.get)
    lines = ["trunk map rel-l2 %.3e; gradients of %d tensors: median %.3e, worst %.3e (%s), all of layer2-4 as one vector %.3e"
             % (ey, len(errs), statistics.median(errs.values()), errs[worst], worst, e_all)]
    for stage in ("resnet.5", "resnet.6", "resnet.7"):
        ks = [k for k in errs if k.startswith(stage)]
        lines.append("  %s: %d tensors, median %.3e, max %.3e" % (stage, len(ks), statistics.median(errs[k] for k in ks), max(errs[k] for k in ks)))
    _report(lines, "well-conditioned whole trunk vs fp64")
    assert ey <= 1e-4, lines[0]
    assert statistics.median(errs.values()) <= 1e-3 and errs[worst] <= 3e-3 and e_all <= 1e-3, lines[0]


def test_baseline_config2_exact_sizes_pure_scn_vs_oracle(dev):
    """BASELINE configs[1] at its exact sizes -- pure_scn, B=32, T=51 (52-wide captions), V=10 000, 1000 tags, dropout
    mask injected -- against the fp64 oracle (oracle/scnattn_ref.py, reference models/decoders/pure_scn.py:87-140):
    predictions 1e-4, every gradient 2e-4 (no ReLU in this decoder: no floors)."""
    import test_gpu_parity_r2 as T2
    from models.decoders.pure_scn import PureSCN
    torch.manual_seed(32)
    B, V, L = 32, 10000, 52
    m = PureSCN(512, 512, 512, 1000, V, dropout=0.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(14)
    x = torch.rand(B, 8, 8, 2048, generator=g)
    tags = torch.rand(B, 1000, generator=g)
    lens = torch.full((B,), L)
    caps = T2._synthetic_caps(B, V, L, lens, g)
    caplens = lens.unsqueeze(1)
    mask = (torch.rand(B, L - 1, 512, generator=g) > 0.5).float() * 2.0
    si = torch.arange(B)
    r64 = T2._oracle_run("pure_scn", sd, x, tags, caps, caplens, mask, si, torch.float64)
    hip = T2._hip_run("pure_scn", m, x, tags, caps, caplens, mask, si, dev)
    assert hip[0].shape == (B, 51, V)
    T2._compare("pure_scn", m, hip, r64, None, "BASELINE config 2 exact sizes (pure_scn, T=51, V=10000)")


@pytest.mark.parametrize("kind,B,lens", [("attention_scn", 32, "ragged"), ("attention_scn", 7, "ragged"), ("pure_scn", 32, "full")])
def test_cell_kernels_fused_into_the_skinny_launches_are_bit_identical(dev, kind, B, lens):
    """K1: the decode step's element-wise kernels (scn_mix_fwd, lstm_fwd; scn_mix_bwd, gate_bwd, lstm_bwd -- reference
    models/scn_cell.py:62-154 and the gate of models/decoders/attention_scn.py:147-150) run INSIDE the skinny launch that
    feeds them, by the workgroup that arrives last at each 32-column unit (option dec_tail; measured slower than their own
    launches, so default 0: profiles/r03_decode_step_fused_cell_kernels_A_B.txt).  Both paths share csrc/scn_elem.h's
    arithmetic (contraction off) and sum the split-K slabs in slab order, so predictions, alphas, loss and EVERY
    gradient must be bit-identical -- on ragged caption lengths
    (the shrinking batch b_t: the fused LSTM backward of step t-1 runs in step t's launch, whose product has fewer rows),
    a batch that is not a multiple of anything, and the attention-less decoder."""
    import copy
    import test_gpu_parity_r2 as T2
    from scnattn import functional as SF
    torch.manual_seed(5)
    V, L = 1200, 19
    if kind == "attention_scn":
        from models.decoders.attention_scn import AttentionSCN
        m0 = AttentionSCN(512, 512, 512, 512, 1000, V, dropout=0.5)
    else:
        from models.decoders.pure_scn import PureSCN
        m0 = PureSCN(512, 512, 512, 1000, V, dropout=0.5)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 8, 8, 2048, generator=g)
    tags = torch.rand(B, 1000, generator=g)
    ln = torch.full((B,), L) if lens == "full" else torch.randint(3, L + 1, (B,), generator=g).sort(descending=True).values
    caps = T2._synthetic_caps(B, V, L, ln, g)
    mask = (torch.rand(B, int(ln.max()) - 1, 512, generator=g) > 0.5).float() * 2.0
    si = torch.arange(B)
    runs = []
    try:
        for tail in (0, 1, 1):     # reference: their own launches
            SF.set_option("dec_tail", tail)
            preds, alphas, loss, dx, mg = T2._hip_run(kind, copy.deepcopy(m0), x, tags, caps, ln.unsqueeze(1), mask, si, dev)
            torch.cuda.synchronize()
            runs.append((preds.detach().clone(), None if alphas is None else alphas.detach().clone(), loss.detach().clone(), dx.clone(),
                         {k: p.grad.clone() for k, p in mg.named_parameters() if p.grad is not None}))
    finally:
        SF.set_option("dec_tail", 0)
    ref = runs[0]
    assert ref[4] and float(ref[0].abs().sum()) > 0
    for r in runs[1:]:
        assert torch.equal(r[0], ref[0]) and torch.equal(r[2], ref[2]) and torch.equal(r[3], ref[3])
        if ref[1] is not None:
            assert torch.equal(r[1], ref[1])
        for k, v in ref[4].items():
            assert torch.equal(r[4][k], v), k


def test_encoder_tagger_forward_at_the_train_steps_size(dev):
    """EncoderTagger.forward (reference models/encoders/tagger.py:34-47, called at
    trains/attention_scn.py:214) at 32 x 3 x 256 x 256, training mode (batch statistics; injected dropout mask), on the
    GPU (stem + fused Bottlenecks + Linear on the hand-written kernels) against the same weights run by torch ops on the
    CPU in fp64.  The randomly initialised 152-layer trunk amplifies fp32 rounding (torch's own CPU fp32 forward is
    ~1e-3 from fp64 here), so fp64 is the anchor and CPU fp32 the yardstick: the GPU's distance to fp64 must not exceed
    5x the CPU fp32's, and 5e-3 absolutely.  PARITY UNPINNED against the reference (its trunk is torchvision's)."""
    import test_gpu_parity_r2 as T2
    from models.encoders.tagger import EncoderTagger
    torch.manual_seed(3)
    m = EncoderTagger(semantic_size=1000, dropout=0.15, channels_last=True)
    with torch.no_grad():
        m.linear.weight.mul_(4.0)
    B = 32
    x = torch.randn(B, 3, 256, 256)
    mask = (torch.rand(B, 2048) > 0.15).float() / 0.85
    cpu = copy.deepcopy(m).double().train()
    cpu32 = copy.deepcopy(m).train()
    with torch.no_grad():
        feat = cpu.resnet(x.double()).reshape(B, -1) * mask.double()
        ref = torch.sigmoid(F.linear(feat, cpu.linear.weight, cpu.linear.bias))
        ref32 = torch.sigmoid(F.linear(cpu32.resnet(x).reshape(B, -1) * mask, cpu32.linear.weight, cpu32.linear.bias))
    e32 = rel_err(ref32, ref)
    g = m.to(dev).train()
    g.dropout = T2._FixedMask(mask)
    with torch.no_grad():
        y = g(x.to(dev))
    assert y.shape == (B, 1000) and float(ref.max() - ref.min()) > 0.2
    e = rel_err(y, ref)
    _report(["tagger 32x3x256x256 train mode: rel_err vs fp64  gpu %.3e  cpu-fp32 %.3e" % (e, e32)], "EncoderTagger.forward at full size")
    # (two fp32 evaluations of this chaotic map land at independent distances from fp64: measured gpu 1.9e-3, cpu 6.3e-4)
    assert e <= max(5 * e32, 1e-3) and e <= 5e-3, (e, e32)
    for (k, b), (_, bc) in zip(g.named_buffers(), cpu.named_buffers()):
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(b, bc) < 1e-3, k


# ------------------------------------------------------------------------------------------------
# C5: the mixed-precision (bf16) trunk -- BASELINE configs[4]
# ------------------------------------------------------------------------------------------------
BF16_OUT = 5e-3      # the repository's stated bf16 tolerance: one bf16 element carries 8 significant bits
BF16_GRAD = 2.5e-2   # gradients: rel-l2 after ~10 bf16 maps in sequence (measured 6e-3 .. 1.8e-2)


def test_bf16_kernels_vs_fp64(dev):
    """csrc/cgemm16.hip (forward / d input of 1x1 and 3x3 convolutions incl. the stride-2 parity classes, statistics
    epilogue, beta accumulation, split-K), csrc/wgrad16.hip (weight gradients through the transposing LDS load: halo-staged
    3x3, gathered rows), the one-launch weight conversion and the BatchNorm kernels on bf16 maps, each against fp64 on the
    bf16-rounded operands: fp32 outputs 2e-5, bf16 outputs 4e-3 (one rounding), statistics 2e-5."""
    _tool().check16()


@pytest.mark.parametrize("name,inplanes,planes,stride,H", [("layer1.0", 64, 64, 1, 32), ("layer2.0", 256, 128, 2, 32),
                                                           ("layer3.1", 1024, 256, 1, 16), ("layer4.0", 1024, 512, 2, 16),
                                                           ("layer4.1", 2048, 512, 1, 8)])
def test_bf16_bottleneck_vs_fp64(dev, name, inplanes, planes, stride, H):
    """One Bottleneck through scnattn/conv16.py -- bf16 maps and operand copies, fp32 accumulation / statistics / master
    weights / weight gradients -- against the SAME module in fp64 on the CPU, masks made unambiguous as in
    test_fused_bottleneck_vs_fp64 (beta + 3.5 / + 6): output within BF16_OUT (rel-l2), every gradient within BF16_GRAD, weight
    gradients come back as fp32 tensors.  PARITY UNPINNED against torchvision (absent)."""
    from scnattn.resnet import Bottleneck, FusedBatchNorm2d
    from scnattn import conv16 as C16
    from torch import nn
    torch.manual_seed(2000 + H + planes)
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False), FusedBatchNorm2d(planes * 4))
    m = Bottleneck(inplanes, planes, stride, down)
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(mod, nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    with torch.no_grad():
        m.bn1.bias.add_(3.5); m.bn2.bias.add_(3.5); m.bn3.bias.add_(6.0)
    m.train()
    N = 4
    x = (torch.relu(torch.randn(N, inplanes, H, H)) + 0.1 * torch.randn(N, inplanes, H, H)).to(torch.bfloat16)
    ref = copy.deepcopy(m).double()
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    wgt = torch.randn_like(yr).to(torch.bfloat16).double()
    (yr * wgt).sum().backward()
    g = copy.deepcopy(m).to(dev).to(memory_format=torch.channels_last).train()
    C16.refresh_weights(g)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert C16.usable(g, xg)
    y = g(xg)
    assert y.dtype == torch.bfloat16
    (y.float() * wgt.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    lines = ["%s bf16: out rel-l2 %.3e  d x %.3e" % (name, rel_l2(y.float(), yr), rel_l2(xg.grad.float(), xr.grad))]
    assert rel_l2(y.float(), yr) <= BF16_OUT, lines[0]
    assert rel_l2(xg.grad.float(), xr.grad) <= BF16_GRAD, lines[0]
    for (k, p), (_, pr) in zip(g.named_parameters(), ref.named_parameters()):
        e = rel_l2(p.grad, pr.grad)
        lines.append("   %-24s grad rel-l2 %.3e (%s)" % (k, e, str(p.grad.dtype).replace("torch.", "")))
        assert p.grad.dtype == torch.float32, k
    _report(lines, "bf16 bottleneck " + name)
    # Bars (measured values in the report): d x and weight gradients 2.5e-2, BatchNorm scales 4e-2 -- about ten bf16 maps
    # (8 significant bits each) lie between the loss and a gradient.  BatchNorm SHIFT gradients are sums that cancel almost
    # completely (with the ReLUs nearly linear here, a shift of bn1 / bn2 is removed again by the next BatchNorm's batch
    # statistics), so their error is measured against the norm of the same layer's SCALE gradient when that is larger.
    refg = {k: pr.grad for k, pr in ref.named_parameters()}
    for k, p in g.named_parameters():
        den = refg[k].norm().item()
        if k.endswith(".bias"):
            den = max(den, refg[k[:-4] + "weight"].norm().item())
        e = (p.grad.double().cpu() - refg[k]).norm().item() / max(den, 1e-300)
        bar = 2.5e-2 if p.dim() > 1 else 4e-2
        assert e <= bar, "%s: %.3e > %.1e" % (k, e, bar)
    for (k, b), (_, br) in zip(g.named_buffers(), ref.named_buffers()):
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(b, br) <= BF16_OUT, k


@pytest.mark.parametrize("inplanes,planes,stride,H", [(64, 64, 1, 32), (256, 128, 2, 32), (1024, 256, 1, 16), (2048, 512, 1, 8)])
def test_bf16_block_driver_is_bit_identical_to_the_per_launch_path(dev, inplanes, planes, stride, H):
    """csrc/block16.cpp (scnattn_block16_fwd / _bwd: one library call per Bottleneck and direction, the default of the bf16
    trunk) enqueues the same kernels on the same operands in the same order as scnattn/conv16.py's per-launch path
    (SCNATTN_BLOCK16=py): output, d x, every weight / gamma / beta gradient and the running statistics must be
    bit-identical, for identity blocks and the three kinds of down-sampling block, over four consecutive steps (a step's
    conditioning shift is the previous step's batch mean on both paths).  Parameters live in a flat buffer
    (utils.optimizer.FusedClampAdam) as in the train step.  (A third path, the same sequences replayed from HIP graphs,
    passed this test too and was removed for being slower: profiles/r03_bf16_block_hip_graph_replay_rejected.txt.)"""
    from scnattn.resnet import Bottleneck, FusedBatchNorm2d
    from scnattn import conv16 as C16
    from utils.optimizer import FusedClampAdam
    from torch import nn
    torch.manual_seed(77 + H + planes)
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False), FusedBatchNorm2d(planes * 4))
    m = Bottleneck(inplanes, planes, stride, down).train()
    N = 8
    xs = [(torch.relu(torch.randn(N, inplanes, H, H)) + 0.1 * torch.randn(N, inplanes, H, H)).to(torch.bfloat16) for _ in range(4)]
    wg = None
    res = {}
    old = C16.BLOCK16
    try:
        for mode in ("py", "c"):
            C16.BLOCK16 = mode
            g = copy.deepcopy(m).to(dev).to(memory_format=torch.channels_last).train()
            opt = FusedClampAdam(list(g.parameters()), lr=0.0, grad_clip=5.0)
            outs = []
            for x in xs:
                C16.refresh_weights(g)
                xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                assert C16.usable(g, xg)
                opt.zero_grad()
                y = g(xg)
                if wg is None:
                    wg = torch.randn_like(y)
                y.backward(wg)
                torch.cuda.synchronize()
                outs.append((y.detach().clone(), xg.grad.clone(), {k: p.grad.clone() for k, p in g.named_parameters()},
                             {k: b.clone() for k, b in g.named_buffers()}))
            res[mode] = outs
    finally:
        C16.BLOCK16 = old
    for mode in ("c",):
        for (y0, dx0, g0, b0), (y1, dx1, g1, b1) in zip(res["py"], res[mode]):
            assert torch.equal(y0, y1) and torch.equal(dx0, dx1), mode
            assert set(g0) == set(g1) and all(v is not None for v in g0.values())
            for k in g0:
                assert torch.equal(g0[k], g1[k]), (mode, k)
            for k in b0:
                assert torch.equal(b0[k], b1[k]), (mode, k)


def test_bf16_train_step_runs_on_the_hand_written_trunk_and_learns(dev):
    """`--dtype bf16` of the harness (BASELINE configs[4] flavour): the trunk must take scnattn/conv16.py (no MIOpen
    autocast fallback: every Bottleneck output is a bf16 map produced by _Bottleneck16Fn), the first-step loss must agree
    with the fp32 step's to BF16_OUT-level accuracy and a few steps must reduce the loss."""
    from trains.harness import TrainStep, synthetic_batch
    from scnattn import conv16 as C16
    res = {}
    for enc_dtype in ("f32", "bf16"):
        ts = TrainStep(device=dev, seed=5, batch_size=8, max_len=10, vocab_size=200, image_size=128, encoder_dtype=enc_dtype,
                       dropout=0.0)
        cfg = ts.cfg
        imgs, tags, caps, caplens = synthetic_batch(8, cfg["vocab_size"], cfg["max_len"], cfg["image_size"], cfg["semantic_dim"],
                                                    torch.device(dev), 3)
        calls = []
        if enc_dtype == "bf16":
            orig = C16._Bottleneck16Fn.apply
            hooks = [m.register_forward_hook(lambda mod, i, o: calls.append(o.dtype)) for m in ts.encoder.resnet.modules()
                     if type(m).__name__ == "Bottleneck"]
        losses = [float(ts.step(imgs, tags, caps, caplens)) for _ in range(6)]
        if enc_dtype == "bf16":
            assert len(calls) == 6 * 50 and all(d == torch.bfloat16 for d in calls), "the bf16 trunk did not run on conv16"
            assert all(p.grad is None or p.grad.dtype == torch.float32 for p in ts.encoder.parameters())
        res[enc_dtype] = losses
    _report(["fp32 losses %s" % ["%.4f" % l for l in res["f32"]], "bf16 losses %s" % ["%.4f" % l for l in res["bf16"]]],
            "bf16 train step vs fp32 train step")
    assert abs(res["bf16"][0] - res["f32"][0]) <= 2e-2 * abs(res["f32"][0])
    assert res["bf16"][-1] < res["bf16"][0] and all(l == l for l in res["bf16"])
