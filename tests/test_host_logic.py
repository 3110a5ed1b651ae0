"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol that
include/scnattn.h declares, argument validation returns error codes (no GPU call is made), module
construction / state_dict keys / error messages match the reference, and the product path refuses to
run without the GPU instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from helpers import load_golden, params_from

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from scnattn import _lib
    return _lib


def test_library_exports_every_declared_symbol():
    L = _lib()
    h = L.lib()
    header = open(os.path.join(ROOT, "include", "scnattn.h")).read()
    declared = set(re.findall(r"\b(scnattn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(h, name), "libscnattn.so does not export " + name
    assert set(L.EXPORTS) <= declared | {"scnattn_last_error"}
    assert h.scnattn_version() == 108


def test_invalid_arguments_return_codes_and_messages():
    L = _lib()
    h = L.lib()
    assert h.scnattn_set_option(b"no_such_option", 1) == -1
    assert b"unknown option" in h.scnattn_last_error()
    d = L.Dims(0, 196, 2048, 512, 512, 512, 512, 1000, 100, 5, 7, 1)   # B = 0
    sv, sc = C.c_size_t(), C.c_size_t()
    assert h.scnattn_seq_workspace(C.byref(d), None, C.byref(sv), C.byref(sc)) == -1
    assert b"positive" in h.scnattn_last_error()
    d = L.Dims(32, 196, 2048, 512, 512, 512, 512, 1000, 10000, 51, 52, 1)
    assert h.scnattn_seq_workspace(C.byref(d), None, C.byref(sv), C.byref(sc)) == 0
    assert sv.value > 0 and sc.value > 0 and sv.value % 256 == 0
    # null operands are rejected before any launch
    assert h.scnattn_sgemm(None, 0, 0, 4, 4, 4, 1.0, None, 4, None, 4, 0.0, None, 4, None, None, 1, 0, 0, 0) == -1
    with pytest.raises(RuntimeError, match="scnattn_sgemm failed"):
        L.call("scnattn_sgemm", None, 0, 0, 4, 4, 4, 1.0, None, 4, None, 4, 0.0, None, 4, None, None, 1, 0, 0, 0)


def test_no_cpu_fallback():
    from models.scn_cell import SCNCell
    from models.attention import Attention
    from models.decoders.attention_scn import AttentionSCN
    cell = SCNCell(8, 4, 3, 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cell(torch.randn(2, 8), torch.rand(2, 3))
    att = Attention(6, 4, 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        att(torch.randn(2, 9, 6), torch.randn(2, 4))
    dec = AttentionSCN(5, 4, 4, 6, 3, 11, encoder_dim=6, dropout=0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(torch.randn(2, 3, 3, 6), torch.rand(2, 3), torch.randint(0, 11, (2, 5)), torch.tensor([[5], [4]]))


def test_scn_cell_api_matches_reference_fixture():
    from models.scn_cell import SCNCell
    d = load_golden("scn_cell")
    I, F4 = d["p.weight_ia"].shape
    S, H = d["p.weight_ib"].shape[0], d["p.weight_ic"].shape[0]
    m = SCNCell(I, H, S, F4 // 4)
    assert repr(m) == str(d["repr"])
    ref_keys = [k[2:] for k in d if k.startswith("p.")]
    assert list(m.state_dict().keys()) == ref_keys
    m.load_state_dict(params_from(d))
    bound = 1.0 / np.sqrt(H)
    m.reset_parameters()
    for p in m.parameters():        # B4: every parameter, biases too, ~ U(-1/sqrt(H), 1/sqrt(H))
        assert p.abs().max().item() <= bound + 1e-7
    msgs = [str(x) for x in d["errors"]]
    B = d["u"].shape[0]
    with pytest.raises(RuntimeError) as e:
        m(torch.randn(B, I + 1), torch.rand(B, S))
    assert str(e.value) == msgs[0]
    with pytest.raises(RuntimeError) as e:
        m(torch.randn(B, I), torch.rand(B, S), (torch.randn(B + 1, H), torch.randn(B + 1, H)))
    assert str(e.value) == msgs[1]
    with pytest.raises(RuntimeError) as e:
        m(torch.randn(B, I), torch.rand(B, S), (torch.randn(B, H + 1), torch.randn(B, H + 1)))
    assert str(e.value) == msgs[2]


@pytest.mark.parametrize("name,kind", [("attention_scn_distinct", "attention_scn"), ("pure_scn_distinct", "pure_scn"),
                                       ("pure_attention_distinct", "pure_attention"), ("attention", "attention")])
def test_state_dict_keys_and_shapes_match_reference(name, kind):
    from models.attention import Attention
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from models.decoders.pure_attention import PureAttention
    d = load_golden(name)
    if kind == "attention":
        A, E = d["p.encoder_att.weight"].shape
        m = Attention(E, d["p.decoder_att.weight"].shape[1], A)
    else:
        V, M = d["p.embedding.weight"].shape
        D, E = d["p.init_h.weight"].shape
        if kind == "pure_attention":
            m = PureAttention(d["p.attention.encoder_att.weight"].shape[0], M, D, V, encoder_dim=E, dropout=0.0)
        else:
            S, F4 = d["p.decode_step.weight_ib"].shape
            if kind == "attention_scn":
                m = AttentionSCN(d["p.attention.encoder_att.weight"].shape[0], M, D, F4 // 4, S, V, encoder_dim=E,
                                 dropout=0.0)
            else:
                m = PureSCN(M, D, F4 // 4, S, V, encoder_dim=E, dropout=0.0)
    ref = {k[2:]: v.shape for k, v in d.items() if k.startswith("p.")}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(mine.keys()) == list(ref.keys())
    assert mine == {k: tuple(s) for k, s in ref.items()}
    m.load_state_dict(params_from(d))   # strict


def test_decoder_init_weights_and_embedding_helpers():
    from models.decoders.attention_scn import AttentionSCN
    m = AttentionSCN(5, 4, 4, 6, 3, 50, encoder_dim=6, dropout=0.3)
    assert m.embedding.weight.abs().max().item() <= 0.1 and m.fc.weight.abs().max().item() <= 0.1
    assert m.fc.bias.abs().max().item() == 0.0
    assert isinstance(m.dropout, torch.nn.Dropout) and abs(m.dropout.p - 0.3) < 1e-12
    m.fine_tune_embeddings(False)
    assert not m.embedding.weight.requires_grad
    m.load_pretrained_embeddings(torch.zeros(50, 4))
    assert m.embedding.weight.abs().max().item() == 0.0 and m.embedding.weight.requires_grad


def test_encoder_structure_matches_torchvision_resnet152_layout():
    from models.encoders.caption import EncoderCaption, Encoder
    enc = EncoderCaption()
    assert issubclass(Encoder, EncoderCaption)
    n_all = sum(p.numel() for p in enc.parameters())
    assert n_all == 58143808                       # ResNet-152 minus fc
    assert sum(p.numel() for p in enc.parameters() if p.requires_grad) == 57918464   # layer2-4 (B13)
    sd = enc.state_dict()
    for k in ("resnet.0.weight", "resnet.1.running_mean", "resnet.4.0.downsample.0.weight",
              "resnet.6.35.conv3.weight", "resnet.7.2.bn3.num_batches_tracked"):
        assert k in sd
    assert sd["resnet.0.weight"].shape == (64, 3, 7, 7) and sd["resnet.7.2.conv3.weight"].shape == (2048, 512, 1, 1)
    enc.fine_tune(False)
    assert not any(p.requires_grad for p in enc.parameters())
    assert enc.enc_image_size == 14 and isinstance(enc.adaptive_pool, torch.nn.AdaptiveAvgPool2d)


def test_tagger_structure():
    from models.encoders.tagger import EncoderTagger
    t = EncoderTagger()
    sd = t.state_dict()
    assert "linear.weight" in sd and sd["linear.weight"].shape == (1000, 2048) and "resnet.7.2.conv3.weight" in sd
    assert isinstance(list(t.resnet.children())[-1], torch.nn.AdaptiveAvgPool2d)
    assert abs(t.dropout.p - 0.15) < 1e-12
    t.fine_tune(False)
    assert not any(p.requires_grad for p in t.resnet.parameters())
    assert all(p.requires_grad for p in t.linear.parameters())   # the reference leaves the head trainable


def test_encoder_trunk_matches_independent_resnet_definition():
    """Numerical cross-check of our ResNet trunk against HuggingFace's ResNetModel (same v1.5 bottleneck
    topology, built from a local config object: no download) with OUR weights copied in by name.
    HF is NOT the reference's dependency (torchvision is absent everywhere), so parity at the encoder
    boundary stays 'unpinned'; this guards the definition against structural mistakes."""
    tr = pytest.importorskip("transformers")
    from scnattn.resnet import resnet152_trunk
    torch.manual_seed(0)
    depths = [2, 2, 3, 2]   # a shallow stack keeps the CPU test fast; the block code is the 3-8-36-3 one
    ours = resnet152_trunk(depths=depths).eval()
    for m in ours.modules():   # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    cfg = tr.ResNetConfig(depths=depths, hidden_sizes=[256, 512, 1024, 2048], layer_type="bottleneck",
                          embedding_size=64, downsample_in_bottleneck=False)
    hf = tr.ResNetModel(cfg).eval()
    sd = ours.state_dict()

    def bn(dst, src):
        return {dst + ".weight": sd[src + ".weight"], dst + ".bias": sd[src + ".bias"],
                dst + ".running_mean": sd[src + ".running_mean"], dst + ".running_var": sd[src + ".running_var"]}

    new = {"embedder.embedder.convolution.weight": sd["0.weight"]}
    new.update(bn("embedder.embedder.normalization", "1"))
    for s_i, nblk in enumerate(depths):
        for j in range(nblk):
            o, h = "%d.%d" % (4 + s_i, j), "encoder.stages.%d.layers.%d" % (s_i, j)
            for c in range(3):
                new["%s.layer.%d.convolution.weight" % (h, c)] = sd["%s.conv%d.weight" % (o, c + 1)]
                new.update(bn("%s.layer.%d.normalization" % (h, c), "%s.bn%d" % (o, c + 1)))
            if o + ".downsample.0.weight" in sd:
                new[h + ".shortcut.convolution.weight"] = sd[o + ".downsample.0.weight"]
                new.update(bn(h + ".shortcut.normalization", o + ".downsample.1"))
    missing, unexpected = hf.load_state_dict(new, strict=False)
    assert not [k for k in missing if "num_batches_tracked" not in k], missing
    assert not unexpected, unexpected
    x = torch.randn(2, 3, 96, 96)
    with torch.no_grad():
        y, yh = ours(x), hf(x).last_hidden_state
    assert y.shape == yh.shape == (2, 2048, 3, 3)
    assert (y - yh).abs().max().item() <= 1e-4 * yh.abs().max().item()


def test_corpus_bleu_known_values():
    """NLTK-free corpus BLEU (utils/metric.py) against hand-computed values of the standard definition."""
    import math
    from utils.metric import corpus_bleu, AverageMeter
    ref = [[1, 2, 3, 4, 5, 6, 7, 8]]
    assert abs(corpus_bleu([ref], [[1, 2, 3, 4, 5, 6, 7, 8]]) - 1.0) < 1e-12
    # hypothesis = first 6 tokens: all n-gram precisions 1, brevity penalty exp(1 - 8/6)
    assert abs(corpus_bleu([ref], [[1, 2, 3, 4, 5, 6]]) - math.exp(1 - 8 / 6)) < 1e-12
    # one substitution in the middle: p1 = 7/8, p2 = 5/7, p3 = 3/6, p4 = 1/5, BP = 1
    want = math.exp(0.25 * (math.log(7 / 8) + math.log(5 / 7) + math.log(3 / 6) + math.log(1 / 5)))
    assert abs(corpus_bleu([ref], [[1, 2, 3, 9, 5, 6, 7, 8]]) - want) < 1e-12
    # clipping + closest reference length + corpus-level pooling over two segments
    refs = [[[1, 1, 2, 3], [1, 2, 3, 4, 5]], [[7, 8, 9, 10]]]
    hyps = [[1, 1, 1, 2, 3], [7, 8, 9, 10]]
    # n=1: seg1 clipped 1x2 + 2 + 3 -> 4 of 5, seg2 4 of 4 -> 8/9 ; n=2: (1,1)x1,(1,2),(2,3) -> 3 of 4, 3 of 3 -> 6/7
    # n=3: (1,1,2),(1,2,3) -> 2 of 3, 2 of 2 -> 4/5 ; n=4: (1,1,2,3) -> 1 of 2, 1 of 1 -> 2/3 ; len 9 vs ref 5+4 -> BP 1
    want = math.exp(0.25 * (math.log(8 / 9) + math.log(6 / 7) + math.log(4 / 5) + math.log(2 / 3)))
    assert abs(corpus_bleu(refs, hyps) - want) < 1e-12
    assert corpus_bleu([ref], [[9, 9, 9, 9]]) == 0.0
    m = AverageMeter(); m.update(2.0, 3); m.update(4.0, 1)
    assert m.val == 4.0 and m.count == 4 and abs(m.avg - 2.5) < 1e-12


def test_corpus_bleu_matches_nltk_golden():
    """utils.metric.corpus_bleu against values produced by NLTK's corpus_bleu itself (the function the
    reference calls, trains/attention_scn.py:377); fixture written by oracle/gen_bleu_golden.py.  Where a
    higher-order precision is 0 NLTK returns ~1e-78 (float_info.min under the log) and we return 0."""
    import json
    from utils.metric import corpus_bleu
    with open(os.path.join(os.path.dirname(__file__), "golden", "bleu_nltk.json")) as fh:
        gold = json.load(fh)
    assert len(gold["cases"]) >= 15
    nonzero = 0
    for c in gold["cases"]:
        got4 = corpus_bleu(c["references"], c["hypotheses"])
        got2 = corpus_bleu(c["references"], c["hypotheses"], weights=(0.5, 0.5))
        assert abs(got4 - c["bleu4"]) <= 1e-12, (got4, c["bleu4"])
        assert abs(got2 - c["bleu2"]) <= 1e-12, (got2, c["bleu2"])
        nonzero += c["bleu4"] > 1e-3
    assert nonzero >= 6


def test_checkpoint_roundtrip_keeps_reference_layout_and_flat_aliasing(tmp_path):
    """utils/checkpoint.py:4-31: whole modules + optimizers pickled in one dict under the reference's keys
    and file names; after loading, the fused optimizer's parameters still alias its flat buffer and a
    `state_dict` feeds utils.loader.load_decoder (utils/loader.py:9-68)."""
    from models.decoders.attention_scn import AttentionSCN
    from utils.checkpoint import save_checkpoint, save_tagger_checkpoint, load_checkpoint
    from utils.loader import load_decoder
    from utils.optimizer import FusedClampAdam
    torch.manual_seed(0)
    dec = AttentionSCN(16, 12, 16, 20, 10, 30, dropout=0.5)      # encoder_dim 2048, as load_decoder assumes
    opt = FusedClampAdam(dec.parameters(), lr=4e-4, grad_clip=5.0)
    opt.step_count = 3
    opt.flat_m.uniform_(-1, 1)
    enc = torch.nn.Linear(4, 4)
    path = save_checkpoint("attention_scn", "tiny", 7, 2, enc, dec, None, opt, 0.25, True, folder=str(tmp_path))
    assert os.path.basename(path) == "checkpoint_attention_scn_tiny.pth.tar"
    assert os.path.exists(str(tmp_path / "BEST_checkpoint_attention_scn_tiny.pth.tar"))
    ck = load_checkpoint(path)
    assert set(ck) == {"epoch", "epochs_since_improvement", "bleu-4", "encoder", "decoder", "encoder_optimizer",
                       "decoder_optimizer"}
    assert ck["epoch"] == 7 and ck["epochs_since_improvement"] == 2 and ck["bleu-4"] == 0.25
    d2, o2 = ck["decoder"], ck["decoder_optimizer"]
    assert type(d2).__module__ == "models.decoders.attention_scn" and ck["encoder_optimizer"] is None
    for (k, a), (_, b) in zip(dec.state_dict().items(), d2.state_dict().items()):
        assert torch.equal(a, b), k
    assert o2.step_count == 3 and torch.equal(o2.flat_m, opt.flat_m)
    lo, hi = o2.flat.flat_p.data_ptr(), o2.flat.flat_p.data_ptr() + 4 * o2.flat.flat_p.numel()
    named = dict(d2.named_parameters())
    assert all(lo <= p.data_ptr() < hi for p in o2.params)           # views of the flat buffer again
    assert {id(p) for p in o2.params} == {id(p) for p in named.values() if p.requires_grad}
    o2.flat.flat_p.zero_()
    assert all(float(p.detach().abs().max()) == 0.0 for p in d2.parameters())
    p2 = save_tagger_checkpoint("tiny", 1, 0, enc, None, 0.5, False, folder=str(tmp_path))
    assert os.path.basename(p2) == "checkpoint_tagger_tiny.pth.tar" and load_checkpoint(p2)["accuracy"] == 0.5
    assert not os.path.exists(str(tmp_path / "BEST_checkpoint_tagger_tiny.pth.tar"))
    fresh = load_decoder("attention_scn", dec.state_dict(), 30, embed_dim=12, attention_dim=16, decoder_dim=16,
                         factored_dim=20, semantic_dim=10)
    assert torch.equal(fresh.fc.weight.cpu(), dec.fc.weight)
    with pytest.raises(ValueError, match="model type not found"):
        load_decoder("nope", {}, 30)


@pytest.mark.parametrize("hin,hout", [(8, 14), (7, 14), (14, 14), (5, 9)])
def test_pool_taps_encode_adaptive_avg_pool(hin, hout):
    """scnattn.functional.PoolTaps (the tables behind scnattn_pool) reproduce AdaptiveAvgPool2d + permute of
    models/encoders/caption.py:41-43 exactly, their transpose lists are consistent, and pools that are not
    up-sampling (windows wider than 2) are refused."""
    from scnattn import functional as SF
    pt = SF.PoolTaps(hin, hin, hout, hout, "cpu")
    x = torch.randn(3, 6, hin, hin, dtype=torch.float64)
    ref = torch.nn.functional.adaptive_avg_pool2d(x, hout).permute(0, 2, 3, 1).reshape(3, hout * hout, 6)
    m = pt.matrix().double()
    got = m @ x.permute(0, 2, 3, 1).reshape(3, hin * hin, 6)
    assert (got - ref).abs().max().item() <= 1e-12
    assert torch.allclose(m.sum(dim=1), torch.ones(hout * hout, dtype=torch.float64))
    mt = torch.zeros(pt.Q, pt.P, dtype=torch.float64)
    for q in range(pt.Q):
        for k in range(pt.qmax):
            if pt.qtap_idx[q, k] >= 0:
                mt[q, pt.qtap_idx[q, k]] += pt.qtap_w[q, k]
    assert torch.equal(mt, m.t())
    assert torch.allclose(pt.col_w.double(), m.sum(dim=0) / pt.P)
    with pytest.raises(ValueError):
        SF.PoolTaps(14, 14, 4, 4, "cpu")


def test_state_dict_checkpoint_feeds_the_reference_tooling_keys(tmp_path):
    """inference.py:118-129 / eval_caption.py:77-85 read `encoder_model_state_dict` / `decoder_model_state_dict`
    (and `model_state_dict` for the tagger); such a file loads with weights_only=True and rebuilds the decoder
    through utils.loader.load_decoder."""
    from models.decoders.pure_scn import PureSCN
    from utils.checkpoint import save_state_dicts
    from utils.loader import load_decoder
    torch.manual_seed(1)
    dec = PureSCN(12, 16, 20, 1000, 30)
    enc = torch.nn.Linear(3, 3)
    path = save_state_dicts(str(tmp_path / "caption.pth"), encoder=enc, decoder=dec, epoch=4)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"encoder_model_state_dict", "decoder_model_state_dict", "epoch"} and ck["epoch"] == 4
    again = load_decoder("pure_scn", ck["decoder_model_state_dict"], 30, embed_dim=12, decoder_dim=16, factored_dim=20)
    for (k, a), (_, b) in zip(dec.state_dict().items(), again.state_dict().items()):
        assert torch.equal(a, b.cpu()), k
    t = save_state_dicts(str(tmp_path / "tagger.pth"), tagger=enc)
    assert set(torch.load(t, weights_only=True)) == {"model_state_dict"}


def test_public_header_is_plain_c(tmp_path):
    """include/scnattn.h is the drop-in boundary: it must compile as C99 on its own (plain pointers and sizes,
    no C++ or torch types), and as C++."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "scnattn.h"\nint main(void) { scnattn_dims d; scnattn_params p; scnattn_pool q; '
                   '(void)d; (void)p; (void)q; return 0; }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)]):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr


def test_synthetic_batch_follows_the_word_map_convention():
    """SURVEY 8d / utils/dataset.py:302-306: <start> = V-2 at column 0, words in [1, V-4], <end> = V-1 closing the
    caption, pad = 0 after it; caplens counts <start> and <end>; the ragged variant varies the lengths."""
    from trains.harness import synthetic_batch, DEFAULTS
    V, ml = 50, 10
    imgs, tags, caps, caplens = synthetic_batch(6, V, ml, 16, 7, "cpu", 3)
    assert imgs.shape == (6, 3, 16, 16) and tags.shape == (6, 7) and caps.shape == (6, ml + 2) and caplens.shape == (6, 1)
    assert (caplens == ml + 2).all() and (caps[:, 0] == V - 2).all() and (caps[:, -1] == V - 1).all()
    assert caps[:, 1:-1].min() >= 1 and caps[:, 1:-1].max() <= V - 4 and (0 <= tags).all() and (tags <= 1).all()
    _, _, caps_r, lens_r = synthetic_batch(16, V, ml, 16, 7, "cpu", 3, ragged=True)
    assert len(set(lens_r.flatten().tolist())) > 1
    for c, n in zip(caps_r, lens_r.flatten().tolist()):
        assert c[0] == V - 2 and c[n - 1] == V - 1 and (c[n:] == 0).all() and (c[1:n - 1] >= 1).all()
    assert DEFAULTS["batch_size"] == 32 and DEFAULTS["decoder_lr"] == 4e-4 and DEFAULTS["encoder_lr"] == 1e-4 \
        and DEFAULTS["grad_clip"] == 5.0 and DEFAULTS["alpha_c"] == 1.0 and DEFAULTS["dropout"] == 0.5   # trains/attention_scn.py:31-61


def test_caption_files_and_metric_optimizer_helpers():
    """scnattn.data.CaptionFiles on the h5py-written fixtures; utils.metric.{AverageMeter, accuracy} and
    utils.optimizer.{clip_gradient, adjust_learning_rate} with the reference's semantics (utils/metric.py:1-39,
    utils/optimizer.py:1-26)."""
    from scnattn.data import CaptionFiles
    from utils.metric import AverageMeter, accuracy
    from utils.optimizer import clip_gradient, adjust_learning_rate
    G = os.path.join(os.path.dirname(__file__), "golden", "hdf5")
    cf = CaptionFiles(G, "tiny_2_cap_per_img_0_min_word_freq", "TRAIN", cpi=None)
    assert cf.cpi == 2 and len(cf) == 6 and cf.imgs.shape == (3, 3, 256, 256) and cf.imgs.dtype == np.uint8
    assert cf.captions.shape == (6, 8) and cf.caplens.shape == (6,) and not cf.imgs.flags.writeable
    with pytest.raises(RuntimeError, match="need more than"):
        CaptionFiles(G, "tiny_2_cap_per_img_0_min_word_freq", "TRAIN", cpi=1)        # 6 captions at 1 per image > 3 images
    m = AverageMeter()
    m.update(2.0, 3)
    m.update(4.0, 1)
    assert m.val == 4.0 and m.sum == 10.0 and m.count == 4 and m.avg == 2.5
    scores = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5]])
    assert accuracy(scores, torch.tensor([1, 2, 2]), 1) == pytest.approx(100.0 * 2 / 3)
    assert accuracy(scores, torch.tensor([1, 2, 2]), 2) == pytest.approx(100.0 * 2 / 3)
    assert accuracy(scores, torch.tensor([1, 1, 2]), 2) == pytest.approx(100.0)
    w = torch.nn.Parameter(torch.zeros(4))
    opt = torch.optim.Adam([w], lr=4e-4)
    w.grad = torch.tensor([-9.0, -1.0, 2.0, 7.0])
    clip_gradient(opt, 5.0)
    assert w.grad.tolist() == [-5.0, -1.0, 2.0, 5.0]
    adjust_learning_rate(opt, 0.8)
    assert opt.param_groups[0]["lr"] == pytest.approx(3.2e-4)


def test_pure_attention_lstm_as_scn_relayout():
    """PureAttention._scn_view_of_lstm: the SCN-shaped weights reproduce torch's LSTMCell pre-activations gate by
    gate (torch order i, f, g, o -> SCN order i, f, o, c), the constant factors are ones / identities, and autograd
    reaches the LSTMCell's own parameters through the re-layout."""
    from models.decoders.pure_attention import PureAttention
    from scnattn._lib import PARAM_FIELDS
    torch.manual_seed(3)
    m = PureAttention(8, 6, 5, 20, encoder_dim=12, dropout=0.0)
    H, I = 5, 6 + 12
    w = dict(zip(PARAM_FIELDS, m._scn_view_of_lstm()))
    assert w["decode_step_weight_ia"].shape == (I, 4 * H) and w["decode_step_weight_ha"].shape == (H, 4 * H)
    assert torch.equal(w["decode_step_weight_ib"], torch.ones(1, 4 * H))
    assert torch.equal(w["decode_step_weight_ic"], torch.eye(H).repeat(1, 4))
    u, h = torch.randn(3, I), torch.randn(3, H)
    pre = u @ w["decode_step_weight_ia"] + w["decode_step_bias_ih"] + h @ w["decode_step_weight_ha"] + w["decode_step_bias_hh"]
    cell = m.decode_step
    ref = u @ cell.weight_ih.t() + cell.bias_ih + h @ cell.weight_hh.t() + cell.bias_hh       # torch blocks: i, f, g, o
    i_, f_, g_, o_ = ref.split(H, dim=1)
    assert torch.allclose(pre, torch.cat([i_, f_, o_, g_], dim=1), atol=1e-6)
    c0 = torch.randn(3, H)
    si, sf, so, sc = pre.split(H, dim=1)
    c1 = torch.sigmoid(sf) * c0 + torch.sigmoid(si) * torch.tanh(sc)
    h1 = torch.sigmoid(so) * torch.tanh(c1)
    h_ref, c_ref = cell(u, (h, c0))
    assert torch.allclose(h1, h_ref, atol=1e-6) and torch.allclose(c1, c_ref, atol=1e-6)
    pre.sum().backward()
    assert cell.weight_ih.grad is not None and cell.weight_hh.grad is not None and cell.bias_ih.grad.abs().sum() > 0
    assert w["embedding_weight"] is m.embedding.weight and w["fc_weight"] is m.fc.weight


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """`bench.py --gpus 8` under a 1-rank environment used to run one GPU and report n_gpus 1.
    Now a rank whose WORLD_SIZE differs from --gpus exits with a message, before any GPU call."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1 but --gpus 2" in r.stderr


def test_bench_starts_its_own_ranks_when_launched_bare(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE starts `python -m torch.distributed.run --nproc-per-node N
    bench.py <same args>` as a child (never an exec) and exits with its status."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, **kw):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")


def test_flat_gradient_view_is_stolen_once_and_accumulated_after():
    """ADVICE r2 (conv.py:285 / :410): a backward node that writes a weight gradient straight into the parameter's
    slice of the flat gradient buffer must (1) return a FRESH alias so that AccumulateGrad steals it instead of
    cloning it, (2) use the slice only for the first gradient of a sweep into an empty .grad -- a second backward
    without zero_grad, or a weight used twice in one graph, must ACCUMULATE (old + new), not double the new one."""
    import torch
    from scnattn import conv as CV
    from scnattn.flat import FlatBuffer

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, x):
            ctx.w, ctx.x = w, x
            return (w * x).sum()

        @staticmethod
        def backward(ctx, g):
            out = CV._grad_out(ctx.w)
            out.copy_(ctx.x * g)          # what the wgrad kernel does: overwrite, never accumulate
            return out, None

    w = torch.nn.Parameter(torch.randn(4, 3))
    flat = FlatBuffer([w])
    gv = flat.gviews[0]
    x1, x2 = torch.randn(4, 3), torch.randn(4, 3)
    Fn.apply(w, x1).backward()
    assert w.grad.data_ptr() == gv.data_ptr(), "autograd cloned the flat view instead of stealing the alias"
    assert torch.equal(gv, x1)
    # second backward WITHOUT zero_grad: accumulate
    Fn.apply(w, x2).backward()
    assert torch.allclose(w.grad, x1 + x2)
    # after gather (p.grad IS the flat view) and no zero_grad: still accumulates
    flat.gather()
    assert w.grad.data_ptr() == gv.data_ptr()
    Fn.apply(w, x2).backward()
    assert torch.allclose(w.grad, x1 + 2 * x2)
    # one weight used twice in one graph
    flat.zero_grad()
    (Fn.apply(w, x1) + Fn.apply(w, x2)).backward()
    assert torch.allclose(w.grad, x1 + x2)
    flat.gather()
    assert torch.allclose(gv, x1 + x2)
