"""Pins oracle/scnattn_ref.py against vectors produced by the reference's own modules
(tests/golden/*.npz, written by oracle/gen_golden.py).  CPU only.  Tolerance: 1e-6 abs on O(1)
values (the restatement is op-for-op, so in practice the match is bit-exact or 1 ulp)."""
import numpy as np
import pytest
import torch

from oracle import scnattn_ref as R
from helpers import load_golden, params_from, t

ATOL = 1e-6


def _close(a, b, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=atol)


def test_scn_cell_forward_backward():
    d = load_golden("scn_cell")
    P = {k: v.requires_grad_(True) for k, v in params_from(d).items()}
    u, s = t(d["u"]).requires_grad_(True), t(d["s"]).requires_grad_(True)
    h0, c0 = t(d["h0"]).requires_grad_(True), t(d["c0"]).requires_grad_(True)
    h, c = R.scn_cell_forward(P, "", u, s, (h0, c0))
    _close(h, d["h"]); _close(c, d["c"])
    ((h * t(d["wh"])).sum() + (c * t(d["wc"])).sum()).backward()
    _close(u.grad, d["du"]); _close(s.grad, d["ds"]); _close(h0.grad, d["dh0"]); _close(c0.grad, d["dc0"])
    for k, v in P.items():
        _close(v.grad, d["g." + k])
    h2, c2 = R.scn_cell_forward(P, "", u.detach(), s.detach(), None)
    _close(h2, d["h_none"]); _close(c2, d["c_none"])


def test_scn_cell_error_messages():
    d = load_golden("scn_cell")
    P = params_from(d)
    B, I = d["u"].shape
    H = d["h0"].shape[1]
    s = t(d["s"])
    msgs = [str(m) for m in d["errors"]]
    with pytest.raises(RuntimeError) as e:
        R.scn_cell_forward(P, "", torch.randn(B, I + 1), s)
    assert str(e.value) == msgs[0]
    with pytest.raises(RuntimeError) as e:
        R.scn_cell_forward(P, "", t(d["u"]), s, (torch.randn(B + 1, H), torch.randn(B + 1, H)))
    assert str(e.value) == msgs[1]
    with pytest.raises(RuntimeError) as e:
        R.scn_cell_forward(P, "", t(d["u"]), s, (torch.randn(B, H + 1), torch.randn(B, H + 1)))
    assert str(e.value) == msgs[2]


def test_attention_forward_backward():
    d = load_golden("attention")
    P = {k: v.requires_grad_(True) for k, v in params_from(d).items()}
    enc, h = t(d["enc"]).requires_grad_(True), t(d["h"]).requires_grad_(True)
    awe, alpha = R.attention_forward(P, "", enc, h)
    _close(awe, d["awe"]); _close(alpha, d["alpha"])
    np.testing.assert_allclose(alpha.sum(1).detach().numpy(), 1.0, atol=1e-6)   # B9
    ((awe * t(d["w_awe"])).sum() + (alpha * t(d["w_alpha"])).sum()).backward()
    _close(enc.grad, d["denc"]); _close(h.grad, d["dh"])
    for k, v in P.items():
        _close(v.grad, d["g." + k])


def _run_decoder(kind, d, sort_ind=None, hoist=False):
    P = {k: v.requires_grad_(True) for k, v in params_from(d).items()}
    enc = t(d["enc"]).requires_grad_(True)
    tags, caps, caplens = t(d["tags"]), t(d["caps"]), t(d["caplens"])
    if kind == "attention_scn":
        preds, caps_s, dl, alphas, si = R.attention_scn_forward(P, enc, tags, caps, caplens,
                                                                sort_ind=sort_ind, hoist=hoist)
    elif kind == "pure_scn":
        preds, caps_s, dl, si = R.pure_scn_forward(P, enc, tags, caps, caplens, sort_ind=sort_ind)
        alphas = None
    else:
        preds, caps_s, dl, alphas, si = R.pure_attention_forward(P, enc, caps, caplens, sort_ind=sort_ind)
    return P, enc, preds, caps_s, dl, alphas, si


@pytest.mark.parametrize("name,kind,inject", [
    ("attention_scn_distinct", "attention_scn", False),
    ("attention_scn_tied", "attention_scn", True),
    ("attention_scn_full", "attention_scn", True),
    ("attention_scn_odd", "attention_scn", True),
    ("pure_scn_distinct", "pure_scn", False),
    ("pure_attention_distinct", "pure_attention", False),
])
def test_decoder_forward_loss_backward(name, kind, inject):
    d = load_golden(name)
    si_in = t(d["sort_ind"]) if inject else None
    P, enc, preds, caps_s, dl, alphas, si = _run_decoder(kind, d, si_in)
    _close(preds, d["preds"])
    assert np.array_equal(caps_s.numpy(), d["caps_sorted"])
    assert list(dl) == list(d["decode_lengths"])
    assert np.array_equal(si.numpy(), d["sort_ind"])
    if alphas is not None:
        _close(alphas, d["alphas"])
    # B8: rows beyond decode length stay exactly zero
    for b, l in enumerate(dl):
        assert preds[b, l:].abs().max().item() == 0.0 if l < preds.size(1) else True
    loss, sc, tg = R.caption_loss(preds, caps_s, dl, alphas, 1.0)
    _close(sc, d["packed_scores"]); assert np.array_equal(tg.numpy(), d["packed_targets"])
    _close(loss, d["loss"])
    loss.backward()
    for k, v in P.items():
        key = "g_raw." + k
        if key in d:
            _close(v.grad, d[key], atol=2e-6)
    _close(enc.grad, d["denc"], atol=2e-6)


def test_hoisted_encoder_att_is_equivalent():
    d = load_golden("attention_scn_distinct")
    _, _, preds, _, _, alphas, _ = _run_decoder("attention_scn", d, hoist=True)
    _close(preds, d["preds"], atol=2e-6); _close(alphas, d["alphas"], atol=2e-6)


def test_q1_tags_not_permuted():
    """Quirk Q1 (attention_scn.py:152): pre-sorting tags changes the result."""
    d = load_golden("attention_scn_distinct")
    P = params_from(d)
    enc, tags, caps, caplens = t(d["enc"]), t(d["tags"]), t(d["caps"]), t(d["caplens"])
    si = t(d["sort_ind"])
    p1 = R.attention_scn_forward(P, enc, tags, caps, caplens)[0]
    p2 = R.attention_scn_forward(P, enc, tags[si], caps, caplens)[0]
    assert (p1 - p2).abs().max().item() > 1e-4


@pytest.mark.parametrize("name,kind", [("attention_scn_distinct", "attention_scn"),
                                       ("pure_scn_distinct", "pure_scn"),
                                       ("pure_attention_distinct", "pure_attention")])
def test_train_step_clamp_adam(name, kind):
    d = load_golden(name)
    P = params_from(d)
    state = {}
    args = dict(enc=t(d["enc"]), tags=t(d["tags"]), caps=t(d["caps"]), caplens=t(d["caplens"]))
    loss, grads, _ = R.decoder_train_step(kind, P, adam_state=state, step=1, **args)
    _close(loss, d["loss"])
    for k, g in grads.items():
        key = "g_clamped." + k
        if key in d:
            _close(g, d[key], atol=2e-6)
            assert g.abs().max().item() <= 5.0
    for k in P:
        _close(P[k], d["p_after." + k], atol=2e-6)
    loss2, _, _ = R.decoder_train_step(kind, P, adam_state=state, step=2, **args)
    _close(loss2, d["loss2"], atol=5e-6)
    for k in P:
        _close(P[k], d["p_after2." + k], atol=5e-6)


def test_topk_accuracy():
    d = load_golden("attention_scn_distinct")
    acc = R.topk_accuracy(t(d["packed_scores"]), t(d["packed_targets"]), 5)
    assert abs(acc - float(d["top5"])) < 1e-9


def test_adaptive_pool_matrix_matches_torch():
    x = torch.randn(2, 5, 8, 8)
    U = R.adaptive_pool_matrix(8, 14)
    y = torch.einsum("ih,bchw,jw->bijc", U, x, U)
    _close(y, R.pool_permute(x, 14).numpy(), atol=1e-6)
    assert np.allclose(U.sum(1).numpy(), 1.0)


@pytest.mark.parametrize("name,kind", [("attention_scn_odd", "attention_scn"), ("pure_scn_distinct", "pure_scn"),
                                       ("pure_attention_distinct", "pure_attention")])
def test_beam_search_oracle_is_stable_across_precisions(name, kind):
    """oracle/beam_ref.py (the reference's sample() with `//`): on the golden weights, sharpened as the GPU test
    does, fp32 and fp64 pick the same beams in the same completion order -- i.e. the discrete search is not decided
    by rounding on these fixtures, so an fp32 HIP implementation can be compared with it sequence for sequence.
    Structure: sequences start with <start>, end with <end>, completion scores are non-increasing per step."""
    from oracle import beam_ref as BR
    d = dict(load_golden(name))
    V = d["p.embedding.weight"].shape[0]
    d["p.fc.weight"] = d["p.fc.weight"] * 10.0
    fb = d["p.fc.bias"].copy()
    fb[V - 1] += 0.2
    d["p.fc.bias"] = fb
    wm = {"<pad>": 0, "<unk>": V - 3, "<start>": V - 2, "<end>": V - 1}
    for i in range(1, V - 3):
        wm["w%d" % i] = i
    done = 0
    for b in range(d["enc"].shape[0]):
        for k in (1, 3, 5):
            res = []
            for dt in (torch.float32, torch.float64):
                P = params_from(d, dtype=dt)
                enc = t(d["enc"], dt)[b:b + 1]
                tags = t(d["tags"], dt)[b:b + 1] if kind != "pure_attention" else None
                try:
                    out, allc = BR.beam_search(kind, P, k, wm, enc, tags, return_all=True)
                except ValueError:
                    res.append(None)
                    continue
                seq = out if kind == "pure_scn" else out[0]
                assert seq[0] == V - 2 and seq[-1] == V - 1 and len(allc) <= k
                assert all(s[0] == V - 2 and s[-1] == V - 1 for s, _ in allc)
                if kind != "pure_scn":
                    assert len(out[1]) == len(seq)
                res.append([s for s, _ in allc])
            assert res[0] == res[1], (b, k)
            done += res[1] is not None
    assert done >= 4
