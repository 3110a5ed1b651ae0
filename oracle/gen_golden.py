"""Generate tests/golden/*.npz by running the reference's OWN modules (build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

The reference's decoder-side modules import ``torchvision`` without using it
(models/scn_cell.py:3, models/attention.py:3); torchvision is not installed in this image, so an
empty module object is registered under that name first.  Nothing else is shimmed; the encoder
(which really needs torchvision + a weight download) is never constructed.

Only DATA is written: seeded inputs, the modules' state_dict tensors, outputs and gradients.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("SCNATTN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _import_reference():
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    sys.path.insert(0, REF)
    from models.scn_cell import SCNCell
    from models.attention import Attention
    from models.decoders.attention_scn import AttentionSCN
    from models.decoders.pure_scn import PureSCN
    from models.decoders.pure_attention import PureAttention
    from utils.optimizer import clip_gradient
    from utils.metric import accuracy
    return SCNCell, Attention, AttentionSCN, PureSCN, PureAttention, clip_gradient, accuracy


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def _sd(mod, prefix="p."):
    return {prefix + k: _np(v) for k, v in mod.state_dict().items()}


def _grads(mod, prefix="g."):
    return {prefix + k: _np(p.grad) for k, p in mod.named_parameters() if p.grad is not None}


def gen_scn_cell(SCNCell):
    torch.manual_seed(11)
    B, I, H, S, Fd = 5, 24, 16, 10, 12
    cell = SCNCell(I, H, S, Fd)
    u = torch.randn(B, I, requires_grad=True)
    s = torch.rand(B, S, requires_grad=True)
    h0 = torch.randn(B, H, requires_grad=True)
    c0 = torch.randn(B, H, requires_grad=True)
    h, c = cell(u, s, (h0, c0))
    wh, wc = torch.randn(B, H), torch.randn(B, H)
    ((h * wh).sum() + (c * wc).sum()).backward()
    d = {"u": _np(u), "s": _np(s), "h0": _np(h0), "c0": _np(c0), "wh": _np(wh), "wc": _np(wc),
         "h": _np(h), "c": _np(c), "du": _np(u.grad), "ds": _np(s.grad), "dh0": _np(h0.grad),
         "dc0": _np(c0.grad), "repr": np.array(repr(cell))}
    d.update(_sd(cell))
    d.update(_grads(cell))
    # hx=None variant (scn_cell.py:93-96)
    cell.zero_grad()
    h2, c2 = cell(u.detach(), s.detach())
    d["h_none"], d["c_none"] = _np(h2), _np(c2)
    # error messages (scn_cell.py:169-184)
    msgs = []
    for bad in (lambda: cell(torch.randn(B, I + 1), s.detach()),
                lambda: cell(u.detach(), s.detach(), (torch.randn(B + 1, H), torch.randn(B + 1, H))),
                lambda: cell(u.detach(), s.detach(), (torch.randn(B, H + 1), torch.randn(B, H + 1)))):
        try:
            bad()
            msgs.append("")
        except RuntimeError as e:
            msgs.append(str(e))
    d["errors"] = np.array(msgs)
    _save("scn_cell", d)


def gen_attention(Attention):
    torch.manual_seed(12)
    B, Pn, E, D, A = 3, 9, 32, 16, 16
    att = Attention(E, D, A)
    enc = torch.randn(B, Pn, E, requires_grad=True)
    h = torch.randn(B, D, requires_grad=True)
    awe, alpha = att(enc, h)
    w1, w2 = torch.randn(B, E), torch.randn(B, Pn)
    ((awe * w1).sum() + (alpha * w2).sum()).backward()
    d = {"enc": _np(enc), "h": _np(h), "w_awe": _np(w1), "w_alpha": _np(w2), "awe": _np(awe),
         "alpha": _np(alpha), "denc": _np(enc.grad), "dh": _np(h.grad)}
    d.update(_sd(att))
    d.update(_grads(att))
    _save("attention", d)


def _caps(B, L, V, lens, gen):
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = lens[b]
        caps[b, 0] = V - 2
        caps[b, 1:n - 1] = torch.randint(1, V - 3, (n - 2,), generator=gen)
        caps[b, n - 1] = V - 1
    return caps


def _decoder_case(kind, ctor, name, lens, seed, dims, train_step_also=False, clip_gradient=None,
                  accuracy=None):
    """Forward + loss + backward through the reference decoder, reference-style loss (trains/*.py)."""
    from torch.nn.utils.rnn import pack_padded_sequence
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    B, Pside, E, V, L = dims["B"], dims["Pside"], dims["E"], dims["V"], dims["L"]
    dec = ctor()
    dec.train()  # dropout=0.0 in ctor, so train()==eval() numerically
    enc = torch.randn(B, Pside, Pside, E, generator=g)
    enc.requires_grad_(True)
    tags = torch.rand(B, dims["S"], generator=g)
    caps = _caps(B, L, V, lens, g)
    caplens = torch.tensor(lens, dtype=torch.long).unsqueeze(1)
    if kind == "pure_attention":
        out = dec(enc, caps, caplens)
    else:
        out = dec(enc, tags, caps, caplens)
    if kind == "pure_scn":
        preds, caps_s, dl, sort_ind = out
        alphas = None
    else:
        preds, caps_s, dl, alphas, sort_ind = out
    targets = caps_s[:, 1:]
    sc = pack_padded_sequence(preds, dl, batch_first=True).data
    tg = pack_padded_sequence(targets, dl, batch_first=True).data
    loss = torch.nn.CrossEntropyLoss()(sc, tg)
    if alphas is not None:
        loss = loss + 1.0 * ((1. - alphas.sum(dim=1)) ** 2).mean()
    d = {"enc": _np(enc), "tags": _np(tags), "caps": _np(caps), "caplens": _np(caplens),
         "preds": _np(preds), "caps_sorted": _np(caps_s), "decode_lengths": np.array(dl),
         "sort_ind": _np(sort_ind), "loss": _np(loss), "packed_scores": _np(sc), "packed_targets": _np(tg)}
    if alphas is not None:
        d["alphas"] = _np(alphas)
    d.update(_sd(dec))
    opt = torch.optim.Adam(params=filter(lambda p: p.requires_grad, dec.parameters()), lr=4e-4)
    opt.zero_grad()
    loss.backward()
    d.update(_grads(dec, "g_raw."))
    d["denc"] = _np(enc.grad)
    if train_step_also:
        clip_gradient(opt, 5.0)
        d.update(_grads(dec, "g_clamped."))
        opt.step()
        d.update(_sd(dec, "p_after."))
        d["top5"] = np.array(accuracy(sc, tg, 5))
        # second step on the same batch so Adam's bias correction at step 2 is pinned too
        if kind == "pure_attention":
            out = dec(enc.detach(), caps, caplens)
        else:
            out = dec(enc.detach(), tags, caps, caplens)
        preds2 = out[0]
        sc2 = pack_padded_sequence(preds2, dl, batch_first=True).data
        loss2 = torch.nn.CrossEntropyLoss()(sc2, tg)
        if kind != "pure_scn":
            loss2 = loss2 + 1.0 * ((1. - out[3].sum(dim=1)) ** 2).mean()
        opt.zero_grad()
        loss2.backward()
        clip_gradient(opt, 5.0)
        opt.step()
        d["loss2"] = _np(loss2)
        d.update(_sd(dec, "p_after2."))
    _save(name, d)


def main():
    SCNCell, Attention, AttentionSCN, PureSCN, PureAttention, clip_gradient, accuracy = _import_reference()
    gen_scn_cell(SCNCell)
    gen_attention(Attention)
    dims = dict(B=4, Pside=3, E=32, V=23, L=10, S=10)
    mk_as = lambda: AttentionSCN(attention_dim=16, embed_dim=12, decoder_dim=16, factored_dim=20,
                                 semantic_dim=10, vocab_size=23, encoder_dim=32, dropout=0.0)
    mk_ps = lambda: PureSCN(embed_dim=12, decoder_dim=16, factored_dim=20, semantic_dim=10,
                            vocab_size=23, encoder_dim=32, dropout=0.0)
    mk_pa = lambda: PureAttention(attention_dim=16, embed_dim=12, decoder_dim=16, vocab_size=23,
                                  encoder_dim=32, dropout=0.0)
    # distinct lengths (sort is unambiguous) -- also the one-train-step fixture
    _decoder_case("attention_scn", mk_as, "attention_scn_distinct", [5, 8, 3, 6], 21, dims,
                  True, clip_gradient, accuracy)
    # tied lengths: sort_ind is captured from the reference run (quirk Q2)
    _decoder_case("attention_scn", mk_as, "attention_scn_tied", [7, 7, 4, 7], 22, dims)
    # all rows full length (the benchmark's shape: b_t == B for every t)
    _decoder_case("attention_scn", mk_as, "attention_scn_full", [10, 10, 10, 10], 23, dims)
    _decoder_case("pure_scn", mk_ps, "pure_scn_distinct", [5, 8, 3, 6], 24, dims, True, clip_gradient, accuracy)
    _decoder_case("pure_attention", mk_pa, "pure_attention_distinct", [5, 8, 3, 6], 25, dims,
                  True, clip_gradient, accuracy)
    # a larger, odd-sized AttentionSCN (dims not multiples of the kernels' tiles)
    dims2 = dict(B=6, Pside=4, E=40, V=37, L=12, S=14)
    mk2 = lambda: AttentionSCN(attention_dim=24, embed_dim=20, decoder_dim=28, factored_dim=36,
                               semantic_dim=14, vocab_size=37, encoder_dim=40, dropout=0.0)
    _decoder_case("attention_scn", mk2, "attention_scn_odd", [12, 9, 9, 4, 7, 3], 26, dims2)


if __name__ == "__main__":
    main()
