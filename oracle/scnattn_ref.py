"""CPU oracle for the SCN+Attention training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file, and only as the checker / timed baseline.
The shipped modules (``indonesian-image-captioning_amd/models/...``) never import it and
raise if the HIP library is missing.

What it is: an op-for-op, *un-hoisted* eager-PyTorch restatement of the
reference's arithmetic (same operation order, same loop structure, so that in
fp32 on CPU it reproduces the reference bit for bit), written as pure functions
over a ``dict`` of tensors keyed by the reference's ``state_dict`` names.

Pinned: ``oracle/gen_golden.py`` imports the reference's own modules in the
build container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors (outputs and all gradients).
The encoder (ResNet-152) lives in third-party torchvision which is absent from
the image, so that boundary is "parity unpinned" (see DESIGN.md).

Reference files followed (paths relative to the reference checkout):
  models/scn_cell.py:52-154          -> scn_cell_forward
  utils/tensor.py:1-42               -> _gate_blocks_1d / _gate_blocks_2d
  models/attention.py:26-44          -> attention_forward
  models/decoders/attention_scn.py:82-158 -> attention_scn_forward
  models/decoders/pure_scn.py:87-140 -> pure_scn_forward
  models/decoders/pure_attention.py:90-151 -> pure_attention_forward
  trains/attention_scn.py:219-252    -> caption_loss / train_step
  utils/optimizer.py:1-11            -> clamp_gradients
  utils/metric.py:25-39              -> topk_accuracy
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------
# utils/tensor.py:1-42 -- the 4-way split that fixes gate order i, f, o, c
# --------------------------------------------------------------------------
def _gate_blocks_1d(v: Tensor, width: int) -> List[Tensor]:
    return [v[g * width:(g + 1) * width] if g < 3 else v[3 * width:] for g in range(4)]


def _gate_blocks_2d(w: Tensor, width: int) -> List[Tensor]:
    # reference splits along dim 1 (front=False is the only mode it ever uses)
    return [w[:, g * width:(g + 1) * width] if g < 3 else w[:, 3 * width:] for g in range(4)]


# --------------------------------------------------------------------------
# models/scn_cell.py:52-154
# --------------------------------------------------------------------------
def scn_cell_forward(P: Params, pre: str, u: Tensor, s: Tensor,
                     hx: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """One SCN LSTM step.  ``pre`` is the state-dict prefix ('' or 'decode_step.')."""
    Wa, Wb, Wc = P[pre + "weight_ia"], P[pre + "weight_ib"], P[pre + "weight_ic"]
    Ha, Hb, Hc = P[pre + "weight_ha"], P[pre + "weight_hb"], P[pre + "weight_hc"]
    F_ = Wa.shape[1] // 4
    H = Wc.shape[0]
    if u.size(1) != Wa.shape[0]:
        raise RuntimeError("input has inconsistent input_size: got {}, expected {}".format(u.size(1), Wa.shape[0]))
    a_blk, b_blk, c_blk = _gate_blocks_2d(Wa, F_), _gate_blocks_2d(Wb, F_), _gate_blocks_2d(Wc, F_)
    bi = _gate_blocks_1d(P[pre + "bias_ih"], H)
    # x side (scn_cell.py:73-91); unsqueeze/squeeze(0) are no-ops for 2-D input
    x = []
    for g in range(4):
        t1 = u @ a_blk[g]
        t2 = (s @ b_blk[g]).unsqueeze(0)
        x.append((((t1 * t2) @ c_blk[g].t()) + bi[g]).squeeze(0))
    if hx is None:
        z = u.new_zeros(u.size(0), H)
        hx = (z, z)
    h_, c_ = hx
    for lab, st in (("[0]", h_), ("[1]", c_)):
        if x[0].size(0) != st.size(0):
            raise RuntimeError("Input batch size {} doesn't match hidden{} batch size {}".format(
                x[0].size(0), lab, st.size(0)))
        if st.size(1) != H:
            raise RuntimeError("hidden{} has inconsistent hidden_size: got {}, expected {}".format(
                lab, st.size(1), H))
    # h side (scn_cell.py:123-154)
    ha_blk, hb_blk, hc_blk = _gate_blocks_2d(Ha, F_), _gate_blocks_2d(Hb, F_), _gate_blocks_2d(Hc, F_)
    bh = _gate_blocks_1d(P[pre + "bias_hh"], H)
    r = []
    for g in range(4):
        pre_g = (h_ @ ha_blk[g]) * (s @ hb_blk[g])
        r.append((pre_g @ hc_blk[g].t()) + x[g] + bh[g])
    i = torch.sigmoid(r[0])
    f = torch.sigmoid(r[1])
    o = torch.sigmoid(r[2])
    chat = torch.tanh(r[3])
    c = f * c_ + i * chat
    h = o * torch.tanh(c)
    return h, c


# --------------------------------------------------------------------------
# models/attention.py:26-44
# --------------------------------------------------------------------------
RELU_PROBE = None   # tests: a list that collects min |att1 + att2| of every attention step (how far the
                    # pre-activations stay from the ReLU's kink); never changes the arithmetic


def _relu(x: Tensor) -> Tensor:
    if RELU_PROBE is not None:
        RELU_PROBE.append(float(x.detach().abs().min()))
    return torch.relu(x)


def attention_forward(P: Params, pre: str, enc: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
    att1 = F.linear(enc, P[pre + "encoder_att.weight"], P[pre + "encoder_att.bias"])
    att2 = F.linear(h, P[pre + "decoder_att.weight"], P[pre + "decoder_att.bias"])
    att = F.linear(_relu(att1 + att2.unsqueeze(1)),
                   P[pre + "full_att.weight"], P[pre + "full_att.bias"]).squeeze(2)
    alpha = torch.softmax(att, dim=1)
    awe = (enc * alpha.unsqueeze(2)).sum(dim=1)
    return awe, alpha


# --------------------------------------------------------------------------
# torch.nn.LSTMCell (used by PureAttention, pure_attention.py:40-41).
# Gate order of torch's LSTMCell is i, f, g, o (documented by PyTorch).
# --------------------------------------------------------------------------
def lstm_cell_forward(P: Params, pre: str, x: Tensor, hx: Tuple[Tensor, Tensor]) -> Tuple[Tensor, Tensor]:
    h_, c_ = hx
    gates = F.linear(x, P[pre + "weight_ih"], P[pre + "bias_ih"]) + \
        F.linear(h_, P[pre + "weight_hh"], P[pre + "bias_hh"])
    i, f, g, o = gates.chunk(4, dim=1)
    c = torch.sigmoid(f) * c_ + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


# --------------------------------------------------------------------------
# decoders
# --------------------------------------------------------------------------
def _prepare(P: Params, enc: Tensor, caps: Tensor, caplens: Tensor, sort_ind: Optional[Tensor]):
    """attention_scn.py:109-131: flatten, sort by length (desc), permute enc+caps, embed, init state."""
    B = enc.size(0)
    E = enc.size(-1)
    enc = enc.reshape(B, -1, E)
    lens = caplens.squeeze(1)
    if sort_ind is None:
        lens, sort_ind = lens.sort(dim=0, descending=True)
    else:  # injected permutation (the sort is not declared stable)
        lens = lens[sort_ind]
    enc = enc[sort_ind]
    caps = caps[sort_ind]
    emb = F.embedding(caps, P["embedding.weight"])
    mean_enc = enc.mean(dim=1)
    h = F.linear(mean_enc, P["init_h.weight"], P["init_h.bias"])
    c = F.linear(mean_enc, P["init_c.weight"], P["init_c.bias"])
    decode_lengths = (lens - 1).tolist()
    return enc, caps, emb, h, c, decode_lengths, sort_ind


def _drop(h: Tensor, t: int, drop_mask: Optional[Tensor]) -> Tensor:
    """Dropout between h and fc (attention_scn.py:154) with an injected, pre-scaled mask (B,T,D)."""
    if drop_mask is None:
        return h
    return h * drop_mask[:h.size(0), t, :]


def attention_scn_forward(P: Params, enc: Tensor, tags: Tensor, caps: Tensor, caplens: Tensor,
                          drop_mask: Optional[Tensor] = None, sort_ind: Optional[Tensor] = None,
                          hoist: bool = False):
    """models/decoders/attention_scn.py:95-158.  Tags are NOT permuted (quirk Q1, line 152).

    ``hoist=False`` is the reference's formulation (encoder_att recomputed every step);
    ``hoist=True`` only moves the time-invariant ``encoder_att`` GEMM out of the loop (used to make
    the CPU baseline affordable at full size -- mathematically identical, same per-element ops).
    """
    enc, caps, emb, h, c, dl, sort_ind = _prepare(P, enc, caps, caplens, sort_ind)
    B, Pn = enc.size(0), enc.size(1)
    V = P["fc.weight"].size(0)
    T = max(dl)
    preds = enc.new_zeros(B, T, V)
    alphas = enc.new_zeros(B, T, Pn)
    att1_all = None
    if hoist:
        att1_all = F.linear(enc, P["attention.encoder_att.weight"], P["attention.encoder_att.bias"])
    for t in range(T):
        bt = sum(l > t for l in dl)
        if hoist:
            att2 = F.linear(h[:bt], P["attention.decoder_att.weight"], P["attention.decoder_att.bias"])
            att = F.linear(_relu(att1_all[:bt] + att2.unsqueeze(1)),
                           P["attention.full_att.weight"], P["attention.full_att.bias"]).squeeze(2)
            alpha = torch.softmax(att, dim=1)
            awe = (enc[:bt] * alpha.unsqueeze(2)).sum(dim=1)
        else:
            awe, alpha = attention_forward(P, "attention.", enc[:bt], h[:bt])
        gate = torch.sigmoid(F.linear(h[:bt], P["f_beta.weight"], P["f_beta.bias"]))
        awe = gate * awe
        h, c = scn_cell_forward(P, "decode_step.", torch.cat([emb[:bt, t, :], awe], dim=1),
                                tags[:bt, :], (h[:bt], c[:bt]))
        preds[:bt, t, :] = F.linear(_drop(h, t, drop_mask), P["fc.weight"], P["fc.bias"])
        alphas[:bt, t, :] = alpha
    return preds, caps, dl, alphas, sort_ind


def pure_scn_forward(P: Params, enc: Tensor, tags: Tensor, caps: Tensor, caplens: Tensor,
                     drop_mask: Optional[Tensor] = None, sort_ind: Optional[Tensor] = None):
    """models/decoders/pure_scn.py:87-140 (4-tuple, no alphas; same un-permuted-tags quirk, line 135)."""
    enc, caps, emb, h, c, dl, sort_ind = _prepare(P, enc, caps, caplens, sort_ind)
    B = enc.size(0)
    V = P["fc.weight"].size(0)
    T = max(dl)
    preds = enc.new_zeros(B, T, V)
    for t in range(T):
        bt = sum(l > t for l in dl)
        h, c = scn_cell_forward(P, "decode_step.", emb[:bt, t, :], tags[:bt, :], (h[:bt], c[:bt]))
        preds[:bt, t, :] = F.linear(_drop(h, t, drop_mask), P["fc.weight"], P["fc.bias"])
    return preds, caps, dl, sort_ind


def pure_attention_forward(P: Params, enc: Tensor, caps: Tensor, caplens: Tensor,
                           drop_mask: Optional[Tensor] = None, sort_ind: Optional[Tensor] = None):
    """models/decoders/pure_attention.py:90-151 (nn.LSTMCell instead of SCNCell, no tags)."""
    enc, caps, emb, h, c, dl, sort_ind = _prepare(P, enc, caps, caplens, sort_ind)
    B, Pn = enc.size(0), enc.size(1)
    V = P["fc.weight"].size(0)
    T = max(dl)
    preds = enc.new_zeros(B, T, V)
    alphas = enc.new_zeros(B, T, Pn)
    for t in range(T):
        bt = sum(l > t for l in dl)
        awe, alpha = attention_forward(P, "attention.", enc[:bt], h[:bt])
        gate = torch.sigmoid(F.linear(h[:bt], P["f_beta.weight"], P["f_beta.bias"]))
        awe = gate * awe
        h, c = lstm_cell_forward(P, "decode_step.", torch.cat([emb[:bt, t, :], awe], dim=1),
                                 (h[:bt], c[:bt]))
        preds[:bt, t, :] = F.linear(_drop(h, t, drop_mask), P["fc.weight"], P["fc.bias"])
        alphas[:bt, t, :] = alpha
    return preds, caps, dl, alphas, sort_ind


# --------------------------------------------------------------------------
# trains/attention_scn.py:219-235 -- loss
# --------------------------------------------------------------------------
def pack_rows(x: Tensor, lengths: Sequence[int]) -> Tensor:
    """``pack_padded_sequence(x, lengths, batch_first=True).data`` for length-sorted rows:
    time-major concatenation of the active rows of every step."""
    T = max(lengths)
    out = []
    for t in range(T):
        bt = sum(l > t for l in lengths)
        out.append(x[:bt, t])
    return torch.cat(out, dim=0)


def caption_loss(preds: Tensor, caps_sorted: Tensor, decode_lengths: Sequence[int],
                 alphas: Optional[Tensor], alpha_c: float = 1.0):
    targets = caps_sorted[:, 1:]
    scores = pack_rows(preds, decode_lengths)
    tgt = pack_rows(targets, decode_lengths)
    loss = F.cross_entropy(scores, tgt)
    if alphas is not None:
        loss = loss + alpha_c * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    return loss, scores, tgt


def topk_accuracy(scores: Tensor, targets: Tensor, k: int) -> float:
    """utils/metric.py:25-39."""
    n = targets.size(0)
    _, ind = scores.topk(k, 1, True, True)
    correct = ind.eq(targets.view(-1, 1).expand_as(ind))
    return correct.view(-1).float().sum().item() * (100.0 / n)


# --------------------------------------------------------------------------
# utils/optimizer.py:1-11 + torch.optim.Adam defaults (trains/attention_scn.py:91-96, 244-252)
# --------------------------------------------------------------------------
def clamp_gradients(grads: Dict[str, Tensor], clip: float) -> None:
    for g in grads.values():
        if g is not None:
            g.clamp_(-clip, clip)


def adam_step(P: Params, grads: Dict[str, Tensor], state: Dict[str, Dict[str, Tensor]],
              lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam (weight_decay 0, amsgrad False) restated: in-place update of P."""
    b1, b2 = betas
    for k, g in grads.items():
        if g is None:
            continue
        st = state.setdefault(k, {"m": torch.zeros_like(P[k]), "v": torch.zeros_like(P[k])})
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        P[k].addcdiv_(st["m"], denom, value=-(lr / bc1))


def decoder_train_step(kind: str, P: Params, enc: Tensor, tags: Optional[Tensor], caps: Tensor,
                       caplens: Tensor, adam_state: Dict, step: int, lr: float = 4e-4,
                       grad_clip: float = 5.0, alpha_c: float = 1.0,
                       drop_mask: Optional[Tensor] = None, sort_ind: Optional[Tensor] = None,
                       enc_requires_grad: bool = False, hoist: bool = False):
    """One decoder-side train step, trains/attention_scn.py:215-252 (encoder handled by the caller).
    Returns (loss, grads-after-clamp, d loss / d enc or None).  Updates P in place."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    enc_l = enc.detach().clone().requires_grad_(enc_requires_grad)
    if kind == "attention_scn":
        preds, caps_s, dl, alphas, _ = attention_scn_forward(leaves, enc_l, tags, caps, caplens,
                                                             drop_mask, sort_ind, hoist=hoist)
    elif kind == "pure_scn":
        preds, caps_s, dl, _ = pure_scn_forward(leaves, enc_l, tags, caps, caplens, drop_mask, sort_ind)
        alphas = None
    elif kind == "pure_attention":
        preds, caps_s, dl, alphas, _ = pure_attention_forward(leaves, enc_l, caps, caplens, drop_mask, sort_ind)
    else:
        raise ValueError(kind)
    loss, _, _ = caption_loss(preds, caps_s, dl, alphas, alpha_c)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else None) for k, v in leaves.items()}
    denc = enc_l.grad
    clamp_gradients(grads, grad_clip)
    with torch.no_grad():
        adam_step(P, grads, adam_state, lr, step)
    return loss.detach(), grads, denc


# --------------------------------------------------------------------------
# AdaptiveAvgPool2d(14) on an 8x8 grid + NHWC permute (models/encoders/caption.py:41-43).
# torch's adaptive pooling window for output i over input n->m is
# [floor(i*n/m), ceil((i+1)*n/m)); restated explicitly so the fused HIP kernel has an oracle.
# --------------------------------------------------------------------------
def adaptive_pool_matrix(n_in: int, n_out: int, dtype=torch.float32) -> Tensor:
    U = torch.zeros(n_out, n_in, dtype=dtype)
    for i in range(n_out):
        lo = (i * n_in) // n_out
        hi = -((-(i + 1) * n_in) // n_out)
        U[i, lo:hi] = 1.0 / (hi - lo)
    return U


def pool_permute(x: Tensor, n_out: int = 14) -> Tensor:
    """(B,C,h,w) -> (B,n_out,n_out,C) == adaptive_avg_pool2d(x, n_out).permute(0,2,3,1)."""
    return F.adaptive_avg_pool2d(x, (n_out, n_out)).permute(0, 2, 3, 1)
