"""CPU oracle for ``sample()`` (beam search).  TEST INFRASTRUCTURE ONLY -- see oracle/scnattn_ref.py.

Restates, over a ``dict`` of tensors keyed by the reference's ``state_dict`` names and on top of the pinned
step functions of ``oracle/scnattn_ref.py``:
    models/decoders/attention_scn.py:160-296   -> beam_search(kind="attention_scn")
    models/decoders/pure_scn.py:142-249        -> beam_search(kind="pure_scn")
    models/decoders/pure_attention.py:153-281  -> beam_search(kind="pure_attention")

PARITY UNPINNED for the search itself: the reference's own ``sample`` raises ``IndexError`` on torch >= 1.5
(``top_k_words / vocab_size`` at attention_scn.py:252 is a true division, so the "index" is a float tensor;
SURVEY.md B16), hence no fixture can be generated from it in this image.  This restatement keeps the
reference's control flow line for line and replaces only that division by ``//`` (what the expression meant
on the torch version the reference was written for).  The per-step arithmetic it calls IS pinned
(attention_forward / scn_cell_forward / lstm_cell_forward against tests/golden/*.npz).

One deliberate difference, stated here so the tests can rely on it: the reference ends with
``max(complete_seqs_scores)`` and therefore raises ``ValueError`` when no beam reached <end> within 50 steps;
this function raises the same error, it does not invent a fallback.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import scnattn_ref as R

Tensor = torch.Tensor


def beam_search(kind: str, P: Dict[str, Tensor], beam_size: int, word_map: Dict[str, int], encoder_out: Tensor,
                tag_out: Optional[Tensor], return_all: bool = False):
    """-> (seq, alphas) for the attention decoders, seq for pure_scn (the reference's return values).
    With ``return_all`` also returns every completed (sequence, score) pair, in completion order."""
    use_att = kind in ("attention_scn", "pure_attention")
    use_tags = kind in ("attention_scn", "pure_scn")
    k = beam_size
    vocab_size = len(word_map)
    enc_image_size = encoder_out.size(1)
    encoder_dim = encoder_out.size(3)
    encoder_out = encoder_out.reshape(1, -1, encoder_dim)
    num_pixels = encoder_out.size(1)
    encoder_out = encoder_out.expand(k, num_pixels, encoder_dim)
    temp_tag_out = tag_out.expand(k, tag_out.size(1)) if use_tags else None
    k_prev_words = torch.LongTensor([[word_map["<start>"]]] * k)
    seqs = k_prev_words
    top_k_scores = torch.zeros(k, 1, dtype=encoder_out.dtype)
    seqs_alpha = torch.ones(k, 1, enc_image_size, enc_image_size, dtype=encoder_out.dtype)
    complete_seqs: List[List[int]] = []
    complete_seqs_alpha: List = []
    complete_seqs_scores: List[float] = []
    step = 1
    mean_enc = encoder_out.mean(dim=1)
    h = F.linear(mean_enc, P["init_h.weight"], P["init_h.bias"])
    c = F.linear(mean_enc, P["init_c.weight"], P["init_c.bias"])
    while True:
        embeddings = F.embedding(k_prev_words, P["embedding.weight"]).squeeze(1)
        if use_att:
            awe, alpha = R.attention_forward(P, "attention.", encoder_out, h)
            alpha = alpha.view(-1, enc_image_size, enc_image_size)
            gate = torch.sigmoid(F.linear(h, P["f_beta.weight"], P["f_beta.bias"]))
            awe = gate * awe
            step_in = torch.cat([embeddings, awe], dim=1)
        else:
            alpha = None
            step_in = embeddings
        if kind == "pure_attention":
            h, c = R.lstm_cell_forward(P, "decode_step.", step_in, (h, c))
        else:
            h, c = R.scn_cell_forward(P, "decode_step.", step_in, temp_tag_out, (h, c))
        scores = F.linear(h, P["fc.weight"], P["fc.bias"])
        scores = F.log_softmax(scores, dim=1)
        scores = top_k_scores.expand_as(scores) + scores
        if step == 1:
            top_k_scores, top_k_words = scores[0].topk(k, 0, True, True)
        else:
            top_k_scores, top_k_words = scores.view(-1).topk(k, 0, True, True)
        prev_word_inds = top_k_words // vocab_size      # the reference's `/` (see the module docstring)
        next_word_inds = top_k_words % vocab_size
        seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], dim=1)
        if use_att:
            seqs_alpha = torch.cat([seqs_alpha[prev_word_inds], alpha[prev_word_inds].unsqueeze(1)], dim=1)
        incomplete_inds = [ind for ind, next_word in enumerate(next_word_inds.tolist())
                           if next_word != word_map["<end>"]]
        complete_inds = sorted(set(range(len(next_word_inds))) - set(incomplete_inds))
        if len(complete_inds) > 0:
            complete_seqs.extend(seqs[complete_inds].tolist())
            if use_att:
                complete_seqs_alpha.extend(seqs_alpha[complete_inds].tolist())
            complete_seqs_scores.extend(top_k_scores[complete_inds].tolist())
        k -= len(complete_inds)
        if k == 0:
            break
        seqs = seqs[incomplete_inds]
        if use_att:
            seqs_alpha = seqs_alpha[incomplete_inds]
        h = h[prev_word_inds[incomplete_inds]]
        c = c[prev_word_inds[incomplete_inds]]
        encoder_out = encoder_out[prev_word_inds[incomplete_inds]]
        if use_tags:
            temp_tag_out = temp_tag_out[prev_word_inds[incomplete_inds]]
        top_k_scores = top_k_scores[incomplete_inds].unsqueeze(1)
        k_prev_words = next_word_inds[incomplete_inds].unsqueeze(1)
        if step > 50:
            break
        step += 1
    i = complete_seqs_scores.index(max(complete_seqs_scores))
    seq = complete_seqs[i]
    out = (seq, complete_seqs_alpha[i]) if use_att else seq
    if return_all:
        return out, list(zip(complete_seqs, complete_seqs_scores))
    return out
