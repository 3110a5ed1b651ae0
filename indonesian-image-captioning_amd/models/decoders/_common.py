"""Logic shared by the three caption decoders (the reference repeats it in each file):
length sort + permutation (attention_scn.py:115-131), the dropout mask that sits between h and fc
(:154), and beam search (:160-296)."""
import torch
import torch.nn.functional as F

from scnattn import functional as SF
from utils.token import start_token, end_token


def sort_by_length(encoder_out, encoded_captions, caption_lengths, sort_ind=None, caplens_host=None):
    """Flatten pixels, sort rows by caption length (descending) and permute images + captions.
    Tags are deliberately NOT touched (the reference indexes the un-permuted tags, :152).
    Returns enc (B,P,E), caps (B,L), decode_lengths (list[int]), sort_ind.
    `caplens_host`: the same caption lengths as a CPU tensor, when the caller has them there anyway (a loader reads
    them from a JSON file): the Python list the loop bounds come from is then computed on the host and the forward
    pass has NO device synchronisation at all -- the host keeps running ahead of the GPU across the encoder / decoder
    boundary.  Without it: one sync, exactly where the reference has its `.tolist()` (:131)."""
    B, E = encoder_out.size(0), encoder_out.size(-1)
    enc = encoder_out.reshape(B, -1, E)
    lens = caption_lengths.squeeze(1)
    sort_given = sort_ind is not None
    if not sort_given:
        lens, sort_ind = lens.sort(dim=0, descending=True, stable=True)
    else:
        lens = lens[sort_ind]
    if caplens_host is not None and not sort_given:
        lh, ph = caplens_host.reshape(-1).cpu().sort(dim=0, descending=True, stable=True)    # same stable order as on the device
        decode_lengths, perm = (lh - 1).tolist(), ph.tolist()
    else:
        # the one host sync of the forward pass, as in the reference (`.tolist()`, :131); the permutation rides
        # along so that a batch that is already in order (fixed-length captions, or a loader that sorts) skips
        # the 51 MB gather of encoder_out and, when the encoder is fine-tuned, its scatter in the backward pass
        host = torch.stack([lens - 1, sort_ind]).tolist()
        decode_lengths, perm = host
    if perm != list(range(B)):
        enc = enc[sort_ind]
        encoded_captions = encoded_captions[sort_ind]
    return enc, encoded_captions, decode_lengths, (lens - 1).to(torch.int32), sort_ind


USE_PREPOOL = True     # set False to force the dense (B, P, E) path even when the encoder attached its source map


def attached_prepool(encoder_out):
    """The un-pooled trunk output (B, h, w, E) that `EncoderCaption.forward` attaches to the pooled tensor it
    returns, if `encoder_out` is still that very tensor, unmodified (same object, same version counter)."""
    if not USE_PREPOOL or encoder_out is None:
        return None
    tag = getattr(encoder_out, "_scn_prepool", None)
    if tag is None:
        return None
    pre, version = tag
    if version != encoder_out._version or pre.shape[0] != encoder_out.shape[0] or pre.shape[-1] != encoder_out.shape[-1]:
        return None
    return pre


def resolve_prepool(encoder_out, prepool, pool_size, attention_dim=None, who="forward"):
    """-> (tensor to sort/permute, PoolTaps or None).  The trunk map (B, h, w, E) is used instead of the pooled
    encoder_out when it was given (or attached by EncoderCaption), lives on the GPU, has vector-friendly widths
    and the pooling is an up-sampling one (windows of at most 2x2); otherwise the dense path is taken."""
    pre = prepool if prepool is not None else attached_prepool(encoder_out)
    pool = None
    if pre is not None and pre.is_cuda and pre.dim() == 4 and pre.shape[-1] % 4 == 0 \
            and (attention_dim is None or attention_dim % 4 == 0):
        out_hw = tuple(encoder_out.shape[1:3]) if encoder_out is not None else (pool_size, pool_size)
        try:
            pool = SF.pool_taps(pre.shape[1], pre.shape[2], out_hw[0], out_hw[1], pre.device)
        except ValueError:
            pool = None
    if pool is None and encoder_out is None:
        raise RuntimeError("%s: encoder_out is None and no usable prepool map was given" % who)
    return (pre if pool is not None else encoder_out), pool


def active_rows(decode_lengths):
    return [sum(l > t for l in decode_lengths) for t in range(max(decode_lengths))]


def make_drop_mask(module, B, T, D, device):
    """Pre-scaled Bernoulli mask for fc(dropout(h)) or None (eval mode / p == 0).  A test can pin the
    mask by setting ``module.drop_mask_override`` (B,T,D)."""
    override = getattr(module, "drop_mask_override", None)
    if override is not None:
        return override.to(device=device, dtype=torch.float32)
    p = module.dropout.p
    if not module.training or p <= 0.0:
        return None
    keep = 1.0 - p
    return torch.empty((B, T, D), device=device, dtype=torch.float32).bernoulli_(keep).div_(keep)


def beam_search(decoder, beam_size, word_map, encoder_out, tag_out, use_attention, use_tags, return_all=False):
    """Beam search shared by AttentionSCN / PureSCN / PureAttention ``sample`` (reference
    attention_scn.py:160-296, pure_scn.py:142-249, pure_attention.py:153-281).  Identical control flow,
    with the unrolled-index split done by FLOOR division (the reference's ``/`` breaks on torch >= 1.5).
    ``return_all`` (tests): also return every completed (sequence, score) pair in completion order."""
    k = beam_size
    vocab_size = len(word_map)
    dev = encoder_out.device
    enc_image_size = encoder_out.size(1)
    encoder_dim = encoder_out.size(3)
    encoder_out = encoder_out.reshape(1, -1, encoder_dim)
    num_pixels = encoder_out.size(1)
    encoder_out = encoder_out.expand(k, num_pixels, encoder_dim).contiguous()
    tags = tag_out.expand(k, tag_out.size(1)).contiguous() if use_tags else None

    k_prev_words = torch.full((k, 1), word_map[start_token], dtype=torch.long, device=dev)
    seqs = k_prev_words
    top_k_scores = torch.zeros(k, 1, device=dev)
    seqs_alpha = torch.ones(k, 1, enc_image_size, enc_image_size, device=dev)
    complete_seqs, complete_seqs_alpha, complete_seqs_scores = [], [], []
    step = 1
    h, c = decoder.init_hidden_state(encoder_out)
    while True:
        embeddings = decoder.embedding(k_prev_words).squeeze(1)
        if use_attention:
            awe, alpha = decoder.attention(encoder_out, h)
            alpha = alpha.view(-1, enc_image_size, enc_image_size)
            gate = torch.sigmoid(SF.linear(h, decoder.f_beta.weight, decoder.f_beta.bias))
            step_in = torch.cat([embeddings, gate * awe], dim=1)
        else:
            alpha = None
            step_in = embeddings
        if use_tags:
            h, c = decoder.decode_step(step_in, tags, (h, c))
        else:
            h, c = decoder.decode_step(step_in, (h, c))
        scores = F.log_softmax(SF.linear(h, decoder.fc.weight, decoder.fc.bias), dim=1)
        scores = top_k_scores.expand_as(scores) + scores
        if step == 1:
            top_k_scores, top_k_words = scores[0].topk(k, 0, True, True)
        else:
            top_k_scores, top_k_words = scores.view(-1).topk(k, 0, True, True)
        prev_word_inds = torch.div(top_k_words, vocab_size, rounding_mode='floor')
        next_word_inds = top_k_words % vocab_size
        seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], dim=1)
        if use_attention:
            seqs_alpha = torch.cat([seqs_alpha[prev_word_inds], alpha[prev_word_inds].unsqueeze(1)], dim=1)
        nxt = next_word_inds.tolist()
        incomplete_inds = [i for i, w in enumerate(nxt) if w != word_map[end_token]]
        complete_inds = sorted(set(range(len(nxt))) - set(incomplete_inds))
        if complete_inds:
            complete_seqs.extend(seqs[complete_inds].tolist())
            if use_attention:
                complete_seqs_alpha.extend(seqs_alpha[complete_inds].tolist())
            complete_seqs_scores.extend(top_k_scores[complete_inds].tolist())
        k -= len(complete_inds)
        if k == 0:
            break
        seqs = seqs[incomplete_inds]
        if use_attention:
            seqs_alpha = seqs_alpha[incomplete_inds]
        keep = prev_word_inds[incomplete_inds]
        h, c = h[keep], c[keep]
        encoder_out = encoder_out[keep]
        if use_tags:
            tags = tags[keep]
        top_k_scores = top_k_scores[incomplete_inds].unsqueeze(1)
        k_prev_words = next_word_inds[incomplete_inds].unsqueeze(1)
        if step > 50:
            break
        step += 1
    if not complete_seqs_scores:  # nothing reached <end> within 50 steps: fall back to the best open beam
        complete_seqs = seqs.tolist()
        complete_seqs_scores = top_k_scores.view(-1).tolist()
        if use_attention:
            complete_seqs_alpha = seqs_alpha.tolist()
    i = complete_seqs_scores.index(max(complete_seqs_scores))
    seq = complete_seqs[i]
    out = (seq, complete_seqs_alpha[i]) if use_attention else seq
    if return_all:
        return out, list(zip(complete_seqs, complete_seqs_scores))
    return out
