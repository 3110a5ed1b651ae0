"""AttentionSCN on MI355X: drop-in for the reference's models/decoders/attention_scn.py.

Same constructor, submodule/parameter names (state_dict keys), ``forward`` signature and 5-tuple
return, ``sample``, ``init_hidden_state``, ``init_weights``, ``load_pretrained_embeddings`` and
``fine_tune_embeddings``.  ``forward`` does the length sort on the host exactly like the reference
(:115-131) and then hands the entire teacher-forced recurrence (:142-156), dropout, fc and the
bookkeeping of the shrinking batch to ONE call into libscnattn (scnattn_seq_fwd); its gradient is one
call too (scnattn_seq_bwd).  Outputs are allocated on the input's device (the reference uses an
import-time global device, which breaks one-process-per-GPU data parallelism).
"""
import torch
from torch import nn

from models.attention import Attention
from models.scn_cell import SCNCell
from models.decoders import _common
from scnattn import functional as SF
from scnattn._lib import PARAM_FIELDS


class AttentionSCN(nn.Module):
    def __init__(self, attention_dim, embed_dim, decoder_dim, factored_dim, semantic_dim, vocab_size,
                 encoder_dim=2048, dropout=0.5):
        super().__init__()
        self.attention_dim = attention_dim
        self.embed_dim = embed_dim
        self.encoder_dim = encoder_dim
        self.decoder_dim = decoder_dim
        self.factored_dim = factored_dim
        self.semantic_dim = semantic_dim
        self.vocab_size = vocab_size
        self.attention = Attention(encoder_dim, decoder_dim, attention_dim)
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.dropout = nn.Dropout(p=dropout)
        self.decode_step = SCNCell(embed_dim + encoder_dim, decoder_dim, semantic_dim, factored_dim, bias=True)
        self.init_h = nn.Linear(encoder_dim, decoder_dim)
        self.init_c = nn.Linear(encoder_dim, decoder_dim)
        self.f_beta = nn.Linear(decoder_dim, encoder_dim)
        self.sigmoid = nn.Sigmoid()
        self.fc = nn.Linear(decoder_dim, vocab_size)
        self.init_weights()

    def init_weights(self):
        self.embedding.weight.data.uniform_(-0.1, 0.1)
        self.fc.bias.data.fill_(0)
        self.fc.weight.data.uniform_(-0.1, 0.1)

    def load_pretrained_embeddings(self, embeddings):
        self.embedding.weight = nn.Parameter(embeddings)

    def fine_tune_embeddings(self, fine_tune=True):
        for p in self.embedding.parameters():
            p.requires_grad = fine_tune

    def init_hidden_state(self, encoder_out):
        """(B,P,E) -> h, c (B,D): mean over pixels then two linear maps (reference :82-93)."""
        mean_encoder_out = encoder_out.mean(dim=1)
        h = SF.linear(mean_encoder_out, self.init_h.weight, self.init_h.bias)
        c = SF.linear(mean_encoder_out, self.init_c.weight, self.init_c.bias)
        return h, c

    def forward(self, encoder_out, semantic_input, encoded_captions, caption_lengths, sort_ind=None, prepool=None,
                pool_size=14, caplens_host=None):
        """Reference signature (attention_scn.py:95) plus optional extras.  `prepool`: the encoder trunk's map
        (B, h, w, E) of which `encoder_out` is the AdaptiveAvgPool2d(pool_size) (models/encoders/caption.py:41-43);
        given it -- explicitly, or attached by this build's EncoderCaption to the tensor it returned -- the
        attention runs on the h*w source pixels and the pooled map is not read (same numbers by linearity,
        SURVEY 8d); `encoder_out` may then be None."""
        src, pool = _common.resolve_prepool(encoder_out, prepool, pool_size, self.attention_dim, "AttentionSCN.forward")
        enc, caps, decode_lengths, dl_dev, sort_ind = _common.sort_by_length(
            src, encoded_captions, caption_lengths, sort_ind, caplens_host)
        B, E = enc.shape[0], enc.shape[2]
        P = pool.P if pool is not None else enc.shape[1]
        T = max(decode_lengths)
        dims = (B, P, E, self.attention_dim, self.decoder_dim, self.factored_dim, self.embed_dim,
                self.semantic_dim, self.vocab_size, T, caps.size(1), 1)
        mask = _common.make_drop_mask(self, B, T, self.decoder_dim, enc.device)
        weights = _collect_weights(self)
        predictions, alphas = SF.decoder_sequence(dims, _common.active_rows(decode_lengths), enc, semantic_input,
                                                  caps, dl_dev, mask, weights, pool)
        return predictions, caps, decode_lengths, alphas, sort_ind

    def sample(self, beam_size, word_map, encoder_out, tag_out):
        return _common.beam_search(self, beam_size, word_map, encoder_out, tag_out, use_attention=True, use_tags=True)


_KEY_OF_FIELD = {
    "attention_encoder_att_weight": "attention.encoder_att.weight",
    "attention_encoder_att_bias": "attention.encoder_att.bias",
    "attention_decoder_att_weight": "attention.decoder_att.weight",
    "attention_decoder_att_bias": "attention.decoder_att.bias",
    "attention_full_att_weight": "attention.full_att.weight",
    "attention_full_att_bias": "attention.full_att.bias",
    "embedding_weight": "embedding.weight",
    "decode_step_weight_ia": "decode_step.weight_ia", "decode_step_weight_ib": "decode_step.weight_ib",
    "decode_step_weight_ic": "decode_step.weight_ic", "decode_step_weight_ha": "decode_step.weight_ha",
    "decode_step_weight_hb": "decode_step.weight_hb", "decode_step_weight_hc": "decode_step.weight_hc",
    "decode_step_bias_ih": "decode_step.bias_ih", "decode_step_bias_hh": "decode_step.bias_hh",
    "init_h_weight": "init_h.weight", "init_h_bias": "init_h.bias",
    "init_c_weight": "init_c.weight", "init_c_bias": "init_c.bias",
    "f_beta_weight": "f_beta.weight", "f_beta_bias": "f_beta.bias",
    "fc_weight": "fc.weight", "fc_bias": "fc.bias",
}


def _collect_weights(module):
    """Parameters in the order of the C struct scnattn_params (None where the decoder has none)."""
    named = dict(module.named_parameters())
    return [named.get(_KEY_OF_FIELD[f]) for f in PARAM_FIELDS]
