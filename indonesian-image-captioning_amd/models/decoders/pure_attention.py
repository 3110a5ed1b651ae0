"""PureAttention on MI355X: drop-in for the reference's models/decoders/pure_attention.py
(Show-Attend-Tell: soft attention + a plain `nn.LSTMCell`, no tags).

The teacher-forced forward runs through the same whole-sequence HIP drivers as AttentionSCN: an LSTM cell is
the SCN cell with the tag factors switched off.  With  s.Wb = s.Hb = 1  (one constant "tag", weights of ones)
and  Wc_g = Hc_g = I  (factored_dim = decoder_dim), the SCN pre-activation
    r_g = ((u.Wa_g) * (s.Wb_g)) . Wc_g^T + b_ih_g + ((h.Ha_g) * (s.Hb_g)) . Hc_g^T + b_hh_g        (scn_cell.py:73-144)
is  u.Wa_g + h.Ha_g + b_ih_g + b_hh_g, the LSTMCell's (pure_attention.py:60, 140-141), once the LSTMCell's
[4H, I] / [4H, H] matrices are transposed and its gate blocks (torch order i, f, g, o) put in the SCN order
(i, f, o, c).  The identity products are exact in fp32, autograd carries the gradients back through that
re-layout to `decode_step.weight_ih` etc., and the state_dict keys stay torch's.  `forward_stepwise` is the
literal loop (attention + Linears in libscnattn, torch's LSTMCell), kept as an in-box cross-check; beam search
(`sample`) uses the cell step by step as the reference does."""
import torch
from torch import nn

from models.attention import Attention
from models.decoders import _common
from models.decoders.attention_scn import _KEY_OF_FIELD
from scnattn import functional as SF
from scnattn._lib import PARAM_FIELDS


class PureAttention(nn.Module):
    def __init__(self, attention_dim, embed_dim, decoder_dim, vocab_size, encoder_dim=2048, dropout=0.5):
        super().__init__()
        self.encoder_dim = encoder_dim
        self.attention_dim = attention_dim
        self.embed_dim = embed_dim
        self.decoder_dim = decoder_dim
        self.vocab_size = vocab_size
        self.attention = Attention(encoder_dim, decoder_dim, attention_dim)
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.dropout = nn.Dropout(p=dropout)
        self.decode_step = nn.LSTMCell(embed_dim + encoder_dim, decoder_dim, bias=True)
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
        mean_encoder_out = encoder_out.mean(dim=1)
        return (SF.linear(mean_encoder_out, self.init_h.weight, self.init_h.bias),
                SF.linear(mean_encoder_out, self.init_c.weight, self.init_c.bias))

    def _scn_view_of_lstm(self):
        """The LSTMCell as SCN-cell weights in scnattn_params order (see the module docstring)."""
        H = self.decoder_dim
        cell = self.decode_step
        dev = cell.weight_ih.device
        idx = torch.cat([torch.arange(0, 2 * H), torch.arange(3 * H, 4 * H), torch.arange(2 * H, 3 * H)]).to(dev)
        ones = torch.ones(1, 4 * H, device=dev)
        eye4 = torch.eye(H, device=dev).repeat(1, 4)
        scn = {
            "decode_step_weight_ia": cell.weight_ih.index_select(0, idx).t().contiguous(),
            "decode_step_weight_ib": ones, "decode_step_weight_ic": eye4,
            "decode_step_weight_ha": cell.weight_hh.index_select(0, idx).t().contiguous(),
            "decode_step_weight_hb": ones, "decode_step_weight_hc": eye4,
            "decode_step_bias_ih": cell.bias_ih.index_select(0, idx),
            "decode_step_bias_hh": cell.bias_hh.index_select(0, idx),
        }
        named = dict(self.named_parameters())
        return [scn[f] if f in scn else named.get(_KEY_OF_FIELD[f]) for f in PARAM_FIELDS]

    def forward(self, encoder_out, encoded_captions, caption_lengths, sort_ind=None, prepool=None, pool_size=14,
                caplens_host=None):
        src, pool = _common.resolve_prepool(encoder_out, prepool, pool_size, self.attention_dim, "PureAttention.forward")
        enc, caps, decode_lengths, dl_dev, sort_ind = _common.sort_by_length(
            src, encoded_captions, caption_lengths, sort_ind, caplens_host)
        B, E = enc.shape[0], enc.shape[2]
        P = pool.P if pool is not None else enc.shape[1]
        T = max(decode_lengths)
        H = self.decoder_dim
        dims = (B, P, E, self.attention_dim, H, H, self.embed_dim, 1, self.vocab_size, T, caps.size(1), 1)
        mask = _common.make_drop_mask(self, B, T, H, enc.device)
        ones = torch.ones(B, 1, device=enc.device)                 # the single constant "tag"
        predictions, alphas = SF.decoder_sequence(dims, _common.active_rows(decode_lengths), enc, ones, caps, dl_dev,
                                                  mask, self._scn_view_of_lstm(), pool)
        return predictions, caps, decode_lengths, alphas, sort_ind

    def forward_stepwise(self, encoder_out, encoded_captions, caption_lengths, sort_ind=None):
        """The reference's loop as written (pure_attention.py:120-149), one timestep at a time."""
        enc, caps, decode_lengths, _, sort_ind = _common.sort_by_length(
            encoder_out, encoded_captions, caption_lengths, sort_ind)
        B, P, _ = enc.shape
        T = max(decode_lengths)
        embeddings = self.embedding(caps)
        h, c = self.init_hidden_state(enc)
        mask = _common.make_drop_mask(self, B, T, self.decoder_dim, enc.device)
        predictions = torch.zeros(B, T, self.vocab_size, device=enc.device)
        alphas = torch.zeros(B, T, P, device=enc.device)
        for t, bt in enumerate(_common.active_rows(decode_lengths)):
            awe, alpha = self.attention(enc[:bt], h[:bt])
            gate = torch.sigmoid(SF.linear(h[:bt], self.f_beta.weight, self.f_beta.bias))
            h, c = self.decode_step(torch.cat([embeddings[:bt, t, :], gate * awe], dim=1), (h[:bt], c[:bt]))
            hd = h if mask is None else h * mask[:bt, t, :]
            predictions[:bt, t, :] = SF.linear(hd, self.fc.weight, self.fc.bias)
            alphas[:bt, t, :] = alpha
        return predictions, caps, decode_lengths, alphas, sort_ind

    def sample(self, beam_size, word_map, encoder_out):
        return _common.beam_search(self, beam_size, word_map, encoder_out, None, use_attention=True, use_tags=False)
