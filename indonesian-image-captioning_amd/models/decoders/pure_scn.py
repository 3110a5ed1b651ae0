"""PureSCN on MI355X: drop-in for the reference's models/decoders/pure_scn.py (SCN cell fed by the word
embedding only; the image enters through the initial state).  Same API and 4-tuple return; the
recurrence (:128-138) is the has_att = 0 flavour of scnattn_seq_fwd / scnattn_seq_bwd."""
from torch import nn

from models.scn_cell import SCNCell
from models.decoders import _common
from models.decoders.attention_scn import _collect_weights
from scnattn import functional as SF


class PureSCN(nn.Module):
    def __init__(self, embed_dim, decoder_dim, factored_dim, semantic_dim, vocab_size, encoder_dim=2048, dropout=0.5):
        super().__init__()
        self.embed_dim = embed_dim
        self.encoder_dim = encoder_dim
        self.decoder_dim = decoder_dim
        self.factored_dim = factored_dim
        self.semantic_dim = semantic_dim
        self.vocab_size = vocab_size
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.dropout = nn.Dropout(p=dropout)
        self.decode_step = SCNCell(embed_dim, decoder_dim, semantic_dim, factored_dim, bias=True)
        self.init_h = nn.Linear(encoder_dim, decoder_dim)
        self.init_c = nn.Linear(encoder_dim, decoder_dim)
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

    def forward(self, encoder_out, semantic_input, encoded_captions, caption_lengths, sort_ind=None, prepool=None,
                pool_size=14, caplens_host=None):
        """Reference signature (pure_scn.py:87) plus the optional trunk map (see AttentionSCN.forward): only the
        initial state reads the encoder output here, and the pixel mean of the pooled map is a weighted mean of
        the un-pooled one."""
        src, pool = _common.resolve_prepool(encoder_out, prepool, pool_size, None, "PureSCN.forward")
        enc, caps, decode_lengths, dl_dev, sort_ind = _common.sort_by_length(
            src, encoded_captions, caption_lengths, sort_ind, caplens_host)
        B, E = enc.shape[0], enc.shape[2]
        P = pool.P if pool is not None else enc.shape[1]
        T = max(decode_lengths)
        dims = (B, P, E, 0, self.decoder_dim, self.factored_dim, self.embed_dim, self.semantic_dim,
                self.vocab_size, T, caps.size(1), 0)
        mask = _common.make_drop_mask(self, B, T, self.decoder_dim, enc.device)
        predictions, _ = SF.decoder_sequence(dims, _common.active_rows(decode_lengths), enc, semantic_input, caps,
                                             dl_dev, mask, _collect_weights(self), pool)
        return predictions, caps, decode_lengths, sort_ind

    def sample(self, beam_size, word_map, encoder_out, tag_out):
        return _common.beam_search(self, beam_size, word_map, encoder_out, tag_out, use_attention=False, use_tags=True)
