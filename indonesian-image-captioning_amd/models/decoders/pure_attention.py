"""PureAttention on MI355X: drop-in for the reference's models/decoders/pure_attention.py
(Show-Attend-Tell: soft attention + a plain LSTMCell, no tags).  BASELINE config 1 uses it as
plumbing only, so the loop stays step-by-step: the attention (models/attention.py) and every Linear
run in libscnattn; the LSTMCell is torch's own (its state_dict keys must stay torch's)."""
import torch
from torch import nn

from models.attention import Attention
from models.decoders import _common
from scnattn import functional as SF


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

    def forward(self, encoder_out, encoded_captions, caption_lengths, sort_ind=None):
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
