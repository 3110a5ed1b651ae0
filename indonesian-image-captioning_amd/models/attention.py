"""Soft attention on MI355X: drop-in for the reference's models/attention.py (same submodule
names -> same state_dict keys).  forward(encoder_out (B,P,E), decoder_hidden (B,D)) returns the
attention-weighted encoding (B,E) and the weights (B,P), computed by libscnattn:
att1/att2 on the MFMA sgemm, scores -> softmax -> weighted sum in the streaming attention kernels."""
from torch import nn

from scnattn import functional as SF


class Attention(nn.Module):
    def __init__(self, encoder_dim, decoder_dim, attention_dim):
        super().__init__()
        self.encoder_att = nn.Linear(encoder_dim, attention_dim)
        self.decoder_att = nn.Linear(decoder_dim, attention_dim)
        self.full_att = nn.Linear(attention_dim, 1)
        self.relu = nn.ReLU()            # kept for module-tree parity; the kernels apply them
        self.softmax = nn.Softmax(dim=1)

    def forward(self, encoder_out, decoder_hidden):
        return SF.attention(encoder_out, decoder_hidden, self.encoder_att.weight, self.encoder_att.bias,
                            self.decoder_att.weight, self.decoder_att.bias, self.full_att.weight,
                            self.full_att.bias)
