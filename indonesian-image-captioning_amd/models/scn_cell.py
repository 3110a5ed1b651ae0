"""SCNCell on MI355X: drop-in for the reference's models/scn_cell.py (same constructor, parameters,
state_dict keys, error messages and repr); the arithmetic runs in libscnattn's HIP kernels.

    x_g = ((u . Wa_g) * (s . Wb_g)) . Wc_g^T + b_ih_g            reference scn_cell.py:73-91
    r_g = ((h . Ha_g) * (s . Hb_g)) . Hc_g^T + x_g + b_hh_g      reference scn_cell.py:134-144
    i,f,o = sigmoid(r_i,r_f,r_o); c~ = tanh(r_c); c' = f*c + i*c~; h' = o*tanh(c')   (:146-152)

Inside ``AttentionSCN.forward`` / ``PureSCN.forward`` the cell is not called step by step: the whole
recurrence is one C call (scnattn_seq_fwd).  This module is the stand-alone entry (beam search,
unit use) and owns the parameters.
"""
import math

import torch
from torch import nn

from scnattn import functional as SF


class SCNCell(nn.Module):
    def __init__(self, input_size, hidden_size, semantic_size, factor_size, bias=True):
        super().__init__()
        self.factor_size = factor_size
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.semantic_size = semantic_size
        shapes = (("weight_ia", input_size), ("weight_ib", semantic_size), ("weight_ic", hidden_size),
                  ("weight_ha", hidden_size), ("weight_hb", semantic_size), ("weight_hc", hidden_size))
        for name, rows in shapes:  # registration order == the reference's state_dict order
            setattr(self, name, nn.Parameter(torch.empty(rows, 4 * factor_size)))
        if bias:
            self.bias_ih = nn.Parameter(torch.empty(4 * hidden_size))
            self.bias_hh = nn.Parameter(torch.empty(4 * hidden_size))
        else:
            self.register_parameter('bias_ih', None)
            self.register_parameter('bias_hh', None)
        self.reset_parameters()

    def reset_parameters(self):
        # every parameter, biases included, ~ U(-1/sqrt(H), 1/sqrt(H))  (reference :156-159)
        bound = 1.0 / math.sqrt(self.hidden_size)
        for p in self.parameters():
            nn.init.uniform_(p, -bound, bound)

    def extra_repr(self):
        return '{}, {}'.format(self.input_size, self.hidden_size)

    def check_forward_input(self, input):
        if input.size(1) != self.input_size:
            raise RuntimeError("input has inconsistent input_size: got {}, expected {}".format(
                input.size(1), self.input_size))

    def check_forward_hidden(self, input, hx, hidden_label=''):
        if input.size(0) != hx.size(0):
            raise RuntimeError("Input batch size {} doesn't match hidden{} batch size {}".format(
                input.size(0), hidden_label, hx.size(0)))
        if hx.size(1) != self.hidden_size:
            raise RuntimeError("hidden{} has inconsistent hidden_size: got {}, expected {}".format(
                hidden_label, hx.size(1), self.hidden_size))

    def forward(self, wemb_input, semantic_input, hx=None):
        self.check_forward_input(wemb_input)
        if hx is None:
            zeros = wemb_input.new_zeros(wemb_input.size(0), self.hidden_size, requires_grad=False)
            hx = (zeros, zeros)
        for label, state in (('[0]', hx[0]), ('[1]', hx[1])):
            self.check_forward_hidden(wemb_input, state, label)
        x = SF.scn_input(wemb_input, semantic_input, self.weight_ia, self.weight_ib, self.weight_ic, self.bias_ih)
        return self.recurrent_step(*x, semantic_input, hx)

    def recurrent_step(self, x_i, x_f, x_o, x_c, semantic_input, hx):
        h_, c_ = hx
        return SF.scn_recurrent(x_i, x_f, x_o, x_c, semantic_input, h_, c_, self.weight_ha, self.weight_hb,
                                self.weight_hc, self.bias_hh)
