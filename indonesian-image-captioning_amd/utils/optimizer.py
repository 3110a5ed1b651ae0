"""Gradient clamp + learning-rate helpers (reference: utils/optimizer.py:1-24), plus the fused
MI355X optimizer used by the benchmark harness.

``clip_gradient`` / ``adjust_learning_rate`` keep the reference's signatures and work with any torch
optimizer, so trains/attention_scn.py:244-252 runs unchanged.  ``FusedClampAdam`` does the same
element-wise clamp(+-grad_clip) followed by torch.optim.Adam's update in ONE HIP kernel over a flat fp32
buffer that also backs the gradients (so a data-parallel all-reduce needs no packing copy)."""
import torch

from scnattn import functional as SF


def clip_gradient(optimizer, grad_clip):
    """Element-wise clamp of every gradient to [-grad_clip, grad_clip] (NOT norm clipping)."""
    for group in optimizer.param_groups:
        for param in group['params']:
            if param.grad is not None:
                param.grad.data.clamp_(-grad_clip, grad_clip)


def adjust_learning_rate(optimizer, shrink_factor):
    print("\nDECAYING learning rate.")
    for param_group in optimizer.param_groups:
        param_group['lr'] = param_group['lr'] * shrink_factor
    print("The new learning rate is %f\n" % (optimizer.param_groups[0]['lr'],))


class FusedClampAdam:
    """Adam(lr, betas, eps) with the reference's element-wise gradient clamp fused in.

    Parameters and gradients live in a scnattn.flat.FlatBuffer; ``step(grad_scale)`` first multiplies
    gradients by ``grad_scale`` (1/world_size after a SUM all-reduce), clamps to +-grad_clip, then
    applies torch.optim.Adam's update -- one HIP kernel launch for the whole model."""

    def __init__(self, params, lr, grad_clip=None, betas=(0.9, 0.999), eps=1e-8):
        from scnattn.flat import FlatBuffer
        self.flat = FlatBuffer(params)
        self.params = self.flat.params
        self.param_groups = [{'params': self.params, 'lr': lr}]
        self.grad_clip, self.betas, self.eps = grad_clip, betas, eps
        self.step_count = 0
        self.flat_m = torch.zeros_like(self.flat.flat_p)
        self.flat_v = torch.zeros_like(self.flat.flat_p)

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        self.flat.gather()
        self.step_count += 1
        SF.clamp_adam_(self.flat.flat_p, self.flat.flat_g, self.flat_m, self.flat_v, self.param_groups[0]['lr'],
                       self.step_count, self.grad_clip, self.betas[0], self.betas[1], self.eps, grad_scale)
