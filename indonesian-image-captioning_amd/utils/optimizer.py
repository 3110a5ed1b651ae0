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

    Parameters are re-homed into one flat buffer (``p.data`` becomes a view) and each ``p.grad`` is a
    view into one flat gradient buffer; ``step(grad_scale)`` first multiplies gradients by
    ``grad_scale`` (1/world_size after a sum all-reduce)."""

    def __init__(self, params, lr, grad_clip=None, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "FusedClampAdam: no trainable parameters"
        dev = self.params[0].device
        self.param_groups = [{'params': self.params, 'lr': lr}]
        self.grad_clip, self.betas, self.eps = grad_clip, betas, eps
        self.step_count = 0
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 63) // 64 * 64   # 256-byte granules
        self.numel = n
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets = offs
        self._gviews = []
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                # keep each tensor's own (dense) stride order, e.g. channels-last conv weights
                view = self.flat_p[o:o + p.numel()].as_strided(p.shape, p.data.stride())
                view.copy_(p.data)
                p.data = view
                gv = self.flat_g[o:o + p.numel()].as_strided(p.shape, p.data.stride())
                self._gviews.append(gv)
                p.grad = gv

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        for p, gv in zip(self.params, self._gviews):
            p.grad = gv

    def _gather_stray_grads(self):
        for p, gv in zip(self.params, self._gviews):
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv

    def step(self, grad_scale=1.0):
        self._gather_stray_grads()
        self.step_count += 1
        SF.clamp_adam_(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.param_groups[0]['lr'],
                       self.step_count, self.grad_clip, self.betas[0], self.betas[1], self.eps, grad_scale)
