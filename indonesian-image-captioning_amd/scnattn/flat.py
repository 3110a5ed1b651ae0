"""Flat parameter / gradient storage shared by the fused optimizer and the data-parallel reducer.

All trainable parameters of a module are re-homed into ONE contiguous fp32 buffer (``p.data`` becomes
a view) and every ``p.grad`` is a view into ONE contiguous gradient buffer.  Consequences:
  * clamp + Adam is a single kernel launch over the flat buffers (utils/optimizer.py);
  * a gradient all-reduce bucket is just a slice of the flat gradient buffer -- no packing copies;
  * ``zero_grad`` is one memset.
Device-agnostic (the gloo CPU tests use it too)."""
import torch


class FlatBuffer:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "FlatBuffer: no trainable parameters"
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64   # 256-byte granules keep every view 16-byte aligned
        self.numel = n
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.gviews = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                # keep each tensor's own dense stride order (e.g. channels-last conv weights)
                view = self.flat_p[o:o + p.numel()].as_strided(p.shape, p.data.stride())
                view.copy_(p.data)
                p.data = view
                gv = self.flat_g[o:o + p.numel()].as_strided(p.shape, p.data.stride())
                self.gviews.append(gv)
                p.grad = gv

    def zero_grad(self):
        self.flat_g.zero_()
        for p, gv in zip(self.params, self.gviews):
            p.grad = gv

    def gather_stray_grads(self):
        """If something replaced a .grad (e.g. zero_grad(set_to_none=True) by foreign code) copy it back."""
        for p, gv in zip(self.params, self.gviews):
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv
