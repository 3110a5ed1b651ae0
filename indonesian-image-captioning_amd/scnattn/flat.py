"""Flat parameter / gradient storage shared by the fused optimizer and the data-parallel reducer.

All trainable parameters of a module are re-homed into ONE contiguous fp32 buffer (``p.data`` becomes
a view).  Gradients are gathered into ONE contiguous gradient buffer with multi-tensor copies:
``zero_grad`` sets every ``.grad`` to None so autograd hands over ("steals") each freshly computed
gradient without an accumulate kernel, and ``gather`` copies a whole bucket of them into the flat
buffer with one ``torch._foreach_copy_`` (measured on the ResNet-152 + decoder step: 566 per-parameter
``grad += new`` launches = 4.1 ms per step otherwise).  Consequences:
  * clamp + Adam is a single kernel launch over the flat buffers (utils/optimizer.py);
  * a gradient all-reduce bucket is a slice of the flat gradient buffer;
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
                self.gviews.append(self.flat_g[o:o + p.numel()].as_strided(p.shape, p.data.stride()))
                # producers that can write a gradient in place (the fused Bottleneck's weight gradients) take this
                # view as their output: the optimizer's gather then has nothing to copy for that parameter
                p._scn_flat_grad = self.gviews[-1]
                p.grad = None

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def gather(self, indices=None, join=True, record_stream=None):
        """Copy the gradients autograd produced for parameters `indices` (default: all) into the flat
        buffer and make each ``.grad`` the flat view.  Parameters that received no gradient get zeros.
        `join=False`: the caller runs on the side stream that produced the weight gradients itself (data-parallel
        buckets), so nothing has to wait; `record_stream` then keeps the source tensors alive for that stream."""
        if self.flat_g.is_cuda and join:
            from . import conv as _conv
            _conv.join_side_streams()     # weight gradients of the fused Bottleneck run on a side stream
        idx = range(len(self.params)) if indices is None else indices
        src, dst = [], []
        for i in idx:
            p, gv = self.params[i], self.gviews[i]
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                if record_stream is not None:
                    p.grad.record_stream(record_stream)
                src.append(p.grad)
                dst.append(gv)
            p.grad = gv
        if src:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)

    # kept for callers of the previous name
    gather_stray_grads = gather
