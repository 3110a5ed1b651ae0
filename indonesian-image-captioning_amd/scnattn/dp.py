"""Data-parallel gradient reduction: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm, over xGMI; "gloo" in the CPU tests), bucketed SUM all-reduce of the flat gradient
buffer, launched from post-accumulate-grad hooks so that buckets overlap the rest of backward.

The reference has no distributed code at all; the scheme follows SURVEY.md 8e: the
decoder's gradients appear first (its whole backward is one C call), then ResNet layer4 -> layer2, so
buckets are cut in flat order and fire as soon as every parameter in them has its gradient.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are sized (default 32 MiB) so that a
ring step moves MiB-sized chunks per link, and the division by world size is folded into the fused
optimizer kernel (grad_scale) instead of a separate pass."""
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat, bucket_bytes=32 << 20, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = []      # (start, end, [param indices])
        start, idxs, limit = 0, [], max(1, bucket_bytes // 4)
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            idxs.append(i)
            end = o + (p.numel() + 63) // 64 * 64
            if end - start >= limit:
                self.buckets.append((start, end, idxs))
                start, idxs = end, []
        if idxs:
            self.buckets.append((start, flat.numel, idxs))
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self.pending = [0] * len(self.buckets)
        self.works = []
        self.enabled = self.world > 1
        self._hooks = []
        for i, p in enumerate(flat.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def _make_hook(self, i):
        def hook(param):
            if not self.enabled:
                return
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        s, e, idxs = self.buckets[b]
        self.flat.gather(idxs)            # one multi-tensor copy of the bucket's gradients
        self.works.append(dist.all_reduce(self.flat.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                          async_op=True))
        self.launched[b] = True

    def reset(self):
        """Call before each backward."""
        self.pending = [len(idxs) for (_, _, idxs) in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.works = []

    def finish(self):
        """Call after backward: reduce buckets whose hooks never all fired (unused parameters), wait
        for every all-reduce, and return the factor the optimizer must scale gradients by."""
        if not self.enabled:
            return 1.0
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)           # parameters whose hooks never fired are reduced as zeros
        for w in self.works:
            w.wait()
        self.works = []
        return 1.0 / self.world


def broadcast_parameters(flat, src=0, group=None):
    """Make every rank start from rank `src`'s weights (one broadcast of the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat_p, src=src, group=group)
