"""Data-parallel gradient reduction: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm, over xGMI; "gloo" in the CPU tests), bucketed SUM all-reduce of the flat gradient
buffer, launched from post-accumulate-grad hooks so that buckets overlap the rest of backward.

The reference has no distributed code at all; the scheme follows SURVEY.md 8e: the
decoder's gradients appear first (its whole backward is one C call), then ResNet layer4 -> layer2, so
buckets are cut in flat order and fire as soon as every parameter in them has its gradient.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are sized (default 32 MiB) so that a
ring step moves MiB-sized chunks per link, and the division by world size is folded into the fused
optimizer kernel (grad_scale) instead of a separate pass."""
import ctypes as C
import os

import torch
import torch.distributed as dist


class CabiComm:
    """The library's own RCCL communicator (include/scnattn.h: scnattn_dp_comm_*): communication stream + events live
    in libscnattn, buckets are reduced in place and the compute stream only waits in `finish()`.  The 128-byte RCCL id
    travels from rank 0 through the already initialised torch.distributed group (any backend: it is 128 bytes of host
    data).  Used by GradReducer when SCNATTN_DP_BACKEND=cabi (or backend="cabi")."""

    _shared = {}

    def __init__(self, device):
        from . import _lib
        self.h = _lib.lib()
        self._lib = _lib
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(self.h.scnattn_dp_unique_id(buf), "scnattn_dp_unique_id")
        if world > 1:
            t = torch.tensor(list(buf.raw), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.to(device)
            dist.broadcast(t, src=0)
            buf = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        torch.cuda.set_device(device)
        handle = C.c_void_p()
        _lib.check(self.h.scnattn_dp_comm_create(buf, world, rank, C.byref(handle)), "scnattn_dp_comm_create")
        self.handle, self.world, self.device = handle, world, device
        # reduce on a stream that is known to run beside both the compute stream and the weight-gradient stream
        # (scnattn/conv.py::_concurrent_stream: HIP streams share a handful of hardware queues)
        from . import conv as _conv
        self.stream, self.probe = _conv._concurrent_stream(device, beside=(_conv._side(device).stream,))
        _lib.check(self.h.scnattn_dp_comm_set_stream(handle, C.c_void_p(self.stream.cuda_stream)),
                   "scnattn_dp_comm_set_stream")

    @classmethod
    def get(cls, device):
        key = str(device)
        if key not in cls._shared:
            cls._shared[key] = cls(device)
        return cls._shared[key]

    def _stream(self):
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))

    def allreduce(self, t):
        self._lib.check(self.h.scnattn_dp_comm_allreduce_bucket(self.handle, self._stream(), C.c_void_p(t.data_ptr()),
                                                                t.numel()), "scnattn_dp_comm_allreduce_bucket")

    def finish(self):
        self._lib.check(self.h.scnattn_dp_comm_finish(self.handle, self._stream()), "scnattn_dp_comm_finish")

    def close(self):
        if self.handle:
            self.h.scnattn_dp_comm_destroy(self.handle)
            self.handle = None
            CabiComm._shared.pop(str(self.device), None)


class GradReducer:
    def __init__(self, flat, bucket_bytes=32 << 20, group=None, backend=None):
        self.flat = flat
        self.group = group
        backend = backend or os.environ.get("SCNATTN_DP_BACKEND", "torch")
        self.cabi = CabiComm.get(flat.flat_g.device) if (backend == "cabi" and flat.flat_g.is_cuda) else None
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
        if self.flat.flat_g.is_cuda:
            # The bucket's gather copy and its all-reduce are enqueued on the SIDE stream that also carries the trunk's
            # weight gradients (scnattn/conv.py): that stream first waits for what the main stream has produced so far
            # (an event), the main stream itself never waits for a bucket -- it carries on with the d-input chain --
            # and only `finish()` joins.  (Joining the main stream here instead cost 18 % of the step at one rank.)
            from . import conv as _conv
            dev = self.flat.flat_g.device
            side = _conv._side(dev)
            main = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event()
            ev.record(main)
            side.stream.wait_event(ev)
            with torch.cuda.stream(side.stream):
                self.flat.gather(idxs, join=False, record_stream=side.stream)
                if self.cabi is not None:     # RCCL through the C ABI: library-owned comm stream, event-ordered
                    self.cabi.allreduce(self.flat.flat_g[s:e])
                else:
                    self.works.append(dist.all_reduce(self.flat.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                      async_op=True))
            side.mark()
        else:
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
        if self.flat.flat_g.is_cuda:
            from . import conv as _conv
            _conv.join_side_streams()
            if self.cabi is not None:
                self.cabi.finish()
        return 1.0 / self.world


def broadcast_parameters(flat, src=0, group=None):
    """Make every rank start from rank `src`'s weights (one broadcast of the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat_p, src=src, group=group)
