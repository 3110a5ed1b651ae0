"""Device-side input pipeline for the caption train/validation loops.

Reference: `CaptionDataset` (datasets/caption.py:19-65) behind a `torch.utils.data.DataLoader`
(trains/attention_scn.py:121-130, `shuffle=True`, `workers = 1  # only 1 works with h5py`): every sample
is a float image made on the host (`imgs[i // cpi] / 255.`, then `Normalize(mean, std)`), collated and
copied to the GPU as fp32.

MI355X-first shape of the same thing:
  * the uint8 image rows are memory-mapped straight out of the HDF5 file (`h5lite`, no h5py);
  * `resident=True` (default whenever the dataset fits the HBM budget — 118k COCO images are 23 GB of
    288): the whole uint8 dataset, all captions and lengths are uploaded once; a step's batch is ONE
    launch of `scnattn_u8_gather_normalize` (gather rows by index + /255 + Normalize + layout/type the
    encoder wants) plus two index_selects for the captions — no host work and no PCIe traffic per step;
  * otherwise batches are staged: a producer thread gathers the batch rows into pinned uint8 buffers
    (4x less PCIe traffic than the reference's fp32 tensors), copies them on a side stream ahead of the
    consumer, and the same kernel normalises them on the consumer's stream;
  * data parallel: every rank draws the same per-epoch permutation and takes a strided slice of it
    (wrap-padded to equal length, as DistributedSampler does), so ranks stay in lock-step.
The arithmetic is bit-identical to the reference's (table lookup of its own per-value results,
`normalize_lut`).  There is no CPU fallback: batches are produced on the GPU.
"""
import json
import os
import queue
import threading

import numpy as np
import torch

from . import h5lite
from ._lib import call, ptr, require_cuda, stream_of

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # trains/attention_scn.py:121-122
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_lut(mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """lut[c][v] = the reference's value for pixel byte v in channel c: `torch.FloatTensor(v / 255.)`
    (numpy float64 divide, rounded to float32; datasets/caption.py:51) then torchvision Normalize
    `(t - mean[c]) / std[c]` with float32 mean/std tensors (float32 subtract, float32 IEEE divide)."""
    v = (np.arange(256, dtype=np.uint8) / 255.).astype(np.float32)
    m = np.asarray(mean, dtype=np.float32)
    s = np.asarray(std, dtype=np.float32)
    if m.ndim != 1 or m.shape != s.shape:
        raise ValueError("mean/std must be per-channel sequences of equal length")
    if (s == 0).any():
        raise ValueError("std evaluated to zero after conversion to torch.float32, leading to division by zero.")
    return ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


def identity_lut(channels=3):
    """Only the `/ 255.` of datasets/caption.py:51 (transform=None)."""
    v = (np.arange(256, dtype=np.uint8) / 255.).astype(np.float32)
    return np.repeat(v[None, :], channels, axis=0).copy()


def gather_normalize(src, idx, lut, n_out=None, dtype=torch.float32, channels_last=False, out=None):
    """src: uint8 [N, C, H, W] on the GPU; idx: int64 [n] rows to take (None: rows 0..n_out-1);
    lut: float32 [C, 256] on the GPU.  Returns the normalised batch [n, C, H, W] (`channels_last` selects
    the memory format, not the logical shape — what `EncoderCaption(channels_last=True)` consumes)."""
    require_cuda(src, idx, lut)
    if src.dtype != torch.uint8 or src.dim() != 4 or not src.is_contiguous():
        raise RuntimeError("gather_normalize: src must be a contiguous uint8 [N, C, H, W] tensor")
    N, Cn, H, W = src.shape
    if lut.dtype != torch.float32 or tuple(lut.shape) != (Cn, 256) or not lut.is_contiguous():
        raise RuntimeError("gather_normalize: lut must be float32 [C, 256]")
    if idx is not None:
        if idx.dtype != torch.int64 or idx.dim() != 1 or not idx.is_contiguous():
            raise RuntimeError("gather_normalize: idx must be a contiguous int64 vector")
        n_out = idx.numel()
    elif n_out is None:
        n_out = N
    if n_out > N and idx is None:
        raise RuntimeError("gather_normalize: n_out exceeds the source rows")
    if dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("gather_normalize: dtype must be float32 or bfloat16")
    if out is None:
        out = torch.empty((n_out, Cn, H, W), device=src.device, dtype=dtype,
                          memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    call("scnattn_u8_gather_normalize", stream_of(src), ptr(src), N, ptr(idx), n_out, Cn, H * W, ptr(lut), ptr(out),
         int(dtype == torch.bfloat16), int(bool(channels_last)))
    return out


class CaptionFiles:
    """The on-disk artefacts of one split, as written by utils/dataset.py:303-414."""

    def __init__(self, data_folder, data_name, split, cpi=5):
        assert split in {"TRAIN", "VAL", "TEST"}
        self.split = split
        self.h = h5lite.File(os.path.join(data_folder, split + "_IMAGES_" + data_name + ".hdf5"))
        ds = self.h["images"]
        if ds.dtype != np.uint8 or len(ds.shape) != 4:
            raise RuntimeError("'images' must be a uint8 (N, C, H, W) dataset, found %r" % (ds,))
        self.imgs = ds.array                       # zero-copy view of the file
        self.cpi = cpi if cpi else int(self.h.attrs["captions_per_image"])   # datasets/caption.py:32
        with open(os.path.join(data_folder, split + "_CAPTIONS_" + data_name + ".json")) as j:
            self.captions = np.asarray(json.load(j), dtype=np.int64)
        with open(os.path.join(data_folder, split + "_CAPLENS_" + data_name + ".json")) as j:
            self.caplens = np.asarray(json.load(j), dtype=np.int64)
        if self.captions.ndim != 2 or self.caplens.shape != (self.captions.shape[0],):
            raise RuntimeError("captions / caplens files are inconsistent")
        if (len(self.captions) - 1) // self.cpi >= self.imgs.shape[0]:
            raise RuntimeError("%d captions at %d per image need more than the %d stored images"
                               % (len(self.captions), self.cpi, self.imgs.shape[0]))

    def __len__(self):
        return len(self.captions)      # datasets/caption.py:48


def epoch_order(n, epoch, seed, shuffle, rank=0, world=1):
    """Sample indices this rank visits in `epoch`: one global permutation (same on every rank), wrap-padded
    to a multiple of `world`, rank r takes positions r, r+world, ..."""
    if shuffle:
        order = np.random.RandomState((seed * 1000003 + epoch) % (2 ** 31)).permutation(n)
    else:
        order = np.arange(n)
    if world > 1:
        total = -(-n // world) * world
        if total > n:
            order = np.concatenate([order, order[:total - n]])
        order = order[rank::world]
    return order.astype(np.int64)


class DeviceBatchLoader:
    """Iterable of device batches `(imgs, caps, caplens)` for TRAIN or `(imgs, caps, caplens, allcaps)`
    otherwise — the tuples the reference's loops unpack (trains/attention_scn.py:205, 304)."""

    def __init__(self, data_folder, data_name, split, batch_size, device, cpi=5, shuffle=True, seed=0, rank=0,
                 world=1, dtype=torch.float32, channels_last=True, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                 resident=None, hbm_budget_bytes=64 << 30, prefetch=2, drop_last=False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatchLoader produces batches on an MI355X device; there is no CPU fallback "
                               "(datasets.caption.CaptionDataset is the host-side reader)")
        self.files = CaptionFiles(data_folder, data_name, split, cpi)
        self.split, self.batch_size, self.shuffle, self.seed = split, int(batch_size), shuffle, seed
        self.rank, self.world, self.dtype, self.channels_last = rank, world, dtype, channels_last
        self.drop_last, self.prefetch, self.epoch = drop_last, max(1, int(prefetch)), 0
        lut = identity_lut(self.files.imgs.shape[1]) if mean is None else normalize_lut(mean, std)
        if lut.shape[0] != self.files.imgs.shape[1]:
            raise ValueError("mean/std have %d channels, the images %d" % (lut.shape[0], self.files.imgs.shape[1]))
        self.lut = torch.from_numpy(lut).to(self.device)
        self.caps_dev = torch.from_numpy(self.files.captions).to(self.device)
        self.caplens_dev = torch.from_numpy(self.files.caplens).to(self.device)
        nbytes = self.files.imgs.nbytes
        self.resident = (nbytes <= hbm_budget_bytes) if resident is None else bool(resident)
        self.imgs_dev = self._upload() if self.resident else None

    # ---- one-off upload of the uint8 dataset ---------------------------------------------------------------
    def _upload(self, chunk_rows=256):
        src = self.files.imgs
        N = src.shape[0]
        dev = torch.empty(src.shape, dtype=torch.uint8, device=self.device)
        stage = [torch.empty((min(chunk_rows, N),) + src.shape[1:], dtype=torch.uint8).pin_memory() for _ in range(2)]
        evs = [None, None]
        for k, a in enumerate(range(0, N, chunk_rows)):
            b = min(a + chunk_rows, N)
            s = k & 1
            if evs[s] is not None:
                evs[s].synchronize()
            np.copyto(stage[s].numpy()[:b - a], src[a:b])
            dev[a:b].copy_(stage[s][:b - a], non_blocking=True)
            evs[s] = torch.cuda.Event()
            evs[s].record(torch.cuda.current_stream(self.device))
        torch.cuda.current_stream(self.device).synchronize()
        return dev

    # ---- epoch bookkeeping --------------------------------------------------------------------------
    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _order(self):
        return epoch_order(len(self.files), self.epoch, self.seed, self.shuffle, self.rank, self.world)

    def __len__(self):
        n = -(-len(self.files) // self.world) if self.world > 1 else len(self.files)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _tuple(self, imgs, sel):
        caps = self.caps_dev.index_select(0, sel)
        caplens = self.caplens_dev.index_select(0, sel).unsqueeze(1)          # LongTensor([caplen]) per sample
        if self.split == "TRAIN":
            return imgs, caps, caplens
        cpi = self.files.cpi                                                    # datasets/caption.py:62-63
        first = torch.div(sel, cpi, rounding_mode="floor") * cpi
        rows = first.unsqueeze(1) + torch.arange(cpi, device=self.device)
        return imgs, caps, caplens, self.caps_dev[rows]

    def __iter__(self):
        order = self._order()
        nb = len(self)
        if self.resident:
            order_dev = torch.from_numpy(order).to(self.device)              # once per epoch
            img_rows = torch.div(order_dev, self.files.cpi, rounding_mode="floor")
            for b in range(nb):
                sl = slice(b * self.batch_size, min((b + 1) * self.batch_size, len(order)))
                sel = order_dev[sl]
                imgs = gather_normalize(self.imgs_dev, img_rows[sl].contiguous(), self.lut, dtype=self.dtype,
                                        channels_last=self.channels_last)
                yield self._tuple(imgs, sel)
            return
        yield from self._iter_staged(order, nb)

    # ---- staged mode: pinned uint8 batches copied ahead on a side stream ------------------------------
    def _iter_staged(self, order, nb):
        B = self.batch_size
        src = self.files.imgs
        nslots = self.prefetch + 2
        pinned = [torch.empty((B,) + src.shape[1:], dtype=torch.uint8).pin_memory() for _ in range(nslots)]
        staged = [torch.empty((B,) + src.shape[1:], dtype=torch.uint8, device=self.device) for _ in range(nslots)]
        copied = [None] * nslots          # event: H2D of the slot finished (side stream)
        released = [None] * nslots        # event: consumer's kernel has read the slot (consumer stream)
        q = queue.Queue(maxsize=self.prefetch)
        side = torch.cuda.Stream(device=self.device)
        stop = threading.Event()
        rows_all = order // self.files.cpi

        def produce():
            try:
                torch.cuda.set_device(self.device)
                for b in range(nb):
                    if stop.is_set():
                        return
                    s = b % nslots
                    rows = rows_all[b * B:(b + 1) * B]
                    if copied[s] is not None:
                        copied[s].synchronize()                    # pinned[s] is free again
                    np.take(src, rows, axis=0, out=pinned[s].numpy()[:len(rows)], mode="clip")   # unbuffered; rows are valid by construction
                    with torch.cuda.stream(side):
                        if released[s] is not None:
                            side.wait_event(released[s])           # staged[s] no longer being read
                        staged[s][:len(rows)].copy_(pinned[s][:len(rows)], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    copied[s] = ev
                    q.put((b, s, len(rows), ev))
                q.put(None)
            except BaseException as e:      # surface producer failures in the consumer
                q.put(e)

        th = threading.Thread(target=produce, name="scnattn-batch-stager", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                b, s, n, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                imgs = gather_normalize(staged[s], None, self.lut, n_out=n, dtype=self.dtype,
                                        channels_last=self.channels_last)
                rel = torch.cuda.Event()
                rel.record(cur)
                released[s] = rel
                sel = torch.from_numpy(order[b * B:b * B + n]).to(self.device, non_blocking=True)
                yield self._tuple(imgs, sel)
        finally:
            stop.set()
            while th.is_alive():            # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
            torch.cuda.current_stream(self.device).synchronize()
