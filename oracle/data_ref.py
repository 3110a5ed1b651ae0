"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of the reference's input pipeline for one batch:
  * `CaptionDataset.__getitem__` — datasets/caption.py:49-65: `torch.FloatTensor(imgs[i // cpi] / 255.)`,
    optional transform, `LongTensor(captions[i])`, `LongTensor([caplens[i]])`, and for VAL/TEST the `cpi`
    captions of the image;
  * `TagDataset.__getitem__` — datasets/tag.py:46-55;
  * `transforms.Normalize(mean, std)` — trains/attention_scn.py:121-126.  torchvision is a third-party
    dependency that is absent here (README names it without a version); its published algorithm is
    `tensor.sub_(mean[:, None, None]).div_(std[:, None, None])` with mean/std as tensors of the input dtype;
  * the DataLoader's default collate: `torch.stack` per field.
Pinned by: HDF5 bytes written by real h5py (tests/golden/hdf5, oracle/gen_hdf5_golden.py — array contents as
h5py reads them); the arithmetic above is plain IEEE fp32/fp64 and is restated, not imported.
"""
import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def normalize(t, mean=MEAN, std=STD):
    m = torch.as_tensor(mean, dtype=t.dtype)
    s = torch.as_tensor(std, dtype=t.dtype)
    return (t - m[:, None, None]) / s[:, None, None]


def image_item(u8_chw, mean=MEAN, std=STD):
    """u8_chw: numpy uint8 (C, H, W) as h5py returns it."""
    img = torch.FloatTensor(np.asarray(u8_chw) / 255.)
    return img if mean is None else normalize(img, mean, std)


def caption_item(imgs, captions, caplens, i, cpi, split, mean=MEAN, std=STD):
    img = image_item(imgs[i // cpi], mean, std)
    caption = torch.LongTensor(captions[i])
    caplen = torch.LongTensor([caplens[i]])
    if split == "TRAIN":
        return img, caption, caplen
    first = (i // cpi) * cpi
    return img, caption, caplen, torch.LongTensor(captions[first:first + cpi])


def collate(items):
    return tuple(torch.stack(field) for field in zip(*items))


def caption_batch(imgs, captions, caplens, indices, cpi, split, mean=MEAN, std=STD):
    return collate([caption_item(imgs, captions, caplens, int(i), cpi, split, mean, std) for i in indices])
