"""Drop-in for the reference's `datasets/caption.py:9-65` (same constructor, same items), reading the
HDF5 image file through `scnattn.h5lite` instead of h5py.

This is the host-side, one-sample-at-a-time interface a `torch.utils.data.DataLoader` expects; the train
loop of this build uses `scnattn.data.DeviceBatchLoader` instead (whole uint8 batches normalised on the
GPU), which yields the same numbers.  Differences from the reference: `split == 'TRAIN'` is compared by
value (the reference's `is 'TRAIN'`, caption.py:58, relies on string interning), and any number of
DataLoader workers works because every worker maps the file itself.
"""
import json
import os

import torch
from torch.utils.data import Dataset

from scnattn import h5lite


class CaptionDataset(Dataset):
    r"""A PyTorch Dataset class to be used in a PyTorch DataLoader to create batches.

    Arguments
        data_folder (string): folder where data files are stored
        data_name (string): base name of processed datasets
        split (string): split, one of 'TRAIN', 'VAL', or 'TEST'
        transform (callable): image transform pipeline
        cpi (int): captions per image; falsy -> the file's `captions_per_image` attribute
    """

    def __init__(self, data_folder, data_name, split, transform=None, cpi=5):
        self.split = split
        assert self.split in {'TRAIN', 'VAL', 'TEST'}
        self._path = os.path.join(data_folder, self.split + '_IMAGES_' + data_name + '.hdf5')
        self.h = h5lite.File(self._path)
        self.imgs = self.h['images']
        self.cpi = cpi if cpi else int(self.h.attrs['captions_per_image'])
        with open(os.path.join(data_folder, self.split + '_CAPTIONS_' + data_name + '.json'), 'r') as j:
            self.captions = json.load(j)
        with open(os.path.join(data_folder, self.split + '_CAPLENS_' + data_name + '.json'), 'r') as j:
            self.caplens = json.load(j)
        self.transform = transform
        self.dataset_size = len(self.captions)

    def __getstate__(self):            # DataLoader workers re-open the map instead of pickling it
        st = dict(self.__dict__)
        st['h'] = st['imgs'] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.h = h5lite.File(self._path)
        self.imgs = self.h['images']

    def __getitem__(self, i):
        # the Nth caption corresponds to the (N // captions_per_image)th image
        img = torch.FloatTensor(self.imgs[i // self.cpi] / 255.)
        if self.transform is not None:
            img = self.transform(img)
        caption = torch.LongTensor(self.captions[i])
        caplen = torch.LongTensor([self.caplens[i]])
        if self.split == 'TRAIN':
            return img, caption, caplen
        # validation / test: also all `cpi` captions of the image, for BLEU-4
        first = (i // self.cpi) * self.cpi
        all_captions = torch.LongTensor(self.captions[first:first + self.cpi])
        return img, caption, caplen, all_captions

    def __len__(self):
        return self.dataset_size


class Normalize:
    """`torchvision.transforms.Normalize(mean, std)` for a (C, H, W) float tensor (torchvision is not part
    of this stack): `(tensor - mean[:, None, None]) / std[:, None, None]` with mean/std in the tensor's
    dtype — what trains/attention_scn.py:121-126 passes as `transform`."""

    def __init__(self, mean, std):
        self.mean, self.std = tuple(mean), tuple(std)

    def __call__(self, tensor):
        mean = torch.as_tensor(self.mean, dtype=tensor.dtype, device=tensor.device)
        std = torch.as_tensor(self.std, dtype=tensor.dtype, device=tensor.device)
        if (std == 0).any():
            raise ValueError('std evaluated to zero after conversion to {}, leading to division by zero.'
                             .format(tensor.dtype))
        return (tensor - mean[:, None, None]) / std[:, None, None]

    def __repr__(self):
        return self.__class__.__name__ + '(mean={0}, std={1})'.format(self.mean, self.std)
