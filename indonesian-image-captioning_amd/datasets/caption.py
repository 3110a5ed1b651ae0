"""Drop-in for the reference's `datasets/caption.py:9-65` (same constructor, same items), reading the
HDF5 image file through `scnattn.h5lite` instead of h5py.

This is the host-side, one-sample-at-a-time interface a `torch.utils.data.DataLoader` expects; the train
loop of this build uses `scnattn.data.DeviceBatchLoader` instead (whole uint8 batches normalised on the
GPU), which yields the same numbers.  Differences from the reference: `split == 'TRAIN'` is compared by
value (the reference's `is 'TRAIN'`, caption.py:58, relies on string interning), and any number of
DataLoader workers works because every worker maps the file itself.
"""
import json

import torch
from torch.utils.data import Dataset

from datasets._mapped import SPLITS, MappedArrays, split_file, unit_float_image


class CaptionDataset(Dataset):
    """CaptionDataset(data_folder, data_name, split, transform=None, cpi=5): caption i with the image it
    belongs to (image i // cpi); a falsy `cpi` takes the file's `captions_per_image` attribute."""

    def __init__(self, data_folder, data_name, split, transform=None, cpi=5):
        assert split in SPLITS
        self.split, self.transform = split, transform
        self._arrays = MappedArrays(imgs=(split_file(data_folder, split, 'IMAGES', data_name, '.hdf5'), 'images'))
        self.cpi = cpi if cpi else int(self._arrays.imgs_file.attrs['captions_per_image'])
        with open(split_file(data_folder, split, 'CAPTIONS', data_name, '.json')) as j:
            self.captions = json.load(j)          # whole files in memory, as the reference
        with open(split_file(data_folder, split, 'CAPLENS', data_name, '.json')) as j:
            self.caplens = json.load(j)
        self.dataset_size = len(self.captions)

    @property
    def imgs(self):
        return self._arrays.imgs

    @property
    def h(self):
        return self._arrays.imgs_file

    def __len__(self):
        return self.dataset_size

    def __getitem__(self, i):
        image_index = i // self.cpi
        img = unit_float_image(self.imgs[image_index])
        if self.transform is not None:
            img = self.transform(img)
        item = (img, torch.LongTensor(self.captions[i]), torch.LongTensor([self.caplens[i]]))
        if self.split == 'TRAIN':
            return item
        # validation / test: all captions of the image too, for BLEU-4
        first = image_index * self.cpi
        return item + (torch.LongTensor(self.captions[first:first + self.cpi]),)


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
