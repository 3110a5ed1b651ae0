"""Drop-in for the reference's `datasets/tag.py:9-58` (the tagger's training set): image + ground-truth
tag vector per item, read through `scnattn.h5lite` instead of h5py."""
import os

import torch
from torch.utils.data import Dataset

from scnattn import h5lite


class TagDataset(Dataset):
    r"""Arguments
        data_folder: folder where data files are stored
        data_name: base name of processed datasets
        split: split, one of 'TRAIN', 'VAL', or 'TEST'
        transform: image transform pipeline
    """

    def __init__(self, data_folder, data_name, split, transform=None):
        self.split = split
        assert self.split in {'TRAIN', 'VAL', 'TEST'}
        self._paths = (os.path.join(data_folder, self.split + '_IMAGES_' + data_name + '.hdf5'),
                       os.path.join(data_folder, self.split + '_TAGS_' + data_name + '.hdf5'))
        self._open()
        self.transform = transform
        self.dataset_size = len(self.tags)

    def _open(self):
        self.h = h5lite.File(self._paths[0])
        self.imgs = self.h['images']
        self.t = h5lite.File(self._paths[1])
        self.tags = self.t['tags']

    def __getstate__(self):
        st = dict(self.__dict__)
        st['h'] = st['imgs'] = st['t'] = st['tags'] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._open()

    def __getitem__(self, i):
        img = torch.FloatTensor(self.imgs[i] / 255.)
        if self.transform is not None:
            img = self.transform(img)
        tags = torch.FloatTensor(self.tags[i])
        return img, tags

    def __len__(self):
        return self.dataset_size
