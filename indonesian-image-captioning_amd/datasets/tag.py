"""Drop-in for the reference's `datasets/tag.py` (the tagger's training set): item i = (image i scaled to
[0, 1] and transformed, ground-truth tag vector i), read through `scnattn.h5lite` instead of h5py."""
import torch
from torch.utils.data import Dataset

from datasets._mapped import SPLITS, MappedArrays, split_file, unit_float_image


class TagDataset(Dataset):
    """TagDataset(data_folder, data_name, split, transform=None) — constructor and items of reference
    datasets/tag.py:20-58; `len` is the number of tag rows."""

    def __init__(self, data_folder, data_name, split, transform=None):
        assert split in SPLITS
        self.split, self.transform = split, transform
        self._arrays = MappedArrays(imgs=(split_file(data_folder, split, 'IMAGES', data_name, '.hdf5'), 'images'),
                                    tags=(split_file(data_folder, split, 'TAGS', data_name, '.hdf5'), 'tags'))
        self.dataset_size = len(self._arrays.tags)

    @property
    def imgs(self):
        return self._arrays.imgs

    @property
    def tags(self):
        return self._arrays.tags

    def __len__(self):
        return self.dataset_size

    def __getitem__(self, i):
        img = unit_float_image(self.imgs[i])
        if self.transform is not None:
            img = self.transform(img)
        return img, torch.FloatTensor(self.tags[i])
