"""Shared plumbing of the drop-in datasets: memory-mapped HDF5 members that survive pickling (DataLoader
workers re-open the map themselves, so any `num_workers` works — the reference notes "only 1 works with
h5py", trains/attention_scn.py:49)."""
import os

import torch

from scnattn import h5lite

SPLITS = ('TRAIN', 'VAL', 'TEST')


def split_file(folder, split, kind, data_name, ext):
    return os.path.join(folder, '%s_%s_%s%s' % (split, kind, data_name, ext))


class MappedArrays:
    """name -> (path, dataset key); attribute `name` is the h5lite dataset, `name_file` its open file."""

    def __init__(self, **members):
        self._members = members
        self._open()

    def _open(self):
        for name, (path, key) in self._members.items():
            f = h5lite.File(path)
            setattr(self, name + '_file', f)
            setattr(self, name, f[key])

    def __getstate__(self):
        return {'_members': self._members}

    def __setstate__(self, state):
        self._members = state['_members']
        self._open()


def unit_float_image(u8_chw):
    """`torch.FloatTensor(img / 255.)` of the reference (datasets/caption.py:51, datasets/tag.py:48): float64
    divide in numpy, rounded to float32."""
    return torch.FloatTensor(u8_chw / 255.)
