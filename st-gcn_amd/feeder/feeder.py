"""Dataset of pre-converted skeleton clips, the contract of the reference's feeder.feeder.Feeder
(feeder/feeder.py:22-85 of hensyou/st-gcn): `data_path` is a float32 .npy of shape (N, C, T, V, M) (memory-mapped),
`label_path` a pickle of (sample_name list, label list) as tools/ntu_gendata.py:78-94 writes them; an item is
(array (C, T', V, M), int label) after the optional window / move augmentation.
"""
import io
import pickle

import numpy as np
import torch

from . import tools

_ALLOWED = {
    ("builtins", "list"), ("builtins", "tuple"), ("builtins", "str"), ("builtins", "int"), ("builtins", "dict"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "bytes"), ("builtins", "float"), ("builtins", "bool"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "scalar"), ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
}


class _DataOnlyUnpickler(pickle.Unpickler):
    """Label files are plain lists of names and integers; nothing else is allowed to be constructed."""

    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("label file names %s.%s: only plain data is loaded" % (module, name))


def load_label_file(path):
    with open(path, "rb") as f:
        sample_name, label = _DataOnlyUnpickler(io.BytesIO(f.read())).load()
    return list(sample_name), list(label)


class Feeder(torch.utils.data.Dataset):
    """Arguments as the reference's Feeder: data_path, label_path, random_choose (random window of window_size
    frames), random_move (random affine motion of the x-y plane), window_size (> 0 without random_choose: pad
    shorter clips), debug (first 100 clips), mmap."""

    def __init__(self, data_path, label_path, random_choose=False, random_move=False, window_size=-1, debug=False,
                 mmap=True):
        self.debug = debug
        self.data_path = data_path
        self.label_path = label_path
        self.random_choose = random_choose
        self.random_move = random_move
        self.window_size = window_size
        self.load_data(mmap)

    def load_data(self, mmap):
        self.sample_name, self.label = load_label_file(self.label_path)
        self.data = np.load(self.data_path, mmap_mode="r") if mmap else np.load(self.data_path)
        if self.debug:
            self.label = self.label[0:100]
            self.data = self.data[0:100]
            self.sample_name = self.sample_name[0:100]
        self.N, self.C, self.T, self.V, self.M = self.data.shape

    def __len__(self):
        return len(self.label)

    def __getitem__(self, index):
        data = np.array(self.data[index])
        label = self.label[index]
        if self.random_choose:
            data = tools.random_choose(data, self.window_size)
        elif self.window_size > 0:
            data = tools.auto_pading(data, self.window_size)
        if self.random_move:
            data = tools.random_move(data)
        return data, label
