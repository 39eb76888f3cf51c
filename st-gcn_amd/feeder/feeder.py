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
    shorter clips), debug (first 100 clips), mmap.

    An item runs the clip through a list of stages fixed at construction (window stage, then move stage);
    `item_with_draws` returns the untouched clip together with the random draws of those stages instead, for batches
    whose arithmetic runs on the GPU (stgcn_rt.ops.augment_batch)."""

    DEBUG_CLIPS = 100

    def __init__(self, data_path, label_path, random_choose=False, random_move=False, window_size=-1, debug=False,
                 mmap=True):
        self.debug = debug
        self.data_path = data_path
        self.label_path = label_path
        self.random_choose = random_choose
        self.random_move = random_move
        self.window_size = window_size
        self.load_data(mmap)

    def _stages(self):
        stages = []
        if self.random_choose:
            stages.append(lambda clip: tools.random_choose(clip, self.window_size))
        elif self.window_size > 0:
            stages.append(lambda clip: tools.auto_pading(clip, self.window_size))
        if self.random_move:
            stages.append(tools.random_move)
        return stages

    def load_data(self, mmap):
        names, labels = load_label_file(self.label_path)
        clips = np.load(self.data_path, mmap_mode="r" if mmap else None)
        keep = self.DEBUG_CLIPS if self.debug else len(labels)
        self.sample_name, self.label, self.data = names[:keep], labels[:keep], clips[:keep]
        self.N, self.C, self.T, self.V, self.M = self.data.shape

    def __len__(self):
        return len(self.label)

    def __getitem__(self, index):
        clip = np.array(self.data[index])
        for stage in self._stages():
            clip = stage(clip)
        return clip, self.label[index]

    def item_with_draws(self, index, rng=None, nprng=None):
        """-> (raw clip (C,T,V,M), label, shift or None, (a, s, tx, ty) or None): what __getitem__ would draw, in the
        same order, without applying it."""
        size = self.window_size
        shift = move = None
        if self.random_choose:
            shift = tools.choose_shift(self.T, size, rng=rng)
        elif 0 < self.T < size:
            shift = 0
        if self.random_move:
            move = tools.move_params(size if shift is not None else self.T, rng=rng, nprng=nprng)
        return np.array(self.data[index]), self.label[index], shift, move
