"""Skeleton-sequence augmentation, the functions of the reference's feeder/tools.py (hensyou/st-gcn) that its
NTU / Kinetics feeders call, on arrays shaped (C, T, V, M).

Random draws are separated from the arithmetic: every `random_*` function takes an optional `rng`
(random.Random-like for integer draws, numpy Generator/RandomState-like for `choice`) and has a deterministic
counterpart taking the drawn values, which is what the parity tests and the batched GPU stage
(stgcn_rt.ops.augment_batch) share.
"""
import random as _random

import numpy as np


def downsample(data, step, random_sample=True, rng=None):
    """Every step-th frame from a random (or the first) phase (tools.py:5-8)."""
    begin = (rng or np.random).randint(step) if random_sample else 0
    return data[:, begin::step, :, :]


def temporal_slice(data, step):
    """Groups of `step` consecutive frames folded into the person axis: (C, T, V, M) -> (C, T // step, V, step * M)
    with out[c, g, v, s * M + m] = data[c, g * step + s, v, m] (tools.py:11-15).  T must be a multiple of `step`.
    The reference's own version divides with `/` and cannot run under Python 3, so no fixture of it exists: this
    function is checked by hand-computed cases only."""
    C, T, V, M = data.shape
    if T % step != 0:
        raise ValueError("temporal_slice: %d frames are no multiple of step %d" % (T, step))
    G = T // step
    return data.reshape(C, G, step, V, M).transpose((0, 1, 3, 2, 4)).reshape(C, G, V, step * M)


def mean_subtractor(data, mean):
    """Subtract `mean` from every frame up to the last non-empty one, in place (tools.py:18-28).  mean == 0
    returns None, as the reference does."""
    if mean == 0:
        return None
    valid = (data != 0).sum(axis=3).sum(axis=2).sum(axis=0) > 0
    end = len(valid) - valid[::-1].argmax()
    data[:, :end, :, :] = data[:, :end, :, :] - mean
    return data


def pad_to(data, size, begin=0):
    """Zero array of `size` frames with `data` placed at frame `begin` (float64, as np.zeros gives)."""
    C, T, V, M = data.shape
    out = np.zeros((C, size, V, M))
    out[:, begin:begin + T, :, :] = data
    return out


def auto_pading(data, size, random_pad=False, rng=None):
    """Shorter sequences are zero-padded to `size` frames, at a random offset if asked (tools.py:31-39)."""
    T = data.shape[1]
    if T < size:
        begin = (rng or _random).randint(0, size - T) if random_pad else 0
        return pad_to(data, size, begin)
    return data


def choose_shift(T, size, auto_pad=True, rng=None):
    """The draw of random_choose as a frame shift: output frame t shows input frame t + shift (outside: zeros).
    None means the sequence is returned as it is."""
    if T == size or (T < size and not auto_pad):
        return None
    if T < size:
        return -(rng or _random).randint(0, size - T)
    return (rng or _random).randint(0, T - size)


def random_choose(data, size, auto_pad=True, rng=None):
    """A random window of `size` frames; shorter sequences are padded at a random offset (tools.py:42-54)."""
    T = data.shape[1]
    shift = choose_shift(T, size, auto_pad, rng)
    if shift is None:
        return data
    if shift < 0 or T < size:
        return pad_to(data, size, -shift)
    return data[:, shift:shift + size, :, :]


def move_params(T, angle_candidate=(-10., -5., 0., 5., 10.), scale_candidate=(0.9, 1.0, 1.1),
                transform_candidate=(-0.2, -0.1, 0.0, 0.1, 0.2), move_time_candidate=(1,), rng=None, nprng=None):
    """The draws of random_move turned into per-frame rotation angle (radians), scale and translation
    (tools.py:57-94): values are drawn at `move_time`+1 nodes and linearly interpolated in between
    (np.linspace per segment, end point included, as the reference does)."""
    move_time = (rng or _random).choice(list(move_time_candidate))
    node = np.arange(0, T, T * 1.0 / move_time).round().astype(int)
    node = np.append(node, T)
    n = len(node)
    nprng = nprng or np.random
    A = nprng.choice(list(angle_candidate), n)
    S = nprng.choice(list(scale_candidate), n)
    Tx = nprng.choice(list(transform_candidate), n)
    Ty = nprng.choice(list(transform_candidate), n)
    a, s, tx, ty = np.zeros(T), np.zeros(T), np.zeros(T), np.zeros(T)
    for i in range(n - 1):
        seg = slice(node[i], node[i + 1])
        m = node[i + 1] - node[i]
        a[seg] = np.linspace(A[i], A[i + 1], m) * np.pi / 180
        s[seg] = np.linspace(S[i], S[i + 1], m)
        tx[seg] = np.linspace(Tx[i], Tx[i + 1], m)
        ty[seg] = np.linspace(Ty[i], Ty[i + 1], m)
    return a, s, tx, ty


def apply_move(data, a, s, tx, ty):
    """x' = s (cos a · x − sin a · y) + tx,  y' = s (sin a · x + cos a · y) + ty on channels 0 and 1 of every frame,
    in place (tools.py:96-107); other channels (confidence / depth) are untouched."""
    ca, sa = (np.cos(a) * s)[:, None, None], (np.sin(a) * s)[:, None, None]
    x, y = data[0].copy(), data[1].copy()
    data[0] = ca * x - sa * y + tx[:, None, None]
    data[1] = sa * x + ca * y + ty[:, None, None]
    return data


def random_move(data, angle_candidate=(-10., -5., 0., 5., 10.), scale_candidate=(0.9, 1.0, 1.1),
                transform_candidate=(-0.2, -0.1, 0.0, 0.1, 0.2), move_time_candidate=(1,), rng=None, nprng=None):
    """Random smooth rotation / scaling / translation of the x-y plane (tools.py:57-109)."""
    a, s, tx, ty = move_params(data.shape[1], angle_candidate, scale_candidate, transform_candidate,
                               move_time_candidate, rng, nprng)
    return apply_move(data, a, s, tx, ty)


def random_shift(data, rng=None):
    """The non-empty frame range moved to a random position (tools.py:112-124)."""
    C, T, V, M = data.shape
    out = np.zeros(data.shape)
    valid = (data != 0).sum(axis=3).sum(axis=2).sum(axis=0) > 0
    begin = valid.argmax()
    end = len(valid) - valid[::-1].argmax()
    size = end - begin
    bias = (rng or _random).randint(0, T - size)
    out[:, bias:bias + size, :, :] = data[:, begin:end, :, :]
    return out
