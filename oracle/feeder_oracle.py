"""TEST INFRASTRUCTURE -- CPU restatement of the reference's clip augmentation (hensyou/st-gcn feeder/tools.py),
with the random draws passed in, frame by frame as the reference loops.  Imported only by tests/ (and never by the
product path).  Pinned by tests/golden/feeder.npz: outputs of the reference's own feeder/tools.py on closed-form
clips with seeded generators (tests/golden/make_golden.py gen_feeder; feeder.tools needs only numpy and random --
feeder.feeder itself wants torchvision and is not importable), plus hand-checkable cases in tests/test_feeder.py.
"""
import numpy as np


def window(data, size, shift):
    """random_choose / auto_pading (tools.py:31-54) with the drawn begin: output frame t = input frame t + shift,
    zeros outside."""
    C, T, V, M = data.shape
    out = np.zeros((C, size, V, M))
    for t in range(size):
        src = t + shift
        if 0 <= src < T:
            out[:, t] = data[:, src]
    return out


def move(data, a, s, tx, ty):
    """The transformation loop of random_move (tools.py:91-107): theta = [[cos a * s, -sin a * s], [sin a * s,
    cos a * s]] per frame, applied to channels 0:2 of every (joint, body), then the translation."""
    out = np.array(data, dtype=np.float64, copy=True)
    C, T, V, M = out.shape
    for t in range(T):
        theta = np.array([[np.cos(a[t]) * s[t], -np.sin(a[t]) * s[t]], [np.sin(a[t]) * s[t], np.cos(a[t]) * s[t]]])
        xy = out[0:2, t, :, :].reshape(2, -1)
        new_xy = theta.dot(xy)
        new_xy[0] += tx[t]
        new_xy[1] += ty[t]
        out[0:2, t, :, :] = new_xy.reshape(2, V, M)
    return out
