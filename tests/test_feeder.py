"""Input side of the path: feeder contract, augmentation arithmetic (CPU), checkpoint interchange, evaluation
loop; and the batched GPU augmentation against the oracle (gpu)."""
import os
import pickle
import random
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
sys.path.insert(0, ROOT)

from feeder import tools                                    # noqa: E402
from feeder.feeder import Feeder, load_label_file           # noqa: E402
from oracle import feeder_oracle as FO                      # noqa: E402
from stgcn_rt import checkpoint as CK                       # noqa: E402
from stgcn_rt.evaluate import evaluate, topk_accuracy       # noqa: E402


from tests.golden import cases as GC                        # noqa: E402

_GOLD = np.load(os.path.join(ROOT, "tests", "golden", "feeder.npz"))


def _ours(fn, d, kw, s_py, s_np):
    """Our feeder.tools function on clip d with the generators the reference drew from, seeded alike."""
    rng, nprng = random.Random(s_py), np.random.RandomState(s_np)
    if fn == "random_choose":
        return tools.random_choose(d, kw["size"], kw.get("auto_pad", True), rng=rng)
    if fn == "auto_pading":
        return tools.auto_pading(d, kw["size"], kw.get("random_pad", False), rng=rng)
    if fn == "random_move":
        return tools.random_move(d, rng=rng, nprng=nprng, **{k: tuple(v) for k, v in kw.items()})
    if fn == "random_shift":
        return tools.random_shift(d, rng=rng)
    if fn == "downsample":
        return tools.downsample(d, kw["step"], kw.get("random_sample", True), rng=nprng)
    if fn == "mean_subtractor":
        return tools.mean_subtractor(d, kw["mean"])
    if fn == "pipeline":
        return tools.random_move(np.array(tools.random_choose(d, kw["size"], rng=rng)), rng=rng, nprng=nprng)
    raise KeyError(fn)


@pytest.mark.parametrize("case", GC.FEEDER_CASES, ids=[c[0] for c in GC.FEEDER_CASES])
def test_tools_match_reference_fixtures(case):
    """feeder.tools against the outputs of the reference's feeder/tools.py (tests/golden/feeder.npz, written by
    tests/golden/make_golden.py with `random` / `numpy.random` seeded per case)."""
    name, fn, shape, gap, kw, s_py, s_np = case
    d = GC.feeder_clip(name, shape, gap)
    got = _ours(fn, d.copy(), kw, s_py, s_np)
    want = _GOLD[name]
    if want.size == 0:
        assert got is None
        return
    assert got.shape == want.shape
    if fn in ("random_move", "pipeline"):
        assert np.abs(got - want).max() < 1e-12            # same formula, products summed in another order
    else:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("case", [c for c in GC.FEEDER_CASES if c[1] in ("random_choose", "random_move", "pipeline")],
                         ids=lambda c: c[0])
def test_feeder_oracle_matches_reference_fixtures(case):
    """The augmentation oracle (window / move with the draws passed in) reproduces the reference's outputs when it
    is handed the draws the reference made."""
    name, fn, shape, gap, kw, s_py, s_np = case
    d = GC.feeder_clip(name, shape, gap)
    rng, nprng = random.Random(s_py), np.random.RandomState(s_np)
    out = d
    if fn in ("random_choose", "pipeline"):
        shift = tools.choose_shift(shape[1], kw["size"], kw.get("auto_pad", True), rng=rng)
        if shift is not None:
            out = FO.window(out, kw["size"], shift)
    if fn in ("random_move", "pipeline"):
        mk = {k: tuple(v) for k, v in kw.items() if k != "size"}
        out = FO.move(out, *tools.move_params(out.shape[1], rng=rng, nprng=nprng, **mk))
    assert out.shape == _GOLD[name].shape and np.abs(out - _GOLD[name]).max() < 1e-12


def clip(C=3, T=20, V=5, M=2, seed=0):
    return np.random.RandomState(seed).randn(C, T, V, M)


def test_window_matches_oracle_and_reference_semantics():
    d = clip(T=20)
    # longer than the window: a crop starting at the drawn frame
    rng = random.Random(3)
    begin = random.Random(3).randint(0, 20 - 8)
    out = tools.random_choose(d, 8, rng=rng)
    assert out.shape == (3, 8, 5, 2) and np.array_equal(out, d[:, begin:begin + 8])
    assert np.array_equal(out, FO.window(d, 8, begin))
    # shorter: zero padding at the drawn offset; equal: untouched
    rng = random.Random(5)
    b2 = random.Random(5).randint(0, 30 - 20)
    out = tools.random_choose(d, 30, rng=rng)
    assert out.shape == (3, 30, 5, 2) and np.array_equal(out, FO.window(d, 30, -b2))
    assert out[:, :b2].sum() == 0 and np.array_equal(out[:, b2:b2 + 20], d)
    assert tools.random_choose(d, 20) is d
    assert tools.random_choose(d, 30, auto_pad=False) is d
    # auto_pading: at the front unless random_pad
    p = tools.auto_pading(d, 25)
    assert np.array_equal(p[:, :20], d) and p[:, 20:].sum() == 0 and tools.auto_pading(d, 10) is d


def test_move_matches_oracle():
    d = clip(T=16, seed=1)
    a, s, tx, ty = tools.move_params(16, rng=random.Random(1), nprng=np.random.RandomState(2))
    # one segment: linear from the first to the second draw, end point included
    assert a.shape == (16,) and abs((a[1] - a[0]) - (a[-1] - a[0]) / 15) < 1e-12
    want = FO.move(d, a, s, tx, ty)
    got = tools.apply_move(d.copy(), a, s, tx, ty)
    assert np.abs(got - want).max() < 1e-12
    assert np.array_equal(got[2], d[2])                       # the third channel is not a coordinate
    # hand-checkable: quarter turn, scale 2, shift (1, -1) sends (1, 0) to (1, 1)
    one = np.zeros((3, 1, 1, 1)); one[0] = 1.0
    r = tools.apply_move(one, np.array([np.pi / 2]), np.array([2.0]), np.array([1.0]), np.array([-1.0]))
    assert np.allclose(r[:2, 0, 0, 0], [1.0, 1.0])
    # two segments: three nodes
    a3 = tools.move_params(10, move_time_candidate=(2,), rng=random.Random(0), nprng=np.random.RandomState(0))[0]
    assert a3.shape == (10,)


def test_other_tools():
    d = clip(T=12)
    assert tools.downsample(d, 3, random_sample=False).shape == (3, 4, 5, 2)
    z = np.zeros((3, 12, 5, 2)); z[:, 2:7] = d[:, 2:7]
    sh = tools.random_shift(z, rng=random.Random(4))
    bias = random.Random(4).randint(0, 12 - 5)
    assert np.array_equal(sh[:, bias:bias + 5], d[:, 2:7]) and sh.sum() == pytest.approx(d[:, 2:7].sum())
    m = z.copy()
    tools.mean_subtractor(m, 0.5)
    assert np.allclose(m[:, :7], z[:, :7] - 0.5) and np.array_equal(m[:, 7:], z[:, 7:])
    assert tools.mean_subtractor(z, 0) is None


def test_temporal_slice_by_hand():
    """tools.py:11-15 of the reference (which divides with `/` and cannot run under Python 3: parity unpinned by
    reference fixtures): out[c, g, v, s * M + m] = data[c, g * step + s, v, m]."""
    C, T, V, M, step = 2, 6, 3, 2, 3
    d = np.arange(C * T * V * M, dtype=np.float64).reshape(C, T, V, M)
    out = tools.temporal_slice(d, step)
    assert out.shape == (C, T // step, V, step * M)
    for c in range(C):
        for g in range(T // step):
            for v in range(V):
                for s in range(step):
                    for m in range(M):
                        assert out[c, g, v, s * M + m] == d[c, g * step + s, v, m]
    assert np.array_equal(tools.temporal_slice(d, 1), d)
    with pytest.raises(ValueError):
        tools.temporal_slice(d, 4)


def test_feeder_contract(tmp_path):
    N = 6
    data = np.random.RandomState(0).randn(N, 3, 20, 5, 2).astype(np.float32)
    np.save(tmp_path / "d.npy", data)
    names = ["S%03d.skeleton" % i for i in range(N)]
    labels = [int(i % 4) for i in range(N)]
    with open(tmp_path / "l.pkl", "wb") as f:
        pickle.dump((names, labels), f)
    fd = Feeder(str(tmp_path / "d.npy"), str(tmp_path / "l.pkl"))
    assert len(fd) == N and (fd.N, fd.C, fd.T, fd.V, fd.M) == (N, 3, 20, 5, 2)
    x, y = fd[2]
    assert np.array_equal(x, data[2]) and y == 2
    fw = Feeder(str(tmp_path / "d.npy"), str(tmp_path / "l.pkl"), random_choose=True, random_move=True, window_size=8)
    x, _ = fw[1]
    assert x.shape == (3, 8, 5, 2)
    batch = next(iter(torch.utils.data.DataLoader(fw, batch_size=3)))
    assert tuple(batch[0].shape) == (3, 3, 8, 5, 2) and batch[1].dtype == torch.int64
    assert Feeder(str(tmp_path / "d.npy"), str(tmp_path / "l.pkl"), window_size=30)[0][0].shape == (3, 30, 5, 2)

    # a label file is data: anything that asks for code is refused
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with open(tmp_path / "evil.pkl", "wb") as f:
        pickle.dump((names, [Evil()]), f)
    with pytest.raises(pickle.UnpicklingError):
        load_label_file(str(tmp_path / "evil.pkl"))


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(4, 3)
        self.bn = torch.nn.BatchNorm1d(3)
        self.register_buffer("A", torch.eye(3))

    def forward(self, x):
        return self.bn(self.fc(x.reshape(x.shape[0], -1)))


def test_checkpoint_interchange(tmp_path):
    torch.manual_seed(0)
    m = _Tiny()
    p = str(tmp_path / "w.pt")
    saved = CK.save_model(torch.nn.DataParallel(m) if False else m, p)
    assert list(saved.keys()) == list(m.state_dict().keys()) and "A" in saved
    # DataParallel-style names load as well; the last `module.` and everything before it goes
    torch.save(OrderedDict(("module." + k, v) for k, v in saved.items()), p)
    m2 = _Tiny()
    logs = []
    CK.load_weights(m2, p, log=logs.append)
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v), k
    assert logs[0].startswith("Load weights from")
    # ignored prefixes keep the model's own values; so do entries the file lacks
    m3 = _Tiny()
    keep = m3.fc.weight.detach().clone()
    CK.load_weights(m3, p, ignore_weights="fc", log=logs.append)
    assert torch.equal(m3.fc.weight, keep) and torch.equal(m3.bn.weight, m.bn.weight)
    assert any(l.startswith("Filter [fc] remove weights [fc.weight]") for l in logs)
    assert any(l.startswith("Can not find weights [fc.") for l in logs)
    # what the model does not have is an error, as with load_state_dict
    torch.save(OrderedDict(list(saved.items()) + [("extra.w", torch.zeros(1))]), p)
    with pytest.raises(RuntimeError):
        CK.load_weights(_Tiny(), p)


def test_model_from_config_builds_the_reference_config(tmp_path):
    cfg = tmp_path / "train.yaml"
    cfg.write_text("model: net.st_gcn.Model\nmodel_args:\n  in_channels: 3\n  num_class: 60\n  dropout: 0.5\n"
                   "  edge_importance_weighting: True\n  graph_args:\n    layout: 'ntu-rgb+d'\n    strategy: 'spatial'\n")
    m = CK.model_from_config(str(cfg))
    sd = m.state_dict()
    assert tuple(sd["A"].shape) == (3, 25, 25) and tuple(sd["fcn.weight"].shape) == (60, 256, 1, 1)
    assert len(m.st_gcn_networks) == 10 and len(m.edge_importance) == 10


def test_evaluate_and_topk():
    res = np.array([[0.1, 0.7, 0.2], [0.5, 0.3, 0.2], [0.2, 0.3, 0.5]])
    assert topk_accuracy(res, [1, 2, 0], 1) == pytest.approx(1 / 3)
    assert topk_accuracy(res, [1, 2, 0], 2) == pytest.approx(1 / 3)
    assert topk_accuracy(res, [1, 1, 1], 2) == pytest.approx(1.0)
    torch.manual_seed(0)
    m = _Tiny()
    xs = torch.randn(10, 4)
    ys = torch.randint(0, 3, (10,))
    loader = [(xs[:6], ys[:6]), (xs[6:], ys[6:])]
    out = evaluate(m, loader, torch.device("cpu"), show_topk=(1, 3))
    assert not m.training and out["result"].shape == (10, 3) and np.array_equal(out["label"], ys.numpy())
    with torch.no_grad():
        full = m(xs)
    assert np.allclose(out["result"], full.numpy(), atol=1e-6)
    ce = torch.nn.CrossEntropyLoss()
    assert out["mean_loss"] == pytest.approx(np.mean([ce(full[:6], ys[:6]).item(), ce(full[6:], ys[6:]).item()]), rel=1e-5)
    assert out["topk"][3] == 1.0 and out["topk"][1] == pytest.approx((full.argmax(1) == ys).float().mean().item())


def test_item_with_draws_is_getitem_unapplied(tmp_path):
    N, T, W = 4, 20, 12
    data = np.random.RandomState(3).randn(N, 3, T, 5, 2).astype(np.float32)
    np.save(tmp_path / "d.npy", data)
    with open(tmp_path / "l.pkl", "wb") as f:
        pickle.dump((["a"] * N, list(range(N))), f)
    fd = Feeder(str(tmp_path / "d.npy"), str(tmp_path / "l.pkl"), random_choose=True, random_move=True, window_size=W)
    for i in range(N):
        random.seed(40 + i); np.random.seed(50 + i)
        want, lab = fd[i]
        raw, lab2, shift, move = fd.item_with_draws(i, rng=random.Random(40 + i), nprng=np.random.RandomState(50 + i))
        assert lab == lab2 and np.array_equal(raw, data[i])
        got = FO.move(FO.window(raw.astype(np.float64), W, shift), *move)
        assert np.abs(got - want).max() < 1e-6
    fp = Feeder(str(tmp_path / "d.npy"), str(tmp_path / "l.pkl"), window_size=30)
    raw, _, shift, move = fp.item_with_draws(0)
    assert shift == 0 and move is None and np.array_equal(FO.window(raw, 30, 0), fp[0][0])


@pytest.mark.gpu
def test_augment_batch_gpu_matches_reference_fixtures():
    """stgcn_augment (window + move for a batch in one launch) against the reference's Feeder.__getitem__ pipeline
    outputs (tests/golden/feeder.npz `pipeline_*`, `choose_*`, `move_*`), draws reproduced from the recorded seeds."""
    from stgcn_rt import ops as P
    for name, fn, shape, gap, kw, s_py, s_np in GC.FEEDER_CASES:
        if fn not in ("pipeline", "random_choose", "random_move"):
            continue
        d = GC.feeder_clip(name, shape, gap)
        rng, nprng = random.Random(s_py), np.random.RandomState(s_np)
        shift, move, W = None, None, shape[1]
        if fn in ("pipeline", "random_choose"):
            shift = tools.choose_shift(shape[1], kw["size"], kw.get("auto_pad", True), rng=rng)
            if shift is not None:
                W = kw["size"]
        if fn in ("pipeline", "random_move"):
            mk = {k: tuple(v) for k, v in kw.items() if k != "size"}
            move = tuple(np.asarray(v)[None] for v in tools.move_params(W, rng=rng, nprng=nprng, **mk))
        xd = torch.from_numpy(d.astype(np.float32))[None].cuda()
        got = P.augment_batch(xd, window=W, shift=None if shift is None else [shift], move=move)[0].cpu().numpy()
        want = _GOLD[name]
        assert got.shape == want.shape, name
        assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max()), (name, np.abs(got - want).max())


@pytest.mark.gpu
def test_augment_batch_gpu_matches_oracle():
    from stgcn_rt import ops as P
    N, C, T, V, M, W = 5, 3, 40, 25, 2, 24
    rs = np.random.RandomState(7)
    x = rs.randn(N, C, T, V, M).astype(np.float32)
    shifts = [tools.choose_shift(T, W, rng=random.Random(i)) for i in range(N)]
    params = [tools.move_params(W, rng=random.Random(10 + i), nprng=np.random.RandomState(20 + i)) for i in range(N)]
    move = tuple(np.stack([p[j] for p in params]) for j in range(4))
    xd = torch.from_numpy(x).cuda()
    got = P.augment_batch(xd, window=W, shift=shifts, move=move).cpu().numpy()
    for n in range(N):
        want = FO.move(FO.window(x[n].astype(np.float64), W, shifts[n]), *params[n])
        assert np.abs(got[n] - want).max() < 1e-5, n
    # padding (window longer than the clip) and window only
    shp = [-3, 0, -10, -1, -7]
    got = P.augment_batch(xd, window=50, shift=shp).cpu().numpy()
    for n in range(N):
        assert np.array_equal(got[n], FO.window(x[n], 50, shp[n]).astype(np.float32)), n
    # move only, padded frames are moved too (zeros -> the translation), as the reference does after padding
    a = np.zeros((N, T)); s = np.ones((N, T)); tx = np.full((N, T), 0.25); ty = np.full((N, T), -0.5)
    got = P.augment_batch(xd, move=(a, s, tx, ty)).cpu().numpy()
    assert np.allclose(got[:, 0], x[:, 0] + 0.25, atol=1e-6) and np.allclose(got[:, 1], x[:, 1] - 0.5, atol=1e-6)
    assert np.array_equal(got[:, 2], x[:, 2])
