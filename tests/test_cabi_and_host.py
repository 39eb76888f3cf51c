"""CPU: the C-ABI library loads and exports every symbol include/stgcn_hip.h declares (no compute
calls without a GPU); the drop-in modules keep the reference's interface; host-side packing logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "stgcn_hip.h")).read()
    return sorted(set(re.findall(r"\b(stgcn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from stgcn_rt import lib
    assert os.path.exists(lib.lib_path()), "build the library first: make -C st-gcn_amd/csrc"
    h = ctypes.CDLL(lib.lib_path())
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(h, n), n
    # the ctypes table binds exactly the declared entry points
    assert sorted(lib.EXPORTS) == names
    l = lib.load()
    assert l.stgcn_version() >= 1 and l.stgcn_arch() == b"gfx950"


def test_launch_state_is_kept_per_device():
    """Compute-unit counts and the large-LDS opt-ins are tables over device ordinals (a process may drive several
    GPUs, as nn.DataParallel does: processor/io.py:86-87 of the reference), not process-wide flags."""
    from stgcn_rt import lib
    l = lib.load()
    cus, opt = ctypes.c_int(-1), ctypes.c_int(-1)
    for d in (0, 1, 7, 63):
        assert l.stgcn_launch_state(d, ctypes.byref(cus), ctypes.byref(opt)) == 0
        assert cus.value == 0 and opt.value == 0            # nothing launched in this process: every slot untouched
    assert l.stgcn_launch_state(64, ctypes.byref(cus), ctypes.byref(opt)) == 1
    assert l.stgcn_launch_state(-1, None, None) == 1
    # no launcher keeps a process-wide "already set" flag any more
    src = os.path.join(ROOT, "st-gcn_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith(".hip"):
            txt = open(os.path.join(src, f)).read()
            assert not re.search(r"static\s+bool\s+\w+(\[\d+\])?\s*=", txt), f
            assert not re.search(r"\b256\s*/\s*\(?n", txt), f  # grids are sized from the device's CU count


def test_no_cpu_fallback():
    from net.st_gcn import Model
    from net.utils.tgcn import ConvTemporalGraphical
    m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 8, 25, 2))
    g = ConvTemporalGraphical(8, 8, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 8, 4, 25), torch.zeros(3, 25, 25))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "st-gcn_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(d, f)).read(), os.path.join(d, f)


@pytest.mark.parametrize("layout,ncls,V", [("ntu-rgb+d", 60, 25), ("openpose", 400, 18)])
def test_model_interface_matches_reference(layout, ncls, V):
    from net.st_gcn import Model, st_gcn
    from oracle import stgcn_oracle as O
    m = Model(3, ncls, dict(layout=layout, strategy="spatial"), True, dropout=0.5)
    shapes = O.model_state_shapes(3, ncls, 3, V)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert sum(p.numel() for p in m.parameters()) == {60: 3098832, 400: 3177140}[ncls]
    assert m.graph.edge and m.A.shape == (3, V, V) and torch.all(m.edge_importance[0] == 1)
    # weights_init of the reference keys on these class-name substrings (processor/recognition.py:21-33)
    names = [mod.__class__.__name__ for mod in m.modules()]
    assert sum("Conv2d" in n for n in names) == 10 + 10 + 2 + 1 and sum("BatchNorm" in n for n in names) == 1 + 20 + 2
    assert not any("Conv2d" in n or "Conv1d" in n for n in ["ConvTemporalGraphical"])
    # block 0 never drops, the others take the dropout kwarg (st_gcn.py:43-55)
    assert m.st_gcn_networks[0].tcn[4].p == 0 and m.st_gcn_networks[5].tcn[4].p == 0.5
    m2 = Model(3, ncls, dict(layout=layout, strategy="uniform"), False)
    assert m2.edge_importance == [1] * 10 and "edge_importance.0" not in m2.state_dict()
    with pytest.raises(AssertionError):
        st_gcn(3, 8, (8, 3))                       # even temporal kernel
    with pytest.raises(ValueError):
        Model(3, ncls, dict(layout="nope"), True)


def test_checkpoint_roundtrip_with_reference_keys(tmp_path):
    """Flat OrderedDict of CPU tensors, `module.` prefix stripped: the wire format torchlight writes
    (torchlight/torchlight/io.py:57-90, 101-107 of the reference)."""
    from collections import OrderedDict
    from net.st_gcn import Model
    m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True)
    ck = OrderedDict(("module." + k, v.cpu()) for k, v in m.state_dict().items())
    p = str(tmp_path / "ck.pt")
    torch.save(ck, p)
    loaded = torch.load(p, weights_only=True)
    stripped = OrderedDict((k.split("module.")[-1], v) for k, v in loaded.items())
    m2 = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True)
    m2.load_state_dict(stripped, strict=True)


def test_weight_packing():
    from stgcn_rt import ops as P
    K, Cout, Cin = 3, 4, 3
    W = torch.arange(K * Cout * Cin, dtype=torch.float32).reshape(K * Cout, Cin, 1, 1)
    wp = P.pack_gcn_w(W, K, 8, torch.float32)
    assert wp.shape == (K, Cout, 8) and wp[1, 2, 1] == W[1 * Cout + 2, 1, 0, 0] and wp[:, :, 3:].abs().sum() == 0
    wt = P.pack_gcn_wt(W, K, 8, torch.float32)
    assert wt.shape == (K, 8, Cout) and wt[2, 1, 3] == W[2 * Cout + 3, 1, 0, 0] and wt[:, 3:].abs().sum() == 0
    Wc = torch.arange(5 * 6 * 9, dtype=torch.float32).reshape(5, 6, 9, 1)
    assert P.pack_conv_w(Wc, torch.float32)[7, 4, 2] == Wc[4, 2, 7, 0]
    assert P.pack_conv_wt(Wc, torch.float32)[7, 2, 4] == Wc[4, 2, 7, 0]
    assert P.pad_to(3, 16) == 16 and P.pad_to(64, 16) == 64 and P.kstep(torch.bfloat16) == 16 and P.kchunk(torch.float32) == 32


def test_graph_matches_golden(golden):
    from net.utils.graph import Graph
    for key in golden["graph"].files:
        lay, strat, mh, dil = key.split("|")
        g = Graph(lay, strat, int(mh), int(dil))
        assert np.array_equal(g.A, golden["graph"][key]), key
