"""CPU: the oracle restatement against the golden vectors produced by the reference's own
modules (tests/golden/make_golden.py).  Storage precision of the fixtures is float32."""
import numpy as np
import pytest
import torch

from oracle import stgcn_oracle as O
from tests.golden import cases as C

TOL = 2e-6   # float32 storage of float64 reference outputs


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("case", C.GRAPH_CASES, ids=lambda c: "%s-%s-%d-%d" % c)
def test_graph(golden, case):
    lay, strat, mh, dil = case
    ref = golden["graph"]["%s|%s|%d|%d" % case]
    g = O.build_graph(lay, strat, mh, dil)
    assert g["A"].shape == ref.shape
    assert np.array_equal(g["A"], ref)          # bit-exact float64
    # every column of sum_k A_k sums to one (column-normalised adjacency)
    assert np.allclose(g["A"].sum(axis=(0, 1)), 1.0)


def test_graph_errors():
    with pytest.raises(ValueError):
        O.build_graph("nope", "spatial")
    with pytest.raises(ValueError):
        O.build_graph("openpose", "nope")


@pytest.mark.parametrize("ci", range(len(C.GCN_CASES)), ids=[c[0] for c in C.GCN_CASES])
def test_gcn_op(golden, ci):
    case = C.GCN_CASES[ci]
    x, A, W, b, dy = C.gcn_inputs(case, 1000 + 10 * ci)
    y = O.gcn_forward(x, A, W, b)
    dx, dW, db, dA = O.gcn_backward(x, A, W, b, dy)
    for k, v in dict(y=y, dx=dx, dW=dW, db=db, dA=dA).items():
        assert rel(C.compact(v), golden["gcn_op"][case[0] + "/" + k]) < TOL, k


@pytest.mark.parametrize("ci", range(len(C.GCN_T_CASES)), ids=[c[0] for c in C.GCN_T_CASES])
def test_gcn_with_temporal_kernel(golden, ci):
    """ConvTemporalGraphical with t_kernel_size > 1, stride, padding, dilation (net/utils/tgcn.py:44-66)."""
    case = C.GCN_T_CASES[ci]
    stride, pad, dil = case[8], case[9], case[10]
    x, A, W, b, dy = C.gcn_t_inputs(case, 7000 + 10 * ci)
    y = O.gcn_temporal_forward(x, A, W, b, stride, pad, dil)
    dx, dW, db, dA = O.gcn_temporal_backward(x, A, W, b, dy, stride, pad, dil)
    for k, v in dict(y=y, dx=dx, dW=dW, db=db, dA=dA).items():
        assert rel(C.compact(v), golden["gcn_temporal"][case[0] + "/" + k]) < TOL, k


@pytest.mark.parametrize("ci", range(len(C.BLOCK_CASES)), ids=[c[0] for c in C.BLOCK_CASES])
def test_block(golden, ci):
    case = C.BLOCK_CASES[ci]
    name, lay, cin, cout, stride, residual, B, T, training = case
    x, A, p, bufs, dout, kind = C.block_inputs(case, 5000 + 100 * ci)
    out, cache, nb = O.block_forward(x, A, p, bufs, stride, kind, training)
    dx, dA, g = O.block_backward(dout, p, cache, stride, kind, training)
    mine = {"out": out, "dx": dx, "dA": dA}
    mine.update({"g:" + k: v for k, v in g.items()})
    mine.update({"buf:" + k: v for k, v in nb.items()})
    G = golden["block"]
    checked = 0
    for key in G.files:
        if not key.startswith(name + "/"):
            continue
        k = key[len(name) + 1:]
        if training and k in ("g:tcn.2.bias", "g:residual.0.bias"):
            assert np.abs(C.compact(mine[k])).max() < 1e-6   # exactly zero up to rounding
        else:
            assert rel(C.compact(mine[k]), G[key]) < TOL, k
        checked += 1
    assert checked >= 11


@pytest.mark.parametrize("ci", [0, 1, 2], ids=[c[0] for c in C.MODEL_CASES[:3]])
def test_model_small(golden, ci):
    _check_model(golden, ci)


@pytest.mark.parametrize("ci", [3], ids=[C.MODEL_CASES[3][0]])
def test_model_full_ntu(golden, ci):
    _check_model(golden, ci)


_K12 = [i for i, c in enumerate(C.MODEL_CASES) if C.strategy_of(c[0]) != "spatial"]


@pytest.mark.parametrize("ci", _K12, ids=[C.MODEL_CASES[i][0] for i in _K12])
def test_model_uniform_distance(golden, ci):
    """K = 1 / 2 partitions (uniform / distance strategies, graph.py:85-94 of the reference)."""
    _check_model(golden, ci)


def _check_model(golden, ci):
    name, lay, ncls, N, T, M, training = C.MODEL_CASES[ci]
    g = O.build_graph(lay, C.strategy_of(name))
    sd = O.det_state(3, ncls, g["A"], True)
    x, labels = C.model_inputs(ci)
    logits, cache, newb = O.model_forward(sd, x, training)
    loss, dlog = O.cross_entropy(logits, labels)
    G = golden["model"]
    assert rel(logits, G[name + "/logits"]) < 1e-9
    assert abs(loss.item() - float(G[name + "/loss"])) < 1e-9
    if training:
        grads = O.model_backward(sd, cache, dlog, True)
        for k in O.param_names(sd):
            gr = grads[k]
            ref_norm = float(G[name + "/gnorm:" + k])
            if k.endswith("tcn.2.bias") or k.endswith("residual.0.bias"):
                assert gr.abs().max().item() < 1e-9
                continue
            assert abs(gr.norm().item() - ref_norm) <= 1e-8 * ref_norm + 1e-14, k
            flat = gr.reshape(-1)
            samp = flat[:: max(1, flat.numel() // 257)].numpy()
            assert rel(samp, G[name + "/gsample:" + k]) < 1e-7, k
        for key in G.files:
            if key.startswith(name + "/buf:"):
                assert rel(newb[key.split("buf:")[1]], G[key]) < 1e-9
    if name.endswith("small_eval"):
        o2, f2 = O.model_forward(sd, x, False, return_feature=True)
        assert rel(o2, G[name + "/feat_output"]) < TOL
        assert rel(f2, G[name + "/feat_feature"]) < TOL


def test_sgd_steps(golden):
    G = golden["sgd"]
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True)
    moms = {}
    for step in range(3):
        x = O.det_input(2, 3, 24, 25, 2, tag=1200 + step)
        labels = torch.tensor([(5 * i + step) % 60 for i in range(2)])
        logits, cache, newb = O.model_forward(sd, x, True)
        loss, dlog = O.cross_entropy(logits, labels)
        assert abs(loss.item() - float(G["loss%d" % step])) < 1e-7
        grads = O.model_backward(sd, cache, dlog, True)
        prm = {k: sd[k] for k in O.param_names(sd)}
        O.sgd_nesterov_step(prm, grads, moms, 0.1)
        sd.update(prm)
        sd.update(newb)
    for k in ("fcn.bias", "edge_importance.3", "st_gcn_networks.0.gcn.conv.weight"):
        assert rel(sd[k], G["p:" + k]) < 1e-6


def test_dropout_mask_rate():
    m = O.dropout_keep_mask(200000, 1234, 0.5)
    assert abs(m.mean() - 0.5) < 0.01
    m2 = O.dropout_keep_mask(200000, 1235, 0.5)
    assert (m != m2).mean() > 0.4
    assert O.dropout_keep_mask(100, 1, 0.0).min() == 1.0


def test_one_relu_decision_moves_gradients():
    """Conditioning of the whole-model gradient cases: in float64 alone, switching the ONE ReLU decision whose
    pre-activation is closest to zero (|pre| < 1e-4 rms: the forward does not notice) moves some parameter-gradient
    tensor by more than 1e-3 relative L2.  The gradients of these two-clip cases are small remainders of cancelling
    sums, which is why the GPU tests evaluate the oracle with the kernels' own decisions and bound, separately, how
    far from zero a differently decided pre-activation may be."""
    name, lay, ncls, N, T, M, training = C.MODEL_CASES[0]
    g = O.build_graph(lay, "spatial")
    sd = O.det_state(3, ncls, g["A"], True)
    x, labels = C.model_inputs(0)
    lo, cache, _ = O.model_forward(sd, x, True)
    _, dlog = O.cross_entropy(lo, labels)
    g0 = O.model_backward(sd, cache, dlog, True)
    masks = {i: (bc["hmask"].clone(), bc["omask"].clone()) for i, bc in enumerate(cache["blocks"])}
    pre = cache["blocks"][2]["h1"]
    idx = pre.abs().reshape(-1).argmin()
    assert float(pre.reshape(-1)[idx].abs() / pre.pow(2).mean().sqrt()) < 1e-4
    flat = masks[2][0].reshape(-1)
    flat[idx] = ~flat[idx]
    lo1, c1, _ = O.model_forward(sd, x, True, masks=masks)
    g1 = O.model_backward(sd, c1, dlog, True)
    assert float((lo1 - lo).norm() / lo.norm()) < 1e-6
    moved = max(float((g1[k] - g0[k]).norm() / g0[k].norm()) for k in g0
                if not (k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")))
    assert moved > 1e-3, moved


def test_stock_torch_step_matches_golden(golden):
    """oracle/torch_step.py (the CPU baseline bench.py times: the library operators the reference delegates to, as a
    function of a flat state dict) against the reference's golden vectors: logits and gradient samples of the small
    NTU training case, eval logits of the Kinetics-layout case, and the three SGD-nesterov steps."""
    from oracle import torch_step as TS
    G = golden["model"]
    for ci in (0, 2, 8):
        name, lay, ncls, N, T, M, training = C.MODEL_CASES[ci]
        g = O.build_graph(lay, C.strategy_of(name))
        sd = O.det_state(3, ncls, g["A"], True)
        x, labels = C.model_inputs(ci)
        st = TS.Stepper(sd, dropout=0.0)
        logits = TS.forward(st.sd, x, training)
        assert float((logits.detach() - torch.tensor(G[name + "/logits"])).abs().max()) < 1e-9 * max(1.0, float(np.abs(G[name + "/logits"]).max()))
        if training:
            torch.nn.functional.cross_entropy(logits, labels).backward()
            for k in st.names:
                flat = st.sd[k].grad.reshape(-1)
                got = flat[:: max(1, flat.numel() // 257)].numpy()
                ref = G[name + "/gsample:" + k]
                assert np.abs(got - ref).max() <= 1e-9 * max(1e-3, np.abs(ref).max()), k
    S = golden["sgd"]
    g = O.build_graph("ntu-rgb+d", "spatial")
    st = TS.Stepper(O.det_state(3, 60, g["A"], True), lr=0.1)
    for step in range(3):
        x = O.det_input(2, 3, 24, 25, 2, tag=1200 + step)
        labels = torch.tensor([(5 * i + step) % 60 for i in range(2)])
        assert abs(float(st.step(x, labels)) - float(S["loss%d" % step])) < 1e-9
    for k in ("fcn.bias", "edge_importance.3", "st_gcn_networks.0.gcn.conv.weight"):
        assert np.abs(st.sd[k].detach().numpy() - S["p:" + k]).max() < 1e-9
