"""GPU parity tests of the drop-in modules (ConvTemporalGraphical, st_gcn, Model) against the
golden vectors produced by the reference's own modules (tests/golden/*.npz), on the same
closed-form inputs and weights.  float32: 1e-4; bfloat16: 2e-2 (relative to the max-abs of the
reference tensor; for whole-model gradients in bf16, relative L2 error per tensor)."""
import numpy as np
import pytest
import torch

from oracle import stgcn_oracle as O
from tests.golden import cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def net():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on the MI355X box)")
    import net.st_gcn as m
    import net.utils.tgcn as t
    from stgcn_rt import lib
    lib.load()
    return m, t


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def cpu64(t):
    return t.detach().to("cpu", torch.float64)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", range(len(C.GCN_CASES)), ids=[c[0] for c in C.GCN_CASES])
def test_conv_temporal_graphical(net, golden, ci, dtype):
    _, tg = net
    case = C.GCN_CASES[ci]
    name, lay, strat, dense, B, Cin, Cout, T = case
    x, A, W, b, dy = C.gcn_inputs(case, 1000 + 10 * ci)
    K = A.shape[0]
    m = tg.ConvTemporalGraphical(Cin, Cout, K).to(DEV)
    with torch.no_grad():
        m.conv.weight.copy_(W.reshape(m.conv.weight.shape))
        m.conv.bias.copy_(b)
    xr = x.to(DEV).to(dtype).requires_grad_(True)
    Ar = A.to(DEV).float().requires_grad_(True)
    y, A_out = m(xr, Ar)
    assert A_out is Ar and tuple(y.shape) == (B, Cout, T, A.shape[1])
    y.backward(dy.to(DEV).to(dtype))
    G = golden["gcn_op"]
    tol = TOL[dtype]
    got = dict(y=y, dx=xr.grad, dW=m.conv.weight.grad.reshape(W.shape), db=m.conv.bias.grad, dA=Ar.grad)
    for k, v in got.items():
        assert rel(C.compact(cpu64(v)), G[name + "/" + k]) < tol, (k, rel(C.compact(cpu64(v)), G[name + "/" + k]))


def test_conv_temporal_graphical_contract(net):
    _, tg = net
    m = tg.ConvTemporalGraphical(8, 16, 3).to(DEV)
    x = torch.zeros(1, 8, 4, 25, device=DEV)
    with pytest.raises(AssertionError):
        m(x, torch.zeros(2, 25, 25, device=DEV))
    assert list(m.state_dict().keys()) == ["conv.weight", "conv.bias"]
    assert tuple(m.conv.weight.shape) == (48, 8, 1, 1)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", range(len(C.BLOCK_CASES)), ids=[c[0] for c in C.BLOCK_CASES])
def test_st_gcn_block(net, golden, ci, dtype):
    mod, _ = net
    case = C.BLOCK_CASES[ci]
    name, lay, cin, cout, stride, residual, B, T, training = case
    x, A, p, bufs, dout, kind = C.block_inputs(case, 5000 + 100 * ci)
    K = A.shape[0]
    m = mod.st_gcn(cin, cout, (9, K), stride, dropout=0, residual=residual).to(DEV)
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in list(p.items()) + list(bufs.items()):
            sd[k].copy_(v)
    m.train(training)
    xr = x.to(DEV).to(dtype).requires_grad_(True)
    Ar = A.to(DEV).float().requires_grad_(True)
    y, _ = m(xr, Ar)
    y.backward(dout.to(DEV).to(dtype))
    got = {"out": y, "dx": xr.grad, "dA": Ar.grad}
    for k, prm in m.named_parameters():
        got["g:" + k] = prm.grad
    for k in bufs:
        got["buf:" + k] = m.state_dict()[k]
    G = golden["block"]
    tol = TOL[dtype]
    n = 0
    for key in G.files:
        if not key.startswith(name + "/"):
            continue
        k = key[len(name) + 1:]
        ref = G[key]
        val = C.compact(cpu64(got[k]))
        n += 1
        if training and k in ("g:tcn.2.bias", "g:residual.0.bias"):
            # exactly zero in exact arithmetic (bias ahead of a train-mode BN): bound by the noise floor
            wkey = "g:tcn.2.weight" if k == "g:tcn.2.bias" else "g:residual.0.weight"
            assert np.abs(val).max() < tol * np.abs(G[name + "/" + wkey]).max() * 50, k
            continue
        # gradients that sum a relu-masked signal are compared in relative L2 for bf16
        err = rel(val, ref) if dtype == torch.float32 else rel_l2(val, ref)
        assert err < tol, (k, err)
    assert n >= 11
    if training:
        assert int(m.tcn[0].num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", range(len(C.MODEL_CASES)), ids=[c[0] for c in C.MODEL_CASES])
def test_model(net, golden, ci, dtype):
    mod, _ = net
    name, lay, ncls, N, T, M, training = C.MODEL_CASES[ci]
    g = O.build_graph(lay, "spatial")
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    x, labels = C.model_inputs(ci)
    m = mod.Model(3, ncls, dict(layout=lay, strategy="spatial"), True, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train(training)
    logits = m(x.float().to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, labels.to(DEV))
    G = golden["model"]
    tol = TOL[dtype]
    assert tuple(logits.shape) == (N, ncls)
    assert rel(cpu64(logits), G[name + "/logits"]) < tol
    assert abs(loss.item() - float(G[name + "/loss"])) < tol * max(1.0, abs(float(G[name + "/loss"])))
    if training:
        loss.backward()
        worst = ("", 0.0)
        for k, prm in m.named_parameters():
            gr = cpu64(prm.grad)
            if k.endswith("tcn.2.bias") or k.endswith("residual.0.bias"):
                continue                                   # zero up to rounding noise
            ref_norm = float(G[name + "/gnorm:" + k])
            flat = gr.reshape(-1)
            samp = flat[:: max(1, flat.numel() // 257)].numpy()
            ref = G[name + "/gsample:" + k]
            e = rel(samp, ref) if dtype == torch.float32 else rel_l2(samp, ref)
            en = abs(gr.norm().item() - ref_norm) / (ref_norm + 1e-30)
            if max(e, en) > worst[1]:
                worst = (k, max(e, en))
            if name + "/g:" + k in G.files:
                full = G[name + "/g:" + k]
                e2 = rel(gr.numpy(), full) if dtype == torch.float32 else rel_l2(gr.numpy(), full)
                if e2 > worst[1]:
                    worst = (k + " (full)", e2)
        assert worst[1] < tol * (1 if dtype == torch.bfloat16 else 3), worst
        rsd = m.state_dict()
        for key in G.files:
            if key.startswith(name + "/buf:"):
                assert rel(cpu64(rsd[key.split("buf:")[1]]), G[key]) < tol, key
        assert int(rsd["data_bn.num_batches_tracked"]) == 1
    if name.endswith("small_eval"):
        out, feat = m.extract_feature(x.float().to(DEV))
        assert rel(cpu64(out), G[name + "/feat_output"]) < tol
        assert rel(cpu64(feat), G[name + "/feat_feature"]) < tol


def test_model_dropout_and_contract(net):
    mod, _ = net
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5).to(DEV)
    assert m.st_gcn_networks[0].tcn[4].p == 0 and m.st_gcn_networks[1].tcn[4].p == 0.5
    x = torch.randn(2, 3, 16, 25, 2, device=DEV)
    m.train()
    a = m(x)
    b = m(x)
    assert torch.isfinite(a).all() and (a - b).abs().max() > 0     # fresh mask every call
    m.eval()
    c, d = m(x), m(x)
    assert torch.equal(c, d)
    m2 = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="uniform"), False).to(DEV)
    assert m2.edge_importance == [1] * 10 and m2.A.shape[0] == 1
    assert torch.isfinite(m2(x)).all()
    with pytest.raises(RuntimeError):
        m(x.cpu())
