"""GPU parity tests of the drop-in modules (ConvTemporalGraphical, st_gcn, Model) against the
golden vectors produced by the reference's own modules (tests/golden/*.npz), on the same
closed-form inputs and weights.  float32: 1e-4; bfloat16: 2e-2 (relative to the max-abs of the
reference tensor; for whole-model gradients in bf16, relative L2 error per tensor)."""
import os

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


# A reduced-precision evaluation decides a ReLU differently from float64 only where the float64 pre-activation is
# within its own error of zero: bound on |pre-activation| / rms(pre-activation tensor) over such elements.
FLIP_BOUND = {torch.float32: 1e-4, torch.bfloat16: 0.5}


def _relu_flips(plain_cache, masks):
    """(count, largest |float64 pre-activation| / rms) over the elements whose ReLU decision in `masks`
    ({block: (inner, final)}, the kernels') differs from the plain float64 evaluation's."""
    n, worst = 0, 0.0
    for i, bc in enumerate(plain_cache["blocks"]):
        for mine, theirs, pre in ((masks[i][0], bc["hmask"], bc["h1"]), (masks[i][1], bc["omask"], bc["s"])):
            flip = mine.bool() != theirs.bool()
            k = int(flip.sum())
            if k:
                n += k
                worst = max(worst, float(pre[flip].abs().max() / pre.pow(2).mean().sqrt()))
    return n, worst


def _sample(g):
    """The elements of a gradient tensor the model fixtures keep (tests/golden/make_golden.py: every
    max(1, numel // 257)-th)."""
    flat = g.reshape(-1)
    return flat[:: max(1, flat.numel() // 257)]


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


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", range(len(C.GCN_T_CASES)), ids=[c[0] for c in C.GCN_T_CASES])
def test_conv_temporal_graphical_with_temporal_kernel(net, golden, ci, dtype):
    """t_kernel_size > 1 with stride, padding and dilation (net/utils/tgcn.py:44-66) against the reference's own
    outputs and gradients."""
    _, tg = net
    case = C.GCN_T_CASES[ci]
    name, lay, strat, B, Cin, Cout, T, kt, stride, pad, dil = case
    x, A, W, b, dy = C.gcn_t_inputs(case, 7000 + 10 * ci)
    K = A.shape[0]
    m = tg.ConvTemporalGraphical(Cin, Cout, K, t_kernel_size=kt, t_stride=stride, t_padding=pad, t_dilation=dil).to(DEV)
    assert tuple(m.conv.weight.shape) == (K * Cout, Cin, kt, 1)
    with torch.no_grad():
        m.conv.weight.copy_(W.reshape(m.conv.weight.shape))
        m.conv.bias.copy_(b)
    xr = x.to(DEV).to(dtype).requires_grad_(True)
    Ar = A.to(DEV).float().requires_grad_(True)
    y, A_out = m(xr, Ar)
    assert A_out is Ar and tuple(y.shape) == tuple(dy.shape)
    y.backward(dy.to(DEV).to(dtype))
    G = golden["gcn_temporal"]
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


def _capture():
    import stgcn_rt.functional as F
    F.CAPTURE = []
    return F


def _masks_from_capture(F, nblocks):
    """CAPTURE is filled in backward order (last block first)."""
    cap = F.CAPTURE
    F.CAPTURE = None
    assert len(cap) == nblocks
    return {nblocks - 1 - j: (c["h_mask"].cpu(), c["out_mask"].cpu()) for j, c in enumerate(cap)}


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", range(len(C.BLOCK_CASES)), ids=[c[0] for c in C.BLOCK_CASES])
def test_st_gcn_block(net, golden, ci, dtype):
    """Block forward + backward.  float32 is checked against the reference's golden vectors; both dtypes
    are checked against the oracle evaluated on the inputs as the kernels see them (rounded to the storage
    dtype) and with the kernels' own ReLU decisions (elements within rounding error of zero)."""
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
    F = _capture()
    y, _ = m(xr, Ar)
    y.backward(dout.to(DEV).to(dtype))
    masks = tuple(mk[:, :cout] for mk in _masks_from_capture(F, 1)[0])   # odd widths run padded: keep the real channels
    got = {"out": y, "dx": xr.grad, "dA": Ar.grad}
    for k, prm in m.named_parameters():
        got["g:" + k] = prm.grad
    for k in bufs:
        got["buf:" + k] = m.state_dict()[k]
    tol = TOL[dtype]
    zero_bias = ("g:tcn.2.bias", "g:residual.0.bias") if training else ()

    if dtype == torch.float32:
        G = golden["block"]
        n = 0
        for key in G.files:
            if not key.startswith(name + "/"):
                continue
            k = key[len(name) + 1:]
            n += 1
            if k in zero_bias:
                continue
            assert rel(C.compact(cpu64(got[k])), G[key]) < tol, k
        assert n >= 11

    # oracle with the kernels' view of the inputs and their ReLU decisions
    xq = x.to(dtype).double()
    dq = dout.to(dtype).double()
    out_o, cache, nb = O.block_forward(xq, A, p, bufs, stride, kind, training, None, 1.0, masks[0], masks[1])
    dx_o, dA_o, g_o = O.block_backward(dq, p, cache, stride, kind, training)
    ref = {"out": out_o, "dx": dx_o, "dA": dA_o}
    ref.update({"g:" + k: v for k, v in g_o.items()})
    ref.update({"buf:" + k: v for k, v in nb.items()})
    for k, r in ref.items():
        if k in zero_bias:
            # exactly zero in exact arithmetic (a bias ahead of a train-mode BatchNorm): bound by the noise floor
            wk = "g:tcn.2.weight" if k == "g:tcn.2.bias" else "g:residual.0.weight"
            assert cpu64(got[k]).abs().max() < 50 * tol * ref[wk].abs().max(), k
            continue
        err = rel(cpu64(got[k]), r) if dtype == torch.float32 else rel_l2(cpu64(got[k]), r)
        assert err < tol, (k, err)
    if training:
        assert int(m.tcn[0].num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", range(len(C.MODEL_CASES)), ids=[c[0] for c in C.MODEL_CASES])
def test_model(net, golden, ci, dtype):
    """Whole model.  Logits / loss / running statistics against the reference's golden vectors; gradients
    against the oracle evaluated with the kernels' own ReLU decisions (a float64 and a float32 evaluation
    of a ReLU network disagree on elements whose pre-activation is within float32 rounding of zero, and with
    only a few thousand rows per channel one such element moves a gradient by about 1e-3)."""
    mod, _ = net
    name, lay, ncls, N, T, M, training = C.MODEL_CASES[ci]
    strat = C.strategy_of(name)
    g = O.build_graph(lay, strat)
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    x, labels = C.model_inputs(ci)
    m = mod.Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train(training)
    F = _capture() if training else None
    logits = m(x.float().to(DEV))
    loss = torch.nn.functional.cross_entropy(logits, labels.to(DEV))
    G = golden["model"]
    tol = TOL[dtype]
    assert tuple(logits.shape) == (N, ncls)
    assert rel(cpu64(logits), G[name + "/logits"]) < tol
    assert abs(loss.item() - float(G[name + "/loss"])) < tol * max(1.0, abs(float(G[name + "/loss"])))
    if training:
        loss.backward()
        masks = _masks_from_capture(F, 10)
        rsd = m.state_dict()
        for key in G.files:
            if key.startswith(name + "/buf:"):
                assert rel(cpu64(rsd[key.split("buf:")[1]]), G[key]) < tol, key
        assert int(rsd["data_bn.num_batches_tracked"]) == 1
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        skip = lambda k: k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")      # zero up to rounding noise
        got = {k: cpu64(prm.grad) for k, prm in m.named_parameters() if not skip(k)}

        # 1. The kernels' ReLU decisions against the plain float64 evaluation: they may differ only where the float64
        #    pre-activation is within the storage format's rounding error of zero.
        nflip, ratio = _relu_flips(O.model_forward(sd64, x, True)[1], masks)
        print("ReLU decisions differing from the float64 evaluation: %d, largest |pre-activation| / rms among them: %.2e" % (nflip, ratio))
        assert ratio < FLIP_BOUND[dtype], (nflip, ratio)

        # 2. Directly against the reference's own gradients (its ReLU decisions, its logits): every 257th element of
        #    every tensor is in the fixtures.  One differing decision switches one element's gradient path on or off,
        #    which moves the small tensors of these two-clip cases (edge_importance: 73 sums) by up to 1e-2; with no
        #    differing decision (it happens: the count varies with the last bits of the BatchNorm coefficients) what is left
        #    is the float32 forward: logits off by 1e-6 put a relative error of about 1e-5 into d loss / d logits, and the
        #    backward of these two-clip cases cancels 13-68x, so single tensors sit up to 1e-3 from the reference's
        #    (kin_small_train_oddT: 1.0e-3 on BN1's beta of block 0 with no decision differing, while check 3 below,
        #    which hands the GPU's own d loss / d logits to the oracle, agrees to 4e-6).  The case with a margin on every
        #    ReLU input is the one held to 1e-4 directly.
        dd = sorted(((rel_l2(_sample(got[k]), G[name + "/gsample:" + k]), k) for k in got), reverse=True)
        direct = dd[0][0]
        print("worst gradient rel-L2 deviations from the reference's golden samples:", dd[:4])
        #    A bf16 evaluation decides about 0.5 % of the ReLUs differently (pre-activations within its error of zero, bound
        #    asserted above), so its gradients are those of a slightly different piecewise-linear function: on these
        #    ill-conditioned cases (each gradient is what is left of sums that cancel 10-70x) they sit 30-60 % from the
        #    float64 ones.  tests/test_oracle_golden.py::test_one_relu_decision_moves_gradients shows the same
        #    sensitivity inside float64 alone.  bf16 is therefore bounded against the decision-consistent oracle below.
        assert direct < (2e-3 + 1.5e-2 * nflip if dtype == torch.float32 else 1.0), (direct, nflip)
        if "margin" in name and dtype == torch.float32:
            # every ReLU input of this case is at least 1.8e-5 rms away from zero in float64 (the fixture holds the
            # margin), so float32 takes the reference's decisions and its gradients match the reference's own, with no
            # decision handed over
            assert float(G[name + "/relu_margin"]) > 1.5e-5
            assert nflip == 0 and direct < tol, (nflip, direct)

        # 3. Against the oracle evaluated with the kernels' decisions and the GPU's d loss / d logits (the softmax turns
        #    an absolute logit error e into a relative probability error e, which says nothing about the backward kernels).
        lo, cache, _ = O.model_forward(sd64, x, True, masks=masks)
        _, dlog = O.cross_entropy(cpu64(logits), labels)
        ref_g = O.model_backward(sd64, cache, dlog, True)
        errs = {k: rel_l2(got[k], ref_g[k]) for k in got}
        worst = max(errs.items(), key=lambda kv: kv[1])
        print("worst gradient rel-L2 error vs mask-consistent oracle:", worst)
        if dtype == torch.float32:
            assert worst[1] < tol, worst
        else:
            # bfloat16.  The yardstick is the REFERENCE's own bfloat16 run: its Model under
            # torch.autocast("cpu", dtype=torch.bfloat16) on this case, its gradients compared with the float64
            # evaluation that takes ITS ReLU decisions and ITS d loss / d logits -- exactly the comparison made here for
            # the kernels (tests/golden/make_golden.py::gen_bf16_reference -> bf16_ref.npz).  The reference's worst
            # tensor deviates by 3.5-5.0e-2 and its median tensor by 1.8-2.1e-2 over the six training cases: activations,
            # weights and the gradients between operators are rounded to 8 significant bits ~70 times by the last block.
            # Bound: every tensor within 2e-2 or PER_TENSOR x the reference's deviation of that tensor (the two runs are
            # different realisations of the same rounding noise), the worst and the median tensor within 1.2 x the
            # reference's worst and median.
            R = golden["bf16_ref"]
            refdev = {k: float(R[name + "/dev:" + k]) for k in got}
            med = lambda d: sorted(d.values())[len(d) // 2]
            ratio = max(((errs[k] / refdev[k], k) for k in got if errs[k] > tol), default=(0.0, ""))
            print("reference under bf16 autocast vs float64: worst %.2e, median %.2e; kernels: worst %.2e, median %.2e; "
                  "largest kernel/reference ratio among tensors above 2e-2: %.2f (%s)"
                  % (max(refdev.values()), med(refdev), worst[1], med(errs), ratio[0], ratio[1]))
            PER_TENSOR = 2.0           # a 73-term tensor (edge_importance, K = 1) of ours measured 1.67 x the reference's
            for k in got:
                assert errs[k] < max(tol, PER_TENSOR * refdev[k]), (k, errs[k], refdev[k])
            assert worst[1] < 1.2 * max(refdev.values()), (worst, max(refdev.values()))
            assert med(errs) < max(tol, 1.2 * med(refdev)), (med(errs), med(refdev))
            if name in ("ntu_small_train", "kin_small_train_oddT"):
                # diagnostic only: an exact evaluation of the reference's formulas that rounds to bfloat16 where the
                # kernels store or feed an MFMA (oracle.storage_rounding) -- this build's own model of the format's cost
                with O.storage_rounding(torch.bfloat16):
                    _, cache_q, _ = O.model_forward(sd64, x, True, masks=masks)
                    fmt_g = O.model_backward(sd64, cache_q, dlog, True)
                fmt = {k: rel_l2(fmt_g[k], ref_g[k]) for k in got}
                print("bf16-storage oracle vs float64 oracle (diagnostic): worst %.2e, median %.2e" % (max(fmt.values()), med(fmt)))
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


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_trainer_steps_match_oracle(net, dtype):
    """REC_Processor.train's iteration (processor/recognition.py:84-97) through ddp.Trainer: three SGD-nesterov steps
    (lr 0.1, momentum 0.9, weight decay 1e-4, dropout 0) on the small NTU model, in lockstep with the oracle's
    training loop evaluated with the kernels' ReLU decisions of the same step.  Every step re-packs the weights the
    optimizer just changed: a stale pack or a lost gradient shows up in the second loss."""
    mod, _ = net
    from stgcn_rt.ddp import Trainer
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    x = O.det_input(2, 3, 24, 25, 2, tag=777)
    labels = torch.tensor([5, 17])
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    names = O.param_names(sd64)
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.0, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train()
    tr = Trainer(m, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    xd, ld = x.float().to(DEV), labels.to(DEV)
    moms, losses, ref_losses = {}, [], []
    for _ in range(3):
        F = _capture()
        losses.append(float(tr.step(xd, ld).detach()))
        masks = _masks_from_capture(F, 10)
        logits, cache, newb = O.model_forward(sd64, x, True, masks=masks)
        loss, dlog = O.cross_entropy(logits, labels)
        grads = O.model_backward(sd64, cache, dlog, True)
        prm = {k: sd64[k] for k in names}
        O.sgd_nesterov_step(prm, grads, moms, 0.1)
        sd64.update(prm)
        sd64.update(newb)
        ref_losses.append(float(loss))
    ltol = 1e-4 if dtype == torch.float32 else 5e-2
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < ltol * max(1.0, abs(b)), (losses, ref_losses)
    assert ref_losses[2] < ref_losses[0]                     # the loop does train
    rsd = m.state_dict()
    worst = ("", 0.0)
    for k in names:
        e = rel_l2(cpu64(rsd[k]), sd64[k])
        if e > worst[1]:
            worst = (k, e)
    print("worst parameter rel-L2 deviation after 3 steps:", worst, "losses", losses, ref_losses)
    assert worst[1] < (1e-4 if dtype == torch.float32 else 5e-2), worst
    assert int(rsd["data_bn.num_batches_tracked"]) == 3 and int(rsd["st_gcn_networks.4.residual.1.num_batches_tracked"]) == 3


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_training_learns_a_separable_task(net, dtype):
    """End to end: feeder-style augmentation on the GPU, Trainer steps (forward, cross-entropy, backward, SGD-nesterov)
    -- on clips whose class is written into the motion of one joint, the loss has to fall and the fit to become exact.
    Not a parity check; it guards against an update path that matches nothing in particular but still 'runs'."""
    m, _ = net
    from feeder import tools
    from stgcn_rt import ops as P
    from stgcn_rt.ddp import Trainer
    from stgcn_rt.evaluate import evaluate
    import random
    torch.manual_seed(0)
    N, T, W, V, M, ncls = 16, 40, 32, 25, 2, 4
    rs = np.random.RandomState(0)
    labels = np.arange(N) % ncls
    x = 0.05 * rs.randn(N, 3, T, V, M).astype(np.float32)
    tt = np.arange(T, dtype=np.float32)
    for n in range(N):                                   # class c: joint 3 oscillates with frequency (c+1)
        x[n, 0, :, 3, :] += np.sin(2 * np.pi * (labels[n] + 1) * tt / T)[:, None]
        x[n, 1, :, 7, :] += np.cos(2 * np.pi * (labels[n] + 1) * tt / T)[:, None]
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.from_numpy(labels).long().to(DEV)
    model = m.Model(3, ncls, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.0, compute_dtype=dtype).to(DEV)

    def weights_init(mod):                               # processor/recognition.py:21-33
        cn = mod.__class__.__name__
        if cn.find("Conv1d") != -1 or cn.find("Conv2d") != -1:
            mod.weight.data.normal_(0.0, 0.02)
            if mod.bias is not None:
                mod.bias.data.fill_(0)
        elif cn.find("BatchNorm") != -1:
            mod.weight.data.normal_(1.0, 0.02)
            mod.bias.data.fill_(0)
    model.apply(weights_init)
    model.train()
    tr = Trainer(model, lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    losses = []
    for it in range(60):
        shifts = [tools.choose_shift(T, W, rng=random.Random(100 * it + n)) for n in range(N)]
        params = [tools.move_params(W, angle_candidate=(-5., 0., 5.), scale_candidate=(0.95, 1.0, 1.05),
                                    transform_candidate=(-0.05, 0.0, 0.05), rng=random.Random(it * 7 + n),
                                    nprng=np.random.RandomState(it * 13 + n)) for n in range(N)]
        move = tuple(np.stack([p[j] for p in params]) for j in range(4))
        batch = P.augment_batch(xd, window=W, shift=shifts, move=move)
        losses.append(float(tr.step(batch, yd).item()))
    assert np.isfinite(losses).all()
    print("losses", [round(v, 3) for v in losses[::6]], [round(v, 3) for v in losses[-5:]])
    assert np.mean(losses[-5:]) < 0.35 * np.mean(losses[:3]), (losses[:3], losses[-5:])
    # The weights moved fast during the last steps and the running statistics (momentum 0.1) lag behind them: let
    # them settle on the final weights with forward passes only, so that eval mode sees what training saw.
    with torch.no_grad():
        for it in range(40):
            model(P.augment_batch(xd, window=W, shift=[it % (T - W + 1)] * N))
    res = evaluate(model, [(P.augment_batch(xd, window=W, shift=[4] * N), yd)], DEV, show_topk=(1,))
    print("top1", res["topk"][1], "eval loss", res["mean_loss"])
    assert res["topk"][1] >= 0.9, res["topk"]


def _fresh_trainer(mod, dtype, dropout, graph, lr=0.01, wd=1e-4, split=None):
    from stgcn_rt.ddp import Trainer
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=dropout, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train()
    return m, Trainer(m, lr=lr, momentum=0.9, weight_decay=wd, nesterov=True, graph=graph, split=split)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_trainer_graph_replay_matches_launch_by_launch(net, dtype):
    """The step captured into a HIP graph (forward, loss, backward, gradient gather; optimizer launch outside) against
    the same trainer stepping launch by launch: same losses and parameters over six steps on changing batches (the
    replayed steps read their batch from the graph's static buffers), running statistics and counters included."""
    mod, _ = net
    xs = [O.det_input(2, 3, 24, 25, 2, tag=3100 + i).float().to(DEV) for i in range(6)]
    ls = [torch.tensor([(3 * i + 1) % 60, (7 * i + 2) % 60], device=DEV) for i in range(6)]
    res = {}
    for graph in (False, True):
        # a small learning rate: at 0.01 and above, two launch-by-launch runs of this two-clip problem already drift
        # apart by 5e-4 in three steps from the last-bit differences of the BatchNorm atomics
        m, tr = _fresh_trainer(mod, dtype, 0.0, graph, lr=1e-3)
        init = {k: cpu64(v).clone() for k, v in m.state_dict().items()}
        losses = [float(tr.step(x, l).detach()) for x, l in zip(xs, ls)]
        assert tr.graph_ready() == graph
        res[graph] = (losses, {k: cpu64(v) for k, v in m.state_dict().items()}, init)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) < tol * max(1.0, abs(b)), (res[True][0], res[False][0])
    # What six steps changed, replayed against launch by launch.  Single small tensors may differ by a few per cent
    # between ANY two runs (one ReLU decision at a pre-activation within rounding of zero falls the other way: see
    # tests/test_oracle_golden.py::test_one_relu_decision_moves_gradients), so the bound is on the whole update.
    worst, ua, ub = ("", 0.0), [], []
    for k, v in res[False][1].items():
        if not v.is_floating_point():
            assert torch.equal(res[True][1][k], v), k
        elif k != "A":
            da, db = res[True][1][k] - res[True][2][k], v - res[False][2][k]
            worst = max(worst, (k, rel_l2(da, db)), key=lambda t: t[1])
            if "running" in k:
                assert rel_l2(da, db) < tol * 10, k
            else:
                ua.append(da.reshape(-1))
                ub.append(db.reshape(-1))
    whole = rel_l2(torch.cat(ua), torch.cat(ub))
    print("six-step parameter update, graph replay vs launch by launch: whole vector %.2e, worst tensor %s %.2e" % (whole, worst[0], worst[1]))
    assert whole < (5e-3 if dtype == torch.float32 else 5e-2), (whole, worst)
    assert int(res[True][1]["data_bn.num_batches_tracked"]) == 6


def test_trainer_graph_replay_draws_fresh_dropout_masks(net):
    """Every launch argument of a replayed step is frozen in the graph; the dropout masks still change from step to
    step because the kernel mixes a device-side step count into its seed.  With lr = 0 and no weight decay the
    parameters never move, so on one fixed batch the loss varies only through the masks."""
    mod, _ = net
    m, tr = _fresh_trainer(mod, torch.float32, 0.5, True, lr=0.0, wd=0.0)
    x = O.det_input(2, 3, 24, 25, 2, tag=3200).float().to(DEV)
    l = torch.tensor([4, 9], device=DEV)
    losses = [float(tr.step(x, l).detach()) for _ in range(7)]
    assert tr.graph_ready() and int(m._drop_step) == 7
    assert len({round(v, 6) for v in losses[2:]}) == 5, losses          # five replays, five different masks
    m2, tr2 = _fresh_trainer(mod, torch.float32, 0.5, False, lr=0.0, wd=0.0)
    losses2 = [float(tr2.step(x, l).detach()) for _ in range(7)]
    # the same seeds and step counts launch by launch: the same masks, hence the same losses
    assert max(abs(a - b) for a, b in zip(losses, losses2)) < 1e-4, (losses, losses2)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_training_repeats_bit_for_bit_in_every_execution_mode(net, dtype):
    """Six SGD steps with dropout from one state: launch by launch or replayed from a HIP graph, with the
    parameter-gradient kernels on the main stream or forked to their second stream (fork and join become graph edges),
    and run twice -- every parameter ends bit-identical.  Every reduction of a step is fixed-order or an exact
    double-precision atomic sum, and the dropout masks depend on seed, block, rank and step count only."""
    mod, _ = net
    import stgcn_rt.functional as F
    x = O.det_input(4, 3, 48, 25, 2, tag=3250).float().to(DEV)
    l = torch.tensor([4, 9, 33, 58], device=DEV)
    was = F.OVERLAP_WGRAD
    runs = []
    try:
        for graph, overlap in ((False, False), (False, False), (True, False), (False, True), (True, True)):
            F.OVERLAP_WGRAD = overlap
            torch.manual_seed(99)
            m, tr = _fresh_trainer(mod, dtype, 0.5, graph, lr=0.01)
            losses = [float(tr.step(x, l).detach()) for _ in range(6)]
            torch.cuda.synchronize()
            assert tr.graph_ready() == graph
            names = {id(p): k for k, p in m.named_parameters()}
            runs.append(((graph, overlap), losses, {names[id(p)]: tr.flat.flat[o:o + n].detach().clone()
                                                     for p, o, n in zip(tr.flat.params, tr.flat.offsets, tr.flat.sizes)}))
            del m, tr
    finally:
        F.OVERLAP_WGRAD = was
    _, losses0, ref = runs[0]
    assert losses0[-1] < losses0[0] and all(bool(torch.isfinite(v).all()) for v in ref.values())
    for mode, losses, got in runs[1:]:
        assert losses == losses0, (mode, losses, losses0)
        differ = [k for k in ref if not torch.equal(got[k], ref[k])]
        assert not differ, (mode, differ[:5])


def test_model_dropout_matches_oracle(net):
    """Whole model with dropout 0.5 in train mode, forward and backward, against the oracle drawing the same masks:
    block i's seed is st_gcn._seed(), advanced by the device-side step count times 0x85EBCA6B (second call here)."""
    mod, _ = net
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    x = O.det_input(2, 3, 24, 25, 2, tag=3300)
    labels = torch.tensor([11, 50])
    torch.manual_seed(4321)
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train()
    m(x.float().to(DEV))                                 # step count 0 -> 1; running statistics move: reload them
    m.load_state_dict(sd, strict=True)
    assert int(m._drop_step) == 1
    seeds = [(blk._seed() + 1 * 0x85EBCA6B) & 0xFFFFFFFF for blk in m.st_gcn_networks]
    F = _capture()
    logits = m(x.float().to(DEV))
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    masks = _masks_from_capture(F, 10)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    lo, cache, _ = O.model_forward(sd64, x, True, dropout=0.5, seeds=seeds, masks=masks)
    assert rel(cpu64(logits), lo) < 1e-4
    _, dlog = O.cross_entropy(cpu64(logits), labels)
    ref_g = O.model_backward(sd64, cache, dlog, True)
    for k, prm in m.named_parameters():
        if not (k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")):
            assert rel_l2(cpu64(prm.grad), ref_g[k]) < 1e-4, k


def test_checkpoint_file_to_golden_logits(net, golden, tmp_path):
    """The checkpoint wire format end to end on the GPU (torchlight/torchlight/io.py:57-107 of the reference): a file
    written by save_model from a DataParallel-style wrapper (every key carries `module.`), loaded with load_weights
    into a fresh model on the GPU, must reproduce the reference's golden eval logits; with ignore_weights='fcn' the
    classifier keeps the fresh model's values and the rest still arrives."""
    mod, _ = net
    from stgcn_rt import checkpoint as CK
    ci = [c[0] for c in C.MODEL_CASES].index("ntu_small_eval")
    name, lay, ncls, N, T, M, _ = C.MODEL_CASES[ci]
    g = O.build_graph(lay, "spatial")
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    src = mod.Model(3, ncls, dict(layout=lay, strategy="spatial"), True)
    src.load_state_dict(sd, strict=True)
    wrapper = torch.nn.Module()
    wrapper.module = src                                   # state_dict keys: module.<name>, as under nn.DataParallel
    path = str(tmp_path / "epoch1_model.pt")
    saved = CK.save_model(wrapper, path)
    assert all(not k.startswith("module.") for k in saved) and "A" in saved
    raw = torch.load(path, weights_only=True)
    torch.save({"module." + k: v for k, v in raw.items()}, path)       # an older file that kept the prefix
    x, _ = C.model_inputs(ci)
    for dtype in DTYPES:
        m = mod.Model(3, ncls, dict(layout=lay, strategy="spatial"), True, compute_dtype=dtype).to(DEV)
        CK.load_weights(m, path)
        m.eval()
        with torch.no_grad():
            logits = m(x.float().to(DEV))
        assert rel(cpu64(logits), golden["model"][name + "/logits"]) < TOL[dtype]
    m = mod.Model(3, ncls, dict(layout=lay, strategy="spatial"), True).to(DEV)
    fresh_fc = m.fcn.weight.detach().clone()
    logs = []
    CK.load_weights(m, path, ignore_weights="fcn", log=logs.append)
    assert torch.equal(m.fcn.weight, fresh_fc) and any("Filter [fcn]" in l for l in logs)
    assert torch.equal(m.st_gcn_networks[3].tcn[2].weight.cpu(), sd["st_gcn_networks.3.tcn.2.weight"])


@pytest.mark.parametrize("lay,strat,ncls", [("ntu-rgb+d", "uniform", 60), ("ntu-rgb+d", "distance", 60), ("ntu-rgb+d", "spatial", 60),
                                            ("openpose", "uniform", 400), ("openpose", "distance", 400), ("openpose", "spatial", 400)])
def test_large_batch_bf16_inference_sweep(net, lay, strat, ncls):
    """Large-batch bf16 inference across the partition strategies (K = 1 / 2 / 3) and both layouts: in eval mode a
    clip's logits do not depend on what else is in the batch, so the clips of a 1024-clip (once per layout; 128 otherwise) batch at T = 300 must come out
    as they do in two-clip batches -- whose logits the golden eval cases pin for every strategy -- and the first two
    clips are those golden inputs themselves."""
    mod, _ = net
    V = 25 if lay == "ntu-rgb+d" else 18
    g = O.build_graph(lay, strat)
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    m = mod.Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=torch.bfloat16).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.eval()
    N = 1024 if (lay, strat) in (("ntu-rgb+d", "distance"), ("openpose", "uniform")) else 128    # the sweep's own N once per layout
    x = torch.tensor(O.det_uniform((N, 3, 300, V, 2), 8800 + V)).float().to(DEV)
    probes = (0, N // 2 - 2, N - 2)
    with torch.no_grad():
        big = m(x)
        pairs_default = torch.cat([m(x[i:i + 2]) for i in probes])
        # a two-clip batch does not fill the chip, so by default its 128 / 256-channel convolutions take another kernel
        # than the large batch's (smaller tiles, another order of additions); with the grid condition lifted both run
        # the same kernels
        os.environ["STGCN_CONV_WIDE_ANY_GRID"] = "1"
        try:
            pairs = torch.cat([m(x[i:i + 2]) for i in probes])
        finally:
            del os.environ["STGCN_CONV_WIDE_ANY_GRID"]
    assert torch.isfinite(big).all()
    ref = torch.cat([big[i:i + 2] for i in probes])
    assert float((ref - pairs).abs().max() / pairs.abs().max()) < 1e-6      # same kernels, same rows: same bits up to tile order
    assert float((ref - pairs_default).abs().max() / pairs.abs().max()) < 1e-3      # other kernels: bf16 roundings fall differently
    # and against the float64 oracle for one clip pair (bf16 tolerance)
    lo, _, _ = O.model_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x[:2].cpu().double(), False)
    assert rel(cpu64(big[:2]), lo) < TOL[torch.bfloat16]


@pytest.mark.parametrize("ci", [i for i, c in enumerate(C.MODEL_CASES) if ("uniform" in c[0] or "distance" in c[0]) and not c[6]],
                         ids=[c[0] for c in C.MODEL_CASES if ("uniform" in c[0] or "distance" in c[0]) and not c[6]])
def test_uniform_distance_eval_logits_bf16_and_f32(net, golden, ci):
    """K = 1 / 2 eval logits against the reference's golden vectors in both precisions (graph.py:85-94)."""
    mod, _ = net
    name, lay, ncls, N, T, M, _ = C.MODEL_CASES[ci]
    strat = C.strategy_of(name)
    g = O.build_graph(lay, strat)
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    x, _ = C.model_inputs(ci)
    for dtype in DTYPES:
        m = mod.Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=dtype).to(DEV)
        m.load_state_dict(sd, strict=True)
        m.eval()
        with torch.no_grad():
            assert rel(cpu64(m(x.float().to(DEV))), golden["model"][name + "/logits"]) < TOL[dtype]


@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_gradient_wrt_input_clip(net, training):
    """The reference Model is differentiable w.r.t. its input (saliency, adversarial inputs); so is this one:
    d loss / d x through data_bn against the oracle, and tensors of another device than the current one are refused."""
    mod, _ = net
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    x = O.det_input(2, 3, 24, 25, 2, tag=3400)
    labels = torch.tensor([2, 40])
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train(training)
    xd = x.float().to(DEV).requires_grad_(True)
    F = _capture()
    logits = m(xd)
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    masks = _masks_from_capture(F, 10)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    lo, cache, _ = O.model_forward(sd64, x, training, masks=masks)
    _, dlog = O.cross_entropy(cpu64(logits), labels)
    _, dx = O.model_backward(sd64, cache, dlog, training, input_grad=True)
    assert xd.grad is not None and tuple(xd.grad.shape) == tuple(x.shape)
    assert rel_l2(cpu64(xd.grad), dx) < 1e-4


def test_bias_gradients_survive_the_next_forward(net):
    """Gradient accumulation over two micro-batches with torch's own accumulation (the drop-in Model under
    torch.optim): the conv-bias gradients of the first backward must not alias memory that the second forward
    clears.  Eval-mode BatchNorm makes those gradients non-zero."""
    mod, _ = net
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True).to(DEV)
    m.load_state_dict(sd, strict=True)
    xs = [O.det_input(2, 3, 24, 25, 2, tag=3500 + i).float().to(DEV) for i in range(2)]
    labels = torch.tensor([1, 2], device=DEV)
    m.train()
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.modules.batchnorm._BatchNorm)]:
        bn.eval()                                           # fine-tuning with frozen statistics
    singles = []
    for x in xs:
        m.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x), labels).backward()
        singles.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
    m.zero_grad(set_to_none=True)
    for x in xs:
        torch.nn.functional.cross_entropy(m(x), labels).backward()
    for k, p in m.named_parameters():
        if k.endswith("tcn.2.bias") or k.endswith("residual.0.bias"):
            want = singles[0][k] + singles[1][k]
            assert want.abs().max() > 0
            assert float((p.grad - want).abs().max() / want.abs().max()) < 1e-5, k


@pytest.mark.parametrize("N,Cc,T,V,M", [(3, 3, 20, 25, 2), (2, 3, 33, 18, 1), (2, 8, 9, 25, 2), (1, 3, 7, 25, 12), (2, 20, 5, 18, 2)],
                         ids=["ntu", "kinetics_m1", "c8", "vm300", "vc360"])
def test_data_bn_any_person_and_channel_count(net, N, Cc, T, V, M):
    """Model.forward's input normalisation (net/st_gcn.py:72-78) as its own unit: BatchNorm1d over v*C + c with
    statistics over (n, m, t), output in the blocks' channels-last layout, parameter gradients; person and channel
    counts on both sides of the fixed-order kernels' limit (V*M, V*C <= 256).  The statistics repeat bit for bit."""
    import stgcn_rt.functional as F
    x = torch.tensor(O.det_uniform((N, Cc, T, V, M), 3801, -2.0, 3.0), dtype=torch.float64)
    gamma = torch.tensor(O.det_uniform((V * Cc,), 3802, 0.5, 1.5), dtype=torch.float64)
    beta = torch.tensor(O.det_uniform((V * Cc,), 3803, -0.3, 0.3), dtype=torch.float64)
    dy = torch.tensor(O.det_uniform((N * M, Cc, T, V), 3804), dtype=torch.float64)
    xr = x.permute(0, 4, 3, 1, 2).reshape(N * M, V * Cc, T)                     # the reference's view for BatchNorm1d
    zeros, ones = torch.zeros(V * Cc, dtype=torch.float64), torch.ones(V * Cc, dtype=torch.float64)
    yr, cache, (rm_ref, rv_ref) = O.bn_forward(xr, gamma, beta, zeros, ones, True)
    want = yr.reshape(N, M, V, Cc, T).permute(0, 1, 3, 4, 2).reshape(N * M, Cc, T, V)
    dyr = dy.reshape(N, M, Cc, T, V).permute(0, 1, 4, 2, 3).reshape(N * M, V * Cc, T)
    _, dg_ref, db_ref = O.bn_backward(xr, gamma, cache, dyr, True)
    outs = []
    for rep in range(2):
        g = gamma.float().to(DEV).requires_grad_(True)
        b = beta.float().to(DEV).requires_grad_(True)
        rm, rv = zeros.float().to(DEV), ones.float().to(DEV)
        y = F.DataBnFn.apply(x.float().to(DEV), g, b, rm, rv, True, torch.float32, 8 * ((Cc + 7) // 8))
        assert tuple(y.shape) == (N * M, 8 * ((Cc + 7) // 8), T, V)
        assert rel(cpu64(y[:, :Cc]), want) < 1e-5 and float(y.detach()[:, Cc:].abs().max() if y.shape[1] > Cc else 0) == 0
        assert rel(cpu64(rm), rm_ref) < 1e-5 and rel(cpu64(rv), rv_ref) < 1e-5
        dyp = torch.zeros_like(y)
        dyp[:, :Cc] = dy.float().to(DEV)
        y.backward(dyp)
        assert rel(cpu64(g.grad), dg_ref) < 1e-4 and rel(cpu64(b.grad), db_ref) < 1e-4
        outs.append((y.detach().clone(), rm.clone(), g.grad.clone()))
    if V * M <= 256:
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    if V * M <= 256 and V * Cc <= 256:               # the gradient sums use the forward's mean and rstd
        assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_weight_gradients_on_the_side_stream(net, dtype):
    """With functional.OVERLAP_WGRAD a block's backward launches its parameter-gradient kernels on a second stream, beside
    the data-gradient chain, and joins before it returns: every gradient equals the single-stream run's bit for bit, at a batch whose kernels are long enough to overlap and
    over several backward passes in a row (a stale or early read would show as a large difference)."""
    mod, _ = net
    import stgcn_rt.functional as F
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    x = O.det_input(8, 3, 64, 25, 2, tag=3700).float().to(DEV)
    labels = torch.arange(8, device=DEV) * 7 % 60
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.0, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.train()
    was = F.OVERLAP_WGRAD
    runs = {True: [], False: []}
    try:
        for overlap in (True, False, True):
            F.OVERLAP_WGRAD = overlap
            for rep in range(3):
                m.zero_grad(set_to_none=True)
                torch.nn.functional.cross_entropy(m(x), labels).backward()
            torch.cuda.synchronize()
            runs[overlap].append({k: p.grad.detach().double().cpu() for k, p in m.named_parameters()})
    finally:
        F.OVERLAP_WGRAD = was
    # runs of one batch repeat bit for bit (fixed-order reductions; double atomics of float32 partials are exact), so
    # any difference is the side stream's
    ref = runs[False][0]
    keys = list(ref)
    for got in runs[True]:
        differ = [k for k in keys if not torch.equal(got[k], ref[k])]
        assert not differ, differ[:5]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("ci", [i for i, c in enumerate(C.MODEL_CASES) if not c[6]], ids=[c[0] for c in C.MODEL_CASES if not c[6]])
def test_folded_inference_path(net, golden, ci, dtype):
    """Inference under torch.no_grad() takes the path with BatchNorm folded into the convolution weights and the
    add + ReLU fused into the temporal convolution (2-3 launches per block): logits against the reference's golden
    eval vectors for every layout / partition strategy, extract_feature against its fixtures, and the folded
    parameters must follow an in-place change of a running statistic."""
    mod, _ = net
    from stgcn_rt import ops as P
    name, lay, ncls, N, T, M, _ = C.MODEL_CASES[ci]
    strat = C.strategy_of(name)
    g = O.build_graph(lay, strat)
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    x, _ = C.model_inputs(ci)
    m = mod.Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True)
    m.eval()
    xd = x.float().to(DEV)
    tol = TOL[dtype]
    P.TIMING, P.TIMING_NAMES = {}, None
    with torch.no_grad():
        assert m._foldable()
        logits = m(xd)
    launches = sum(v[0] for v in P.timing_summary().values())
    P.TIMING = None
    assert launches <= 10 * 2 + 2, launches                  # gcn + conv per block, + the two 1x1 residual convolutions
    assert rel(cpu64(logits), golden["model"][name + "/logits"]) < tol
    general = m(xd)                                           # grad enabled: the general path (separate BatchNorm stages)
    assert not m._foldable() and rel(cpu64(logits), cpu64(general)) < tol
    if name.endswith("small_eval"):
        with torch.no_grad():
            out, feat = m.extract_feature(xd)
        assert rel(cpu64(out), golden["model"][name + "/feat_output"]) < tol
        assert rel(cpu64(feat), golden["model"][name + "/feat_feature"]) < tol
    # a running statistic changes in place: the next forward must re-fold
    with torch.no_grad():
        m.st_gcn_networks[2].tcn[3].running_var.mul_(1.7)
        m.st_gcn_networks[4].residual[1].running_mean.add_(0.05)
        after = m(xd)
    sd2 = {k: (v.double().clone() if v.is_floating_point() else v) for k, v in sd.items()}
    sd2["st_gcn_networks.2.tcn.3.running_var"] *= 1.7
    sd2["st_gcn_networks.4.residual.1.running_mean"] += 0.05
    lo, _, _ = O.model_forward(sd2, x, False)
    assert rel(cpu64(after), lo) < tol and rel(cpu64(after), cpu64(logits)) > 1e-4


def test_folded_plan_follows_graph_replayed_training(net):
    """Train (steps replayed from a HIP graph: no Python op touches a parameter or a running statistic, so no tensor
    version advances), evaluate under no_grad, train on, evaluate again: the second evaluation must see the current
    parameters and running statistics, i.e. agree with the general (unfolded) path on them and with the oracle, and
    differ from the first.  Also: after a replay every p.grad is the parameter's slice of the flat gradient."""
    mod, _ = net
    m, tr = _fresh_trainer(mod, torch.float32, 0.0, True, lr=0.05)
    xs = [O.det_input(2, 3, 24, 25, 2, tag=3400 + i).float().to(DEV) for i in range(4)]
    l = torch.tensor([4, 9], device=DEV)
    xe = O.det_input(2, 3, 24, 25, 2, tag=3450)
    xed = xe.float().to(DEV)

    def evaluate():
        m.eval()
        with torch.no_grad():
            assert m._foldable()
            folded = m(xed).clone()
        general = m(xed).detach().clone()                       # grad enabled: separate BatchNorm stages
        sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in m.state_dict().items()}
        lo, _, _ = O.model_forward(sd64, xe, False)
        m.train()
        return folded, general, lo

    for i in range(4):
        tr.step(xs[i % 4], l)
    assert tr.graph_ready()
    for p, v in zip(tr.flat.params, tr.flat.views):
        assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
    f1, g1, o1 = evaluate()
    assert rel(cpu64(f1), o1) < 1e-4 and rel(cpu64(f1), cpu64(g1)) < 1e-4
    # more replayed steps WITHOUT leaving eval/train through Model.train(): the trainer's own invalidation must do
    m.eval()
    with torch.no_grad():
        m(xed)                                                  # a plan exists
    for b in m.modules():
        b.training = True                                       # back to train mode behind Model.train()'s back
    for i in range(4):
        tr.step(xs[i % 4], l)
    for b in m.modules():
        b.training = False
    with torch.no_grad():
        assert m._foldable()
        f2 = m(xed).clone()
    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in m.state_dict().items()}
    o2, _, _ = O.model_forward(sd64, xe, False)
    assert rel(cpu64(f2), o2) < 1e-4, rel(cpu64(f2), o2)
    assert rel(cpu64(f2), cpu64(f1)) > 1e-3                     # the parameters did move


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_two_segment_graph_equals_one_graph(net, dtype):
    """With more than one rank the trainer captures a step as TWO graphs (forward + the backward of the head and the last
    blocks; the rest of the backward) and all-reduces the first segment's gradients while the second is replayed.  The
    split itself, forced here without a process group: six steps with dropout end in the same parameters, bit for bit,
    as the one-graph trainer and as the launch-by-launch trainer."""
    mod, _ = net
    x = O.det_input(4, 3, 48, 25, 2, tag=3270).float().to(DEV)
    l = torch.tensor([4, 9, 33, 58], device=DEV)
    runs = []
    for graph, split in ((False, None), (True, False), (True, True)):
        torch.manual_seed(99)
        m, tr = _fresh_trainer(mod, dtype, 0.5, graph, lr=0.01, split=split)
        losses = [float(tr.step(x, l).detach()) for _ in range(6)]
        torch.cuda.synchronize()
        assert tr.graph_ready() == graph
        if split:
            ent = next(iter(tr._graphs.values()))
            assert isinstance(ent[0], tuple) and 0 < ent[0][2] < tr.flat.flat.numel()        # two graphs, a cut inside the buffer
            assert ent[0][2] > tr.flat.flat.numel() // 2                                   # most parameters travel early
            for p, v in zip(tr.flat.params, tr.flat.views):
                assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
        runs.append((losses, tr.flat.flat.detach().clone(), tr.flat.grad.detach().clone()))
        del m, tr
    for losses, params, grads in runs[1:]:
        assert losses == runs[0][0], (losses, runs[0][0])
        assert torch.equal(params, runs[0][1]) and torch.equal(grads, runs[0][2])
