"""Generate tests/golden/*.npz by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
It imports net.st_gcn / net.utils.tgcn / net.utils.graph (and feeder.tools, which needs only numpy and
random) from /root/reference, feeds
them closed-form deterministic inputs and weights (oracle.stgcn_oracle.det_uniform,
no library RNG) in float64, and stores the outputs.  Inputs are NOT stored: tests
regenerate them from the same closed form.  Nothing from the reference's source is
stored; only numeric outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import stgcn_oracle as O  # noqa: E402

from net.st_gcn import Model as RefModel, st_gcn as RefBlock  # noqa: E402
from net.utils.graph import Graph as RefGraph  # noqa: E402
from net.utils.tgcn import ConvTemporalGraphical as RefGCN  # noqa: E402

torch.set_default_dtype(torch.float64)

from tests.golden.cases import (GRAPH_CASES, GCN_CASES, GCN_T_CASES, gcn_t_inputs, BLOCK_CASES, MODEL_CASES, FEEDER_CASES, t64, compact,  # noqa: E402
                                gcn_inputs, block_inputs, model_inputs, feeder_clip, strategy_of)


def gen_graphs():
    out = {}
    for lay, strat, mh, dil in GRAPH_CASES:
        g = RefGraph(layout=lay, strategy=strat, max_hop=mh, dilation=dil)
        key = "%s|%s|%d|%d" % (lay, strat, mh, dil)
        out[key] = g.A
        mine = O.build_graph(lay, strat, mh, dil)
        assert mine["A"].shape == g.A.shape and np.array_equal(mine["A"], g.A), key
        assert mine["edge"] == g.edge and mine["center"] == g.center
        assert np.array_equal(mine["hop_dis"], g.hop_dis)
    np.savez_compressed(os.path.join(HERE, "graph.npz"), **out)
    print("graph: %d cases, oracle identical" % len(out))


def gen_gcn():
    out = {}
    for ci, case in enumerate(GCN_CASES):
        name = case[0]
        x, A, W, b, dy = gcn_inputs(case, 1000 + 10 * ci)
        K = A.shape[0]
        m = RefGCN(x.shape[1], W.shape[0] // K, K)
        with torch.no_grad():
            m.conv.weight.copy_(W.reshape(m.conv.weight.shape))
            m.conv.bias.copy_(b)
        xr = x.clone().requires_grad_(True)
        Ar = A.clone().requires_grad_(True)
        y, _ = m(xr, Ar)
        y.backward(dy)
        ref = dict(y=y.detach(), dx=xr.grad, dW=m.conv.weight.grad.reshape(W.shape), db=m.conv.bias.grad, dA=Ar.grad)
        my = O.gcn_forward(x, A, W, b)
        mdx, mdW, mdb, mdA = O.gcn_backward(x, A, W, b, dy)
        for k, v in dict(y=my, dx=mdx, dW=mdW, db=mdb, dA=mdA).items():
            err = (v - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-30)
            assert err < 1e-12, (name, k, err)
            out[name + "/" + k] = compact(ref[k])
    np.savez_compressed(os.path.join(HERE, "gcn_op.npz"), **out)
    print("gcn op: %d cases, oracle <1e-12" % len(GCN_CASES))


def gen_gcn_temporal():
    """ConvTemporalGraphical with t_kernel_size > 1, strides, padding and dilation (net/utils/tgcn.py:44-66)."""
    out = {}
    for ci, case in enumerate(GCN_T_CASES):
        name, lay, strat, B, Cin, Cout, T, kt, stride, pad, dil = case
        x, A, W, b, dy = gcn_t_inputs(case, 7000 + 10 * ci)
        K = A.shape[0]
        m = RefGCN(Cin, Cout, K, t_kernel_size=kt, t_stride=stride, t_padding=pad, t_dilation=dil).double()
        with torch.no_grad():
            m.conv.weight.copy_(W.reshape(m.conv.weight.shape))
            m.conv.bias.copy_(b)
        xr = x.clone().requires_grad_(True)
        Ar = A.clone().requires_grad_(True)
        y, _ = m(xr, Ar)
        y.backward(dy)
        ref = dict(y=y.detach(), dx=xr.grad, dW=m.conv.weight.grad.reshape(W.shape), db=m.conv.bias.grad, dA=Ar.grad)
        my = O.gcn_temporal_forward(x, A, W, b, stride, pad, dil)
        mdx, mdW, mdb, mdA = O.gcn_temporal_backward(x, A, W, b, dy, stride, pad, dil)
        for k, v in dict(y=my, dx=mdx, dW=mdW, db=mdb, dA=mdA).items():
            err = (v - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-30)
            assert err < 1e-12, (name, k, err)
            out[name + "/" + k] = compact(ref[k])
    np.savez_compressed(os.path.join(HERE, "gcn_temporal.npz"), **out)
    print("gcn with temporal kernel: %d cases, oracle <1e-12" % len(GCN_T_CASES))


def gen_block():
    out = {}
    for ci, case in enumerate(BLOCK_CASES):
        name, lay, cin, cout, stride, residual, B, T, training = case
        x, A, p, bufs, dout, kind = block_inputs(case, 5000 + 100 * ci)
        K = A.shape[0]
        m = RefBlock(cin, cout, (9, K), stride, dropout=0, residual=residual)
        sd = m.state_dict()
        for k, v in list(p.items()) + list(bufs.items()):
            sd[k].copy_(v)
        m.train(training)
        xr = x.clone().requires_grad_(True)
        Ar = A.clone().requires_grad_(True)
        y, _ = m(xr, Ar)
        y.backward(dout)
        ref = {"out": y.detach(), "dx": xr.grad, "dA": Ar.grad}
        for k, prm in m.named_parameters():
            ref["g:" + k] = prm.grad
        for k in bufs:
            ref["buf:" + k] = m.state_dict()[k].clone()
        mo, cache, nb = O.block_forward(x, A, p, bufs, stride, kind, training)
        mdx, mdA, mg = O.block_backward(dout, p, cache, stride, kind, training)
        mine = {"out": mo, "dx": mdx, "dA": mdA}
        mine.update({"g:" + k: v for k, v in mg.items()})
        mine.update({"buf:" + k: v for k, v in nb.items()})
        for k, v in ref.items():
            scale = v.abs().max().item() + 1e-30
            err = (mine[k].reshape(v.shape) - v).abs().max().item() / scale
            # conv bias grads ahead of a train-mode BN are ~0 (pure rounding): compare absolutely
            if training and k in ("g:tcn.2.bias", "g:residual.0.bias"):
                assert mine[k].abs().max().item() < 1e-9 and v.abs().max().item() < 1e-9, (name, k)
            else:
                assert err < 1e-10, (name, k, err)
            out[name + "/" + k] = compact(v)
    np.savez_compressed(os.path.join(HERE, "block.npz"), **out)
    print("block: %d cases, oracle <1e-10" % len(BLOCK_CASES))


def gen_model():
    out = {}
    for ci, (name, lay, ncls, N, T, M, training) in enumerate(MODEL_CASES):
        strat = strategy_of(name)
        g = O.build_graph(lay, strat)
        V = g["num_node"]
        sd = O.det_state(3, ncls, g["A"], True)
        x, labels = model_inputs(ci)
        m = RefModel(3, ncls, dict(layout=lay, strategy=strat), True)
        m.double()
        missing = m.load_state_dict(sd, strict=True)
        m.train(training)
        logits = m(x)
        loss = torch.nn.functional.cross_entropy(logits, labels)
        out[name + "/logits"] = logits.detach().numpy()
        out[name + "/loss"] = loss.detach().numpy()
        ml, cache, newb = O.model_forward(sd, x, training)
        mloss, dlog = O.cross_entropy(ml, labels)
        e = (ml - logits.detach()).abs().max().item() / logits.abs().max().item()
        assert e < 1e-9, (name, "logits", e)
        assert abs(mloss.item() - loss.item()) < 1e-10
        if training:
            margin = min((pre.abs().min() / pre.pow(2).mean().sqrt()).item() for bc in cache["blocks"] for pre in (bc["h1"], bc["s"]))
            out[name + "/relu_margin"] = np.array(margin)  # smallest |pre-activation| / rms over every ReLU input
            if "margin" in name:
                assert margin > 1.5e-5, (name, margin)
            loss.backward()
            mg = O.model_backward(sd, cache, dlog, True)
            for k, prm in m.named_parameters():
                gr = prm.grad
                if k.endswith("tcn.2.bias") or k.endswith("residual.0.bias") :
                    assert mg[k].abs().max().item() < 1e-9, (name, k)
                else:
                    e = (mg[k] - gr).abs().max().item() / (gr.abs().max().item() + 1e-30)
                    assert e < 1e-8, (name, k, e)
                out[name + "/gnorm:" + k] = np.array(gr.norm().item())
                flat = gr.reshape(-1)
                out[name + "/gsample:" + k] = flat[:: max(1, flat.numel() // 257)].numpy().astype(np.float64)
                if k.startswith("edge_importance") or k.startswith("data_bn") or k.startswith("fcn.bias"):
                    out[name + "/g:" + k] = gr.numpy()
            rsd = m.state_dict()
            for k in newb:
                e = (newb[k] - rsd[k]).abs().max().item() / (rsd[k].abs().max().item() + 1e-30)
                assert e < 1e-9, (name, k, e)
            for k in ("data_bn.running_mean", "data_bn.running_var",
                      "st_gcn_networks.0.tcn.0.running_mean", "st_gcn_networks.0.tcn.0.running_var",
                      "st_gcn_networks.4.residual.1.running_var",
                      "st_gcn_networks.9.tcn.3.running_mean", "st_gcn_networks.9.tcn.3.running_var"):
                out[name + "/buf:" + k] = rsd[k].numpy()
        if name.endswith("small_eval"):
            o2, f2 = m.extract_feature(x)
            mo, mf = O.model_forward(sd, x, False, return_feature=True)
            assert (mo - o2.detach()).abs().max().item() < 1e-9 and (mf - f2.detach()).abs().max().item() < 1e-9
            out[name + "/feat_output"] = o2.detach().numpy().astype(np.float32)
            out[name + "/feat_feature"] = f2.detach().numpy().astype(np.float32)
        print("model %-22s ok (loss %.6f)" % (name, loss.item()))
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)


def gen_sgd():
    """3 SGD-nesterov steps (recognition.py:46-53, 84-97) on the small NTU model, dropout 0."""
    lay, ncls, N, T, M = "ntu-rgb+d", 60, 2, 24, 2
    g = O.build_graph(lay, "spatial")
    sd = O.det_state(3, ncls, g["A"], True)
    m = RefModel(3, ncls, dict(layout=lay, strategy="spatial"), True).double()
    m.load_state_dict(sd)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    osd = {k: v.clone() for k, v in sd.items()}
    moms = {}
    out = {}
    for step in range(3):
        x = O.det_input(N, 3, T, 25, M, tag=1200 + step)
        labels = torch.tensor([(5 * i + step) % ncls for i in range(N)])
        loss = torch.nn.functional.cross_entropy(m(x), labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        ml, cache, newb = O.model_forward(osd, x, True)
        mloss, dlog = O.cross_entropy(ml, labels)
        mg = O.model_backward(osd, cache, dlog, True)
        names = O.param_names(osd)
        prm = {k: osd[k] for k in names}
        O.sgd_nesterov_step(prm, mg, moms, 0.1)
        osd.update(prm)
        osd.update(newb)
        out["loss%d" % step] = np.array(loss.item())
        assert abs(mloss.item() - loss.item()) < 1e-8, (step, mloss.item(), loss.item())
    rsd = m.state_dict()
    for k in O.param_names(sd):
        e = (osd[k] - rsd[k]).abs().max().item() / (rsd[k].abs().max().item() + 1e-30)
        assert e < 1e-7, (k, e)
        out["pnorm:" + k] = np.array(rsd[k].norm().item())
    for k in ("fcn.bias", "edge_importance.3", "st_gcn_networks.0.gcn.conv.weight"):
        out["p:" + k] = rsd[k].numpy()
    np.savez_compressed(os.path.join(HERE, "sgd.npz"), **out)
    print("sgd: 3 steps ok, losses", [float(out["loss%d" % i]) for i in range(3)])


def gen_bf16_reference():
    """What bfloat16 costs the REFERENCE: its own Model (float32 parameters) run under
    torch.autocast("cpu", dtype=torch.bfloat16) on the training cases -- convolutions and the einsum in bfloat16, their
    outputs stored in bfloat16 -- and its parameter gradients compared, tensor by tensor, with the float64 evaluation
    that takes the SAME ReLU decisions (read from the run with forward hooks) and the same d loss / d logits.  Stored:
    the relative L2 deviation of every gradient tensor, and of the logits from the float64 reference logits.  The GPU
    tests bound the kernels' bfloat16 gradients by these numbers, measured the same way."""
    out = {}
    for ci, (name, lay, ncls, N, T, M, training) in enumerate(MODEL_CASES):
        if not training:
            continue
        strat = strategy_of(name)
        g = O.build_graph(lay, strat)
        sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
        x, labels = model_inputs(ci)
        m = RefModel(3, ncls, dict(layout=lay, strategy=strat), True).float()
        m.load_state_dict(sd, strict=True)
        m.train()
        masks = {}

        def catch(i, slot):
            def hook(mod, inp, res):
                masks.setdefault(i, [None, None])[slot] = (res.detach() > 0).clone()
            return hook
        for i, blk in enumerate(m.st_gcn_networks):
            blk.tcn[1].register_forward_hook(catch(i, 0))
            blk.relu.register_forward_hook(catch(i, 1))
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits = m(x.float())
            loss = torch.nn.functional.cross_entropy(logits, labels)
        loss.backward()
        assert logits.dtype == torch.bfloat16
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        plain, _, _ = O.model_forward(sd64, x, True)
        _, cache, _ = O.model_forward(sd64, x, True, masks={i: tuple(v) for i, v in masks.items()})
        _, dlog = O.cross_entropy(logits.detach().double(), labels)
        ref_g = O.model_backward(sd64, cache, dlog, True)
        devs = []
        for k, prm in m.named_parameters():
            if k.endswith("tcn.2.bias") or k.endswith("residual.0.bias"):
                continue                                   # zero up to rounding noise behind a train-mode BatchNorm
            d = ((prm.grad.double() - ref_g[k]).norm() / (ref_g[k].norm() + 1e-30)).item()
            out[name + "/dev:" + k] = np.array(d)
            devs.append(d)
        out[name + "/logits_dev"] = np.array(((logits.detach().double() - plain).abs().max() / plain.abs().max()).item())
        devs.sort()
        print("bf16 reference %-22s logits %.2e, gradients worst %.2e median %.2e" % (name, float(out[name + "/logits_dev"]), devs[-1], devs[len(devs) // 2]))
    np.savez_compressed(os.path.join(HERE, "bf16_ref.npz"), **out)


def gen_feeder():
    """feeder/tools.py of the reference on closed-form clips, with the module-level generators it draws from
    (random, numpy.random) seeded per case: the tests hand the same seeds to random.Random / numpy RandomState."""
    import random
    import feeder.tools as RT
    out = {}
    for name, fn, shape, gap, kw, s_py, s_np in FEEDER_CASES:
        d = feeder_clip(name, shape, gap)
        random.seed(s_py)
        np.random.seed(s_np)
        if fn == "pipeline":
            r = RT.random_move(np.array(RT.random_choose(d.copy(), kw["size"])))
        else:
            r = getattr(RT, fn)(d.copy(), **kw)
        out[name] = np.zeros(0) if r is None else np.asarray(r, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "feeder.npz"), **out)
    print("feeder: %d cases" % len(out))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # e.g. `make_golden.py feeder model`: regenerate only those files
        for what in sys.argv[1:]:
            {"graph": gen_graphs, "gcn": gen_gcn, "gcn_temporal": gen_gcn_temporal, "block": gen_block, "model": gen_model, "sgd": gen_sgd,
             "feeder": gen_feeder, "bf16": gen_bf16_reference}[what]()
        sys.exit(0)
    gen_feeder()
    gen_graphs()
    gen_gcn()
    gen_gcn_temporal()
    gen_block()
    gen_model()
    gen_sgd()
    gen_bf16_reference()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
