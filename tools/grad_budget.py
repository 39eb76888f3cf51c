#!/usr/bin/env python
"""Where does the bf16 gradient error of the whole model come from?

Runs one training step of a golden model case in bf16 on the GPU with every block's tensors captured, evaluates the
float64 oracle with the kernels' ReLU decisions, and prints per block

  forward      rel-L2 error of the stored activations x, y, o
  backward     rel-L2 error of the gradients that travel between kernels: dout, dO, gh, dy, dx
  parameter gradients, three ways each:
     total      GPU result vs oracle
     inherited  float64 evaluation of the SAME contraction from the GPU's own (bf16) operands vs oracle
                -- what any exact kernel would return on these operands
     kernel     GPU result vs that float64 evaluation -- what the kernel itself adds
     cancel     sum |terms| / |sum terms| of the contraction (how much the sum cancels)

  python tools/grad_budget.py [case index, default 0 = ntu_small_train] [dtype bf16|fp32]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

from oracle import stgcn_oracle as O
from tests.golden import cases as C


def rl2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def c64(t):
    return None if t is None else t.detach().to("cpu", torch.float64)


def main():
    ci = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dtype = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "fp32" else torch.bfloat16
    import net.st_gcn as mod
    import stgcn_rt.functional as F
    name, lay, ncls, N, T, M, training = C.MODEL_CASES[ci]
    strat = C.strategy_of(name)
    g = O.build_graph(lay, strat)
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    x, labels = C.model_inputs(ci)
    dev = "cuda:0"
    m = mod.Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=dtype).to(dev)
    m.load_state_dict(sd, strict=True)
    m.train()
    F.CAPTURE, F.CAPTURE_TENSORS = [], True
    logits = m(x.float().to(dev))
    loss = torch.nn.functional.cross_entropy(logits, labels.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    cap, F.CAPTURE, F.CAPTURE_TENSORS = F.CAPTURE, None, False
    nb = len(cap)
    caps = {nb - 1 - j: c for j, c in enumerate(cap)}
    masks = {i: (c["h_mask"].cpu(), c["out_mask"].cpu()) for i, c in caps.items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, cache, _ = O.model_forward(sd64, x, True, masks=masks)
    _, dlog = O.cross_entropy(c64(logits), labels)
    ref_g = O.model_backward(sd64, cache, dlog, True)
    got_g = {k: c64(p.grad) for k, p in m.named_parameters()}

    # oracle backward again, block by block, keeping what travels between the kernels
    Nn, Cc, Tt, V, Mm = cache["dims"]
    Bc, Cf, Tf, Vf = cache["feat_shape"]
    Wf = sd64["fcn.weight"].reshape(ncls, -1)
    dh = ((dlog @ Wf) / (Mm * Tf * Vf))[:, None, :, None, None].expand(Nn, Mm, Cf, Tf, Vf).reshape(Bc, Cf, Tf, Vf)
    print("case %s, %s; logits rel err %.2e" % (name, str(dtype).split(".")[-1],
                                                  rl2(c64(logits), O.model_forward(sd64, x, True)[0])))
    hdr = "blk | fwd: x y o | bwd: dout dO gh dy dx"
    print(hdr)
    rows = []
    for i in reversed(range(nb)):
        bc = cache["blocks"][i]
        cp = caps[i]
        p, _ = O._block_view(sd64, i)
        kind, stride = bc["kind"], bc["stride"]
        dout_o = dh
        ds = dout_o * bc["omask"].double()
        do, dg2_o, _ = O.bn_backward(bc["o"], p["tcn.3.weight"], bc["c2"], ds, True)
        Wt = p["tcn.2.weight"][..., 0]
        dhh, dWt_o, _ = O.tconv_backward(bc["h"], Wt, do, stride)
        dh1 = dhh * bc["hmask"].double()
        dy_o, dg1_o, _ = O.bn_backward(bc["y"], p["tcn.0.weight"], bc["c1"], dh1, True)
        Wg = p["gcn.conv.weight"].reshape(p["gcn.conv.weight"].shape[0], -1)
        dx_o, dWg_o, _, dA_o = O.gcn_backward(bc["x"], bc["A"], Wg, p["gcn.conv.bias"], dy_o)
        if kind == "identity":
            dx_o = dx_o + ds
        elif kind == "conv":
            dr, _, _ = O.bn_backward(bc["r"], p["residual.1.weight"], bc["cr"], ds, True)
            dxr, _, _ = O.tconv_backward(bc["x"], p["residual.0.weight"][..., 0], dr, stride, pad=0)
            dx_o = dx_o + dxr
        dh = dx_o

        cin = bc["x"].shape[1]
        xg = c64(cp["x"])[:, :cin]
        yg, og = c64(cp["y"]), c64(cp["o"])
        if cp["dout"] is not None:
            doutg = c64(cp["dout"])
        else:
            doutg = c64(cp["dpool"])[:, :, None, None].expand_as(dout_o)
        dOg, ghg, dyg = c64(cp["dO"]), c64(cp["gh"]), c64(cp["dy"])
        dxg = c64(cp["dx"])[:, :cin]
        print("%2d  | %.1e %.1e %.1e | %.1e %.1e %.1e %.1e %.1e" % (
            i, rl2(xg, bc["x"]), rl2(yg, bc["y"]), rl2(og, bc["o"]),
            rl2(doutg, dout_o), rl2(dOg, do), rl2(ghg, dh1), rl2(dyg, dy_o), rl2(dxg, dx_o)))

        pre = "st_gcn_networks.%d." % i
        # --- dA: contraction of z = W x + b with dy, from the GPU's own operands in float64
        Wq = Wg.to(dtype).double() if dtype == torch.bfloat16 else Wg
        _, dWg_i, _, dA_i = O.gcn_backward(xg, bc["A"], Wq, p["gcn.conv.bias"], dyg)
        dA_g = c64(cp["dA"])
        K = bc["A"].shape[0]
        Cout = Wg.shape[0] // K
        z = torch.einsum("koc,bctv->bkotv", Wq.reshape(K, Cout, -1), xg) + p["gcn.conv.bias"].reshape(1, K, Cout, 1, 1)
        absum = torch.einsum("bkotv,botw->kvw", z.abs(), dyg.abs())
        Amask = sd64["A"] != 0
        canc = float(absum[Amask].sum() / (dA_i[Amask].abs().sum() + 1e-300))
        A0 = sd64["A"]
        rows.append(("edge_importance.%d" % i, rl2(dA_g * A0, dA_o * A0), rl2(dA_i * A0, dA_o * A0), rl2(dA_g * A0, dA_i * A0), canc))
        rows.append((pre + "gcn.conv.weight", rl2(got_g[pre + "gcn.conv.weight"], dWg_o), rl2(dWg_i, dWg_o),
                     rl2(got_g[pre + "gcn.conv.weight"].reshape(dWg_i.shape), dWg_i), float("nan")))
        # --- d gamma2 = sum dq * ohat from the GPU's o (bf16), its mean / rstd and its incoming gradient
        bn1, bn2, bnr = cp["bn"]
        dsg = doutg * cp["out_mask"].cpu().double()
        ohat = (og - c64(bn2.mean)[None, :, None, None]) * c64(bn2.rstd)[None, :, None, None]
        dg2_i = (dsg * ohat).sum(dim=(0, 2, 3))
        canc2 = float((dsg * ohat).abs().sum() / (dg2_i.abs().sum() + 1e-300))
        rows.append((pre + "tcn.3.weight", rl2(got_g[pre + "tcn.3.weight"], dg2_o), rl2(dg2_i, dg2_o), rl2(got_g[pre + "tcn.3.weight"], dg2_i), canc2))
        # --- d gamma1 = sum gh * yhat
        yhat = (yg - c64(bn1.mean)[None, :, None, None]) * c64(bn1.rstd)[None, :, None, None]
        dg1_i = (ghg * yhat).sum(dim=(0, 2, 3))
        canc1 = float((ghg * yhat).abs().sum() / (dg1_i.abs().sum() + 1e-300))
        rows.append((pre + "tcn.0.weight", rl2(got_g[pre + "tcn.0.weight"], dg1_o), rl2(dg1_i, dg1_o), rl2(got_g[pre + "tcn.0.weight"], dg1_i), canc1))
        # --- temporal conv weight gradient from the GPU's h = relu(bn1(y)) (rounded as the kernel's prologue does) and dO
        hg = (yg * c64(bn1.scale)[None, :, None, None] + c64(bn1.shift)[None, :, None, None]) * cp["h_mask"].cpu().double()
        if dtype == torch.bfloat16:
            hg = hg.to(torch.bfloat16).double()
        _, dWt_i, _ = O.tconv_backward(hg, Wt, dOg, stride)
        gk = got_g[pre + "tcn.2.weight"][..., 0]
        rows.append((pre + "tcn.2.weight", rl2(gk, dWt_o), rl2(dWt_i, dWt_o), rl2(gk, dWt_i), float("nan")))
    print("\n%-36s %9s %9s %9s %8s" % ("gradient", "total", "inherited", "kernel", "cancel"))
    for r in rows:
        print("%-36s %9.2e %9.2e %9.2e %8.1f" % r)
    if dtype == torch.bfloat16:
        # the same formulas with bf16 rounding at the kernels' storage points: what is left is the kernels' arithmetic
        with O.storage_rounding(torch.bfloat16):
            lq, cq, _ = O.model_forward(sd64, x, True, masks=masks)
            _, dlq = O.cross_entropy(c64(logits), labels)
            ref_q = O.model_backward(sd64, cq, dlq, True)
        skip = lambda k: k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")
        print("\nlogits vs bf16-storage oracle: %.2e" % rl2(c64(logits), lq))
        print("per block stored x: GPU vs bf16-storage oracle:", " ".join("%.1e" % rl2(c64(caps[i]["x"])[:, :cq["blocks"][i]["x"].shape[1]], cq["blocks"][i]["x"]) for i in range(nb)))
        eq = sorted(((rl2(got_g[k], ref_q[k]), k) for k in got_g if not skip(k)), reverse=True)
        print("GPU vs bf16-storage oracle (sorted, worst 12 of %d; median %.2e):" % (len(eq), eq[len(eq) // 2][0]))
        for e, k in eq[:12]:
            print("  %.2e  %s" % (e, k))
        ei = sorted(((rl2(ref_q[k], ref_g[k]), k) for k in got_g if not skip(k)), reverse=True)
        print("bf16-storage oracle vs float64 oracle (the price of the format; worst 6, median %.2e):" % ei[len(ei) // 2][0])
        for e, k in ei[:6]:
            print("  %.2e  %s" % (e, k))
    print("\nall parameters, total error vs oracle (sorted):")
    errs = sorted(((rl2(got_g[k], ref_g[k]), k) for k in got_g if not (k.endswith("tcn.2.bias") or k.endswith("residual.0.bias"))), reverse=True)
    for e, k in errs[:16]:
        print("  %.2e  %s" % (e, k))


if __name__ == "__main__":
    main()
