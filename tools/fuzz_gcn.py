"""Random-shape sweep of the graph-convolution entry points (fused forward with statistics, input gradient with and
without addend, weight and bias gradient, adjacency gradient) (bf16, or f32 on request) against the CPU oracle: channel widths in
multiples of 16, both skeleton layouts, the three partition strategies and frame counts drawn at random.

    python tools/fuzz_gcn.py [cases] [seed] [bf16|f32]
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
from oracle import stgcn_oracle as O          # noqa: E402
from stgcn_rt import lib as L                  # noqa: E402
from stgcn_rt import ops as R                  # noqa: E402

DEV = "cuda:0"
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
TOL = 1e-4 if dt == torch.float32 else 2e-2


def t64(shape, tag, lo=-1.0, hi=1.0):
    return torch.tensor(O.det_uniform(shape, tag, lo, hi), dtype=torch.float64)


def q(x):
    return x.to(dt).to(torch.float64)


def dev(x):
    return x.to(DEV).to(dt).contiguous(memory_format=torch.channels_last)


def rel(a, b):
    a, b = a.detach().to("cpu", torch.float64), b.detach().to("cpu", torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    L.load()
    graphs = {(lay, st): O.build_graph(lay, st)["A"] for lay in ("ntu-rgb+d", "openpose") for st in ("spatial", "uniform", "distance")}
    bad = 0
    for i in range(n):
        lay, st = rng.choice(list(graphs))
        A0 = graphs[(lay, st)]
        K, V = A0.shape[0], A0.shape[1]
        B = rng.choice([1, 2, 3])
        Cin = rng.choice([16, 32, 48, 64, 128, 256, 96, 80, 160])
        Cout = rng.choice([16, 32, 64, 128, 256, 48, 208, 96])
        T = rng.choice([1, 2, 5, 12, 40, 150])
        with_bias = rng.random() < 0.7
        case = (lay, st, B, Cin, Cout, T, with_bias)
        A = torch.tensor(A0) * t64(A0.shape, 40 + i, 0.5, 1.5) if rng.random() < 0.5 else t64(A0.shape, 40 + i, -0.5, 0.5)
        x = q(t64((B, Cin, T, V), 1000 + i))
        W = q(t64((K * Cout, Cin), 2000 + i) * (1.5 / np.sqrt(Cin)))
        b = t64((K * Cout,), 3000 + i, -0.2, 0.2) if with_bias else None
        dy = q(t64((B, Cout, T, V), 4000 + i))
        Aq = q(A)
        y = O.gcn_forward(x, Aq, W, b)
        dx, dW, db, dA = O.gcn_backward(x, Aq, W, b, dy)
        Ad = A.to(DEV).float().contiguous()
        W4 = W.reshape(K * Cout, Cin, 1, 1).to(DEV).float()
        bd = None if b is None else b.to(DEV).float()
        xd, dyd = dev(x), dev(dy)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
        wp, wt = R.pack_gcn_w(W4, K, Cin, dt), R.pack_gcn_wt(W4, K, Cin, dt)
        errs = {"fwd": rel(R.gcn_fused(xd, wp, Ad, Cout, bias=bd, stats=stats, adj_t=True), y)}
        errs["mean"] = rel(stats[:Cout].cpu() / (B * T * V), y.mean(dim=(0, 2, 3))) / 2
        errs["dx"] = rel(R.gcn_fused(dyd, wt, Ad, Cin), dx)
        addt = q(t64((B, Cin, T, V), 5000 + i))
        errs["dx+add"] = rel(R.gcn_fused(dyd, wt, Ad, Cin, add=dev(addt)), dx + addt)
        try:
            dWk, dbk = R.gcn_wgrad(xd, dyd, Ad)
            errs["dW"] = rel(dWk.reshape(K * Cout, Cin), dW)
            errs["db"] = rel(dbk, db if db is not None else O.gcn_backward(x, Aq, W, torch.zeros(K * Cout, dtype=torch.float64), dy)[2])
        except RuntimeError as e:                # widths the entry point refuses (the modules pad before they call it)
            print("   gcn_wgrad refuses", case, str(e)[-40:])
        try:
            errs["dA"] = rel(R.gcn_agrad(xd, wp, bd, dyd), dA)
        except RuntimeError as e:
            print("   gcn_agrad refuses", case, str(e)[-40:])
        torch.cuda.synchronize()
        worst = max(errs.values())
        flag = "" if worst < TOL and all(np.isfinite(v) for v in errs.values()) else "   <-- FAIL"
        bad += bool(flag)
        print(i, case, " ".join("%s %.1e" % kv for kv in errs.items()) + flag, flush=True)
    print("FUZZ_DONE failures", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
