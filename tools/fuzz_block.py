"""Random-configuration sweep of the st_gcn block module (forward and backward, train and eval mode, both storage
types) against the CPU oracle: channel widths inside and outside the kernels' native multiples, both strides, the three
residual kinds, both skeleton layouts.  What the per-kernel sweeps cannot see -- padding, dispatch between the
kernels, the order of the launches -- goes through here.

    python tools/fuzz_block.py [cases] [seed]
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
from tests.golden import cases as C            # noqa: E402
import net.st_gcn as mod                       # noqa: E402
import stgcn_rt.functional as F               # noqa: E402

DEV = "cuda:0"
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def cpu64(t):
    return t.detach().to("cpu", torch.float64)


def err(a, b, dtype):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    if dtype == torch.float32:
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(n):
        lay = rng.choice(["ntu-rgb+d", "openpose"])
        cin = rng.choice([3, 16, 24, 40, 64, 72, 96, 128, 160, 256])
        cout = rng.choice([16, 40, 64, 96, 128, 192, 256])
        stride = rng.choice([1, 2])
        residual = rng.random() < 0.8
        B, T = rng.choice([1, 2, 3]), rng.choice([4, 9, 16, 31, 40])
        training = rng.random() < 0.75
        dtype = rng.choice([torch.float32, torch.bfloat16])
        case = ("fuzz%d" % i, lay, cin, cout, stride, residual, B, T, training)
        x, A, p, bufs, dout, kind = C.block_inputs(case, 70000 + 100 * i)
        K = A.shape[0]
        m = mod.st_gcn(cin, cout, (9, K), stride, dropout=0, residual=residual).to(DEV)
        sd = m.state_dict()
        with torch.no_grad():
            for k, v in list(p.items()) + list(bufs.items()):
                sd[k].copy_(v)
        m.train(training)
        xr = x.to(DEV).to(dtype).requires_grad_(True)
        Ar = A.to(DEV).float().requires_grad_(True)
        F.CAPTURE = []
        y, _ = m(xr, Ar)
        y.backward(dout.to(DEV).to(dtype))
        torch.cuda.synchronize()
        cap, F.CAPTURE = F.CAPTURE, None
        masks = tuple(mk[:, :cout] for mk in (cap[0]["h_mask"].cpu(), cap[0]["out_mask"].cpu()))
        got = {"out": y, "dx": xr.grad, "dA": Ar.grad}
        for k, prm in m.named_parameters():
            got["g:" + k] = prm.grad
        for k in bufs:
            got["buf:" + k] = m.state_dict()[k]
        out_o, cache, nb = O.block_forward(x.to(dtype).double(), A, p, bufs, stride, kind, training, None, 1.0, masks[0], masks[1])
        dx_o, dA_o, g_o = O.block_backward(dout.to(dtype).double(), p, cache, stride, kind, training)
        ref = {"out": out_o, "dx": dx_o, "dA": dA_o}
        ref.update({"g:" + k: v for k, v in g_o.items()})
        ref.update({"buf:" + k: v for k, v in nb.items()})
        zero_bias = ("g:tcn.2.bias", "g:residual.0.bias") if training else ()
        worst, wk = 0.0, ""
        for k, r in ref.items():
            if k in zero_bias:                   # exactly zero in exact arithmetic: bounded by the noise floor
                w = "g:tcn.2.weight" if k == "g:tcn.2.bias" else "g:residual.0.weight"
                e = float(cpu64(got[k]).abs().max() / (50 * ref[w].abs().max())) * TOL[dtype]
            else:
                e = err(cpu64(got[k]), r, dtype)
            if not np.isfinite(e) or e > worst:
                worst, wk = e, k
        ok = np.isfinite(worst) and worst < TOL[dtype]
        bad += not ok
        print(i, case[1:], kind, str(dtype)[6:], "worst %.2e at %s" % (worst, wk), "" if ok else "  <-- FAIL", flush=True)
    print("FUZZ_DONE failures", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
