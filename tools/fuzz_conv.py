"""Random-shape sweep of the temporal-convolution entry points (forward with statistics, transposed form, weight
gradient) (bf16, or f32 on request) against the CPU oracle: channel widths, strides, tap counts, joint counts and frame counts drawn at
random, so that every dispatch boundary between the convolution kernels is crossed many times.

    python tools/fuzz_conv.py [cases] [seed] [bf16|f32]
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
    bad = 0
    for i in range(n):
        B = rng.choice([1, 2, 3, 5])
        Cin = rng.choice([64, 128, 256])
        Cout = rng.choice([64, 128, 192, 256, 320, 72])
        kt = rng.choice([1, 1, 9, 9, 3, 5, 7, 4, 6])
        stride = rng.choice([1, 2, 2])
        V = rng.choice([1, 7, 18, 25, 25, 32])
        T = rng.choice([1, 2, 3, 9, 20, 37, 64, 150])
        pad = (kt - 1) // 2
        pro = kt > 1 and rng.random() < 0.5
        case = (B, Cin, Cout, T, V, kt, stride, pad, pro)
        if (T + 2 * pad - kt) // stride + 1 < 1:            # an even tap count on a clip shorter than the kernel: no output frame
            continue
        print(i, case, end=" ", flush=True)
        x = q(t64((B, Cin, T, V), 100 + i))
        W = q(t64((Cout, Cin, kt), 200 + i) * (1.5 / np.sqrt(Cin * kt)))
        b = t64((Cout,), 300 + i, -0.2, 0.2)
        sc, sh = t64((Cin,), 400 + i, 0.5, 1.5), t64((Cin,), 500 + i, -0.3, 0.3)
        h = torch.clamp_min(x * sc[None, :, None, None] + sh[None, :, None, None], 0) if pro else x
        if pro:
            h = q(h)
        ref = O.tconv_forward(h, W, b, stride, pad)
        To = ref.shape[2]
        do = q(t64((B, Cout, To, V), 600 + i))
        dh, dW, _ = O.tconv_backward(h, W, do, stride, pad)
        W4 = W[..., None].to(DEV).float()
        pr = (sc.to(DEV).float(), sh.to(DEV).float()) if pro else None
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
        out = R.conv_rows(dev(x), R.pack_conv_w(W4, dt), Cout, To, bias=b.to(DEV).float(), pro=pr, stats=stats, ntaps=kt, stride=stride, pad=pad)
        errs = {"fwd": rel(out, ref)}
        nrow = B * To * V
        errs["mean"] = rel(stats[:Cout].cpu() / nrow, ref.mean(dim=(0, 2, 3))) / 2
        errs["wgrad"] = rel(R.conv_wgrad(dev(x), dev(do), kt, stride, pad, pro=pr)[..., 0], dW)
        if Cout % 64 == 0:
            errs["dgrad"] = rel(R.conv_rows(dev(do), R.pack_conv_wt(W4, dt), Cin, T, ntaps=kt, stride=stride, pad=pad, transposed=True), dh)
        torch.cuda.synchronize()
        worst = max(errs.values())
        flag = "" if worst < TOL and all(np.isfinite(v) for v in errs.values()) else "   <-- FAIL"
        bad += bool(flag)
        print(" ".join("%s %.1e" % kv for kv in errs.items()) + flag, flush=True)
    print("FUZZ_DONE failures", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
