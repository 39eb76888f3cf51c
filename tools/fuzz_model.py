"""Random-configuration sweep of the whole model against the CPU oracle: batch, clip length, persons, layout, partition
strategy and input channels drawn at random; train-mode logits in both storage types, eval-mode logits (the
BatchNorm-folded inference path), and in float32 every parameter gradient (the oracle evaluated with the kernels' own
ReLU decisions, as the tests do).

    python tools/fuzz_model.py [cases] [seed]
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
from oracle import stgcn_oracle as O          # noqa: E402
from net.st_gcn import Model                   # noqa: E402
import stgcn_rt.functional as F               # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(n):
        lay = rng.choice(["ntu-rgb+d", "openpose"])
        strat = rng.choice(["spatial", "uniform", "distance"])
        cin = rng.choice([2, 3, 3, 4])
        ncls = rng.choice([7, 60, 400])
        N, T, M = rng.choice([1, 2, 3, 5]), rng.choice([8, 13, 24, 40, 75]), rng.choice([1, 2, 3])
        g = O.build_graph(lay, strat)
        V = g["A"].shape[1]
        sd = O.det_state(cin, ncls, g["A"], True, dtype=torch.float32)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        x = O.det_input(N, cin, T, V, M, tag=9000 + i)
        labels = torch.tensor([(7 * j + 3) % ncls for j in range(N)])
        ref_train, _, _ = O.model_forward(sd64, x, True)
        ref_eval, _, _ = O.model_forward(sd64, x, False)
        errs = {}
        for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
            tag = str(dtype)[6:]
            m = Model(cin, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=dtype).to(DEV)
            m.load_state_dict(sd)
            m.eval()
            with torch.no_grad():
                errs["eval " + tag] = rel(m(x.float().to(DEV)), ref_eval) / tol
            m.train()
            F.CAPTURE = []
            logits = m(x.float().to(DEV))
            errs["train " + tag] = rel(logits, ref_train) / tol
            loss = torch.nn.functional.cross_entropy(logits, labels.to(DEV))
            loss.backward()
            torch.cuda.synchronize()
            cap, F.CAPTURE = F.CAPTURE, None
            if dtype == torch.float32:
                masks = {len(cap) - 1 - j: (c["h_mask"].cpu(), c["out_mask"].cpu()) for j, c in enumerate(cap)}
                _, cache, _ = O.model_forward(sd64, x, True, masks=masks)
                _, dlog = O.cross_entropy(logits.detach().cpu().double(), labels)
                ref_g = O.model_backward(sd64, cache, dlog, True)
                worst, wk = 0.0, ""
                for k, prm in m.named_parameters():
                    if k not in ref_g or prm.grad is None:
                        continue
                    r = ref_g[k]
                    if float(r.norm()) < 1e-9 * max(1.0, float(sum(v.norm() for v in ref_g.values()))):
                        continue                 # exactly zero in exact arithmetic (bias ahead of a train-mode BatchNorm)
                    e = ((prm.grad.detach().cpu().double() - r).norm() / r.norm()).item()
                    if e > worst:
                        worst, wk = e, k
                errs["grads f32 (%s)" % wk] = worst / 1e-3
        worst = max(errs.values())
        ok = worst < 1.0 and all(v == v for v in errs.values())
        bad += not ok
        print(i, (lay, strat, cin, ncls, N, T, M), " ".join("%s %.2f" % kv for kv in errs.items()), "" if ok else "  <-- FAIL", flush=True)
    print("FUZZ_DONE failures", bad, "(numbers are errors in units of their tolerance)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
