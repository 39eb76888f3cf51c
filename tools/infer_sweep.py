"""Large-batch bf16 inference sweep: eval-mode forward at N = 1024 clips (T = 300, M = 2) for both skeleton layouts and
the three partition strategies (K = 3 / 1 / 2), with a check that a clip's logits inside the big batch equal those of
the same clips run as a two-clip batch.  Uses the oracle only for the deterministic weights and the graph.

    python tools/infer_sweep.py
"""
import os
import sys
import time

import torch

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(_ROOT, "st-gcn_amd"))
sys.path.insert(0, _ROOT)
from oracle import stgcn_oracle as O
from net import st_gcn
DEV = "cuda:0"
for lay, ncls, V in (("ntu-rgb+d", 60, 25), ("openpose", 400, 18)):
    for strat in ("spatial", "uniform", "distance"):
        g = O.build_graph(lay, strat)
        sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
        m = st_gcn.Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=torch.bfloat16).to(DEV)
        m.load_state_dict(sd, strict=True)
        m.eval()
        N = 1024
        x = torch.randn(N, 3, 300, V, 2, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        with torch.no_grad():
            full = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                full = m(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            worst = 0.0
            for i in (0, 510, 1022):
                part = m(x[i:i + 2].contiguous())
                worst = max(worst, ((full[i:i + 2] - part).abs().max() / part.abs().max()).item())
        print(lay, strat, "K=%d" % g["A"].shape[0], "N=1024 eval: %.1f ms, %.0f clips/s, finite %s, per-clip rel diff vs 2-clip batches %.2e, mem %.1f GB"
              % (dt * 1e3, N / dt, bool(torch.isfinite(full).all()), worst, torch.cuda.max_memory_allocated() / 2**30), flush=True)
        del m, x, full
        torch.cuda.empty_cache()
