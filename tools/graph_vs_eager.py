"""Diagnostic: six training steps launch by launch (twice) and through the captured HIP graph; per-tensor deviation
of the six-step update between the runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd")):
    sys.path.insert(0, _p)
import torch
from oracle import stgcn_oracle as O
import net.st_gcn as mod
from stgcn_rt.ddp import Trainer
DEV = "cuda:0"
dtype = torch.float32
xs = [O.det_input(2, 3, 24, 25, 2, tag=3100 + i).float().to(DEV) for i in range(6)]
ls = [torch.tensor([(3 * i + 1) % 60, (7 * i + 2) % 60], device=DEV) for i in range(6)]
def run(graph):
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.0, compute_dtype=dtype).to(DEV)
    m.load_state_dict(sd, strict=True); m.train()
    tr = Trainer(m, lr=1e-3, graph=graph)
    init = {k: v.detach().double().cpu().clone() for k, v in m.state_dict().items()}
    for x, l in zip(xs, ls): tr.step(x, l)
    return {k: v.detach().double().cpu() - init[k] for k, v in m.state_dict().items() if v.is_floating_point() and k != "A"}
a, b, c = run(False), run(False), run(True)
def top(u, v):
    d = sorted(((float((u[k] - v[k]).norm() / (v[k].norm() + 1e-300)), k) for k in u), reverse=True)
    return d[:5]
print("eager vs eager", top(a, b))
print("graph vs eager", top(c, a))
