import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
import torch
from oracle import stgcn_oracle as O
from net.st_gcn import Model
dev = "cuda:0"
dtype = torch.float32
g = O.build_graph("ntu-rgb+d", "spatial")
sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
x = O.det_input(2, 3, 24, 25, 2, tag=900).float().to(dev)
labels = torch.tensor([3, 10]).to(dev)
sync = len(sys.argv) > 1 and sys.argv[1] == "sync"
ref = None
for it in range(12):
    m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
    m.load_state_dict(sd); m.train()
    logits = m(x)
    if sync: torch.cuda.synchronize()
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    cur = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    if ref is None:
        ref = cur
        continue
    worst = max(((cur[k]-ref[k]).norm()/(ref[k].norm()+1e-30)).item() for k in cur if not (k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")))
    first = [k for k in cur if not (k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")) and ((cur[k]-ref[k]).norm()/(ref[k].norm()+1e-30)).item() > 1e-5]
    print("iter %d worst relL2 vs iter0: %.2e   deviating tensors: %d  last-in-order: %s" % (it, worst, len(first), first[-1] if first else None))
