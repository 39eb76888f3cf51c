import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
import torch
from oracle import stgcn_oracle as O
from net.st_gcn import Model
from stgcn_rt import ops as P
dev = "cuda:0"
dtype = torch.float32
g = O.build_graph("ntu-rgb+d", "spatial")
sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
x = O.det_input(2, 3, 24, 25, 2, tag=900).float().to(dev)
labels = torch.tensor([3, 10]).to(dev)
rec = []
names = ("gcn_fused", "conv_rows", "block_out_fwd", "block_out_bwd", "bn_bwd_apply", "bn_finalize")
for n in names:
    def mk(n, orig):
        def f(*a, **k):
            o = orig(*a, **k)
            if n == "bn_finalize":
                rec.append((n, torch.stack([o.mean, o.rstd, o.scale, o.shift]).clone()))
            elif n == "block_out_bwd":
                rec.append((n + ".dO", o[0].clone()))
                rec.append((n + ".stats2", o[3].clone()))
            else:
                rec.append((n, o.clone()))
            return o
        return f
    setattr(P, n, mk(n, getattr(P, n)))
ref = None
for it in range(10):
    rec.clear()
    m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
    m.load_state_dict(sd); m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    cur = list(rec)
    if ref is None:
        ref = cur; continue
    msg = []
    for i, ((n, a), (n2, b)) in enumerate(zip(cur, ref)):
        d = (a.double() - b.double()).abs().max().item()
        rel = d / (b.double().abs().max().item() + 1e-30)
        if rel > 1e-5:
            nz = int(((a > 0) != (b > 0)).sum())
            msg.append("#%d %s %s rel %.1e signflips %d nbad %d" % (i, n, tuple(a.shape), rel, nz, int(((a.double()-b.double()).abs() > 1e-5*b.double().abs().max()).sum())))
    print("iter", it, "first deviating records:", msg[:4])
