import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
import torch
from oracle import stgcn_oracle as O
from net.st_gcn import Model
from stgcn_rt import ops as P
dev = "cuda:0"
g = O.build_graph("ntu-rgb+d", "spatial")
sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
x = O.det_input(2, 3, 24, 25, 2, tag=900).float().to(dev)
labels = torch.tensor([3, 10]).to(dev)
rec = {}
orig = P.block_out_bwd
def f(dout, out, o, r, bn2, g2, bnr, gr, want_g, p, seed, training):
    res = orig(dout, out, o, r, bn2, g2, bnr, gr, want_g, p, seed, training)
    if tuple(o.shape) == (4, 256, 6, 25) and r is not None:
        rec["in"] = [t.clone() for t in (dout, out, o, r, bn2.mean, bn2.rstd, bnr.mean, bnr.rstd, g2, gr)]
        rec["out"] = [res[0].clone(), res[1].clone(), res[3].clone(), res[4].clone()]
        rec["ptrs"] = [t.data_ptr() for t in (dout, out, o, r, res[3], res[0], res[1])]
    return res
P.block_out_bwd = f
ref = None
for it in range(10):
    m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True).to(dev)
    m.load_state_dict(sd); m.train()
    loss = torch.nn.functional.cross_entropy(m(x), labels); loss.backward(); torch.cuda.synchronize()
    cur = dict(rec)
    if ref is None:
        ref = cur; continue
    din = [int((a != b).sum()) for a, b in zip(cur["in"], ref["in"])]
    dou = [((a.double()-b.double()).abs().max()/b.double().abs().max()).item() for a, b in zip(cur["out"], ref["out"])]
    s2 = (cur["out"][2] - ref["out"][2])
    print("iter", it, "input mismatches", din, "output rel dev", ["%.1e" % v for v in dou], "stats2 bad idx", (s2.abs() > 1e-9).nonzero().tolist()[:4],
          "ptr offsets", [hex(p & 0xFFFFF) for p in cur["ptrs"]])
