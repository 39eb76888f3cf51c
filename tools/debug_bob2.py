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
cap = {}
orig = P.block_out_bwd
def f(dout, out, o, r, bn2, g2, bnr, gr, want_g, p, seed, training):
    if tuple(o.shape) == (4, 256, 6, 25) and r is not None and "a" not in cap:
        cap["a"] = (dout.clone(), out.clone(), o.clone(), r.clone(), bn2, g2.clone(), bnr, gr.clone())
    return orig(dout, out, o, r, bn2, g2, bnr, gr, want_g, p, seed, training)
P.block_out_bwd = f
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True).to(dev)
m.load_state_dict(sd); m.train()
loss = torch.nn.functional.cross_entropy(m(x), labels); loss.backward(); torch.cuda.synchronize()
dout, out, o, r, bn2, g2, bnr, gr = cap["a"]
gd = (dout.double() * (out > 0))
oh = (o.double() - bn2.mean.double()[None, :, None, None]) * bn2.rstd.double()[None, :, None, None]
ref = torch.stack([gd.sum(dim=(0, 2, 3)), (gd * oh).sum(dim=(0, 2, 3))])
print("nan/inf in inputs:", [bool(~torch.isfinite(t).all()) for t in (dout, out, o, r)])
states = {}
for it in range(200):
    dO, dR, gg, s2, sr, dbo, dbr = orig(dout, out, o, r, bn2, g2, bnr, gr, False, 0.0, 1, True)
    torch.cuda.synchronize()
    d = (s2 - ref)
    bad = (d.abs() > 1e-6 * ref.abs().max()).nonzero().tolist()
    key = str(bad)
    if key not in states:
        states[key] = 0
        print("iter", it, "bad entries", bad, [ (d[i, j].item(), ref[i, j].item()) for i, j in bad][:4])
        for i, j in bad[:1]:
            col = gd[:, j].reshape(-1)
            # does the deviation equal one (or several) element contributions?
            dv = d[0, j].item() if i == 0 else None
            if dv is not None:
                k = (col + dv).abs().argmin().item(); k2 = (col - dv).abs().argmin().item()
                print("   dev %.6e ; closest -elem %.6e (row %d), closest +elem %.6e (row %d)" % (dv, col[k].item(), k, col[k2].item(), k2))
    states[key] += 1
print(states)
