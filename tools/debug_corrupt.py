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
x = O.det_input(2, 3, 24, 25, 2, tag=900)
labels = torch.tensor([3, 10])
stash = {}
orig_f, orig_b = P.block_out_fwd, P.block_out_bwd
def f(o, res, bn2, bnr, p, seed):
    out = orig_f(o, res, bn2, bnr, p, seed)
    torch.cuda.synchronize()
    stash[out.data_ptr()] = dict(o=o.clone(), out=out.clone(), res=None if res is None else res.clone(),
                                 bn2=[t.clone() for t in (bn2.mean, bn2.rstd, bn2.scale, bn2.shift)],
                                 bnr=None if bnr is None else [t.clone() for t in (bnr.mean, bnr.rstd, bnr.scale, bnr.shift)])
    return out
def b(dout, out, o, r, bn2, g2, bnr, gr, want_g, p, seed, training):
    torch.cuda.synchronize()
    s = stash[out.data_ptr()]
    def chk(name, a, c):
        n = int((a != c).sum())
        if n:
            idx = (a != c).nonzero()[:4].tolist()
            print("CORRUPTED %s shape %s: %d elements differ from forward-time value, first %s" % (name, tuple(a.shape), n, idx))
    chk("out", out, s["out"]); chk("o", o, s["o"])
    if r is not None: chk("r", r, s["res"])
    for nm, a, c in zip(("mean2", "rstd2", "scale2", "shift2"), (bn2.mean, bn2.rstd, bn2.scale, bn2.shift), s["bn2"]): chk(nm, a, c)
    if bnr is not None:
        for nm, a, c in zip(("meanr", "rstdr", "scaler", "shiftr"), (bnr.mean, bnr.rstd, bnr.scale, bnr.shift), s["bnr"]): chk(nm, a, c)
    print("block_out_bwd C=%d rows=%d checked" % (o.shape[1], o.shape[0]*o.shape[2]*o.shape[3]))
    return orig_b(dout, out, o, r, bn2, g2, bnr, gr, want_g, p, seed, training)
P.block_out_fwd, P.block_out_bwd = f, b
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
m.load_state_dict(sd); m.train()
logits = m(x.float().to(dev))
loss = torch.nn.functional.cross_entropy(logits, labels.to(dev))
loss.backward()
torch.cuda.synchronize()
