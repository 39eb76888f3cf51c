import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
import torch
from oracle import stgcn_oracle as O
from net.st_gcn import Model
from stgcn_rt import ops as P
import stgcn_rt.functional as F
dev = "cuda:0"
dtype = torch.float32
g = O.build_graph("ntu-rgb+d", "spatial")
sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
x = O.det_input(2, 3, 24, 25, 2, tag=900)
labels = torch.tensor([3, 10])
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ref_logits, cache, _ = O.model_forward(sd64, x, True)
ref_loss, dlog = O.cross_entropy(ref_logits, labels)
N, C, T_, V, M = cache["dims"]; Bc, Cf, Tf, Vf = cache["feat_shape"]
Wf = sd64["fcn.weight"].reshape(60, -1)
dh = ((dlog @ Wf) / (M * Tf * Vf))[:, None, :, None, None].expand(N, M, Cf, Tf, Vf).reshape(Bc, Cf, Tf, Vf)
ref = {}
for i in reversed(range(10)):
    bc = cache["blocks"][i]; p, _ = O._block_view(sd64, i)
    if i == 7:
        dout = dh
        ds = dout * (bc["out"] > 0)
        do, _, _ = O.bn_backward(bc["o"], p["tcn.3.weight"], bc["c2"], ds, True)
        hh = torch.clamp_min(bc["h1"], 0)
        dhh, _, _ = O.tconv_backward(hh, p["tcn.2.weight"][..., 0], do, 2)
        dh1 = dhh * (bc["h1"] > 0)
        dy, _, _ = O.bn_backward(bc["y"], p["tcn.0.weight"], bc["c1"], dh1, True)
        dr, _, _ = O.bn_backward(bc["r"], p["residual.1.weight"], bc["cr"], ds, True)
        dxr, _, _ = O.tconv_backward(bc["x"], p["residual.0.weight"][..., 0], dr, 2, pad=0)
        ref = dict(dout=dout, dO=do, dR=dr, gh=dh1, dy=dy, dxr=dxr)
    dh, dA, gb = O.block_backward(dh, p, bc, bc["stride"], bc["kind"], True)
    if i == 7:
        ref["dx"] = dh
rec = []
def wrap(name):
    orig = getattr(P, name)
    def f(*a, **k):
        o = orig(*a, **k)
        torch.cuda.synchronize()
        if name == "block_out_bwd":
            rec.append((name, a[0].detach().clone(), o[0].detach().clone(), None if o[1] is None else o[1].detach().clone()))
        else:
            rec.append((name, o.detach().clone()))
        return o
    setattr(P, name, f)
for n in ("block_out_bwd", "conv_rows", "bn_bwd_apply", "gcn_fused"):
    wrap(n)
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
m.load_state_dict(sd); m.train()
logits = m(x.float().to(dev))
nfwd = len(rec)
loss = torch.nn.functional.cross_entropy(logits, labels.to(dev))
loss.backward()
torch.cuda.synchronize()
bw = rec[nfwd:]
# find block 7's backward: block_out_bwd with dR not None and dO shape [4,256,6,25]
idx = [i for i, r in enumerate(bw) if r[0] == "block_out_bwd" and r[3] is not None and tuple(r[2].shape) == (4, 256, 6, 25)][0]
def cmp(name, t, r):
    t = t.cpu().double(); d = (t - r).abs()
    bad = d > 1e-4 * r.abs().max()
    print("%-6s relmax %.2e  nbad %d" % (name, (d.max()/r.abs().max()).item(), int(bad.sum())), "first bad idx", bad.nonzero()[:3].tolist())
cmp("dout", bw[idx][1], ref["dout"]); cmp("dO", bw[idx][2], ref["dO"]); cmp("dR", bw[idx][3], ref["dR"])
print([ (r[0], tuple(r[1].shape)) for r in bw[idx+1: idx+6]])
cmp("gh", bw[idx+1][1], ref["gh"]) if False else None
# order in backward: conv_rows (gh, RELU_BWD) -> bn_bwd_apply (dy, in place on gh) -> conv_rows (dxr) -> gcn_fused (dx)
cmp("dy(gh buffer after apply)", bw[idx+1][1], ref["dy"])
cmp("dy", bw[idx+2][1], ref["dy"]); cmp("dxr", bw[idx+3][1], ref["dxr"]); cmp("dx", bw[idx+4][1], ref["dx"])
