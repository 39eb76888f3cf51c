import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
import torch
from oracle import stgcn_oracle as O
from net.st_gcn import Model
from stgcn_rt import ops as P
import stgcn_rt.functional as F
dev = "cuda:0"
dtype = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] == "bf16" else torch.float32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
g = O.build_graph("ntu-rgb+d", "spatial")
sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
x = O.det_input(2, 3, T, 25, 2, tag=900)
labels = torch.tensor([3, 10])
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
rec = []
def wrap(name):
    orig = getattr(P, name)
    def f(*a, **k):
        o = orig(*a, **k)
        if name == "block_out_bwd":
            rec.append((name, o[0].detach().clone()))
        elif name == "bn_bwd_apply":
            rec.append((name, o.detach().clone()))
        else:
            rec.append((name, o.detach().clone()))
        return o
    setattr(P, name, f)
for n in ("block_out_bwd", "conv_rows", "bn_bwd_apply", "gcn_fused"):
    wrap(n)
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
m.load_state_dict(sd); m.train()
F.CAPTURE = []
logits = m(x.float().to(dev))
nfwd = len(rec)
fwd = rec[:nfwd]
loss = torch.nn.functional.cross_entropy(logits, labels.to(dev))
loss.backward(); torch.cuda.synchronize()
cap = F.CAPTURE; F.CAPTURE = None
masks = {9 - j: (c["h_mask"].cpu(), c["out_mask"].cpu()) for j, c in enumerate(cap)}
lo, cache, _ = O.model_forward(sd64, x, True, masks=masks)
_, dlog = O.cross_entropy(lo, labels)
N, C, T_, V, M = cache["dims"]; Bc, Cf, Tf, Vf = cache["feat_shape"]
Wf = sd64["fcn.weight"].reshape(60, -1)
dh = ((dlog @ Wf) / (M * Tf * Vf))[:, None, :, None, None].expand(N, M, Cf, Tf, Vf).reshape(Bc, Cf, Tf, Vf)
def l2(a, b):
    a = a.cpu().double(); return ((a - b).norm() / b.norm()).item()
# forward activations of each block: y (gcn_fused), o (conv_rows 9 tap), out
print("logits rel", l2(logits.detach(), lo))
bw = rec[nfwd:]
pos = 0
for i in reversed(range(10)):
    bc = cache["blocks"][i]; p, _ = O._block_view(sd64, i)
    ds = dh * bc["omask"]
    do, _, _ = O.bn_backward(bc["o"], p["tcn.3.weight"], bc["c2"], ds, True)
    dhh, _, _ = O.tconv_backward(bc["h"], p["tcn.2.weight"][..., 0], do, bc["stride"])
    dh1 = dhh * bc["hmask"]
    dy, _, _ = O.bn_backward(bc["y"], p["tcn.0.weight"], bc["c1"], dh1, True)
    dx, dA, gb = O.block_backward(dh, p, bc, bc["stride"], bc["kind"], True)
    # GPU records for this block: block_out_bwd, conv_rows(gh), bn_bwd_apply(dy), [conv_rows(dxr)], gcn_fused(dx)
    assert bw[pos][0] == "block_out_bwd"
    e_dO = l2(bw[pos][1], do); e_gh = l2(bw[pos+1][1], dh1); e_dy = l2(bw[pos+2][1], dy)
    pos += 3
    if bc["kind"] == "conv": pos += 1
    e_dx = l2(bw[pos][1], dx); pos += 1
    print("block %d: dO %.4f  gh %.4f  dy %.4f  dx %.4f | norms dout %.2e dO %.2e (ratio %.3f)  gh %.2e dy %.2e (ratio %.3f)" % (
        i, e_dO, e_gh, e_dy, e_dx, ds.norm().item(), do.norm().item()/p["tcn.3.weight"].abs().mean().item()/bc["c2"][1].mean().item(), do.norm().item()/ds.norm().item(),
        dh1.norm().item(), dy.norm().item(), dy.norm().item()/dh1.norm().item()))
    dh = dx
