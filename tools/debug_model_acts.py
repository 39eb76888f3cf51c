import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
import torch
from oracle import stgcn_oracle as O
from net.st_gcn import Model
dev = "cuda:0"
dtype = torch.float32 if len(sys.argv) < 2 or sys.argv[1] == "f32" else torch.bfloat16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
g = O.build_graph("ntu-rgb+d", "spatial")
sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
x = O.det_input(2, 3, T, 25, 2, tag=900)
labels = torch.tensor([3, 10])
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ref_logits, cache, _ = O.model_forward(sd64, x, True)
ref_loss, dlog = O.cross_entropy(ref_logits, labels)
# oracle backward with recorded activations' grads
N, C, T_, V, M = cache["dims"]; Bc, Cf, Tf, Vf = cache["feat_shape"]
Wf = sd64["fcn.weight"].reshape(60, -1)
dh = ((dlog @ Wf) / (M * Tf * Vf))[:, None, :, None, None].expand(N, M, Cf, Tf, Vf).reshape(Bc, Cf, Tf, Vf)
ref_dout = {}
for i in reversed(range(10)):
    ref_dout[i] = dh
    bc = cache["blocks"][i]
    p, _ = O._block_view(sd64, i)
    dh, dA, gb = O.block_backward(dh, p, bc, bc["stride"], bc["kind"], True)
ref_dout[-1] = dh
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
m.load_state_dict(sd); m.train()
got, outs = {}, {}
def mk(i):
    def hook(mod, inp, out):
        outs[i] = out[0]
        out[0].register_hook(lambda gr: got.__setitem__(i, gr.detach().clone()))
    return hook
for i, blk in enumerate(m.st_gcn_networks):
    blk.register_forward_hook(mk(i))
logits = m(x.float().to(dev))
loss = torch.nn.functional.cross_entropy(logits, labels.to(dev))
loss.backward()
for i in range(10):
    o = outs[i].detach().cpu().double(); r = cache["blocks"][i]["out"]
    gr = got[i].cpu().double(); rg = ref_dout[i]
    d = (gr - rg)
    print("block %d out relmax %.2e | dout relL2 %.2e relmax %.2e | err mean/|err| %.3f  nbad(>1e-3 max) %d of %d" % (
        i, ((o-r).abs().max()/r.abs().max()).item(), (d.norm()/rg.norm()).item(), (d.abs().max()/rg.abs().max()).item(),
        (d.mean()/ (d.abs().mean()+1e-30)).item(), int((d.abs() > 1e-3*rg.abs().max()).sum()), d.numel()))
    if (d.abs().max()/rg.abs().max()).item() > 1e-3:
        idx = (d.abs() > 1e-3*rg.abs().max()).nonzero()[:12]
        print("   bad idx (b,c,t,v):", idx.tolist())
        for b_, c_, t_, v_ in idx.tolist()[:6]:
            print("      got %.6e ref %.6e  out_here(next block input) ref %.6e" % (gr[b_,c_,t_,v_], rg[b_,c_,t_,v_], r[b_,c_,t_,v_]))
print("---- error pattern of grad wrt block-6 output (dx of block 7)")
gr = got[6].cpu().double(); rg = ref_dout[6]
bad = ((gr - rg).abs() > 1e-4 * rg.abs().max())
print("total bad", int(bad.sum()))
print("by b", bad.sum(dim=(1,2,3)).tolist())
print("by t", bad.sum(dim=(0,1,3)).tolist())
print("by v", bad.sum(dim=(0,1,2)).tolist())
print("by c//32", bad.reshape(bad.shape[0], -1, 32, bad.shape[2], bad.shape[3]).sum(dim=(0,2,3,4)).tolist())
bt = bad.sum(dim=(1,3))
print("by (b,t):"); print(bt)
