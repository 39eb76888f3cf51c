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
ref_g = O.model_backward(sd64, cache, dlog, True)
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, compute_dtype=dtype).to(dev)
m.load_state_dict(sd); m.train()
logits = m(x.float().to(dev))
loss = torch.nn.functional.cross_entropy(logits, labels.to(dev))
loss.backward()
print("logits err", ((logits.detach().cpu().double()-ref_logits).abs().max()/ref_logits.abs().max()).item())
for k, prm in m.named_parameters():
    gr = prm.grad.detach().cpu().double()
    r = ref_g[k]
    print("%-45s relL2 %.3e  relmax %.3e  norm %.3e" % (k, ((gr-r).norm()/(r.norm()+1e-30)).item(), ((gr-r).abs().max()/(r.abs().max()+1e-30)).item(), r.norm().item()))
