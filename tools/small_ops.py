"""Which torch operators (and which source lines) are behind the small launches of one training step."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
from net.st_gcn import Model          # noqa: E402
from stgcn_rt.ddp import Trainer      # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5, compute_dtype=torch.bfloat16).to(dev)
m.train()
tr = Trainer(m)
x = torch.randn(8, 3, 300, 25, 2, device=dev)
y = torch.randint(0, 60, (8,), device=dev)
for _ in range(2):
    tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
ops = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::"):
        st = [s for s in (ev.stack or []) if "st-gcn_amd" in s or "bench" in s or "tools" in s]
        ops[(ev.name, st[0] if st else "?")] += 1
    elif ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None:
        st = [s for s in (ev.stack or []) if "st-gcn_amd" in s or "bench" in s or "tools" in s]
        ops[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in sorted(ops.items(), key=lambda kv: -kv[1])[:60]:
    print("%4d  %-28s %s" % (n, name, where[-110:]))
