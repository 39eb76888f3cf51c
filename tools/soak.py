"""Soak run of the training step at the benchmarked size: many steps on one fixed synthetic batch, printing the loss
and the allocator's footprint along the way (a leak or a drift into non-finite values shows here, not in bench.py).

    python tools/soak.py [bf16_steps] [fp32_steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "st-gcn_amd"))
from net import st_gcn                      # noqa: E402
from stgcn_rt.ddp import Trainer            # noqa: E402

dev = "cuda:0"
steps = {torch.bfloat16: int(sys.argv[1]) if len(sys.argv) > 1 else 300, torch.float32: int(sys.argv[2]) if len(sys.argv) > 2 else 60}


def weights_init(mod):                      # processor/recognition.py:21-33
    cn = mod.__class__.__name__
    if cn.find("Conv1d") != -1 or cn.find("Conv2d") != -1:
        mod.weight.data.normal_(0.0, 0.02)
        if mod.bias is not None:
            mod.bias.data.fill_(0)
    elif cn.find("BatchNorm") != -1:
        mod.weight.data.normal_(1.0, 0.02)
        mod.bias.data.fill_(0)


for dtype, n in steps.items():
    if n <= 0:
        continue
    torch.manual_seed(0)
    m = st_gcn.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5, compute_dtype=dtype).to(dev)
    m.apply(weights_init)
    m.train()
    tr = Trainer(m, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 3, 300, 25, 2, generator=g).to(dev)
    y = torch.randint(0, 60, (64,), generator=g).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        loss = tr.step(x, y)
        if i % max(1, n // 6) == 0 or i == n - 1:
            print(dtype, "step", i, "loss %.4f" % loss.item(), "allocated %.2f GB reserved %.2f GB"
                  % (torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    torch.cuda.synchronize()
    print(dtype, "%.2f ms/step over %d steps (with the loss read-backs)" % ((time.perf_counter() - t0) / n * 1e3, n),
          "parameters finite:", all(torch.isfinite(p).all().item() for p in m.parameters()), flush=True)
    del m, tr
