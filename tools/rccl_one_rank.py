#!/usr/bin/env python
"""The data-parallel trainer over RCCL in a one-rank group (what a box with a single GPU can run of the multi-GPU
path): process-group set-up as bench.py does it, broadcast from rank 0, asynchronous bucket all-reduces launched from
the backward hooks and waited for before the update, the all-reduce of the whole flat gradient after a HIP-graph
replay.  A one-rank sum is the identity and training repeats bit for bit, so after the same steps the parameters must
equal those of the trainer that never touches a collective -- bit for bit.

Launch:  STGCN_DDP_FORCE=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 \
             --master-port 29544 tools/rccl_one_rank.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("STGCN_WATCHDOG", "140")), exit=True)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    print("RCCL_ONE_RANK_GROUP_UP", flush=True)         # from here on a failure is the trainer's, not the rendezvous'

    from oracle import stgcn_oracle as O
    import net.st_gcn as mod
    from stgcn_rt.ddp import Trainer

    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    x = O.det_input(4, 3, 48, 25, 2, tag=6100).float().to(dev)
    labels = torch.tensor([3, 17, 41, 59], device=dev)

    def run(force, graph, dtype):
        os.environ["STGCN_DDP_FORCE"] = "1" if force else "0"
        torch.manual_seed(7)
        m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5, compute_dtype=dtype).to(dev)
        m.load_state_dict(sd, strict=True)
        m.train()
        tr = Trainer(m, lr=0.01, bucket_bytes=1 << 20, graph=graph)
        assert tr.bucketer.distributed == force
        tr.broadcast_from_rank0()
        losses, by_hooks = [], None
        for i in range(6):
            losses.append(float(tr.step(x, labels).detach()))
            if i == 0:
                by_hooks = list(tr.bucketer.launched)           # the first step is always launch by launch
        torch.cuda.synchronize()
        assert tr.graph_ready() == graph
        assert tr.graph_segments() == (2 if (graph and force) else 1 if graph else 0), tr.graph_segments()
        return tr.flat.flat.detach().clone(), losses, by_hooks, len(tr.bucketer.buckets)

    for dtype in (torch.float32, torch.bfloat16):
        ref, l_ref, _, _ = run(False, False, dtype)
        for graph in (False, True):
            got, l_got, by_hooks, nb = run(True, graph, dtype)
            assert nb >= 3 and by_hooks == list(range(nb)), (by_hooks, nb)     # every bucket left from a backward hook, in order
            assert l_got == l_ref, (l_got, l_ref)
            assert torch.equal(got, ref), float((got - ref).abs().max())
            print("%s graph=%s: %d buckets all-reduced over RCCL from the backward hooks%s; losses and parameters equal the "
                  "collective-free trainer bit for bit" % (str(dtype).replace("torch.", ""), graph, nb,
                  " in the two launch-by-launch steps, then the step replayed as two graphs with the late blocks' all-reduce between them" if graph else ""))
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")


if __name__ == "__main__":
    main()
