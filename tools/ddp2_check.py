#!/usr/bin/env python
"""The real ddp.Trainer with world size 2 (replaces nn.DataParallel, processor/io.py:86-87 of the reference).

Launch:  STGCN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
             --master-port 29533 tools/ddp2_check.py
Both ranks may share one GPU (a one-GPU box): rank r uses device r % device_count.  Backend "nccl" (RCCL) when every
rank has its own GPU, gloo otherwise.  Checks, per rank:
  1. logits of the rank's shard against the CPU oracle on that shard (BatchNorm statistics are per replica);
  2. the averaged gradient (all-reduced sum x 1/world) against the MEAN of the per-shard oracle gradients;
  3. buckets were launched from the backward hooks, the first of them before the first block's gradient existed;
  4. after the optimizer step the replicas are bit-identical;
  5. the same through the HIP-graph path (graph=True: replay, then all buckets reduced, then the update), where ranks
     must also draw DIFFERENT dropout masks.
The ranks check 3-5 themselves (rank 0 prints DDP2_RANKS_OK) and write what 1 and 2 need to $STGCN_DDP2_OUT;
`python tools/ddp2_check.py --verify DIR` then evaluates the oracle in one process and prints DDP2_OK.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist


def gather_cpu(t, world, backend, dev):
    """Every rank's copy of the CPU tensor t (a list, rank order)."""
    src = t.contiguous() if backend == "gloo" else t.to(dev)
    out = [torch.zeros_like(src) for _ in range(world)]
    dist.all_gather(out, src)
    return [o.cpu() for o in out]


def rl2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("STGCN_WATCHDOG", "140")), exit=True)    # a stall names its line
    import time
    t00 = time.time()

    def stamp(what):
        if os.environ.get("STGCN_DDP2_VERBOSE"):
            print("[rank %s %.1fs] %s" % (os.environ["RANK"], time.time() - t00, what), flush=True)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    assert ndev >= 1, "needs a GPU"
    dev = torch.device("cuda", rank % ndev)
    torch.cuda.set_device(dev)
    backend = os.environ.get("STGCN_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)

    from oracle import stgcn_oracle as O
    import net.st_gcn as mod
    import stgcn_rt.functional as F
    from stgcn_rt.ddp import Trainer

    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    names = O.param_names(sd64)
    x, labels = shard(rank)                                      # every rank its own shard of the global batch
    out_dir = os.environ.get("STGCN_DDP2_OUT", "/tmp")

    m = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.0).to(dev)
    if rank == 0:
        m.load_state_dict(sd, strict=True)                        # the other ranks get it by broadcast
    m.train()
    tr = Trainer(m, lr=0.1, bucket_bytes=1 << 20)
    tr.broadcast_from_rank0()
    assert len(tr.bucketer.buckets) >= 3
    stamp("trainer built, parameters broadcast")

    # one step, spelled out so that logits and the reduced gradient can be looked at
    tr.flat.zero_grad()
    tr.bucketer.reset()
    F.CAPTURE = []
    logits = m(x.float().to(dev))
    loss = tr.loss_fn(logits, labels.to(dev))
    loss.backward()
    cap, F.CAPTURE = F.CAPTURE, None
    trace = list(tr.bucketer.trace)
    by_hooks = list(tr.bucketer.launched)
    stamp("backward issued")
    scale = tr.bucketer.finish()
    torch.cuda.synchronize()
    stamp("buckets reduced")
    masks = {len(cap) - 1 - j: (c["h_mask"].cpu(), c["out_mask"].cpu()) for j, c in enumerate(cap)}
    order = {id(p): k for k, p in m.named_parameters()}
    got = {order[id(p)]: (tr.flat.grad[o:o + n] * scale).cpu().reshape(p.shape)
           for p, o, n in zip(tr.flat.params, tr.flat.offsets, tr.flat.sizes)}
    # the oracle side of checks 1 and 2 runs in verify(), in ONE process after the ranks have ended (float64 CPU work
    # inside rank processes that share a GPU crawls on the test boxes)
    torch.save(dict(logits=logits.detach().cpu(), masks=masks, avg_grad=got if rank == 0 else None),
               os.path.join(out_dir, "rank%d.pt" % rank))
    stamp("results written")
    # overlap: hooks launched buckets in order, the first one before block 0's gradients existed
    assert by_hooks == sorted(by_hooks) and len(by_hooks) >= 2, by_hooks
    first_block = [i for i, p in enumerate(tr.flat.params) if p is m.st_gcn_networks[0].gcn.conv.weight][0]
    assert trace.index(("launch", 0)) < trace.index(("grad", first_block))
    tr.opt.step(scale)
    torch.cuda.synchronize()

    def replicas_identical(tag):
        mineb = tr.flat.flat.detach().cpu()
        allb = gather_cpu(mineb, world, backend, dev)
        for r in range(world):
            assert torch.equal(allb[r], allb[0]), (tag, r)
    replicas_identical("after the launch-by-launch step")
    if rank == 0:
        print("launch-by-launch: buckets launched by hooks %s of %d, replicas bit-identical after the update" % (by_hooks, len(tr.bucketer.buckets)))

    # the HIP-graph path with dropout: replicas stay identical, ranks draw different masks
    m2 = mod.Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5).to(dev)
    m2.load_state_dict(sd, strict=True)
    m2.train()
    tr = Trainer(m2, lr=0.01, graph=True)
    tr.broadcast_from_rank0()
    xs = O.det_input(2, 3, 24, 25, 2, tag=6000).float().to(dev)   # the SAME clips on every rank here
    ls = torch.tensor([7, 21], device=dev)
    losses = [float(tr.step(xs, ls).detach()) for _ in range(5)]
    torch.cuda.synchronize()
    assert tr.graph_ready() and tr.graph_segments() == 2      # two graphs, the first all-reduce issued between them
    replicas_identical("after graph replays")
    lt = torch.tensor(losses, dtype=torch.float64)
    alll = gather_cpu(lt, world, backend, dev)
    assert float((alll[0] - alll[1]).abs().min()) > 1e-6, ("same clips, same weights: losses differ only by the masks", alll)
    dist.barrier()
    if rank == 0:
        print("graph replay (two graphs per step, the late blocks' all-reduce between them): 5 steps, replicas bit-identical, per-rank losses", [[round(float(v), 4) for v in a] for a in alll])
        print("DDP2_RANKS_OK backend=%s world=%d" % (backend, world))
    dist.destroy_process_group()


def shard(rank):
    from oracle import stgcn_oracle as O
    return O.det_input(2, 3, 24, 25, 2, tag=5000 + rank), torch.tensor([(9 * rank + 4) % 60, (5 * rank + 31) % 60])


def verify(out_dir, world=2):
    """Checks 1 and 2 from the files the ranks wrote: per-shard logits against the oracle on that shard, and the
    averaged gradient against the mean of the per-shard oracle gradients (each evaluated with that rank's ReLU
    decisions and its own d loss / d logits)."""
    from oracle import stgcn_oracle as O
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    names = O.param_names(sd64)
    mean_ref, avg = None, None
    for r in range(world):
        d = torch.load(os.path.join(out_dir, "rank%d.pt" % r), weights_only=True)
        x, labels = shard(r)
        lo, cache, _ = O.model_forward(sd64, x, True, masks=d["masks"])
        e_log = float((d["logits"].double() - lo).abs().max() / lo.abs().max())
        assert e_log < 1e-4, ("logits of shard %d" % r, e_log)
        _, dlog = O.cross_entropy(d["logits"].double(), labels)
        gr = O.model_backward(sd64, cache, dlog, True)
        mean_ref = gr if mean_ref is None else {k: mean_ref[k] + gr[k] for k in names}
        if d["avg_grad"] is not None:
            avg = d["avg_grad"]
    worst = ("", 0.0)
    for k in names:
        if not (k.endswith("tcn.2.bias") or k.endswith("residual.0.bias")):
            worst = max(worst, (k, rl2(avg[k], mean_ref[k] / world)), key=lambda t: t[1])
    assert worst[1] < 1e-4, ("averaged gradient vs mean of per-shard oracle gradients", worst)
    print("verify: per-shard logits within 1e-4 of the oracle; averaged gradient vs mean of per-shard oracle gradients: worst %s %.2e" % worst)
    print("DDP2_OK")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--verify":
        verify(sys.argv[2])
    else:
        main()
