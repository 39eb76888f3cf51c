"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce and flat-buffer bookkeeping that the
multi-GPU path uses (stgcn_rt.ddp), exercised with a stand-in CPU network (the ST-GCN kernels themselves
have no CPU path).  Checks: flat views, bucket partition in reverse execution order, hook-driven launches,
averaged gradients == mean of the per-rank gradients, unused parameters, identical replicas after the step."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _AllImportances(torch.autograd.Function):
    """One autograd node for every block's importance, as functional.EdgeImportanceFn is on the real model: all its
    gradients appear together, after the first block's backward."""

    @staticmethod
    def forward(ctx, *imps):
        return tuple(p.detach().sum() for p in imps)

    @staticmethod
    def backward(ctx, *grads):
        return tuple(torch.ones(3) * (0.0 if g is None else g) for g in grads)


class TinyNet(torch.nn.Module):
    """fcn / st_gcn_networks / data_bn attribute names as in the real model so the ordering helper applies."""

    def __init__(self):
        super().__init__()
        self.data_bn = torch.nn.BatchNorm1d(6)
        self.st_gcn_networks = torch.nn.ModuleList([torch.nn.Linear(6, 32), torch.nn.Linear(32, 32), torch.nn.Linear(32, 16)])
        self.edge_importance = torch.nn.ParameterList([torch.nn.Parameter(torch.ones(3)) for _ in range(3)])
        self.fcn = torch.nn.Linear(16, 5)
        self.unused = torch.nn.Parameter(torch.ones(7))

    def forward(self, x):
        h = self.data_bn(x)
        sums = _AllImportances.apply(*self.edge_importance)
        for lin, imp in zip(self.st_gcn_networks, sums):
            h = torch.relu(lin(h)) * imp
        return self.fcn(h)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stgcn_rt.ddp import FlatParams, GradBucketer, reverse_execution_order
    torch.manual_seed(0)
    net = TinyNet()
    flat = FlatParams(net, reverse_execution_order(net))
    # views: parameters alias the flat buffer; gradient slots alias the flat gradient, .grad starts empty
    for p, v, o, n in zip(flat.params, flat.views, flat.offsets, flat.sizes):
        assert p.data_ptr() == flat.flat.data_ptr() + 4 * o and v.data_ptr() == flat.grad.data_ptr() + 4 * o
        assert p.grad is None and v.shape == p.shape
    # reverse execution order: fcn first, data_bn last (unused trailing)
    assert flat.params[0] is net.fcn.weight and flat.params[-1] is net.unused
    bk = GradBucketer(flat, bucket_bytes=2048)
    assert len(bk.buckets) >= 3 and bk.buckets[0][0] == 0 and bk.buckets[-1][1] == flat.flat.numel()
    dist.broadcast(flat.flat, src=0)
    torch.manual_seed(100 + rank)
    x = torch.randn(8, 6)
    y = torch.randint(0, 5, (8,))
    flat.zero_grad()
    bk.reset()
    loss = torch.nn.functional.cross_entropy(net(x), y)
    loss.backward()
    launched_by_hooks = list(bk.launched)
    local = None
    scale = bk.finish()
    assert scale == 1.0 / world
    # hooks launched buckets in increasing (= reverse execution) order; the bucket holding only-unused params came from finish()
    assert launched_by_hooks == sorted(launched_by_hooks) and len(launched_by_hooks) >= 2
    assert sorted(bk.launched) == list(range(len(bk.buckets)))
    # overlap: the first bucket is on its way before the first block's gradients (the end of backward) exist, and
    # the importances -- one autograd node, ready last -- sit behind every block in the flat order
    first_block = [i for i, p in enumerate(flat.params) if p is net.st_gcn_networks[0].weight][0]
    assert bk.trace.index(("launch", 0)) < bk.trace.index(("grad", first_block))
    ei_idx = [i for i, p in enumerate(flat.params) if any(p is q for q in net.edge_importance)]
    assert min(ei_idx) > first_block and max(ei_idx) < [i for i, p in enumerate(flat.params) if p is net.data_bn.weight][0]
    avg = flat.grad * scale
    # reference: every rank recomputes both ranks' gradients serially
    ref = torch.zeros_like(flat.grad)
    for r in range(world):
        torch.manual_seed(0)
        net2 = TinyNet()
        f2 = FlatParams(net2, reverse_execution_order(net2))
        f2.flat.copy_(flat.flat)
        torch.manual_seed(100 + r)
        xr, yr = torch.randn(8, 6), torch.randint(0, 5, (8,))
        torch.nn.functional.cross_entropy(net2(xr), yr).backward()
        f2.gather(range(len(f2.params)))
        ref += f2.grad / world
    err = (avg - ref).abs().max().item()
    assert err < 1e-6, err
    assert avg[flat.offsets[-2]:].abs().max().item() == 0.0        # the unused parameter
    # a plain SGD step on the averaged gradient leaves both replicas identical
    flat.flat -= 0.1 * avg
    gathered = [torch.zeros_like(flat.flat) for _ in range(world)]
    dist.all_gather(gathered, flat.flat)
    assert torch.equal(gathered[0], gathered[1])
    if rank == 0:
        torch.save({"ok": True, "buckets": len(bk.buckets), "err": err}, out)
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["ok"] and res["buckets"] >= 3


def test_world1_is_a_noop():
    from stgcn_rt.ddp import FlatParams, GradBucketer, reverse_execution_order, adjust_lr
    net = TinyNet()
    flat = FlatParams(net, reverse_execution_order(net))
    bk = GradBucketer(flat)
    loss = net(torch.randn(4, 6)).sum()
    loss.backward()
    assert bk.finish() == 1.0 and bk.handles == []
    for prm, v in zip(flat.params, flat.views):     # finish() gathered every gradient into the flat buffer
        assert torch.equal(v, prm.grad if prm.grad is not None else torch.zeros_like(v))
    assert adjust_lr(0.1, 0, [10, 50]) == 0.1 and abs(adjust_lr(0.1, 10, [10, 50]) - 0.01) < 1e-12 and abs(adjust_lr(0.1, 50, [10, 50]) - 0.001) < 1e-12
