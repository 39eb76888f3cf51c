"""Data-parallel training support: flat parameter / gradient buffers, bucketed gradient
all-reduce overlapped with backward, and the fused SGD update.

Replaces the reference's nn.DataParallel wrap (processor/io.py:86-87 of hensyou/st-gcn) with
one process per GPU: every rank holds a full replica, runs the model on its own shard of the
batch (BatchNorm statistics stay per replica, as under DataParallel), and gradients are
averaged with a few large all-reduces over RCCL/xGMI.  Buckets follow reverse execution order
(fcn, block 9 ... block 0, the edge importances, data_bn) and each is launched as soon as its last gradient has been
accumulated, so communication overlaps the rest of backward.  The optimizer step is one HIP
kernel over the flat buffers (torch.optim.SGD semantics: processor/recognition.py:46-53).
"""
import os

import torch
import torch.distributed as dist


class _Staged:
    """Handle of a collective on a device tensor carried out on a host copy (see _host_staged)."""

    def __init__(self, work, host, dst):
        self.work, self.host, self.dst = work, host, dst

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.host)


def _host_staged(group):
    """Whether collectives on device tensors are carried out on host copies: with the gloo backend (rehearsals of the
    multi-process path on a box with fewer GPUs than ranks) the device tensors never reach the collective library;
    RCCL ("nccl") takes them as they are."""
    return dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    """Asynchronous sum all-reduce of tensor t in place; returns a handle with wait()."""
    if t.is_cuda and _host_staged(group):
        host = t.detach().cpu()                                   # waits for the stream that produced t
        return _Staged(dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group, async_op=True), host, t)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def broadcast0(t, group=None):
    if t.is_cuda and _host_staged(group):
        host = t.detach().cpu()
        dist.broadcast(host, src=0, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=0, group=group)


class FlatParams:
    """Re-homes every parameter of `module` (and its .grad) as a view into one flat float32 buffer."""

    def __init__(self, module, order=None):
        params = [p for p in module.parameters() if p.requires_grad]
        if order is not None:
            params = order(params)
        self.params = params
        self.module = module
        self.sizes = [p.numel() for p in params]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        total = self.offsets[-1]
        dev = params[0].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.mom = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o, n in zip(params, self.offsets, self.sizes):
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = None
        self.steps = 0

        self.views = [self.grad[o:o + n].view(p.shape) for p, o, n in zip(params, self.offsets, self.sizes)]

    def zero_grad(self):
        """Drops every .grad: autograd then hands each gradient over without an accumulation kernel, and
        gather() moves a whole bucket of them into the flat buffer with one multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def gather(self, idxs):
        """flat.grad[slot i] = params[i].grad for i in idxs (zeros where a parameter received no gradient)."""
        dst, src = [], []
        for i in idxs:
            g = self.params[i].grad
            if g is None:
                self.views[i].zero_()
            elif g.data_ptr() != self.views[i].data_ptr():
                dst.append(self.views[i])
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)


def reverse_execution_order(model):
    """Parameter order in which gradients become ready: fcn, block 9 ... block 0, then every edge_importance, then
    data_bn.  The importances come last although each belongs to one block: A * importance is ONE autograd node for
    all blocks (functional.EdgeImportanceFn), whose backward runs only after block 0's -- spread over the blocks'
    buckets they would hold every bucket back until the end of backward."""
    def order(params):
        ids = {id(p) for p in params}
        out = []
        def take(mod):
            for p in mod.parameters():
                if id(p) in ids and all(p is not q for q in out):
                    out.append(p)
        take(model.fcn)
        for i in reversed(range(len(model.st_gcn_networks))):
            take(model.st_gcn_networks[i])
        ei = getattr(model, "edge_importance", None)
        if isinstance(ei, torch.nn.ParameterList):
            for i in reversed(range(len(ei))):
                if id(ei[i]) in ids and all(ei[i] is not q for q in out):
                    out.append(ei[i])
        take(model.data_bn)
        for p in params:                      # anything not covered above
            if all(p is not q for q in out):
                out.append(p)
        return out
    return order


class GradBucketer:
    """Bucketed asynchronous all-reduce of a FlatParams gradient buffer.

    Works with any process group (RCCL on GPUs, gloo on CPU).  Buckets are contiguous slices of the
    flat gradient; a bucket is reduced when every parameter in it has received its gradient."""

    def __init__(self, flat, bucket_bytes=4 << 20, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # collectives run when there is more than one rank -- or, with STGCN_DDP_FORCE=1, in a one-rank group too: the
        # whole RCCL path (hook-launched asynchronous all-reduces, broadcast, waits) on a box with a single GPU
        self.distributed = self.world > 1 or (os.environ.get("STGCN_DDP_FORCE") == "1" and dist.is_available() and dist.is_initialized())
        self.buckets = []                     # (start, end, [param indices])
        start, idxs, nbytes = 0, [], 0
        for i, n in enumerate(flat.sizes):
            idxs.append(i)
            nbytes += 4 * n
            if nbytes >= bucket_bytes:
                self.buckets.append((start, flat.offsets[i + 1], idxs))
                start, idxs, nbytes = flat.offsets[i + 1], [], 0
        if idxs:
            self.buckets.append((start, flat.offsets[-1], idxs))
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self.pending = [0] * len(self.buckets)
        self.handles = []
        self.launched = []
        self.trace = []                       # ("grad", param index) / ("launch", bucket) in the order they happened
        self.hooks_enabled = True             # off while a step is captured into / replayed from a HIP graph
        if self.distributed:
            for i, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))
        self.reset()

    def reset(self):
        self.pending = [len(idx) for (_, _, idx) in self.buckets]
        self.handles = []
        self.launched = []
        self.trace = []

    def _make_hook(self, i):
        def hook(param):
            if not self.hooks_enabled:
                return
            b = self.bucket_of[i]
            self.trace.append(("grad", i))
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        s, e, idxs = self.buckets[b]
        self.launched.append(b)
        self.trace.append(("launch", b))
        self.flat.gather(idxs)
        self.handles.append(all_reduce_sum(self.flat.grad[s:e], self.group))

    def finish(self):
        """Launch whatever was not triggered by hooks (unused parameters), wait for all buckets.
        Gradients hold the SUM over ranks afterwards; the optimizer divides by the world size.
        Single process: one multi-tensor copy brings every gradient into the flat buffer."""
        if self.distributed:
            for b in range(len(self.buckets)):
                if b not in self.launched:
                    self._launch(b)
            for h in self.handles:
                h.wait()
        else:
            self.flat.gather(range(len(self.flat.params)))
        return 1.0 / self.world

    def reduce_flat(self):
        """All buckets of an already gathered flat gradient (a replayed HIP graph leaves it complete): every
        all-reduce is launched, then all are waited for.  Returns the scale the optimizer applies."""
        if self.distributed:
            hs = [all_reduce_sum(self.flat.grad[s:e], self.group) for (s, e, _) in self.buckets]
            for h in hs:
                h.wait()
        return 1.0 / self.world


class SGD:
    """torch.optim.SGD(momentum, nesterov, weight_decay) over the flat buffers, one HIP kernel per step."""

    def __init__(self, flat, lr, momentum=0.9, weight_decay=1e-4, nesterov=True):
        self.flat, self.lr, self.momentum, self.weight_decay, self.nesterov = flat, lr, momentum, weight_decay, nesterov

    def step(self, grad_scale=1.0):
        from . import ops as P
        P.sgd_step(self.flat.flat, self.flat.grad, self.flat.mom, self.lr, self.momentum, self.weight_decay,
                   self.nesterov, self.flat.steps == 0, grad_scale)
        self.flat.steps += 1
        # the kernel wrote the parameters through raw pointers: nothing derived from their old values may survive
        drop = getattr(self.flat.module, "invalidate_plan", None)
        if drop is not None:
            drop()


def adjust_lr(base_lr, epoch, steps):
    """Step schedule of the reference (processor/recognition.py:62-70)."""
    return base_lr * (0.1 ** sum(1 for s in steps if epoch >= s))


class Trainer:
    """One training iteration as REC_Processor.train does it (processor/recognition.py:84-97):
    forward, mean cross-entropy, backward, (all-reduce), SGD-nesterov update.

    graph=True: after two ordinary steps (they size the allocator's pools, the workspaces and the kernels' per-device
    state) the forward + loss + backward + gradient gather of a step are captured ONCE into a HIP graph per input
    shape and replayed from then on: the host issues one graph launch instead of ~200 kernel launches through
    autograd, which is what bounds small per-GPU batches.  Inputs are copied into the graph's static buffers; the
    dropout masks advance through a step count kept on the device; collectives and the optimizer launch stay
    outside the graph (the learning rate may change freely, and no RCCL call is ever captured).  With more than one
    rank the step is captured as two graphs (SPLIT_BLOCK): the gradients of the first backward segment are all-reduced
    while the second is replayed.  If the capture fails the trainer says so on stderr and keeps stepping launch by
    launch."""

    GRAPH_WARMUP = 2
    # With more than one rank the captured step is TWO graphs: forward, loss and the backward of the head and of blocks
    # SPLIT_BLOCK.. (three quarters of the parameters) in the first, the rest of the backward in the second.  The
    # all-reduce of the first segment's gradients is issued between the two replays and runs under the second; no
    # collective is ever captured.
    SPLIT_BLOCK = 7

    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True, bucket_bytes=4 << 20, graph=False, split=None):
        self.model = model
        self.flat = FlatParams(model, reverse_execution_order(model))
        self.bucketer = GradBucketer(self.flat, bucket_bytes)
        self.opt = SGD(self.flat, lr, momentum, weight_decay, nesterov)
        from . import functional as F_
        self.loss_fn = F_.cross_entropy          # nn.CrossEntropyLoss() with its defaults, one launch each way
        self.use_graph = bool(graph)
        self.split = split                    # capture the step as two graphs: None = when there is more than one rank
        self._graphs = {}                     # (data shape, label shape) -> (graph, static data, static label, static loss)
        self._eager_steps = 0
        self._grads_bound = False             # whether every p.grad is the parameter's slice of the flat gradient

    def broadcast_from_rank0(self):
        if self.bucketer.distributed:
            broadcast0(self.flat.flat)
            for b in self.model.buffers():
                broadcast0(b)

    def graph_ready(self):
        return bool(self._graphs)

    def graph_segments(self):
        """Captured graphs per step: 2 when the step was captured in two segments (collectives between them), 1 for one
        graph, 0 without a graph."""
        return max((2 if isinstance(ent[0], tuple) else 1 for ent in self._graphs.values()), default=0)

    def _forward_backward(self, data, label):
        self.flat.zero_grad()
        out = self.model(data)
        loss = self.loss_fn(out, label)
        loss.backward()
        return loss.detach()                  # not the graph's root: a kept loss would keep every autograd node of the step alive

    def _split_point(self):
        """Index into the flat buffers where the parameters of the second backward segment begin, or None when the
        model cannot be split (not an ST-GCN Model, or an order in which the late parameters are no prefix)."""
        m = self.model
        blocks = getattr(m, "st_gcn_networks", None)
        if blocks is None or not hasattr(m, "split_block") or len(blocks) <= self.SPLIT_BLOCK:
            return None
        late = {id(p) for p in m.fcn.parameters()}
        for i in range(self.SPLIT_BLOCK, len(blocks)):
            late |= {id(p) for p in blocks[i].parameters()}
        n = 0
        while n < len(self.flat.params) and id(self.flat.params[n]) in late:
            n += 1
        if n == 0 or n != len([p for p in self.flat.params if id(p) in late]):
            return None
        return n

    def _capture_split(self, data, label, nlate):
        """Two graphs sharing one memory pool (always replayed in capture order): see SPLIT_BLOCK.  Gradients are taken
        with torch.autograd.grad and copied into the flat buffer inside the graphs; the roots of the second segment
        are the activation entering block SPLIT_BLOCK and the late blocks' A * importance tensors, whose gradients the
        first segment returns (the importance product is ONE autograd node for all blocks: it runs in the second)."""
        from . import ops as P
        if P.TIMING is not None:
            return None
        m, flat = self.model, self.flat
        sx, sl = data.clone(), label.clone()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        late, early = flat.params[:nlate], flat.params[nlate:]
        self.bucketer.hooks_enabled = False
        m.split_block = self.SPLIT_BLOCK
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                flat.zero_grad()
                loss = self.loss_fn(m(sx), sl)
                mid = m._mid
                adj = [t for t in (m._a_eff[self.SPLIT_BLOCK:] if m._a_eff is not None else []) if isinstance(t, torch.Tensor) and t.requires_grad]
                roots = [mid] + adj
                gr = torch.autograd.grad(loss, roots + late, allow_unused=True)
                groots, glate = gr[:len(roots)], gr[len(roots):]
                dst = [flat.views[i] for i, g in enumerate(glate) if g is not None]
                torch._foreach_copy_(dst, [g for g in glate if g is not None])
                for i, g in enumerate(glate):
                    if g is None:
                        flat.views[i].zero_()
                loss = loss.detach()
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                keep = [(r, g) for r, g in zip(roots, groots) if g is not None]
                ge = torch.autograd.grad([r for r, _ in keep], early, grad_outputs=[g for _, g in keep], allow_unused=True)
                dst = [flat.views[nlate + i] for i, g in enumerate(ge) if g is not None]
                torch._foreach_copy_(dst, [g for g in ge if g is not None])
                for i, g in enumerate(ge):
                    if g is None:
                        flat.views[nlate + i].zero_()
            del gr, groots, glate, ge, keep, roots, mid, adj
        except Exception as e:                # stay in this process: step launch by launch from now on
            import sys
            print("stgcn Trainer: HIP graph capture (two segments) failed (%s: %s); stepping without a graph" % (type(e).__name__, e), file=sys.stderr)
            if os.environ.get("STGCN_GRAPH_DEBUG"):
                import traceback
                traceback.print_exc()
            self.use_graph = False
            self.bucketer.hooks_enabled = True
            m.split_block = None
            m._mid = m._a_eff = None
            torch.cuda.synchronize()
            return None
        self.bucketer.hooks_enabled = True
        m.split_block = None
        m._mid = m._a_eff = None
        return ((g1, g2, flat.offsets[nlate]), sx, sl, loss)

    def _capture(self, data, label):
        from . import ops as P
        if P.TIMING is not None:
            return None
        want = self.bucketer.distributed and os.environ.get("STGCN_GRAPH_SPLIT", "1") != "0" if self.split is None else self.split
        if want:
            nlate = self._split_point()
            if nlate is not None:
                return self._capture_split(data, label, nlate)
        sx, sl = data.clone(), label.clone()
        g = torch.cuda.CUDAGraph()
        self.bucketer.hooks_enabled = False
        try:
            torch.cuda.synchronize()
            # thread_local: the capture restricts this thread only.  The process group's watchdog thread polls the
            # events of earlier all-reduces with hipEventQuery; under the default (global) mode such a call from
            # another thread during the capture is an error that the watchdog turns into an abort of the process.
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = self._forward_backward(sx, sl)
                self.flat.gather(range(len(self.flat.params)))
        except Exception as e:                # stay in this process: step launch by launch from now on
            import sys
            print("stgcn Trainer: HIP graph capture failed (%s: %s); stepping without a graph" % (type(e).__name__, e), file=sys.stderr)
            self.use_graph = False
            self.bucketer.hooks_enabled = True
            torch.cuda.synchronize()
            return None
        self.bucketer.hooks_enabled = True
        return (g, sx, sl, loss)

    def _graph_for(self, data, label):
        key = (tuple(data.shape), data.dtype, tuple(label.shape))
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._capture(data, label)
            if ent is not None:
                self._graphs[key] = ent
        return ent

    def prepare_graph(self, data, label):
        """Captures the graph for this input shape now (no step is performed) instead of inside the first step() that
        would use it; needs the warm-up steps done.  Returns whether a graph is ready."""
        if self.use_graph and self.model.training and self._eager_steps >= self.GRAPH_WARMUP:
            return self._graph_for(data, label) is not None
        return False

    def step(self, data, label, eager=False):
        """One iteration; returns the loss (a device tensor; with a graph, the graph's own output buffer, overwritten
        by the next replay).  eager=True forces the launch-by-launch path for this call."""
        from . import ops as P
        if self.use_graph and not eager and P.TIMING is None and self.model.training and self._eager_steps >= self.GRAPH_WARMUP:
            ent = self._graph_for(data, label)
            if ent is not None:
                g, sx, sl, loss = ent
                if data.data_ptr() != sx.data_ptr():
                    sx.copy_(data, non_blocking=True)
                if label.data_ptr() != sl.data_ptr():
                    sl.copy_(label, non_blocking=True)
                if isinstance(g, tuple):
                    # two segments: the late parameters' gradients travel while the second segment runs
                    g1, g2, cut = g
                    dist_on = self.bucketer.distributed
                    g1.replay()
                    hs = [all_reduce_sum(self.flat.grad[:cut], self.bucketer.group)] if dist_on else []
                    g2.replay()
                    if dist_on:
                        hs.append(all_reduce_sum(self.flat.grad[cut:], self.bucketer.group))
                    for h in hs:
                        h.wait()
                    scale = 1.0 / self.bucketer.world
                else:
                    g.replay()
                    scale = None
                if not self._grads_bound:
                    # the replay leaves every gradient in the flat buffer only; .grad may still name the tensors of
                    # an earlier launch-by-launch step
                    for p, v in zip(self.flat.params, self.flat.views):
                        p.grad = v
                    self._grads_bound = True
                self.opt.step(self.bucketer.reduce_flat() if scale is None else scale)
                return loss
        self._grads_bound = False
        self.bucketer.reset()
        loss = self._forward_backward(data, label)
        scale = self.bucketer.finish()
        self.opt.step(scale)
        self._eager_steps += 1
        return loss
