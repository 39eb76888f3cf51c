"""ST-GCN model and block on hand-written gfx950 kernels.

Drop-in for net.st_gcn of hensyou/st-gcn (st_gcn.py:9-197): Model(in_channels, num_class,
graph_args, edge_importance_weighting, **kwargs), Model.forward(x[N,C,T,V,M]) -> [N,num_class],
Model.extract_feature, st_gcn(in, out, kernel_size, stride, dropout, residual).forward(x, A)
-> (y, A).  Sub-module names and parameter shapes equal the reference's, so checkpoints
interchange; the nn.Conv2d / nn.BatchNorm2d / nn.Dropout sub-modules are parameter holders
only.  Arithmetic runs in the HIP library through stgcn_rt; there is no CPU path.

Extras over the reference signature (all optional): kwarg `compute_dtype` (torch.float32 or
torch.bfloat16; default float32, or bfloat16 when env STGCN_DTYPE=bf16).
"""
import os

import torch
import torch.nn as nn

from net.utils.graph import Graph
from net.utils.tgcn import ConvTemporalGraphical
from stgcn_rt import ops as P
from stgcn_rt.functional import DataBnFn, EdgeImportanceFn, HeadFn, StGcnBlockFn


def _default_dtype():
    v = os.environ.get("STGCN_DTYPE", "fp32").lower()
    return torch.bfloat16 if v in ("bf16", "bfloat16") else torch.float32


class Model(nn.Module):
    def __init__(self, in_channels, num_class, graph_args, edge_importance_weighting, **kwargs):
        super().__init__()
        self.compute_dtype = kwargs.pop("compute_dtype", None) or _default_dtype()

        self.graph = Graph(**graph_args)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer("A", A)

        K = A.size(0)
        kernel_size = (9, K)
        self.data_bn = nn.BatchNorm1d(in_channels * A.size(1))
        first = {k: v for k, v in kwargs.items() if k != "dropout"}     # block 0 never drops (st_gcn.py:43-45)
        plan = [(in_channels, 64, 1, dict(residual=False, **first)), (64, 64, 1, kwargs), (64, 64, 1, kwargs),
                (64, 64, 1, kwargs), (64, 128, 2, kwargs), (128, 128, 1, kwargs), (128, 128, 1, kwargs),
                (128, 256, 2, kwargs), (256, 256, 1, kwargs), (256, 256, 1, kwargs)]
        self.st_gcn_networks = nn.ModuleList(st_gcn(ci, co, kernel_size, s, **kw) for ci, co, s, kw in plan)
        for i, blk in enumerate(self.st_gcn_networks):
            blk.block_index = i                         # enters the block's dropout seed
        self._drop_step = None                          # device-side count of training forwards (dropout stream position)

        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList(
                [nn.Parameter(torch.ones(self.A.size())) for _ in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)

        self.fcn = nn.Conv2d(256, num_class, kernel_size=1)
        self._packer = None
        self._plan = None                               # (key, folded inference parameters): see _eval_plan
        # A trainer that backpropagates in two segments (stgcn_rt.ddp.Trainer with more than one rank and a HIP graph)
        # sets split_block = i: a training forward then keeps the activation entering block i and the per-block
        # A * importance tensors, the roots of the second segment.
        self.split_block = None
        self._mid = None
        self._a_eff = None

    def _pack_specs(self, dtype):
        specs = []
        for blk in self.st_gcn_networks:
            cin_p = P.pad_to(blk.in_channels, P.kstep(dtype))
            specs.append((blk.gcn.conv.weight, "gcn", blk.gcn.kernel_size, cin_p))
            specs.append((blk.tcn[2].weight, "conv", 0, 0))
            if blk.residual_kind == "conv":
                specs.append((blk.residual[0].weight, "conv", 0, 0))
        return specs

    def _packed_weights(self, dtype, device):
        """All blocks' convolution weights in kernel layout and compute dtype: one launch per forward pass."""
        specs = self._pack_specs(dtype)
        pk = self._packer
        if pk is None or pk.dtype != dtype or pk.buf.device != device or not pk.valid(specs):
            pk = self._packer = P.WeightPacker(specs, dtype, device)
        pk.refresh()
        out = []
        for blk in self.st_gcn_networks:
            res = pk.get(blk.residual[0].weight) if blk.residual_kind == "conv" else None
            out.append((pk.get(blk.gcn.conv.weight), pk.get(blk.tcn[2].weight), res))
        return out

    # ---- inference path: BatchNorm folded away -------------------------------------------------------------------
    # In eval mode every BatchNorm is an affine map with fixed coefficients and Dropout is the identity, so a block is
    #   y = gcn(x);  h = relu(s1 * y + t1);  out = relu(conv(h; s2 * W) + (s2 * b + t2) + res)
    # (net/st_gcn.py:159-171, 191-197 of the reference with model.eval(), as processor/recognition.py:110-143 and
    # Model.extract_feature run it): s2 and the residual BatchNorm's scale go into the convolution weights, the shifts
    # into the biases, and the add + ReLU into the temporal convolution's epilogue.  Two or three launches per block,
    # none for statistics; the folded parameters are rebuilt only when a parameter or running statistic has changed.
    @staticmethod
    def _affine(bn):
        s = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        return s.contiguous(), (bn.bias.detach().float() - bn.running_mean.float() * s).contiguous()

    def invalidate_plan(self):
        """Forget the folded inference parameters.  Called by everything that rewrites parameters or running
        statistics through raw pointers -- the fused SGD kernel, the BatchNorm kernels of a training forward, a replayed
        HIP graph of either -- because such writes never advance a tensor's `_version`, which is all the plan's key can
        see of a tensor's contents."""
        self._plan = None

    def train(self, mode=True):
        self._plan = None                               # every train <-> eval transition starts from a fresh plan
        return super().train(mode)

    def _eval_plan(self, dtype, device):
        tensors = list(self.parameters()) + list(self.buffers())
        key = (dtype, device, tuple((t.data_ptr(), t._version) for t in tensors))
        if self._plan is not None and self._plan[0] == key:
            return self._plan[1]
        with torch.no_grad():
            plan = {"data_bn": self._affine(self.data_bn), "blocks": []}
            for i, blk in enumerate(self.st_gcn_networks):
                K = blk.gcn.kernel_size
                cin_p = P.pad_to(blk.in_channels, P.kstep(dtype))
                imp = self.edge_importance[i]
                A = (self.A * imp).float().contiguous() if isinstance(imp, torch.Tensor) else (self.A * float(imp)).float().contiguous()
                s1, t1 = self._affine(blk.tcn[0])
                s2, t2 = self._affine(blk.tcn[3])
                conv = blk.tcn[2]
                ent = dict(A=A, wg=P.pack_gcn_w(blk.gcn.conv.weight.detach(), K, cin_p, dtype),
                           bg=blk.gcn.conv.bias.detach().float().contiguous(), s1=s1, t1=t1,
                           w2=P.pack_conv_w(conv.weight.detach().float() * s2[:, None, None, None], dtype),
                           b2=(conv.bias.detach().float() * s2 + t2).contiguous(), wr=None, br=None)
                if blk.residual_kind == "conv":
                    sr, tr = self._affine(blk.residual[1])
                    rc = blk.residual[0]
                    ent["wr"] = P.pack_conv_w(rc.weight.detach().float() * sr[:, None, None, None], dtype)
                    ent["br"] = (rc.bias.detach().float() * sr + tr).contiguous()
                plan["blocks"].append(ent)
        self._plan = (key, plan)
        return plan

    def _foldable(self):
        """Inference with nothing to differentiate and every BatchNorm / Dropout in eval mode."""
        if torch.is_grad_enabled() or self.data_bn.training:
            return False
        return not any(b.tcn[0].training or b.tcn[3].training or (b.tcn[4].training and b.tcn[4].p > 0) or
                       (b.residual_kind == "conv" and b.residual[1].training) for b in self.st_gcn_networks)

    def _trunk_folded(self, x, pool_last):
        from stgcn_rt import lib as L
        N, C, T, V, M = x.size()
        dtype = self.compute_dtype
        P.check_device(x, "input")
        x = x.float().contiguous()
        plan = self._eval_plan(dtype, x.device)
        cp = P.pad_to(C, P.kstep(dtype))
        sd, td = plan["data_bn"]
        h = P.new_act(N * M, cp, T, V, dtype, x.device)
        L.call("stgcn_data_bn_apply", L.dtype_code(dtype), L.ptr(x), L.ptr(h), L.ptr(sd), L.ptr(td), N, C, T, V, M, cp, L.stream_ptr())
        for blk, e in zip(self.st_gcn_networks, plan["blocks"]):
            Cout, stride = blk.out_channels, blk.stride
            kt = blk.tcn[2].kernel_size[0]
            Tin = h.shape[2]
            To = (Tin - 1) // stride + 1
            y = P.gcn_fused(h, e["wg"], e["A"], Cout, bias=e["bg"], adj_t=True)
            if blk.residual_kind == "conv":
                res = P.conv_rows(h, e["wr"], Cout, To, bias=e["br"], ntaps=1, stride=stride, pad=0)
            else:
                res = h if blk.residual_kind == "identity" else None
            h = P.conv_rows(y, e["w2"], Cout, To, bias=e["b2"], pro=(e["s1"], e["t1"]), epilogue=L.EPI_ADD_RELU, aux=res,
                            ntaps=kt, stride=stride, pad=(kt - 1) // 2)
        if pool_last:
            B, Cc, To, _ = h.shape
            pooled = torch.empty((B, Cc), dtype=torch.float32, device=h.device)
            L.call("stgcn_pool_fwd", L.dtype_code(dtype), L.ptr(h), L.ptr(pooled), B, To * V, Cc, L.stream_ptr())
            return pooled
        return h

    def _trunk(self, x, pool_last=False):
        if self._foldable():
            return self._trunk_folded(x, pool_last)
        N, C, T, V, M = x.size()
        dtype = self.compute_dtype
        bn = self.data_bn
        self._plan = None                               # this pass may rewrite running statistics through raw pointers
        # Train / eval follows each sub-module's own flag, as in the reference where every nn.BatchNorm / nn.Dropout
        # consults its own .training (a model in train mode with its BatchNorms frozen by .eval() is a common
        # fine-tuning set-up).
        drops = any(b.tcn[4].training and b.tcn[4].p > 0 for b in self.st_gcn_networks)
        if bn.training or drops or any(b.tcn[0].training for b in self.st_gcn_networks):
            P.stats_begin(x.device)                     # one fill clears every statistic buffer of this step
        h = DataBnFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, dtype,
                           P.pad_to(C, P.kstep(dtype)))
        packs = self._packed_weights(dtype, h.device)
        counters = [bn.num_batches_tracked] if bn.training else []
        step = None
        if drops:
            # The dropout masks of a step are drawn from (per-block seed, step count); the count lives on the device
            # and advances with the BatchNorm counters below, so a step replayed from a captured HIP graph draws
            # fresh masks although every launch argument is frozen in the graph.
            if self._drop_step is None or self._drop_step.device != h.device:
                self._drop_step = torch.zeros((), dtype=torch.int64, device=h.device)
            step = self._drop_step
            counters.append(step)
        last = len(self.st_gcn_networks) - 1
        if isinstance(self.edge_importance, nn.ParameterList):
            a_eff = EdgeImportanceFn.apply(self.A, *self.edge_importance)       # A * importance, all blocks at once
        else:
            a_eff = [self.A * importance for importance in self.edge_importance]
        self._mid, self._a_eff = None, None
        for i, gcn in enumerate(self.st_gcn_networks):
            if self.split_block == i and torch.is_grad_enabled():
                # The late blocks take A * importance through a view each: a node of its own per block, at which a
                # backward segment can stop.  (Stopping at the outputs of the ONE node that forms all the products makes
                # the autograd engine treat every block that feeds that node as needed, i.e. run the whole backward.)
                a_eff = list(a_eff)
                for j in range(i, len(a_eff)):
                    if isinstance(a_eff[j], torch.Tensor) and a_eff[j].requires_grad:
                        a_eff[j] = a_eff[j].view(a_eff[j].shape)
                self._mid, self._a_eff = h, a_eff
            # global average pool over (T, V) fused into the last block (st_gcn.py:85 of the reference)
            want_pool = pool_last and i == last
            h, pooled = gcn._run(h, a_eff[i], want_pool, packs=packs[i], counters=counters, drop_step=step)
            if want_pool:
                h = pooled
        if counters:
            torch._foreach_add_(counters, 1)            # every BatchNorm's num_batches_tracked in one launch
        return h

    def forward(self, x):
        N, C, T, V, M = x.size()
        pooled = self._trunk(x, pool_last=True)
        return HeadFn.apply(pooled, self.fcn.weight, self.fcn.bias, N, M)

    def extract_feature(self, x):
        """Inference only: per-position class scores [N,num_class,t,v,M] and features [N,256,t,v,M]."""
        N, C, T, V, M = x.size()
        with torch.no_grad():
            h = self._trunk(x)
            _, c, t, v = h.size()
            feature = h.float().reshape(N, M, c, t, v).permute(0, 2, 3, 4, 1)
            ncls = self.fcn.out_channels
            o = P.conv_rows(P.cl(h), P.pack_conv_w(self.fcn.weight, h.dtype), ncls, t,
                            bias=self.fcn.bias.float().contiguous(), ntaps=1)
            output = o.float().reshape(N, M, ncls, t, v).permute(0, 2, 3, 4, 1)
        return output, feature


class st_gcn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dropout=0, residual=True):
        super().__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0] % 2 == 1
        padding = ((kernel_size[0] - 1) // 2, 0)
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1])
        self.tcn = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, (kernel_size[0], 1), (stride, 1), padding),
            nn.BatchNorm2d(out_channels),
            nn.Dropout(dropout, inplace=True),
        )
        if not residual:
            self.residual_kind = "none"
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self.residual_kind = "identity"
            self.residual = lambda x: x
        else:
            self.residual_kind = "conv"
            self.residual = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                nn.BatchNorm2d(out_channels),
            )
        self.relu = nn.ReLU(inplace=True)
        self.compute_dtype = None
        self.block_index = 0
        self._drop_step = None

    def _seed(self):
        """Dropout seed of this block: a function of torch's seed (torch.manual_seed makes runs repeat; unseeded, torch
        draws a fresh seed per process and so do the masks), the block's position in the model and the process's rank
        (data-parallel replicas draw different masks, as nn.DataParallel replicas do).  The position inside the stream
        of masks is the device-side step count."""
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        return (torch.initial_seed() * 0x9E3779B1 + (self.block_index + 1) * 0x632BE5AB + rank * 0xC2B2AE35) & 0xFFFFFFFF

    def forward(self, x, A):
        return self._run(x, A, False)[0], A

    def _run_padded(self, x, A, cfg, cp, bn1, conv, bn2, rargs, rbufs, dtype):
        """Channel widths that are no multiple of one 128-byte chunk (64 channels in bf16, 32 in float32; no reference
        net has them): the block runs at the padded output width cp, and its input is padded to the width its residual
        branch needs.  Every parameter is zero-padded with differentiable tensor ops, so autograd slices the gradients
        back; a padded channel has weight 0, BatchNorm gamma 0 and beta 0 and therefore stays exactly 0 through both
        BatchNorms, the ReLUs and the residual sum, forward and backward.  Running statistics are updated on padded
        copies and written back."""
        import torch.nn.functional as TF
        K = self.gcn.kernel_size
        cout = self.out_channels
        d = cp - cout
        vec = lambda t: TF.pad(t, (0, d))
        kind = self.residual_kind
        want = cp if kind == "identity" else (P.pad_to(self.in_channels, P.kchunk(dtype)) if kind == "conv" else x.size(1))
        if x.size(1) != want:
            x = P.pad_channels(x, want)
        Wg = self.gcn.conv.weight
        Wg_p = TF.pad(Wg.reshape(K, cout, Wg.shape[1]), (0, want - Wg.shape[1], 0, d)).reshape(K * cp, want, 1, 1)
        bg_p = TF.pad(self.gcn.conv.bias.reshape(K, cout), (0, d)).reshape(-1) if self.gcn.conv.bias is not None else None
        Wt_p = TF.pad(conv.weight, (0, 0, 0, 0, 0, d, 0, d))
        rargs_p = rargs
        if kind == "conv":
            Wr, br, gr, betar = rargs
            rargs_p = (TF.pad(Wr, (0, 0, 0, 0, 0, want - Wr.shape[1], 0, d)), vec(br), vec(gr), vec(betar))
        mean_p = lambda t: None if t is None else TF.pad(t.detach(), (0, d))
        var_p = lambda t: None if t is None else TF.pad(t.detach(), (0, d), value=1.0)
        bufs = [bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var, rbufs[0], rbufs[1]]
        bufs_p = [mean_p(bufs[0]), var_p(bufs[1]), mean_p(bufs[2]), var_p(bufs[3]), mean_p(bufs[4]), var_p(bufs[5])]
        cfg = cfg[:5] + (want,) + cfg[6:]                               # the width the graph-conv weight gradient keeps
        y, pooled = StGcnBlockFn.apply(x, A, Wg_p, bg_p, vec(bn1.weight), vec(bn1.bias), Wt_p, vec(conv.bias),
                                       vec(bn2.weight), vec(bn2.bias), *rargs_p, *bufs_p, cfg, None)
        if cfg[2]:                                                      # training: the kernels updated the padded copies
            with torch.no_grad():
                for dst, src in zip(bufs, bufs_p):
                    if dst is not None:
                        dst.copy_(src[:cout])
        return y[:, :cout], (None if pooled is None else pooled[:, :cout])

    def forward_pooled(self, x, A):
        """-> (y, mean of y over (T, V) as float32 [B, C])."""
        return self._run(x, A, True)

    def _run(self, x, A, want_pool, packs=None, counters=None, drop_step=None):
        """packs: this block's pre-packed weights (Model packs all blocks with one launch); counters: a list that
        collects the num_batches_tracked buffers for a batched increment instead of bumping them here; drop_step: the
        model's device-side step count (a block used on its own keeps one itself)."""
        assert A.size(0) == self.gcn.kernel_size
        dtype = self.compute_dtype or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        cin = self.in_channels
        cin_p = P.pad_to(cin, P.kstep(dtype))
        if x.dtype != dtype:
            x = x.to(dtype)
        if x.size(1) != cin_p:
            x = P.pad_channels(x, cin_p)
        bn1, conv, bn2, drop = self.tcn[0], self.tcn[2], self.tcn[3], self.tcn[4]
        if self.residual_kind == "conv":
            rc, rb = self.residual[0], self.residual[1]
            rargs = (rc.weight, rc.bias, rb.weight, rb.bias)
            rbufs = (rb.running_mean, rb.running_var)
        else:
            rargs, rbufs = (None, None, None, None), (None, None)
        bns = [bn1, bn2] + ([self.residual[1]] if self.residual_kind == "conv" else [])
        if len({b.training for b in bns}) != 1:
            raise NotImplementedError("the BatchNorm modules of one st_gcn block must be all in train mode or all in eval mode")
        bn_train = bn1.training
        drop_p = float(drop.p) if drop.training else 0.0
        own_step = None
        if drop_p > 0 and drop_step is None:
            if self._drop_step is None or self._drop_step.device != x.device:
                self._drop_step = torch.zeros((), dtype=torch.int64, device=x.device)
            drop_step = own_step = self._drop_step
        cfg = (self.stride, self.residual_kind, bn_train, drop_p, (self._seed(), drop_step), cin, want_pool)
        cout = self.out_channels
        cout_p = P.pad_to(cout, P.kchunk(dtype))
        if cout_p != cout or (self.residual_kind == "conv" and cin % P.kchunk(dtype) != 0):
            y, pooled = self._run_padded(x, A, cfg, cout_p, bn1, conv, bn2, rargs, rbufs, dtype)
        else:
            y, pooled = StGcnBlockFn.apply(x, A, self.gcn.conv.weight, self.gcn.conv.bias, bn1.weight, bn1.bias,
                                           conv.weight, conv.bias, bn2.weight, bn2.bias, *rargs,
                                           bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var, *rbufs, cfg, packs)
        tracked = [b.num_batches_tracked for b in bns] if bn_train else []
        if own_step is not None:
            tracked.append(own_step)
        if tracked:
            if counters is not None:
                counters.extend(tracked)
            else:
                torch._foreach_add_(tracked, 1)
        return y, pooled
