"""autograd.Function wrappers: one per reference operator on the hot path.

  GraphConvFn   ConvTemporalGraphical.forward        (net/utils/tgcn.py:57-66)
  StGcnBlockFn  st_gcn.forward                       (net/st_gcn.py:191-197)
  DataBnFn      Model.forward input normalisation    (net/st_gcn.py:72-78)
  HeadFn        pooling + fcn                        (net/st_gcn.py:85-90)
"""
import contextlib
import os

import torch

from . import lib as L
from . import ops as P


# Test hook: when set to a list, every StGcnBlockFn.backward appends the two ReLU decisions it used
# (inner ReLU after BN1, final ReLU) so a checker can evaluate its reference with the same decisions.
CAPTURE = None
CAPTURE_TENSORS = False

# Parameter-gradient kernels of a block's backward (conv_wgrad, gcn_wgrad, gcn_agrad) are off the chain that carries the
# data gradient to the previous block.  With OVERLAP_WGRAD (STGCN_OVERLAP_WGRAD=1) they are launched on a second
# stream: they fork from the current stream once their inputs exist and are joined before the block's backward
# returns, so the MFMA-bound weight-gradient kernels run beside the chain's HBM-bound elementwise kernels instead of
# in front of them.  Every tensor they read outlives the join.  Worth 1.4-2.1 % of a training step; off by default,
# because kernels that share the device stretch each other and per-kernel durations (profiles, the benchmark's
# roofline record) stop describing the kernels.
OVERLAP_WGRAD = os.environ.get("STGCN_OVERLAP_WGRAD", "0") == "1"
# Debugging aid (STGCN_BN_LAUNCHES=1): a block's BatchNorms finished by launches of their own (bn_finalize) instead of
# inside the temporal convolution's and the output stage's launches.
BN_LAUNCHES = os.environ.get("STGCN_BN_LAUNCHES", "0") == "1"
_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(dev.index)
    if s is None:
        s = _SIDE[dev.index] = torch.cuda.Stream(device=dev)
    return s


class _Fork:
    """`with _Fork(side, ev):` runs its body on `side` after the event `ev` of the current stream (no-op when side is
    None).  The event is recorded where the body's inputs are complete; the body is ISSUED later, behind the chain's
    next MFMA kernel, so that the device starts that one first and the body runs beside what follows it.  The caller
    joins with `torch.cuda.current_stream().wait_stream(side)`."""

    def __init__(self, side, ev):
        self.side = side
        self.ev = ev
        self.ctx = None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_event(self.ev)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _act_dtype(dtype):
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be float32 or bfloat16")
    return dtype


class GraphConvFn(torch.autograd.Function):
    """y = einsum('nkctv,kvw->nctw', conv1x1(x).view(n,K,C,t,v), A); x already channel-padded."""

    @staticmethod
    def forward(ctx, x, A, W, b, cin_true):
        P.check_act(x)
        K, V = A.shape[0], A.shape[1]
        dtype = x.dtype
        cin_p = x.shape[1]
        Cout = W.shape[0] // K
        x = P.cl(x)
        A32 = A.detach().to(torch.float32)
        cout_p = P.pad_to(Cout, P.kstep(dtype))          # kernels store whole 16-byte channel vectors
        Wd = W.detach()
        bias = None if b is None else b.detach().float()
        if cout_p != Cout:
            Wd = torch.nn.functional.pad(Wd.reshape(K, Cout, -1), (0, 0, 0, cout_p - Cout)).reshape(K * cout_p, -1, 1, 1)
            if bias is not None:
                bias = torch.nn.functional.pad(bias.reshape(K, Cout), (0, cout_p - Cout)).reshape(-1)
        wp = P.pack_gcn_w(Wd, K, cin_p, dtype)
        y = P.gcn_fused(x, wp, A32.contiguous(), cout_p, bias=None if bias is None else bias.contiguous(), adj_t=True)
        if cout_p != Cout:
            y = P.cl(y[:, :Cout])
        ctx.save_for_backward(x, A32, W, b if b is not None else torch.empty(0))
        ctx.has_bias = b is not None
        ctx.cin_true = cin_true
        return y

    @staticmethod
    def backward(ctx, dy):
        x, A32, W, b = ctx.saved_tensors
        b = b if ctx.has_bias else None
        K = A32.shape[0]
        dtype = x.dtype
        cin_p = x.shape[1]
        Cout = W.shape[0] // K
        cout_p = P.pad_to(Cout, P.kstep(dtype))          # the gradient kernels contract over whole k-steps
        dy = P.pad_channels(dy.to(dtype), cout_p)
        Wd = W.detach()
        bias = None if b is None else b.detach().float()
        if cout_p != Cout:
            Wd = torch.nn.functional.pad(Wd.reshape(K, Cout, -1), (0, 0, 0, cout_p - Cout)).reshape(K * cout_p, -1, 1, 1)
            if bias is not None:
                bias = torch.nn.functional.pad(bias.reshape(K, Cout), (0, cout_p - Cout)).reshape(-1)
        if bias is not None:
            bias = bias.contiguous()
        dx = dA = dW = db = None
        if ctx.needs_input_grad[0]:
            wt = P.pack_gcn_wt(Wd, K, cin_p, dtype)
            dx = P.gcn_fused(dy, wt, A32.contiguous(), cin_p)
        if ctx.needs_input_grad[1]:
            wp = P.pack_gcn_w(Wd, K, cin_p, dtype)
            dA = P.gcn_agrad(x, wp, bias, dy)
        if ctx.needs_input_grad[2] or (b is not None and ctx.needs_input_grad[3]):
            dWp, dbp = P.gcn_wgrad(x, dy, A32.contiguous(), need_bias=b is not None)
            dW = dWp[:, :Cout, :ctx.cin_true].reshape(W.shape)
            if dbp is not None:
                db = dbp.reshape(K, cout_p)[:, :Cout].reshape(-1)
        return dx, dA, dW, db, None


class TemporalConvFn(torch.autograd.Function):
    """Conv2d with a (kt, 1) kernel, stride, zero padding and dilation along T (the projection of
    ConvTemporalGraphical when t_kernel_size > 1, net/utils/tgcn.py:48-60).  x [B, cin_p, T, V] channel-padded to
    whole 128-byte chunks; returns [B, cout_p, T', V] with the output channels padded the same way (the padding is
    zero and the caller keeps it: the next kernel contracts over it)."""

    @staticmethod
    def forward(ctx, x, W, b, cfg, cin_true):
        P.check_act(x)
        kt, stride, pad, dil = cfg
        x = P.cl(x)
        dtype = x.dtype
        B, cin_p, T, V = x.shape
        Cout = W.shape[0]
        cout_p = P.pad_to(Cout, P.kchunk(dtype))         # the input gradient contracts over whole chunks of these
        To = (T + 2 * pad - dil * (kt - 1) - 1) // stride + 1
        if To < 1:
            raise ValueError("temporal kernel longer than the padded clip: T=%d, kernel %d, dilation %d, padding %d" % (T, kt, dil, pad))
        Wp = torch.zeros((cout_p, cin_p, kt, 1), dtype=torch.float32, device=x.device)
        Wp[:Cout, :W.shape[1]] = W.detach().float()
        bias = torch.zeros((cout_p,), dtype=torch.float32, device=x.device)
        if b is not None:
            bias[:Cout] = b.detach().float()
        z = P.conv_rows(x, P.pack_conv_w(Wp, dtype), cout_p, To, bias=bias, ntaps=kt, stride=stride, pad=pad, dil=dil)
        ctx.save_for_backward(x, Wp)
        ctx.cfg = (kt, stride, pad, dil, Cout, cin_true, b is not None, tuple(W.shape))
        return z

    @staticmethod
    def backward(ctx, dz):
        x, Wp = ctx.saved_tensors
        kt, stride, pad, dil, Cout, cin_true, has_bias, wshape = ctx.cfg
        dtype = x.dtype
        B, cin_p, T, V = x.shape
        dz = P.cl(dz.to(dtype))
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = P.conv_rows(dz, P.pack_conv_wt(Wp, dtype), cin_p, T, ntaps=kt, stride=stride, pad=pad, dil=dil, transposed=True)
        if ctx.needs_input_grad[1]:
            dW = P.conv_wgrad(x, dz, kt, stride, pad, dil)[:Cout, :cin_true].reshape(wshape)
        if has_bias and ctx.needs_input_grad[2]:
            db = dz[:, :Cout].float().sum(dim=(0, 2, 3))
        return dx, dW, db, None, None


class StGcnBlockFn(torch.autograd.Function):
    """One st_gcn block.  Positional tensors:
       x, A, Wg, bg, g1, b1, Wt, bt, g2, b2, Wr, br, gr, betar  (the last four None unless conv residual)
       then running stats rm1, rv1, rm2, rv2, rmr, rvr (updated in place when training), the config, and
       optionally the pre-packed weights ((wg, wgT), (wt, wtT), (wr, wrT) or None) of a WeightPacker."""

    @staticmethod
    def forward(ctx, x, A, Wg, bg, g1, b1, Wt, bt, g2, b2, Wr, br, gr, betar,
                rm1, rv1, rm2, rv2, rmr, rvr, cfg, packs=None):
        P.check_act(x)
        stride, kind, training, drop_p, seed, cin_true, want_pool = cfg
        seed, seed_step = seed if isinstance(seed, tuple) else (seed, None)      # (seed, device-side step count)
        ctx.set_materialize_grads(False)
        x = P.cl(x)
        dtype = x.dtype
        B, cin_p, T, V = x.shape
        K = A.shape[0]
        Cout = Wg.shape[0] // K
        kt = Wt.shape[2]
        pad = (kt - 1) // 2
        To = (T + 2 * pad - (kt - 1) - 1) // stride + 1
        dev = x.device
        A32 = A.detach().to(torch.float32).contiguous()
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        stats = P.stats_alloc((3, 2 * Cout), dev) if training else None

        if packs is None:
            packs = ((P.pack_gcn_w(Wg.detach(), K, cin_p, dtype), P.pack_gcn_wt(Wg.detach(), K, cin_p, dtype)),
                     (P.pack_conv_w(Wt.detach(), dtype), P.pack_conv_wt(Wt.detach(), dtype)),
                     (P.pack_conv_w(Wr.detach(), dtype), P.pack_conv_wt(Wr.detach(), dtype)) if kind == "conv" else None)
        ctx.packs = packs
        y = P.gcn_fused(x, packs[0][0], A32, Cout, bias=f32(bg), stats=stats[0] if training else None, adj_t=True)
        if training and not BN_LAUNCHES:
            # train mode: BN1 is finished by the temporal convolution's own launch (from the sums gcn_fused gathered)
            o, bn1 = P.conv_rows_bn(y, packs[1][0], Cout, To, stats[0], f32(g1), f32(b1), rm1, rv1, B * T * V, bias=f32(bt),
                                    stats=stats[1], ntaps=kt, stride=stride, pad=pad)
        else:
            bn1 = P.bn_finalize(stats[0] if training else None, B * T * V, f32(g1), f32(b1), rm1, rv1, training)
            o = P.conv_rows(y, packs[1][0], Cout, To, bias=f32(bt), pro=(bn1.scale, bn1.shift), stats=stats[1] if training else None,
                            ntaps=kt, stride=stride, pad=pad)
        # train mode: BN2 and the residual branch's BatchNorm are finished inside the output stage's launch (it derives
        # their coefficients from the convolutions' sums); eval mode reads the running statistics with bn_finalize
        fold = training and not BN_LAUNCHES and Cout <= 512      # the folded launch keeps its coefficient table in LDS
        bn2 = None if fold else P.bn_finalize(stats[1] if training else None, B * To * V, f32(g2), f32(b2), rm2, rv2, training)

        r, bnr = None, None
        if kind == "conv":
            r = P.conv_rows(x, packs[2][0], Cout, To, bias=f32(br), stats=stats[2] if training else None, ntaps=1, stride=stride, pad=0)
            if not fold:
                bnr = P.bn_finalize(stats[2] if training else None, B * To * V, f32(gr), f32(betar), rmr, rvr, training)
            res = r
        elif kind == "identity":
            res = x
        else:
            res = None
        p = drop_p                                      # already 0 when the Dropout module is in eval mode
        if fold:
            out, relu_mask, bn2, bnr = P.block_out_fwd_bn(o, res, stats[1], f32(g2), f32(b2), rm2, rv2,
                                                          stats[2] if kind == "conv" else None, f32(gr) if kind == "conv" else None,
                                                          f32(betar) if kind == "conv" else None, rmr, rvr, p, seed, seed_step=seed_step)
        elif training or p > 0:
            out, relu_mask = P.block_out_fwd(o, res, bn2, bnr, p, seed, want_mask=True, seed_step=seed_step)
        else:
            out, relu_mask = P.block_out_fwd(o, res, bn2, bnr, p, seed), None
        ctx.relu_mask = relu_mask

        pooled = None
        if want_pool:                                   # mean over (T, V) of the block output, float32 [B, C]
            pooled = torch.empty((B, Cout), dtype=torch.float32, device=dev)
            L.call("stgcn_pool_fwd", L.dtype_code(dtype), L.ptr(out), L.ptr(pooled), B, To * V, Cout, L.stream_ptr())
        ctx.cfg = (stride, kind, training, p, seed, cin_true, kt, pad)
        ctx.bn = (bn1, bn2, bnr)
        ctx.save_for_backward(x, A32, y, o, out, r if r is not None else torch.empty(0),
                              Wg, bg, g1, Wt, g2, Wr if Wr is not None else torch.empty(0), gr if gr is not None else torch.empty(0))
        return out, pooled

    @staticmethod
    def backward(ctx, dout, dpooled):
        x, A32, y, o, out, r, Wg, bg, g1, Wt, g2, Wr, gr = ctx.saved_tensors
        stride, kind, training, p, seed, cin_true, kt, pad = ctx.cfg
        bn1, bn2, bnr = ctx.bn
        packs = ctx.packs
        dtype = x.dtype
        B, cin_p, T, V = x.shape
        K = A32.shape[0]
        Cout = Wg.shape[0] // K
        To = o.shape[2]
        dev = x.device
        if dout is None and dpooled is None:
            return (None,) * 22
        if dout is not None:
            dout = P.cl(dout.to(dtype))
        dpool = None
        if dpooled is not None:                         # d(mean over T*V): constant over (t, v), kept in float32; the kernels scale it
            dpool = dpooled.float().contiguous()
        f32 = lambda t: t.detach().float().contiguous()
        r_ = r if kind == "conv" else None

        # identity residual: its gradient is the masked incoming gradient.  With a stored dout and the forward's bit
        # mask, the graph-conv dX kernel masks dout itself and nothing is written for it.
        mask_in_dx = kind == "identity" and dout is not None and dpool is None and ctx.relu_mask is not None
        side = _side_stream(dev) if OVERLAP_WGRAD else None
        # the partial sums of the block's parameter gradients, and the float32 copies of its bias sums, are produced by
        # one launch at the end of the block (on one stream only: with the second stream every sum follows its own
        # kernel at once)
        with (P.deferred_sums(dev) if side is None else contextlib.nullcontext()):
            dO, dR, g, stats2, statsr, dbt, dbr_f, (dg2, db2, dgr, dbetar) = P.block_out_bwd(
                dout, out, o, r_, bn2, f32(g2), bnr, f32(gr) if kind == "conv" else None, kind == "identity" and not mask_in_dx,
                p, seed, training, dpool=dpool, relu_mask=ctx.relu_mask, want_bn_grads=True, dpool_scale=1.0 / float(To * V))

            # temporal conv: weight grad (input = relu(bn1(y)) recomputed in the prologue), then data grad fused with
            # the ReLU mask and the BN1-backward sums
            stats1 = P.stats_alloc((2 * Cout,), dev)
            dWr = None
            ev = torch.cuda.current_stream().record_event() if side is not None else None
            gh = P.conv_rows(dO, packs[1][1], Cout, T, epilogue=L.EPI_RELU_BWD, aux=y, ep=(bn1.scale, bn1.shift, bn1.mean, bn1.rstd),
                             stats=stats1, ntaps=kt, stride=stride, pad=pad, transposed=True)
            if CAPTURE is not None:
                # the kernels decide by the sign of fmaf(y, scale, shift): y * scale is exact in float64 (at most 48
                # significant bits) and a float64 sum keeps the exact sum's sign, so this is their decision bit for bit
                pre = y.double() * bn1.scale.double()[None, :, None, None] + bn1.shift.double()[None, :, None, None]
                CAPTURE.append(dict(h_mask=pre > 0, out_mask=(out > 0)))
                if CAPTURE_TENSORS:                  # diagnostics (tools/grad_budget.py): what each kernel read and wrote
                    CAPTURE[-1].update(x=x, y=y, o=o, r=r_, dout=dout, dpool=dpool, dO=dO, dR=dR, g=g, gh=gh.clone(),
                                       bn=(bn1, bn2, bnr), relu_mask=ctx.relu_mask)
            with _Fork(side, ev):                        # issued behind the data-gradient kernel: runs beside bn_bwd_apply
                dWt = P.conv_wgrad(y, dO, kt, stride, pad, pro=(bn1.scale, bn1.shift))
                if kind == "conv":
                    dWr = P.conv_wgrad(x, dR, 1, stride, 0)
            dy, dg1, db1 = P.bn_bwd_apply(gh, y, f32(g1), bn1, stats1, training, out=gh, want_grads=True)
            ev = torch.cuda.current_stream().record_event() if side is not None else None

            add = g
            if kind == "conv":
                add = P.conv_rows(dR, packs[2][1], cin_p, T, ntaps=1, stride=stride, pad=0, transposed=True)
            if mask_in_dx:
                dx = P.gcn_fused(dy, packs[0][1], A32, cin_p, add=dout, add_mask=ctx.relu_mask)
            else:
                dx = P.gcn_fused(dy, packs[0][1], A32, cin_p, add=add)
            dA = None
            with _Fork(side, ev):
                dWp, dbg = P.gcn_wgrad(x, dy, A32, need_bias=True)
                dWg = dWp[:, :, :cin_true].reshape(Wg.shape)
                if ctx.needs_input_grad[1]:
                    dA = P.gcn_agrad(x, packs[0][0], f32(bg), dy)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        if CAPTURE is not None and CAPTURE_TENSORS:
            CAPTURE[-1].update(dy=dy, dx=dx, dA=dA, add=add)
        return (dx, dA, dWg, dbg, dg1, db1, dWt, dbt, dg2, db2, dWr, dbr_f, dgr, dbetar,
                None, None, None, None, None, None, None, None)


class EdgeImportanceFn(torch.autograd.Function):
    """A * importance_i for every block at once (net/st_gcn.py:81-82 of the reference multiplies per block): one
    launch forward and one backward (stgcn_mul_many) instead of one multiply per block each way.  Returns one [K,V,V]
    float32 tensor per block."""

    @staticmethod
    def forward(ctx, A, *importances):
        ctx.save_for_backward(A)
        a = A.detach().float().contiguous()
        imps = [p.detach().float().contiguous() for p in importances]
        out = P.mul_many(a, imps) if a.is_cuda else None
        if out is None:
            out = a.unsqueeze(0) * torch.stack(imps)
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        (A,) = ctx.saved_tensors
        if all(g is None for g in grads):
            return (None,) * (1 + len(grads))
        a = A.detach().float().contiguous()
        gl = [None if g is None else g.float().contiguous() for g in grads]
        d = P.mul_many(a, gl) if a.is_cuda else None
        if d is None:
            zero = torch.zeros_like(a)
            d = torch.stack([zero if g is None else g for g in gl]) * a.unsqueeze(0)
        return (None,) + tuple(d.unbind(0))


class CrossEntropyFn(torch.autograd.Function):
    """Mean cross-entropy of the logits (nn.CrossEntropyLoss() with its defaults, processor/recognition.py:44, 91): loss
    and d loss / d logits from one launch."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, dlogits = P.cross_entropy(logits.detach().float().contiguous(), labels.contiguous())
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss()(logits, labels) for float32 logits [N, ncls] on the GPU and int64 labels; anything else goes
    to torch's own."""
    if logits.is_cuda and logits.dim() == 2 and logits.dtype == torch.float32 and labels.dtype == torch.int64 and labels.dim() == 1 \
            and logits.shape[0] <= 8192:
        return CrossEntropyFn.apply(logits, labels)
    return torch.nn.functional.cross_entropy(logits, labels)


class DataBnFn(torch.autograd.Function):
    """x [N,C,T,V,M] float32 -> channels-last activation [N*M, Cp, T, V] after BatchNorm1d(V*C)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, dtype, cp):
        P.check_device(x, "input")
        x = x.float().contiguous()
        N, Cc, T, V, M = x.shape
        dev = x.device
        stats = None
        if training:
            stats = P.stats_alloc((2 * V * Cc,), dev)
            L.call("stgcn_data_bn_stats", L.ptr(x), L.ptr(stats), N, Cc, T, V, M, L.stream_ptr())
        bn = P.bn_finalize(stats, N * M * T, gamma.detach().float().contiguous(), beta.detach().float().contiguous(),
                           rmean, rvar, training)
        out = P.new_act(N * M, cp, T, V, dtype, dev)
        L.call("stgcn_data_bn_apply", L.dtype_code(dtype), L.ptr(x), L.ptr(out), L.ptr(bn.scale), L.ptr(bn.shift),
               N, Cc, T, V, M, cp, L.stream_ptr())
        ctx.save_for_backward(x, gamma)
        ctx.bn = bn
        ctx.training = training
        ctx.dims = (N, Cc, T, V, M, cp)
        return out

    @staticmethod
    def backward(ctx, dx0):
        x, gamma = ctx.saved_tensors
        N, Cc, T, V, M, cp = ctx.dims
        dx0 = P.cl(dx0)
        stats = P.stats_alloc((2 * V * Cc,), x.device)
        L.call("stgcn_data_bn_bwd", L.dtype_code(dx0.dtype), L.ptr(dx0), L.ptr(x), L.ptr(ctx.bn.mean), L.ptr(ctx.bn.rstd), L.ptr(stats),
               N, Cc, T, V, M, cp, L.stream_ptr())
        dg, db = P.stats_to_grads(stats, V * Cc)
        dx = None
        if ctx.needs_input_grad[0]:
            # Gradient w.r.t. the raw clip (saliency maps, adversarial inputs): never needed by a training step, so it is
            # spelled in tensor ops on the 3-channel input instead of a kernel.  Channel index of data_bn is v*C + c.
            ch = lambda t: t.detach().float().reshape(1, V, Cc, 1, 1).permute(0, 2, 3, 1, 4)          # -> [1,C,1,V,1]
            dy = dx0[:, :Cc].float().reshape(N, M, Cc, T, V).permute(0, 2, 3, 4, 1)                   # [N,C,T,V,M]
            scale = ch(gamma) * ch(ctx.bn.rstd)
            if ctx.training:
                n = float(N * M * T)
                xhat = (x - ch(ctx.bn.mean)) * ch(ctx.bn.rstd)
                dx = scale * (dy - ch(db) / n - xhat * (ch(dg) / n))
            else:
                dx = scale * dy
        return dx, dg, db, None, None, None, None, None


class HeadFn(torch.autograd.Function):
    """pooled [N*M, C] float32 (mean over (T,V) of the last block) -> logits [N, num_class]:
    mean over the M persons, then the 1x1 fcn."""

    @staticmethod
    def forward(ctx, pooled, W, b, N, M):
        P.check_device(pooled, "input")
        pooled = pooled.float().contiguous()
        Bc, Cc = pooled.shape
        ncls = W.shape[0]
        dev = pooled.device
        Wf = W.detach().reshape(ncls, Cc).float().contiguous()
        logits = torch.empty((N, ncls), dtype=torch.float32, device=dev)
        L.call("stgcn_head_fc_fwd", L.ptr(pooled), L.ptr(Wf), L.ptr(b.detach().float().contiguous()), L.ptr(logits),
               N, M, Cc, ncls, L.stream_ptr())
        ctx.save_for_backward(pooled, Wf)
        ctx.meta = (N, M, Cc, ncls, W.shape)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        pooled, Wf = ctx.saved_tensors
        N, M, Cc, ncls, wshape = ctx.meta
        dev = pooled.device
        dlogits = dlogits.float().contiguous()
        dW = torch.empty((ncls, Cc), dtype=torch.float32, device=dev)
        db = torch.empty((ncls,), dtype=torch.float32, device=dev)
        dpooled = torch.empty((N * M, Cc), dtype=torch.float32, device=dev)     # every person gets 1/M of its clip's gradient
        L.call("stgcn_head_fc_bwd", L.ptr(dlogits), L.ptr(pooled), L.ptr(Wf), L.ptr(dW), L.ptr(db), None, L.ptr(dpooled), N, M, Cc, ncls, L.stream_ptr())
        return dpooled, dW.reshape(wshape), db, None, None
