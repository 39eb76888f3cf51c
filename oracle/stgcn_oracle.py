"""CPU oracle for the ST-GCN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a checker, not a product path: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The shipped modules never do, and
they raise when the HIP library is missing instead of falling back to this code.

It restates, with explicit forward AND backward formulas written from the math
(no autograd, no nn.Conv2d / nn.BatchNorm / einsum-of-the-reference), what these
reference functions compute:

  graph partitioning           net/utils/graph.py:25-147
  graph convolution            net/utils/tgcn.py:57-66
  st_gcn block                 net/st_gcn.py:143-197
  whole model + head           net/st_gcn.py:29-116
  weights_init / SGD step      processor/recognition.py:21-33, 46-53, 84-97

Parity is pinned: tests/golden/*.npz hold outputs of the reference's own modules
(imported from /root/reference by tests/golden/make_golden.py in the build
container, torch 2.10 CPU, float64) and tests/test_oracle_golden.py checks every
function here against them.

All tensors are torch CPU tensors in the reference's logical layout
(x: [B, C, T, V]); dtype follows the inputs (float64 for pinning, float32 for
the timed CPU baseline).  The graph-convolution and temporal-convolution
functions allocate on their inputs' device, so that the full-size GPU tests can
evaluate these same float64 formulas on whole training-step tensors in seconds.
"""
from __future__ import annotations

import math
from collections import deque

import numpy as np
import torch

# --------------------------------------------------------------------------
# Graph  (net/utils/graph.py:25-147)
# --------------------------------------------------------------------------

_LAYOUTS = {
    # name: (num_node, 1-based?, bone list, centre joint (0-based))
    # net/utils/graph.py:42-50
    "openpose": (18, False,
                 [(4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9),
                  (9, 8), (11, 5), (8, 2), (5, 1), (2, 1), (0, 1), (15, 0),
                  (14, 0), (17, 15), (16, 14)], 1),
    # net/utils/graph.py:51-61
    "ntu-rgb+d": (25, True,
                  [(1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6),
                   (8, 7), (9, 21), (10, 9), (11, 10), (12, 11), (13, 1),
                   (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18),
                   (20, 19), (22, 23), (23, 8), (24, 25), (25, 12)], 20),
    # net/utils/graph.py:62-72
    "ntu_edge": (24, True,
                 [(1, 2), (3, 2), (4, 3), (5, 2), (6, 5), (7, 6), (8, 7),
                  (9, 2), (10, 9), (11, 10), (12, 11), (13, 1), (14, 13),
                  (15, 14), (16, 15), (17, 1), (18, 17), (19, 18), (20, 19),
                  (21, 22), (22, 8), (23, 24), (24, 12)], 2),
}


def graph_edges(layout):
    if layout not in _LAYOUTS:
        raise ValueError("Do Not Exist This Layout.")
    n, one_based, bones, centre = _LAYOUTS[layout]
    off = 1 if one_based else 0
    edge = [(i, i) for i in range(n)] + [(a - off, b - off) for a, b in bones]
    return n, edge, centre


def hop_distance(num_node, edge, max_hop):
    """Shortest-path hop count, inf beyond max_hop (graph.py:124-136), by BFS."""
    nbr = [[] for _ in range(num_node)]
    for i, j in edge:
        if i != j:
            nbr[i].append(j)
            nbr[j].append(i)
    hop = np.full((num_node, num_node), np.inf)
    for s in range(num_node):
        hop[s, s] = 0
        seen = {s}
        q = deque([(s, 0)])
        while q:
            u, d = q.popleft()
            if d == max_hop:
                continue
            for w in nbr[u]:
                if w not in seen:
                    seen.add(w)
                    hop[s, w] = d + 1
                    q.append((w, d + 1))
    return hop


def build_graph(layout="openpose", strategy="uniform", max_hop=1, dilation=1):
    """Returns dict(A[K,V,V] float64, edge, num_node, center, hop_dis)."""
    n, edge, centre = graph_edges(layout)
    hop = hop_distance(n, edge, max_hop)
    valid = list(range(0, max_hop + 1, dilation))
    reach = np.zeros((n, n))
    for h in valid:
        reach[hop == h] = 1.0
    colsum = reach.sum(axis=0)
    inv = np.where(colsum > 0, 1.0 / np.where(colsum > 0, colsum, 1.0), 0.0)
    norm = reach * inv[None, :]                    # A . D^-1  (graph.py:139-147)

    if strategy == "uniform":
        A = norm[None]
    elif strategy == "distance":
        A = np.stack([np.where(hop == h, norm, 0.0) for h in valid])
    elif strategy == "spatial":
        dc = hop[:, centre]                        # distance of each joint to centre
        # entry [j, i]: compare dc[j] with dc[i]   (graph.py:101-112)
        same = dc[:, None] == dc[None, :]
        farther = dc[:, None] > dc[None, :]
        parts = []
        for h in valid:
            at = hop == h
            root = np.where(at & same, norm, 0.0)
            close = np.where(at & ~same & farther, norm, 0.0)
            further = np.where(at & ~same & ~farther, norm, 0.0)
            if h == 0:
                parts.append(root)
            else:
                parts.append(root + close)
                parts.append(further)
        A = np.stack(parts)
    else:
        raise ValueError("Do Not Exist This Strategy")
    return dict(A=A, edge=edge, num_node=n, center=centre, hop_dis=hop)


# --------------------------------------------------------------------------
# Deterministic fills (no library RNG): shared by fixture generation and tests
# --------------------------------------------------------------------------

def det_uniform(shape, tag, lo=-1.0, hi=1.0):
    """Closed-form pseudo-random fill in [lo, hi): splitmix64 of (tag, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64)
             + np.uint64((tag * 0x9E3779B97F4A7C15) % (1 << 64)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


def dropout_keep_mask(numel, seed, p):
    """Keep-mask of the HIP dropout: one 32-bit hash serves elements 2q and 2q+1 (low / high 16 bits); an element is
    kept iff its 16 bits are >= round(p * 65536).

    Same integer hash as csrc (stgcn_common.h: drop_hash, drop_keep_bits).  Returns float64 {0,1}.
    """
    if p <= 0.0:
        return np.ones(numel)
    idx = np.arange(numel, dtype=np.uint64)
    pair = idx >> np.uint64(1)
    with np.errstate(over="ignore"):
        x = (pair & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (pair >> np.uint64(32)).astype(np.uint32)
        x = x ^ np.uint32(seed & 0xFFFFFFFF)
        x = x + hi * np.uint32(0x9E3779B9)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    half = np.where((idx & np.uint64(1)) == 0, x & np.uint32(0xFFFF), x >> np.uint32(16))
    thr = max(1, min(int(p * 65536.0 + 0.5), 65535))
    return (half >= np.uint32(thr)).astype(np.float64)


# --------------------------------------------------------------------------
# Storage rounding.  A reduced-precision implementation keeps activations, weights and the gradients that travel
# between kernels in a 16-bit format and accumulates in float32.  Inside `with storage_rounding(torch.bfloat16):`
# the functions below round to that format at exactly those places (weights as MFMA operands, the projected z and
# A*importance as operands of the aggregation, every stored activation / gradient tensor) and evaluate everything
# else in the inputs' dtype (float64 in the tests).  What is left between such an evaluation and the kernels is the
# kernels' own arithmetic (float32 accumulation order); what separates it from the plain evaluation is the price of
# the storage format.  Outside the context nothing is rounded.
# --------------------------------------------------------------------------

_RND = None


class storage_rounding:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _RND
        self.prev, _RND = _RND, self.dtype
        return self

    def __exit__(self, *exc):
        global _RND
        _RND = self.prev
        return False


def _q(t):
    """t rounded to the storage format (and back to its own dtype); identity outside storage_rounding."""
    if _RND is None or t is None:
        return t
    return t.to(_RND).to(t.dtype)


# --------------------------------------------------------------------------
# Graph convolution  (net/utils/tgcn.py:57-66), t_kernel_size == 1 form
# --------------------------------------------------------------------------

def gcn_forward(x, A, W, b):
    """y[b,c,t,w] = sum_k sum_v (sum_ci W[kC+c,ci] x[b,ci,t,v] + bias[kC+c]) A[k,v,w].

    Evaluated aggregate-first (a different association than the reference).
    x [B,Cin,T,V]; A [K,V,V]; W [K*Cout, Cin]; b [K*Cout] or None.
    """
    K = A.shape[0]
    Cout = W.shape[0] // K
    Wk = W.reshape(K, Cout, -1)
    if _RND is not None:
        # projection first, as the kernels do: z_k = W_k x + b_k is rounded before it meets the (rounded) adjacency
        Wq, Aq = _q(Wk), _q(A)
        y = None
        for k in range(K):
            z = torch.matmul(Wq[k], x.permute(0, 2, 1, 3))                  # [B,T,Cout,V]
            if b is not None:
                z = z + b.reshape(K, Cout)[k][None, None, :, None]
            t = torch.matmul(_q(z), Aq[k])                                   # [B,T,Cout,W]
            y = t if y is None else y + t
        return y.permute(0, 2, 1, 3).contiguous()
    y = None
    for k in range(K):
        xa = torch.matmul(x, A[k])                          # [B,Cin,T,W]
        t = torch.matmul(Wk[k], xa.permute(0, 2, 1, 3))     # [B,T,Cout,W]
        y = t if y is None else y + t
    y = y.permute(0, 2, 1, 3)
    if b is not None:
        colsum = A.sum(dim=1)                               # [K, W]
        y = y + torch.matmul(b.reshape(K, Cout).t(), colsum)[None, :, None, :]
    return y.contiguous()


def gcn_backward(x, A, W, b, dy):
    """Returns dx, dW, db, dA for gcn_forward."""
    K = A.shape[0]
    Cout = W.shape[0] // K
    Wk = W.reshape(K, Cout, -1)
    dx = torch.zeros_like(x)
    dW = torch.zeros_like(Wk)
    dA = torch.zeros_like(A)
    db = None if b is None else torch.zeros(K, Cout, dtype=x.dtype, device=x.device)
    S = dy.sum(dim=(0, 2))                                  # [Cout, W]
    if _RND is not None:
        # the three kernels' operand roundings: dX projects dy through W_k^T first (rounded), then aggregates with
        # A_k^T; dW forms dz_k = dy A_k^T (rounded) and contracts it with x; dA contracts the rounded z_k with dy
        Wq, Aq = _q(Wk), _q(A)
        for k in range(K):
            u = torch.matmul(Wq[k].t(), dy.permute(0, 2, 1, 3))            # [B,T,Cin,W]
            dx += torch.matmul(_q(u), Aq[k].t()).permute(0, 2, 1, 3)
            dz = _q(torch.matmul(dy, Aq[k].t()))                             # [B,Cout,T,V]
            dW[k] = torch.tensordot(dz, x, dims=([0, 2, 3], [0, 2, 3]))
            z = torch.matmul(Wq[k], x.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
            if b is not None:
                z = z + b.reshape(K, Cout)[k][None, :, None, None]
                db[k] = dz.sum(dim=(0, 2, 3))
            dA[k] = torch.tensordot(_q(z), dy, dims=([0, 1, 2], [0, 1, 2]))
        return dx, dW.reshape(W.shape), (None if b is None else db.reshape(-1)), dA
    for k in range(K):
        dz = torch.matmul(dy, A[k].t())                     # [B,Cout,T,V]  grad of projected z_k
        dx += torch.matmul(Wk[k].t(), dz.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
        dW[k] = torch.tensordot(dz, x, dims=([0, 2, 3], [0, 2, 3]))
        z = torch.matmul(Wk[k], x.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)  # [B,Cout,T,V]
        dA[k] = torch.tensordot(z, dy, dims=([0, 1, 2], [0, 1, 2]))
        if b is not None:
            bk = b.reshape(K, Cout)[k]
            db[k] = dz.sum(dim=(0, 2, 3))
            dA[k] += torch.matmul(bk, S)[None, :]           # same for every source joint v
    return dx, dW.reshape(W.shape), (None if b is None else db.reshape(-1)), dA


# --------------------------------------------------------------------------
# Temporal convolution  (stock nn.Conv2d at net/st_gcn.py:162-168, 181-185)
# --------------------------------------------------------------------------

def tconv_out_len(T, kt, stride, pad, dil=1):
    return (T + 2 * pad - dil * (kt - 1) - 1) // stride + 1


def tconv_forward(h, W, b, stride=1, pad=None, dil=1):
    """o[b,co,t',v] = bias[co] + sum_ci sum_dt W[co,ci,dt] h[b,ci,s t' + dt*dil - pad, v].

    W [Cout,Cin,kt]; zero padding outside [0,T).
    """
    Cout, Cin, kt = W.shape
    pad = (kt - 1) // 2 if pad is None else pad
    B, _, T, V = h.shape
    To = tconv_out_len(T, kt, stride, pad, dil)
    o = torch.zeros(B, Cout, To, V, dtype=h.dtype, device=h.device)
    tp = torch.arange(To)
    for dt in range(kt):
        ti = stride * tp + dt * dil - pad
        ok = (ti >= 0) & (ti < T)
        if not bool(ok.any()):
            continue
        src = h[:, :, ti[ok], :]                             # [B,Cin,n,V]
        o[:, :, ok, :] += torch.matmul(W[:, :, dt], src.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    if b is not None:
        o += b[None, :, None, None]
    return o


def tconv_backward(h, W, do, stride=1, pad=None, dil=1, need_bias=True):
    Cout, Cin, kt = W.shape
    pad = (kt - 1) // 2 if pad is None else pad
    B, _, T, V = h.shape
    To = do.shape[2]
    dh = torch.zeros_like(h)
    dW = torch.zeros_like(W)
    tp = torch.arange(To)
    for dt in range(kt):
        ti = stride * tp + dt * dil - pad
        ok = (ti >= 0) & (ti < T)
        if not bool(ok.any()):
            continue
        g = do[:, :, ok, :]                                  # [B,Cout,n,V]
        src = h[:, :, ti[ok], :]
        dW[:, :, dt] = torch.tensordot(g, src, dims=([0, 2, 3], [0, 2, 3]))
        dh[:, :, ti[ok], :] += torch.matmul(W[:, :, dt].t(), g.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    db = do.sum(dim=(0, 2, 3)) if need_bias else None
    return dh, dW, db


# --------------------------------------------------------------------------
# Graph convolution with a temporal kernel  (net/utils/tgcn.py:44-66: Conv2d((t_kernel_size,1), stride, padding,
# dilation) to K*Cout channels, then the einsum over the K partitions)
# --------------------------------------------------------------------------

def gcn_temporal_forward(x, A, W, b, stride=1, pad=0, dil=1):
    """y[b,c,t',w] = sum_k sum_v z[b,kC+c,t',v] A[k,v,w],  z = tconv_forward(x, W, b).  W [K*Cout, Cin, kt]."""
    K = A.shape[0]
    z = tconv_forward(x, W, b, stride, pad, dil)            # [B, K*Cout, T', V]
    Cout = z.shape[1] // K
    y = None
    for k in range(K):
        t = torch.matmul(z[:, k * Cout:(k + 1) * Cout], A[k])
        y = t if y is None else y + t
    return y.contiguous()


def gcn_temporal_backward(x, A, W, b, dy, stride=1, pad=0, dil=1):
    """Returns dx, dW, db, dA for gcn_temporal_forward."""
    K = A.shape[0]
    z = tconv_forward(x, W, b, stride, pad, dil)
    Cout = z.shape[1] // K
    dz = torch.cat([torch.matmul(dy, A[k].t()) for k in range(K)], dim=1)       # [B, K*Cout, T', V]
    dA = torch.stack([torch.tensordot(z[:, k * Cout:(k + 1) * Cout], dy, dims=([0, 1, 2], [0, 1, 2])) for k in range(K)])
    dx, dW, db = tconv_backward(x, W, dz, stride, pad, dil, need_bias=b is not None)
    return dx, dW, db, dA


# --------------------------------------------------------------------------
# BatchNorm over all axes but dim 1 (torch defaults: eps 1e-5, momentum 0.1)
# --------------------------------------------------------------------------

def _bn_axes(x):
    return tuple(i for i in range(x.dim()) if i != 1)


def _bn_shape(x):
    return [1, -1] + [1] * (x.dim() - 2)


def bn_forward(x, gamma, beta, rmean, rvar, training, eps=1e-5, momentum=0.1):
    """Returns y, (mean, rstd, xhat-free cache) and UPDATED running stats (new tensors)."""
    sh = _bn_shape(x)
    if training:
        ax = _bn_axes(x)
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=ax)
        var = ((x - mean.reshape(sh)) ** 2).mean(dim=ax)     # biased, used to normalise
        new_rm = (1 - momentum) * rmean + momentum * mean
        new_rv = (1 - momentum) * rvar + momentum * var * (n / max(n - 1, 1))
    else:
        mean, var = rmean, rvar
        new_rm, new_rv = rmean, rvar
    rstd = 1.0 / torch.sqrt(var + eps)
    y = (x - mean.reshape(sh)) * (rstd * gamma).reshape(sh) + beta.reshape(sh)
    return y, (mean, rstd), (new_rm, new_rv)


def bn_backward(x, gamma, cache, dy, training):
    mean, rstd = cache
    sh = _bn_shape(x)
    ax = _bn_axes(x)
    xhat = (x - mean.reshape(sh)) * rstd.reshape(sh)
    dgamma = (dy * xhat).sum(dim=ax)
    dbeta = dy.sum(dim=ax)
    if training:
        n = x.numel() // x.shape[1]
        dx = (gamma * rstd).reshape(sh) * (dy - (dbeta / n).reshape(sh) - xhat * (dgamma / n).reshape(sh))
    else:
        dx = dy * (gamma * rstd).reshape(sh)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------
# st_gcn block  (net/st_gcn.py:143-197)
# --------------------------------------------------------------------------

def block_param_names(residual_kind):
    names = ["gcn.conv.weight", "gcn.conv.bias",
             "tcn.0.weight", "tcn.0.bias", "tcn.2.weight", "tcn.2.bias",
             "tcn.3.weight", "tcn.3.bias"]
    if residual_kind == "conv":
        names += ["residual.0.weight", "residual.0.bias", "residual.1.weight", "residual.1.bias"]
    return names


def block_residual_kind(cin, cout, stride, residual=True):
    if not residual:
        return "none"
    if cin == cout and stride == 1:
        return "identity"
    return "conv"


def block_forward(x, A, p, bufs, stride, residual_kind, training, keep=None, keep_scale=1.0,
                  h_mask=None, out_mask=None):
    """One st_gcn block.  p: dict of parameters (reference state-dict names, block-relative,
    conv weights in their 4-D torch shapes).  bufs: dict of running stats ('tcn.0.running_mean' ...).
    keep: optional dropout keep-mask [B,C,T',V] (1/0); output is scaled by keep_scale=1/(1-p).
    h_mask / out_mask: optional boolean tensors that REPLACE the two ReLU decisions (inner ReLU after
    the first BatchNorm, final ReLU).  A reduced-precision implementation decides elements whose
    pre-activation is within its rounding error of zero differently from float64; forcing its
    decisions makes forward and backward comparable element by element (the functions are then
    smooth in everything else).  Returns out, cache, new_bufs.
    """
    K = A.shape[0]
    Wg = p["gcn.conv.weight"].reshape(p["gcn.conv.weight"].shape[0], -1)
    y = _q(gcn_forward(x, A, Wg, p["gcn.conv.bias"]))        # stored; BatchNorm statistics are those of the stored values
    nb = {}
    h1, c1, (nb["tcn.0.running_mean"], nb["tcn.0.running_var"]) = bn_forward(
        y, p["tcn.0.weight"], p["tcn.0.bias"], bufs["tcn.0.running_mean"], bufs["tcn.0.running_var"], training)
    hmask = (h1 > 0) if h_mask is None else h_mask
    h = _q(h1 * hmask.to(h1.dtype))
    Wt = p["tcn.2.weight"][..., 0]                           # [C,C,9]
    o = _q(tconv_forward(h, _q(Wt), p["tcn.2.bias"], stride))
    q, c2, (nb["tcn.3.running_mean"], nb["tcn.3.running_var"]) = bn_forward(
        o, p["tcn.3.weight"], p["tcn.3.bias"], bufs["tcn.3.running_mean"], bufs["tcn.3.running_var"], training)
    if keep is not None:
        q = q * keep * keep_scale
    cache = dict(x=x, A=A, y=y, h1=h1, h=h, hmask=hmask, o=o, c1=c1, c2=c2, keep=keep, keep_scale=keep_scale)
    if residual_kind == "identity":
        s = q + x
    elif residual_kind == "conv":
        Wr = p["residual.0.weight"][..., 0]                  # [Cout,Cin,1]
        r = _q(tconv_forward(x, _q(Wr), p["residual.0.bias"], stride, pad=0))
        rb, cr, (nb["residual.1.running_mean"], nb["residual.1.running_var"]) = bn_forward(
            r, p["residual.1.weight"], p["residual.1.bias"],
            bufs["residual.1.running_mean"], bufs["residual.1.running_var"], training)
        cache.update(r=r, cr=cr)
        s = q + rb
    else:
        s = q
    omask = (s > 0) if out_mask is None else out_mask
    out = _q(s * omask.to(s.dtype))
    cache["out"], cache["omask"], cache["s"] = out, omask, s
    return out, cache, nb


def block_backward(dout, p, cache, stride, residual_kind, training):
    """Returns dx, dA, grads (dict with the same names as p)."""
    g = {}
    ds = dout * cache["omask"].to(dout.dtype)
    dq = ds if cache["keep"] is None else ds * cache["keep"] * cache["keep_scale"]
    do, g["tcn.3.weight"], g["tcn.3.bias"] = bn_backward(cache["o"], p["tcn.3.weight"], cache["c2"], dq, training)
    do = _q(do)
    h = cache["h"]
    Wt = p["tcn.2.weight"][..., 0]
    dh, dWt, g["tcn.2.bias"] = tconv_backward(h, _q(Wt), do, stride)
    g["tcn.2.weight"] = dWt[..., None]
    dh1 = _q(dh * cache["hmask"].to(dh.dtype))
    dy, g["tcn.0.weight"], g["tcn.0.bias"] = bn_backward(cache["y"], p["tcn.0.weight"], cache["c1"], dh1, training)
    dy = _q(dy)
    Wg = p["gcn.conv.weight"].reshape(p["gcn.conv.weight"].shape[0], -1)
    dx, dWg, g["gcn.conv.bias"], dA = gcn_backward(cache["x"], cache["A"], Wg, p["gcn.conv.bias"], dy)
    g["gcn.conv.weight"] = dWg.reshape(p["gcn.conv.weight"].shape)
    if residual_kind == "identity":
        dx = dx + _q(ds)
    elif residual_kind == "conv":
        dr, g["residual.1.weight"], g["residual.1.bias"] = bn_backward(
            cache["r"], p["residual.1.weight"], cache["cr"], ds, training)
        dr = _q(dr)
        Wr = p["residual.0.weight"][..., 0]
        dxr, dWr, g["residual.0.bias"] = tconv_backward(cache["x"], _q(Wr), dr, stride, pad=0)
        g["residual.0.weight"] = dWr[..., None]
        dx = dx + _q(dxr)
    return _q(dx), dA, g


# --------------------------------------------------------------------------
# Whole model  (net/st_gcn.py:29-116)
# --------------------------------------------------------------------------

BLOCK_SPECS = [  # (cin or None for in_channels, cout, stride, residual)   st_gcn.py:44-55
    (None, 64, 1, False), (64, 64, 1, True), (64, 64, 1, True), (64, 64, 1, True),
    (64, 128, 2, True), (128, 128, 1, True), (128, 128, 1, True),
    (128, 256, 2, True), (256, 256, 1, True), (256, 256, 1, True),
]


def model_state_shapes(in_channels, num_class, K, V, edge_importance_weighting=True):
    """Ordered {name: shape} of the reference Model.state_dict() (num_batches_tracked included)."""
    sd = {"A": (K, V, V)}

    def bn(prefix, c):
        sd[prefix + ".weight"] = (c,)
        sd[prefix + ".bias"] = (c,)
        sd[prefix + ".running_mean"] = (c,)
        sd[prefix + ".running_var"] = (c,)
        sd[prefix + ".num_batches_tracked"] = ()

    bn("data_bn", in_channels * V)
    for i, (cin, cout, stride, res) in enumerate(BLOCK_SPECS):
        cin = in_channels if cin is None else cin
        pre = "st_gcn_networks.%d." % i
        sd[pre + "gcn.conv.weight"] = (cout * K, cin, 1, 1)
        sd[pre + "gcn.conv.bias"] = (cout * K,)
        bn(pre + "tcn.0", cout)
        sd[pre + "tcn.2.weight"] = (cout, cout, 9, 1)
        sd[pre + "tcn.2.bias"] = (cout,)
        bn(pre + "tcn.3", cout)
        if block_residual_kind(cin, cout, stride, res) == "conv":
            sd[pre + "residual.0.weight"] = (cout, cin, 1, 1)
            sd[pre + "residual.0.bias"] = (cout,)
            bn(pre + "residual.1", cout)
    if edge_importance_weighting:
        for i in range(len(BLOCK_SPECS)):
            sd["edge_importance.%d" % i] = (K, V, V)
    sd["fcn.weight"] = (num_class, 256, 1, 1)
    sd["fcn.bias"] = (num_class,)
    return sd


def det_state(in_channels, num_class, A, edge_importance_weighting=True, dtype=torch.float64):
    """Deterministic library-RNG-free state dict with weights_init-like scales
    (processor/recognition.py:21-33: conv ~ 0.02 wide, BN weight ~ 1 +- 0.02), plus
    non-trivial biases / running stats / edge importance so every term is exercised."""
    K, V = A.shape[0], A.shape[1]
    shapes = model_state_shapes(in_channels, num_class, K, V, edge_importance_weighting)
    sd = {}
    for tag, (name, shp) in enumerate(shapes.items()):
        if name == "A":
            v = np.asarray(A, dtype=np.float64)
        elif name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        elif name.endswith("running_var"):
            v = det_uniform(shp, tag, 0.5, 1.5)
        elif name.endswith("running_mean"):
            v = det_uniform(shp, tag, -0.1, 0.1)
        elif name.startswith("edge_importance"):
            v = det_uniform(shp, tag, 0.5, 1.5)
        elif ".conv.weight" in name or "tcn.2.weight" in name or "residual.0.weight" in name or name == "fcn.weight":
            fan_in = int(np.prod(shp[1:]))
            v = det_uniform(shp, tag, -1.0, 1.0) * (1.5 / math.sqrt(fan_in))
        elif name.endswith(".weight"):                       # BN gamma
            v = det_uniform(shp, tag, 0.8, 1.2)
        else:                                                # biases, BN beta
            v = det_uniform(shp, tag, -0.1, 0.1)
        sd[name] = torch.tensor(v, dtype=dtype)
    return sd


def det_input(N, C, T, V, M, tag=777, dtype=torch.float64):
    return torch.tensor(det_uniform((N, C, T, V, M), tag, -1.0, 1.0), dtype=dtype)


def _block_view(sd, i):
    pre = "st_gcn_networks.%d." % i
    p = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre) and "running" not in k and "num_batches" not in k}
    b = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre) and "running" in k}
    return p, b


def model_forward(sd, x, training, dropout=0.0, seeds=None, return_feature=False, masks=None):
    """x [N,C,T,V,M] -> logits [N,num_class].  sd: full state dict (reference names).
    Returns logits, cache, new_buffers (dict of updated running stats, full names).
    seeds: per-block dropout seeds (blocks 1..9 use dropout when training and dropout>0).
    masks: optional {block index: (h_mask, out_mask)} forwarded to block_forward.
    """
    N, C, T, V, M = x.shape
    A = sd["A"]
    ew = "edge_importance.0" in sd
    newb = {}
    # data_bn over channel index v*C + c   (st_gcn.py:72-78)
    xp = x.permute(0, 4, 3, 1, 2).reshape(N * M, V * C, T)
    xb, cd, (newb["data_bn.running_mean"], newb["data_bn.running_var"]) = bn_forward(
        xp, sd["data_bn.weight"], sd["data_bn.bias"], sd["data_bn.running_mean"], sd["data_bn.running_var"], training)
    h = _q(xb.reshape(N, M, V, C, T).permute(0, 1, 3, 4, 2).reshape(N * M, C, T, V))
    caches = []
    for i, (cin, cout, stride, res) in enumerate(BLOCK_SPECS):
        cin = C if cin is None else cin
        kind = block_residual_kind(cin, cout, stride, res)
        p, b = _block_view(sd, i)
        Ai = A * sd["edge_importance.%d" % i] if ew else A
        keep, scale = None, 1.0
        if training and dropout > 0 and i > 0:               # block 0 never has dropout (st_gcn.py:43,45)
            To = (h.shape[2] - 1) // stride + 1
            # mask indexed by the channels-last flat offset ((b*T+t)*V+v)*C+c, as the HIP kernel does
            m = dropout_keep_mask(N * M * cout * To * V, seeds[i], dropout).reshape(N * M, To, V, cout)
            keep = torch.tensor(m, dtype=h.dtype).permute(0, 3, 1, 2)
            scale = 1.0 / (1.0 - dropout)
        hm, om = masks[i] if masks is not None and i in masks else (None, None)
        h, cache, nb = block_forward(h, Ai, p, b, stride, kind, training, keep, scale, hm, om)
        cache["kind"], cache["stride"] = kind, stride
        caches.append(cache)
        for k, v in nb.items():
            newb["st_gcn_networks.%d.%s" % (i, k)] = v
    Bc, Cf, Tf, Vf = h.shape
    if return_feature:                                       # extract_feature, st_gcn.py:94-116
        Wf = sd["fcn.weight"].reshape(sd["fcn.weight"].shape[0], -1)
        out = torch.matmul(Wf, h.permute(0, 2, 1, 3)).permute(0, 2, 1, 3) + sd["fcn.bias"][None, :, None, None]
        output = out.reshape(N, M, -1, Tf, Vf).permute(0, 2, 3, 4, 1)
        feature = h.reshape(N, M, Cf, Tf, Vf).permute(0, 2, 3, 4, 1)
        return output, feature
    pooled = h.mean(dim=(2, 3)).reshape(N, M, Cf).mean(dim=1)            # [N,256]
    Wf = sd["fcn.weight"].reshape(sd["fcn.weight"].shape[0], -1)
    logits = pooled @ Wf.t() + sd["fcn.bias"]
    cache = dict(x=x, xp=xp, cd=cd, blocks=caches, pooled=pooled, feat_shape=(Bc, Cf, Tf, Vf), dims=(N, C, T, V, M))
    return logits, cache, newb


def cross_entropy(logits, labels):
    """Mean CE (nn.CrossEntropyLoss default).  Returns loss, dlogits."""
    z = logits - logits.max(dim=1, keepdim=True).values
    lse = torch.log(torch.exp(z).sum(dim=1, keepdim=True))
    logp = z - lse
    n = logits.shape[0]
    loss = -logp[torch.arange(n), labels].mean()
    d = torch.exp(logp)
    d[torch.arange(n), labels] -= 1.0
    return loss, d / n


def model_backward(sd, cache, dlogits, training=True, input_grad=False):
    """Returns grads: dict name -> tensor for every parameter of sd; with input_grad, (grads, d loss / d x) with
    x the raw [N,C,T,V,M] clip tensor."""
    N, C, T, V, M = cache["dims"]
    Bc, Cf, Tf, Vf = cache["feat_shape"]
    g = {}
    Wf = sd["fcn.weight"].reshape(sd["fcn.weight"].shape[0], -1)
    g["fcn.weight"] = (dlogits.t() @ cache["pooled"]).reshape(sd["fcn.weight"].shape)
    g["fcn.bias"] = dlogits.sum(dim=0)
    dpooled = dlogits @ Wf                                                 # [N,256]
    dh = (dpooled / (M * Tf * Vf))[:, None, :, None, None].expand(N, M, Cf, Tf, Vf).reshape(Bc, Cf, Tf, Vf)
    ew = "edge_importance.0" in sd
    for i in reversed(range(len(BLOCK_SPECS))):
        bc = cache["blocks"][i]
        p, _ = _block_view(sd, i)
        dh, dA, gb = block_backward(dh, p, bc, bc["stride"], bc["kind"], training)
        for k, v in gb.items():
            if v is not None:
                g["st_gcn_networks.%d.%s" % (i, k)] = v
        if ew:
            g["edge_importance.%d" % i] = dA * sd["A"]
    dxb = dh.reshape(N, M, C, T, V).permute(0, 1, 4, 2, 3).reshape(N * M, V * C, T)
    dxp, g["data_bn.weight"], g["data_bn.bias"] = bn_backward(cache["xp"], sd["data_bn.weight"], cache["cd"], dxb, training)
    if input_grad:
        return g, dxp.reshape(N, M, V, C, T).permute(0, 3, 4, 2, 1).contiguous()
    return g


def sgd_nesterov_step(params, grads, moms, lr, momentum=0.9, weight_decay=1e-4, nesterov=True):
    """torch.optim.SGD update (recognition.py:46-53); moms holds momentum buffers or None
    (first step: buf = grad).  In-place on the dicts' tensors; returns moms."""
    for k in params:
        d = grads[k] + weight_decay * params[k]
        if moms.get(k) is None:
            moms[k] = d.clone()
        else:
            moms[k] = momentum * moms[k] + d
        step = d + momentum * moms[k] if nesterov else moms[k]
        params[k] = params[k] - lr * step
    return moms


def param_names(sd):
    return [k for k in sd if k != "A" and "running" not in k and "num_batches" not in k]
