"""Host side of the HIP path: torch tensors in, C-ABI kernel launches, torch tensors out.

Activations are torch tensors of logical shape [B, C, T, V] in torch.channels_last memory
format (physically [B][T][V][C]); parameters stay float32 in their reference shapes and are
re-packed (tiny tensors) per call.  PyTorch here is plumbing: device memory, streams, autograd
bookkeeping.  Every arithmetic step of the path is a kernel of libstgcn_hip.so.
"""
import contextlib
import ctypes

import torch

from . import lib as L

CL = torch.channels_last
_ws = {}

# Per-kernel timing (bench / profiling aid): when TIMING is a dict, every launch of a kernel whose name is
# in TIMING_NAMES (None = all) is bracketed by events on the current stream, keyed by kernel name + shape.
TIMING = None
TIMING_NAMES = None


def _launch(key, name, *args, entry=None):
    """`entry`: the entry point to call when it is not `name` (a variant timed under the plain entry point's name)."""
    if TIMING is None or (TIMING_NAMES is not None and name not in TIMING_NAMES):
        return L.call(entry or name, *args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    L.call(entry or name, *args)
    e1.record()
    TIMING.setdefault((name,) + tuple(key), []).append((e0, e1))


def timing_summary():
    """{key: (launches, mean ms, total ms)} of the recorded events (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for k, evs in (TIMING or {}).items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[k] = (len(ms), sum(ms) / len(ms), sum(ms))
    return out


def kstep(dtype):
    """Channel granularity of the MFMA kernels: one 32-byte k-step."""
    return 16 if dtype == torch.bfloat16 else 8


def kchunk(dtype):
    """Channel granularity of conv_rows / conv_wgrad: one 128-byte chunk."""
    return 64 if dtype == torch.bfloat16 else 32


def pad_to(n, m):
    return (n + m - 1) // m * m


def workspace(device, nbytes=96 << 20):
    """The device's partial-sum workspace, at least `nbytes` long (it only ever grows)."""
    key = (device.type, device.index)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


# The parameter-gradient entry points (conv_wgrad, gcn_wgrad, gcn_agrad) write partial sums into the workspace and
# add them up with a second, small launch.  Inside `with deferred_sums(device):` the second launches are collected
# and issued as one (a block's backward: one launch instead of three or four).  Every call is given a fresh piece of a
# larger workspace, of the same WS_BYTES the undeferred call sees, so the partials and therefore the results are the
# same bit for bit.
WS_BYTES = 96 << 20
DEFER_WS_BYTES = 4 * WS_BYTES
_DEFER = None


class _SumList:
    def __init__(self, device):
        self.device = device
        self.buf = ctypes.create_string_buffer(L.SUM_LIST_BYTES)
        self.count = ctypes.c_int.from_buffer(self.buf)          # the list's first field: sums waiting
        self.used = ctypes.c_size_t(0)
        self.off = 0

    def run(self):
        if self.count.value:
            L.call("stgcn_sum_list_run", ctypes.addressof(self.buf), L.stream_ptr())
        self.off = 0


@contextlib.contextmanager
def deferred_sums(device):
    global _DEFER
    if _DEFER is not None:                                       # nested: the outer scope issues the sums
        yield _DEFER
        return
    _DEFER = d = _SumList(device)
    try:
        yield d
        d.run()
    finally:
        _DEFER = None


def to_float32(src):
    """float32 copy of a float64 tensor of finished sums: inside deferred_sums the conversion rides in the list's
    launch (the copy is complete once the scope has closed), otherwise it is a launch of its own."""
    d = _DEFER
    if d is None or d.device != src.device or not src.is_contiguous():
        return src.float()
    if d.count.value > 7:
        d.run()
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    L.call("stgcn_sum_list_add_cast", ctypes.addressof(d.buf), L.ptr(src), L.ptr(dst), src.numel())
    return dst


def _two_stage(key, name, device, *args):
    """Calls `name`(*args, workspace, workspace_bytes, stream), or its _deferred form inside deferred_sums."""
    d = _DEFER
    if d is None or d.device != device:
        ws = workspace(device)
        return _launch(key, name, *args, L.ptr(ws), WS_BYTES, L.stream_ptr())
    ws = workspace(device, DEFER_WS_BYTES)
    if d.count.value > 6 or d.off + WS_BYTES > DEFER_WS_BYTES:   # a call adds at most two sums to a list of eight
        d.run()
    _launch(key, name, *args, ws.data_ptr() + d.off, WS_BYTES, ctypes.addressof(d.buf), ctypes.addressof(d.used), L.stream_ptr(),
            entry=name + "_deferred")
    d.off += (d.used.value + 255) // 256 * 256


# Zero-initialised float64 statistic buffers come out of one arena that is cleared with a single fill per training
# step (Model.forward calls stats_begin), instead of one tiny fill kernel per BatchNorm and direction.
_arena = {}


def stats_begin(device, capacity=1 << 17):
    key = (device.type, device.index)
    a = _arena.get(key)
    if a is None or a[0].numel() < capacity:
        a = _arena[key] = [torch.empty(capacity, dtype=torch.float64, device=device), 0]
    a[0].zero_()
    a[1] = 0


def stats_alloc(shape, device):
    """Zeroed float64 tensor of `shape`: a slice of the step's arena when one is open, else a fresh allocation."""
    n = 1
    for d in shape:
        n *= d
    a = _arena.get((device.type, device.index))
    if a is None or a[1] + n > a[0].numel():
        return torch.zeros(shape, dtype=torch.float64, device=device)
    out = a[0][a[1]:a[1] + n].view(shape)
    a[1] += (n + 15) // 16 * 16
    return out


def stats_end(device):
    """Closes the arena (later stats_alloc calls allocate on their own until the next stats_begin)."""
    _arena.pop((device.type, device.index), None)


def cl(x):
    return x.contiguous(memory_format=CL)


def new_act(B, Cc, T, V, dtype, device):
    return torch.empty((B, Cc, T, V), dtype=dtype, device=device, memory_format=CL)


def check_device(x, name="x"):
    """The C entry points launch on HIP's current device and on its current stream: a tensor of another GPU would be
    read through a foreign pointer, unordered against the work that produced it."""
    if not x.is_cuda:
        raise RuntimeError("%s must live on the GPU: the ST-GCN HIP path has no CPU fallback" % name)
    if x.device.index != torch.cuda.current_device():
        raise RuntimeError("%s lives on cuda:%d but the current device is cuda:%d: select the tensor's device first "
                           "(torch.cuda.set_device / `with torch.cuda.device(...)`); one process drives one GPU here"
                           % (name, x.device.index, torch.cuda.current_device()))


def check_act(x, name="x"):
    check_device(x, name)
    if x.dim() != 4:
        raise ValueError("%s must be [B, C, T, V]" % name)


def pad_channels(x, cp):
    """[B,C,T,V] -> [B,cp,T,V] channels-last, zero padded."""
    B, Cc, T, V = x.shape
    if Cc == cp:
        return cl(x)
    out = torch.empty((B, cp, T, V), dtype=x.dtype, device=x.device, memory_format=CL).zero_()
    out[:, :Cc] = x
    return out


# ----------------------------------------------------------------------------- weight packing
def pack_gcn_w(W, K, cin_p, dtype):
    """conv weight [K*Cout, Cin, 1, 1] -> [K][Cout][cin_p] (projection operand)."""
    Cin = W.shape[1]
    w = W.reshape(K, -1, Cin)
    if cin_p != Cin:
        w = torch.nn.functional.pad(w, (0, cin_p - Cin))
    return w.to(dtype).contiguous()


def pack_gcn_wt(W, K, cin_p, dtype):
    """conv weight [K*Cout, Cin, 1, 1] -> [K][cin_p][Cout] (input-gradient operand)."""
    Cin = W.shape[1]
    w = W.reshape(K, -1, Cin).permute(0, 2, 1)
    if cin_p != Cin:
        w = torch.nn.functional.pad(w, (0, 0, 0, cin_p - Cin))
    return w.to(dtype).contiguous()


def pack_conv_w(W, dtype):
    """conv weight [Cout, Cin, kt, 1] -> [kt][Cout][Cin]."""
    return W[..., 0].permute(2, 0, 1).to(dtype).contiguous()


def pack_conv_wt(W, dtype):
    """conv weight [Cout, Cin, kt, 1] -> [kt][Cin][Cout] (transposed form)."""
    return W[..., 0].permute(2, 1, 0).to(dtype).contiguous()


class WeightPacker:
    """Packs all convolution weights of a model with ONE kernel launch per refresh().

    specs: list of (weight tensor, kind, K, cin_p) with kind "gcn" (weight [K*Cout, Cin, 1, 1]) or "conv"
    (weight [Cout, Cin, kt, 1]; K and cin_p ignored).  get(weight) -> (w, wt): the tensors pack_gcn_w / pack_gcn_wt
    or pack_conv_w / pack_conv_wt would return.  The weights are read through their storage pointers: the packer
    must be rebuilt when a parameter is re-allocated (valid() tells)."""

    def __init__(self, specs, dtype, device):
        self.dtype = dtype
        self._ptrs = tuple(W.data_ptr() for W, *_ in specs)
        self._weights = [W for W, *_ in specs]
        total, rows, shapes = 0, [], []
        for W, kind, K, cin_p in specs:
            if kind == "gcn":
                G, Cout, Cin, Cinp = K, W.shape[0] // K, W.shape[1], cin_p
                sg, sco, sci = Cout * Cin, Cin, 1
            else:
                Cout, Cin, G = W.shape[0], W.shape[1], W.shape[2]
                Cinp = Cin
                sg, sco, sci = 1, Cin * G, G
            n = G * Cout * Cinp
            shapes.append((G, Cout, Cinp, total, total + n))
            rows.append([0, 0, 0, G, Cout, Cin, Cinp, sg, sco, sci])
            total += 2 * n
        self.buf = torch.empty(total, dtype=dtype, device=device)
        es = self.buf.element_size()
        self._views = {}
        for (W, *_), row, (G, Cout, Cinp, o_w, o_wt) in zip(specs, rows, shapes):
            n = G * Cout * Cinp
            row[0] = W.data_ptr()
            row[1] = self.buf.data_ptr() + o_w * es
            row[2] = self.buf.data_ptr() + o_wt * es
            self._views[W.data_ptr()] = (self.buf[o_w:o_w + n].view(G, Cout, Cinp), self.buf[o_wt:o_wt + n].view(G, Cinp, Cout))
        self.max_elems = max(r[3] * r[4] * r[6] for r in rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.ndesc = len(rows)

    def valid(self, specs):
        return self._ptrs == tuple(W.data_ptr() for W, *_ in specs)

    def refresh(self):
        L.call("stgcn_pack_weights", L.dtype_code(self.dtype), L.ptr(self.table), self.ndesc, self.max_elems, L.stream_ptr())

    def get(self, W):
        return self._views[W.data_ptr()]


# ----------------------------------------------------------------------------- kernel wrappers
def conv_rows(x, w_packed, Cout, Tout, bias=None, pro=None, epilogue=L.EPI_PLAIN, aux=None, ep=None, stats=None,
              ntaps=1, stride=1, pad=0, dil=1, transposed=False, out=None):
    B, Cin, Tin, V = x.shape
    if out is None:
        out = new_act(B, Cout, Tout, V, x.dtype, x.device)
    ps, pb = pro if pro is not None else (None, None)
    es, eb, em, er = ep if ep is not None else (None, None, None, None)
    _launch((str(x.dtype), "T" if transposed else "F", epilogue, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pro is not None),
            "stgcn_conv_rows", L.dtype_code(x.dtype), L.ptr(x), L.ptr(w_packed), L.ptr(out), L.ptr(bias), L.ptr(ps), L.ptr(pb),
           epilogue, L.ptr(aux), L.ptr(es), L.ptr(eb), L.ptr(em), L.ptr(er), L.ptr(stats),
           B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, 1 if transposed else 0, L.stream_ptr())
    return out


def conv_rows_bn(x, w_packed, Cout, Tout, stats_in, gamma, beta, rmean, rvar, count, bias=None, stats=None,
                 ntaps=1, stride=1, pad=0, dil=1, momentum=0.1, eps=1e-5):
    """Forward temporal convolution whose BatchNorm + ReLU prologue is finished inside the launch (train mode):
    stats_in = the double sums [2*Cin] of x over `count` rows.  -> out, BnParams of the prologue's BatchNorm."""
    B, Cin, Tin, V = x.shape
    out = new_act(B, Cout, Tout, V, x.dtype, x.device)
    bn = BnParams(Cin, x.device)
    _launch((str(x.dtype), "F", L.EPI_PLAIN, B, Tin, Tout, V, Cin, Cout, ntaps, stride, True),
            "stgcn_conv_rows", L.dtype_code(x.dtype), L.ptr(x), L.ptr(w_packed), L.ptr(out), L.ptr(bias),
            L.ptr(stats_in), L.ptr(gamma), L.ptr(beta), L.ptr(rmean), L.ptr(rvar), L.ptr(bn.mean), float(count), float(momentum), float(eps),
            L.ptr(stats), B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil, L.stream_ptr(), entry="stgcn_conv_rows_bn")
    return out, bn


def conv_wgrad(x, dout, ntaps, stride, pad, dil=1, pro=None):
    """-> dW [Cout, Cin, ntaps, 1] float32."""
    B, Cin, Tin, V = x.shape
    Cout, Tout = dout.shape[1], dout.shape[2]
    dW = torch.empty((Cout, Cin, ntaps, 1), dtype=torch.float32, device=x.device)
    ps, pb = pro if pro is not None else (None, None)
    _two_stage((str(x.dtype), B, Tin, Tout, V, Cin, Cout, ntaps, stride), "stgcn_conv_wgrad", x.device,
               L.dtype_code(x.dtype), L.ptr(x), L.ptr(dout), L.ptr(dW), L.ptr(ps), L.ptr(pb),
               B, Tin, Tout, V, Cin, Cout, ntaps, stride, pad, dil)
    return dW


def gcn_fused(x, w_packed, adj, Nout, bias=None, add=None, stats=None, adj_t=False, add_mask=None):
    """x [B,Cin,T,V] -> [B,Nout,T,V];  adj [K][a][b] float32 (out joint a, in joint b), or [K][b][a] with adj_t.
    add_mask: the bit mask block_out_fwd wrote for `add`'s positions; masked-out elements of add count as zero."""
    B, Cin, T, V = x.shape
    K = adj.shape[0]
    out = new_act(B, Nout, T, V, x.dtype, x.device)
    _launch((str(x.dtype), B * T, V, Cin, Nout, K, add is not None), "stgcn_gcn_fused", L.dtype_code(x.dtype), L.ptr(x), L.ptr(w_packed), L.ptr(adj), 1 if adj_t else 0, L.ptr(bias), L.ptr(out), L.ptr(add), L.ptr(add_mask),
           L.ptr(stats), B * T, V, Cin, Nout, K, L.stream_ptr())
    return out


def _kgroups(K):
    return [(k0, min(3, K - k0)) for k0 in range(0, K, 3)]


def gcn_agrad(x, w_packed, bias, dy):
    """-> dA_eff [K,V,V] float32.  w_packed [K][Cout][Cin_p]; bias [K*Cout] or None."""
    B, Cin, T, V = x.shape
    K, Cout = w_packed.shape[0], w_packed.shape[1]
    dA = torch.empty((K, V, V), dtype=torch.float32, device=x.device)
    for k0, kn in _kgroups(K):
        bk = None if bias is None else bias[k0 * Cout:(k0 + kn) * Cout]
        _two_stage((str(x.dtype), B * T, V, Cin, Cout, kn), "stgcn_gcn_agrad", x.device,
                   L.dtype_code(x.dtype), L.ptr(x), L.ptr(w_packed[k0:k0 + kn]), L.ptr(bk), L.ptr(dy), L.ptr(dA[k0:k0 + kn]),
                   B * T, V, Cin, Cout, kn)
    return dA


def gcn_wgrad(x, dy, adj, need_bias=True):
    """-> dW [K,Cout,Cin_p] float32, db [K*Cout] float32.  adj = A_eff [K][v][w]."""
    B, Cin, T, V = x.shape
    Cout = dy.shape[1]
    K = adj.shape[0]
    dW = torch.empty((K, Cout, Cin), dtype=torch.float32, device=x.device)
    db = torch.empty((K * Cout,), dtype=torch.float32, device=x.device) if need_bias else None
    for k0, kn in _kgroups(K):
        dbk = None if db is None else db[k0 * Cout:(k0 + kn) * Cout]
        _two_stage((str(x.dtype), B * T, V, Cin, Cout, kn), "stgcn_gcn_wgrad", x.device,
                   L.dtype_code(x.dtype), L.ptr(x), L.ptr(dy), L.ptr(adj[k0:k0 + kn]), L.ptr(dW[k0:k0 + kn]), L.ptr(dbk),
                   B * T, V, Cin, Cout, kn)
    return dW, db


class BnParams:
    """mean / rstd / scale / shift of one BatchNorm evaluation (float32 [C] each)."""
    __slots__ = ("mean", "rstd", "scale", "shift")

    def __init__(self, Cc, device):
        buf = torch.empty((4, Cc), dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_finalize(stats, count, gamma, beta, rmean, rvar, training, momentum=0.1, eps=1e-5):
    Cc = gamma.numel()
    bp = BnParams(Cc, gamma.device)
    L.call("stgcn_bn_finalize", L.ptr(stats), float(count), L.ptr(gamma), L.ptr(beta), L.ptr(rmean), L.ptr(rvar),
           float(momentum), float(eps), L.ptr(bp.mean), L.ptr(bp.rstd), L.ptr(bp.scale), L.ptr(bp.shift), Cc, 1 if training else 0,
           L.stream_ptr())
    return bp


def stats_to_grads(stats, Cc):
    dg = torch.empty((Cc,), dtype=torch.float32, device=stats.device)
    dbt = torch.empty((Cc,), dtype=torch.float32, device=stats.device)
    L.call("stgcn_stats_to_grads", L.ptr(stats), L.ptr(dg), L.ptr(dbt), Cc, L.stream_ptr())
    return dg, dbt


SEED_STEP_MUL = 0x85EBCA6B


def block_out_fwd(o, res, bn2, bnr, drop_p, seed, want_mask=False, seed_step=None):
    """-> out, or (out, mask) with want_mask: 16 bits per 16-byte piece of out, bit j = element j positive, bit 8+j =
    element j kept by dropout.  seed_step: optional integer tensor on the device (a step counter); the kernel draws
    with seed + seed_step * SEED_STEP_MUL (mod 2^32)."""
    B, Cc, T, V = o.shape
    out = new_act(B, Cc, T, V, o.dtype, o.device)
    mask = torch.empty((B * T * V * Cc * o.element_size() // 16,), dtype=torch.int16, device=o.device) if want_mask else None
    _launch((str(o.dtype), B * T * V, Cc, res is not None, bnr is not None), "stgcn_block_out_fwd", L.dtype_code(o.dtype), L.ptr(o), L.ptr(res), L.ptr(out), L.ptr(mask), L.ptr(bn2.scale), L.ptr(bn2.shift),
           L.ptr(bnr.scale) if bnr is not None else None, L.ptr(bnr.shift) if bnr is not None else None,
           B * T * V, Cc, float(drop_p), int(seed) & 0xFFFFFFFF, L.ptr(seed_step), L.stream_ptr())
    return (out, mask) if want_mask else out


def block_out_fwd_bn(o, res, stats2, gamma2, beta2, rm2, rv2, statsr, gammar, betar, rmr, rvr, drop_p, seed, seed_step=None,
                     momentum=0.1, eps=1e-5):
    """Train-mode output stage with its BatchNorms finished in the launch: stats2 / statsr are the double sums
    [2*C] of o / res (statsr None: res, if any, is added as it is).  -> out, mask, bn2, bnr (BnParams or None)."""
    B, Cc, T, V = o.shape
    out = new_act(B, Cc, T, V, o.dtype, o.device)
    mask = torch.empty((B * T * V * Cc * o.element_size() // 16,), dtype=torch.int16, device=o.device)
    bn2 = BnParams(Cc, o.device)
    bnr = BnParams(Cc, o.device) if statsr is not None else None
    has_r = statsr is not None
    _launch((str(o.dtype), B * T * V, Cc, res is not None, has_r), "stgcn_block_out_fwd", L.dtype_code(o.dtype), L.ptr(o), L.ptr(res), L.ptr(out), L.ptr(mask),
            L.ptr(stats2), L.ptr(gamma2), L.ptr(beta2), L.ptr(rm2), L.ptr(rv2), L.ptr(bn2.mean),
            L.ptr(statsr), L.ptr(gammar) if has_r else None, L.ptr(betar) if has_r else None, L.ptr(rmr) if has_r else None,
            L.ptr(rvr) if has_r else None, L.ptr(bnr.mean) if has_r else None, float(momentum), float(eps),
            B * T * V, Cc, float(drop_p), int(seed) & 0xFFFFFFFF, L.ptr(seed_step), L.stream_ptr(), entry="stgcn_block_out_fwd_bn")
    return out, mask, bn2, bnr


def block_out_bwd(dout, out, o, r, bn2, gamma2, bnr, gammar, want_g, drop_p, seed, training, dpool=None, relu_mask=None,
                  want_bn_grads=False, dpool_scale=1.0):
    """-> dO, dR (or None), g (or None), stats2 [2,C] (sum gd, sum gd*ohat), statsr, dbias_o, dbias_r (float32 [C]);
    with want_bn_grads additionally (dgamma2, dbeta2, dgammar, dbetar) as float32, written by the apply pass.
    The incoming gradient is dout (stored tensor) and/or dpool [B,C] float32 times dpool_scale, constant over (t,v).  The ReLU
    decisions are read from relu_mask (as written by block_out_fwd) when given, else from out."""
    B, Cc, T, V = o.shape
    rows = B * T * V
    acc = stats_alloc((6, Cc), o.device)
    stats2, statsr = acc[0:2], acc[2:4]
    dbo, dbr = acc[4], acc[5]                             # zeroed double sums
    has_r = r is not None
    bng = torch.empty((4, Cc), dtype=torch.float32, device=o.device) if want_bn_grads else None
    args = [L.dtype_code(o.dtype), 0, L.ptr(dout), L.ptr(dpool), T * V, float(dpool_scale), L.ptr(out) if relu_mask is None else None, L.ptr(relu_mask), L.ptr(o), L.ptr(r), L.ptr(bn2.mean), L.ptr(bn2.rstd), L.ptr(gamma2),
            L.ptr(bnr.mean) if has_r else None, L.ptr(bnr.rstd) if has_r else None, L.ptr(gammar) if has_r else None,
            L.ptr(stats2), L.ptr(statsr) if has_r else None, None, None, None, L.ptr(dbo), L.ptr(dbr) if has_r else None,
            L.ptr(bng[0]) if want_bn_grads else None, L.ptr(bng[1]) if want_bn_grads else None,
            L.ptr(bng[2]) if want_bn_grads and has_r else None, L.ptr(bng[3]) if want_bn_grads and has_r else None,
            rows, Cc, float(drop_p), int(seed) & 0xFFFFFFFF, float(rows), 1 if training else 0, L.stream_ptr()]
    _launch((str(o.dtype), 0, rows, Cc, has_r), "stgcn_block_out_bwd", *args)
    dO = new_act(B, Cc, T, V, o.dtype, o.device)
    dR = new_act(B, Cc, T, V, o.dtype, o.device) if has_r else None
    g = new_act(B, Cc, T, V, o.dtype, o.device) if want_g else None
    args[1] = 1
    args[18], args[19], args[20] = L.ptr(dO), L.ptr(dR), L.ptr(g)
    _launch((str(o.dtype), 1, rows, Cc, has_r, want_g), "stgcn_block_out_bwd", *args)
    # the bias sums leave this function (they become .grad of the conv biases): float32 memory of their own, not views
    # of the statistics arena, which the next training forward clears
    dbf = to_float32(acc[4:6])
    dbo, dbr = dbf[0], dbf[1]
    res = (dO, dR, g, stats2, (statsr if has_r else None), dbo, (dbr if has_r else None))
    if want_bn_grads:
        res = res + ((bng[0], bng[1], bng[2] if has_r else None, bng[3] if has_r else None),)
    return res


def bn_bwd_apply(g, y, gamma, bn, stats, training, out=None, want_grads=False):
    """-> dy; with want_grads (dy, dgamma, dbeta): the parameter gradients are the sums in `stats`, converted to
    float32 by the same launch."""
    B, Cc, T, V = y.shape
    if out is None:
        out = new_act(B, Cc, T, V, y.dtype, y.device)
    pg = torch.empty((2, Cc), dtype=torch.float32, device=y.device) if want_grads else None
    _launch((str(y.dtype), B * T * V, Cc), "stgcn_bn_bwd_apply", L.dtype_code(y.dtype), L.ptr(g), L.ptr(y), L.ptr(out), L.ptr(gamma), L.ptr(bn.mean), L.ptr(bn.rstd),
           L.ptr(stats), float(B * T * V), L.ptr(pg[0]) if want_grads else None, L.ptr(pg[1]) if want_grads else None,
           B * T * V, Cc, 1 if training else 0, L.stream_ptr())
    return (out, pg[0], pg[1]) if want_grads else out


def mul_many(a, inputs):
    """-> [n, *a.shape] float32 with out[i] = a * inputs[i] (None: zeros), one launch; None when the inputs are not what
    the kernel takes (more than 16, another dtype, device or layout): the caller then multiplies with torch."""
    n = len(inputs)
    ok = lambda t: t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == a.shape and t.device == a.device)
    if not (1 <= n <= 16 and a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and all(ok(t) for t in inputs)):
        return None
    arr = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in inputs])
    out = torch.empty((n,) + tuple(a.shape), dtype=torch.float32, device=a.device)
    L.call("stgcn_mul_many", L.ptr(a), ctypes.addressof(arr), n, L.ptr(out), a.numel(), L.stream_ptr())
    return out


def cross_entropy(logits, labels):
    """Mean cross-entropy of float32 logits [N, ncls] and int64 labels [N] -> (loss [], d loss / d logits [N, ncls])."""
    N, ncls = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty((N, ncls), dtype=torch.float32, device=logits.device)
    L.call("stgcn_cross_entropy", L.ptr(logits), L.ptr(labels), L.ptr(loss), L.ptr(dlogits), N, ncls, L.stream_ptr())
    return loss, dlogits


def sgd_step(params, grads, mom, lr, momentum, weight_decay, nesterov, first_step, grad_scale=1.0):
    L.call("stgcn_sgd_step", L.ptr(params), L.ptr(grads), L.ptr(mom), params.numel(), float(lr), float(momentum), float(weight_decay),
           1 if nesterov else 0, 1 if first_step else 0, float(grad_scale), L.stream_ptr())


def augment_batch(x, window=None, shift=None, move=None):
    """Batched input augmentation on the GPU: x [N,C,T,V,M] float32 -> [N,C,window or T,V,M].

    shift: int tensor/sequence [N], output frame t shows input frame t + shift[n] (zeros outside) -- the draw of
    feeder.tools.choose_shift per clip.  move: (a, s, tx, ty), each [N, Tout] (angles in radians) -- the draws of
    feeder.tools.move_params per clip; the transformation itself runs in the kernel."""
    check_f32 = x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
    if not check_f32:
        raise RuntimeError("augment_batch wants a contiguous float32 CUDA tensor [N,C,T,V,M]")
    N, Cc, T, V, M = x.shape
    Tout = int(window) if window else T
    out = torch.empty((N, Cc, Tout, V, M), dtype=torch.float32, device=x.device)
    sh = None
    if shift is not None:
        sh = torch.as_tensor(shift, dtype=torch.int32).to(x.device).contiguous()
        if sh.numel() != N:
            raise ValueError("shift wants one entry per clip")
    ca = sa = tx = ty = None
    if move is not None:
        a, s, tx_, ty_ = [torch.as_tensor(v, dtype=torch.float64).reshape(N, Tout) for v in move]
        ca = (torch.cos(a) * s).float().to(x.device).contiguous()
        sa = (torch.sin(a) * s).float().to(x.device).contiguous()
        tx = tx_.float().to(x.device).contiguous()
        ty = ty_.float().to(x.device).contiguous()
    L.call("stgcn_augment", L.ptr(x), L.ptr(out), L.ptr(sh), L.ptr(ca), L.ptr(sa), L.ptr(tx), L.ptr(ty),
           N, Cc, T, Tout, V * M, L.stream_ptr())
    return out
