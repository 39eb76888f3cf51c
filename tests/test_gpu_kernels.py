"""GPU parity tests of the individual HIP kernels, called through the C-ABI, against the CPU
oracle on the same seeded inputs.  Tolerances (max-abs error relative to the max-abs of the
oracle result): float32 1e-4, bfloat16 2e-2 -- the bars the task states."""
import numpy as np
import pytest
import torch

from oracle import stgcn_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m gpu on the MI355X box)")
    from stgcn_rt import lib, ops
    lib.load()
    return ops


def t64(shape, tag, lo=-1.0, hi=1.0):
    return torch.tensor(O.det_uniform(shape, tag, lo, hi), dtype=torch.float64)


def to_dev(x, dtype):
    """[B,C,T,V] float64 CPU -> channels-last device tensor."""
    return x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)


def q(x, dtype):
    """What the kernel actually sees: the oracle input rounded to the storage dtype."""
    return x.to(dtype).to(torch.float64)


def rel(a, b):
    a = a.detach().to("cpu", torch.float64)
    b = b.detach().to("cpu", torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def conv_paths(dtype, width):
    """The launch paths a temporal-convolution test runs: the default choice and, for bf16 with 128 / 256 output
    channels, the whole-width-tile kernel forced onto the small shape (by default it only takes grids that fill the
    chip; the full-size tests reach it that way)."""
    import contextlib
    import os

    @contextlib.contextmanager
    def env(force):
        old = os.environ.pop("STGCN_CONV_WIDE_ANY_GRID", None)
        if force:
            os.environ["STGCN_CONV_WIDE_ANY_GRID"] = "1"
        try:
            yield
        finally:
            os.environ.pop("STGCN_CONV_WIDE_ANY_GRID", None)
            if old is not None:
                os.environ["STGCN_CONV_WIDE_ANY_GRID"] = old

    forced = dtype == torch.bfloat16 and width in (128, 256)
    return [env(f) for f in ((False, True) if forced else (False,))]


def test_probe_tr16(rt):
    from stgcn_rt import lib as L
    out = torch.zeros(64 * 8, dtype=torch.int16, device=DEV)
    L.call("stgcn_probe_tr16", out.data_ptr(), L.stream_ptr())
    got = out.cpu().numpy().reshape(64, 8)
    for lane in range(64):
        h, col = lane >> 5, lane & 31
        exp = [((8 * h + j) << 8) | col for j in range(8)]
        assert list(got[lane]) == exp, (lane, list(got[lane]), exp)


CONV_CASES = [
    # B, Cin, Cout, T, V, ntaps, stride, pad, prologue
    (2, 64, 64, 20, 25, 9, 1, 4, True),
    (1, 128, 128, 37, 25, 9, 2, 4, True),
    (2, 64, 128, 19, 18, 1, 2, 0, False),
    (1, 128, 64, 12, 18, 9, 1, 4, False),
    (1, 256, 60, 6, 25, 1, 1, 0, False),
    (1, 64, 64, 300, 25, 9, 1, 4, True),
    (4, 128, 256, 12, 25, 1, 2, 0, False),
    (4, 256, 256, 6, 25, 9, 1, 4, True),
    (3, 64, 128, 150, 18, 9, 2, 4, True),
    (2, 128, 64, 61, 25, 9, 2, 4, False),
    (3, 128, 128, 130, 25, 9, 1, 4, True),
    (2, 256, 400, 10, 18, 1, 1, 0, False),       # per-position classifier of Kinetics: ragged last column tile
    (2, 128, 80, 20, 25, 9, 1, 4, True),
    (2, 64, 40, 33, 25, 9, 1, 4, True),
    # strided 1-tap (the residual branch): 64-column passes, 256 input channels, odd frame count, several blocks per wave
    (2, 128, 64, 21, 25, 1, 2, 0, False),
    (3, 256, 128, 30, 25, 1, 2, 0, False),
    (40, 64, 128, 299, 25, 1, 2, 0, False),
    (2, 64, 192, 13, 25, 1, 2, 0, False),            # three 64-column passes: outside the streaming kernel's domain
    # stride-2 forward with its taps in two parity groups: a sample shorter than one tile's window, several tiles per sample
    (2, 128, 128, 9, 25, 9, 2, 4, True),
    (1, 256, 256, 150, 25, 9, 2, 4, True),
    (2, 64, 128, 46, 18, 5, 2, 2, False),
    # whole-width tiles (128 / 256 output channels, input stride 1): several tiles per sample and per workgroup, ragged
    # last tile, both joint counts, input width different from the output width, fewer taps; their stride-2 relatives
    # reach the same kernel through the per-parity launches of the transposed form
    (3, 256, 256, 40, 18, 9, 1, 4, True),
    (2, 128, 256, 75, 25, 9, 1, 4, False),
    (5, 256, 128, 47, 25, 5, 1, 2, True),
    (300, 64, 128, 9, 25, 9, 1, 4, True),
    (2, 256, 256, 61, 25, 9, 2, 4, False),
    (2, 128, 128, 150, 18, 7, 2, 3, True),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_rows_forward(rt, case, dtype):
    B, Cin, Cout, T, V, kt, stride, pad, pro = case
    x = q(t64((B, Cin, T, V), 11), dtype)
    W = q(t64((Cout, Cin, kt), 12) * (1.5 / np.sqrt(Cin * kt)), dtype)
    b = t64((Cout,), 13, -0.2, 0.2)
    sc, sh = t64((Cin,), 14, 0.5, 1.5), t64((Cin,), 15, -0.3, 0.3)
    h = torch.clamp_min(x * sc[None, :, None, None] + sh[None, :, None, None], 0) if pro else x
    if pro and dtype == torch.bfloat16:
        h = q(h, dtype)                       # the kernel rounds the activated input before the MFMA
    ref = O.tconv_forward(h, W, b, stride, pad)
    To = ref.shape[2]
    wp = rt.pack_conv_w(W[..., None].to(DEV).float(), dtype)
    for path in conv_paths(dtype, Cout):
        with path:
            stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
            out = rt.conv_rows(to_dev(x, dtype), wp, Cout, To, bias=b.to(DEV).float(),
                               pro=(sc.to(DEV).float(), sh.to(DEV).float()) if pro else None, stats=stats,
                               ntaps=kt, stride=stride, pad=pad)
            torch.cuda.synchronize()
        assert out.shape == ref.shape
        assert rel(out, ref) < TOL[dtype]
        s = stats.cpu()
        n = B * To * V
        assert rel(s[:Cout] / n, ref.mean(dim=(0, 2, 3))) < TOL[dtype] * 2 + 1e-6
        assert rel(s[Cout:] / n, (ref ** 2).mean(dim=(0, 2, 3))) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("with_res", [True, False], ids=["res", "nores"])
@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[1], CONV_CASES[5], CONV_CASES[7], CONV_CASES[8], CONV_CASES[10], CONV_CASES[12]],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_rows_add_relu_epilogue(rt, case, with_res, dtype):
    """out = relu(conv(relu(scale * x + shift)) + bias + residual): the output stage of an eval-mode block with its
    BatchNorms folded into weights and bias, in each of the three convolution kernels (LDS ring, tiled, generic)."""
    from stgcn_rt import lib as L
    B, Cin, Cout, T, V, kt, stride, pad, pro = case
    x = q(t64((B, Cin, T, V), 41), dtype)
    W = q(t64((Cout, Cin, kt), 42) * (1.5 / np.sqrt(Cin * kt)), dtype)
    b = t64((Cout,), 43, -0.2, 0.2)
    sc, sh = t64((Cin,), 44, 0.5, 1.5), t64((Cin,), 45, -0.3, 0.3)
    h = torch.clamp_min(x * sc[None, :, None, None] + sh[None, :, None, None], 0) if pro else x
    if pro and dtype == torch.bfloat16:
        h = q(h, dtype)
    o = O.tconv_forward(h, W, b, stride, pad)
    res = q(t64(tuple(o.shape), 46), dtype) if with_res else None
    ref = torch.clamp_min(o + (res if with_res else 0), 0)
    wp = rt.pack_conv_w(W[..., None].to(DEV).float(), dtype)
    for path in conv_paths(dtype, Cout):
        with path:
            out = rt.conv_rows(to_dev(x, dtype), wp, Cout, o.shape[2], bias=b.to(DEV).float(),
                               pro=(sc.to(DEV).float(), sh.to(DEV).float()) if pro else None, epilogue=L.EPI_ADD_RELU,
                               aux=to_dev(res, dtype) if with_res else None, ntaps=kt, stride=stride, pad=pad)
            torch.cuda.synchronize()
        assert out.shape == ref.shape and float(out.min()) >= 0
        assert rel(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
# the transposed form contracts over Cout: whole 128-byte chunks only
@pytest.mark.parametrize("case", [c for c in CONV_CASES[:4] + CONV_CASES[6:] if c[2] % 64 == 0], ids=lambda c: "x".join(map(str, c)))
def test_conv_rows_transposed_and_wgrad(rt, case, dtype):
    B, Cin, Cout, T, V, kt, stride, pad, pro = case
    x = q(t64((B, Cin, T, V), 21), dtype)
    W = q(t64((Cout, Cin, kt), 22) * (1.5 / np.sqrt(Cin * kt)), dtype)
    sc, sh = t64((Cin,), 24, 0.5, 1.5), t64((Cin,), 25, -0.3, 0.3)
    h = torch.clamp_min(x * sc[None, :, None, None] + sh[None, :, None, None], 0) if pro else x
    if pro and dtype == torch.bfloat16:
        h = q(h, dtype)
    To = O.tconv_out_len(T, kt, stride, pad)
    do = q(t64((B, Cout, To, V), 23), dtype)
    dh, dW, _ = O.tconv_backward(h, W, do, stride, pad)
    # data gradient through the transposed form
    wt = rt.pack_conv_wt(W[..., None].to(DEV).float(), dtype)
    for path in conv_paths(dtype, Cin):
        with path:
            got = rt.conv_rows(to_dev(do, dtype), wt, Cin, T, ntaps=kt, stride=stride, pad=pad, transposed=True)
            torch.cuda.synchronize()
        assert rel(got, dh) < TOL[dtype]
    # weight gradient
    gW = rt.conv_wgrad(to_dev(x, dtype), to_dev(do, dtype), kt, stride, pad,
                       pro=(sc.to(DEV).float(), sh.to(DEV).float()) if pro else None)
    assert rel(gW[..., 0], dW) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 21, 25), (3, 128, 45, 25), (2, 256, 33, 18), (260, 256, 5, 25)], ids=lambda c: "x".join(map(str, c)))
def test_conv_rows_relu_bwd_epilogue(rt, shape, dtype):
    from stgcn_rt import lib as L
    B, Cc, T, V = shape
    kt = 9
    y = q(t64((B, Cc, T, V), 31), dtype)
    do = q(t64((B, Cc, T, V), 32), dtype)
    W = q(t64((Cc, Cc, kt), 33) * 0.05, dtype)
    gamma, beta = t64((Cc,), 34, 0.8, 1.2), t64((Cc,), 35, -0.2, 0.2)
    mean = y.mean(dim=(0, 2, 3))
    rstd = 1.0 / torch.sqrt(y.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    scale, shift = gamma * rstd, beta - mean * gamma * rstd
    sh4 = lambda v: v[None, :, None, None]
    pre = y * sh4(scale) + sh4(shift)
    dh, _, _ = O.tconv_backward(torch.clamp_min(pre, 0), W, do, 1, 4)
    g = dh * (pre > 0)
    xhat = (y - sh4(mean)) * sh4(rstd)
    f = lambda v: v.to(DEV).float().contiguous()
    for path in conv_paths(dtype, Cc):
        with path:
            stats = torch.zeros(2 * Cc, dtype=torch.float64, device=DEV)
            got = rt.conv_rows(to_dev(do, dtype), rt.pack_conv_wt(W[..., None].to(DEV).float(), dtype), Cc, T,
                               epilogue=L.EPI_RELU_BWD, aux=to_dev(y, dtype), ep=(f(scale), f(shift), f(mean), f(rstd)),
                               stats=stats, ntaps=kt, stride=1, pad=4, transposed=True)
            torch.cuda.synchronize()
        assert rel(got, g) < TOL[dtype]
        s = stats.cpu()
        assert rel(s[:Cc], g.sum(dim=(0, 2, 3))) < TOL[dtype] * 3
        assert rel(s[Cc:], (g * xhat).sum(dim=(0, 2, 3))) < TOL[dtype] * 3


GCN_DIRECT = [
    # layout, strategy, dense, B, Cin, Cout, T
    ("ntu-rgb+d", "spatial", False, 2, 64, 64, 12),
    ("ntu-rgb+d", "spatial", True, 1, 64, 128, 10),
    ("openpose", "spatial", True, 2, 128, 128, 9),
    ("openpose", "uniform", False, 2, 16, 64, 11),
    ("ntu-rgb+d", "distance", True, 1, 256, 256, 5),
    ("ntu-rgb+d", "spatial", True, 1, 128, 256, 7),
    ("ntu-rgb+d", "spatial", False, 4, 128, 256, 12),
    ("ntu-rgb+d", "spatial", False, 4, 256, 256, 6),
    ("ntu-rgb+d", "spatial", True, 4, 48, 80, 600),      # odd widths (any-width paths), several batches per workgroup
    ("openpose", "distance", True, 3, 64, 64, 301),
    ("ntu-rgb+d", "spatial", True, 2, 256, 80, 7),         # 8-wave workgroups with a ragged last column tile
    ("openpose", "spatial", False, 3, 128, 208, 9),
    ("ntu-rgb+d", "spatial", False, 8, 256, 256, 150),     # weights-in-registers form: several batches of frames per workgroup
    ("openpose", "uniform", True, 5, 128, 192, 131),       # a column group with spare waves, frame count not a multiple of the batch
    ("ntu-rgb+d", "spatial", True, 2, 96, 64, 7),          # input widths above 64 that are no multiple of 64: the weight
    ("openpose", "distance", False, 1, 80, 48, 9),         # gradient splits them into groups of 48 / 16 channels
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", GCN_DIRECT, ids=lambda c: "-".join(map(str, c)))
def test_gcn_kernels(rt, case, dtype):
    lay, strat, dense, B, Cin, Cout, T = case
    A0 = O.build_graph(lay, strat)["A"]
    K, V = A0.shape[0], A0.shape[1]
    A = t64(A0.shape, 41, -0.5, 0.5) if dense else torch.tensor(A0) * t64(A0.shape, 41, 0.5, 1.5)
    x = q(t64((B, Cin, T, V), 42), dtype)
    W = q(t64((K * Cout, Cin), 43) * (1.5 / np.sqrt(Cin)), dtype)
    b = t64((K * Cout,), 44, -0.2, 0.2)
    dy = q(t64((B, Cout, T, V), 45), dtype)
    Aq = q(A, dtype)                     # the kernels feed A to the MFMA in the storage dtype
    y = O.gcn_forward(x, Aq, W, b)
    dx, dW, db, dA = O.gcn_backward(x, Aq, W, b, dy)

    Ad = A.to(DEV).float()
    W4 = W.reshape(K * Cout, Cin, 1, 1).to(DEV).float()
    bd = b.to(DEV).float()
    xd, dyd = to_dev(x, dtype), to_dev(dy, dtype)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
    yk = rt.gcn_fused(xd, rt.pack_gcn_w(W4, K, Cin, dtype), Ad.contiguous(), Cout, bias=bd, stats=stats, adj_t=True)
    tol = TOL[dtype]
    assert rel(yk, y) < tol
    n = B * T * V
    assert rel(stats.cpu()[:Cout] / n, y.mean(dim=(0, 2, 3))) < 2 * tol + 1e-6
    assert rel(stats.cpu()[Cout:] / n, (y ** 2).mean(dim=(0, 2, 3))) < 2 * tol
    dxk = rt.gcn_fused(dyd, rt.pack_gcn_wt(W4, K, Cin, dtype), Ad.contiguous(), Cin)
    assert rel(dxk, dx) < tol
    if Cin % (16 // xd.element_size()) == 0:
        # addend masked by a block_out_fwd-style bit mask (16 bits per 16-byte piece, low byte = keep element j)
        vec = 16 // xd.element_size()
        addt = q(t64((B, Cin, T, V), 46), dtype)
        gen = torch.Generator().manual_seed(47)
        keep = torch.randint(0, 2, (B * T * V * Cin // vec, vec), generator=gen, dtype=torch.int32)
        mbits = ((keep << torch.arange(vec, dtype=torch.int32)).sum(1) | (0x5A << 8)).to(torch.int16).to(DEV)
        keep_t = keep.reshape(B, T, V, Cin).permute(0, 3, 1, 2).to(addt.dtype)
        dxm = rt.gcn_fused(dyd, rt.pack_gcn_wt(W4, K, Cin, dtype), Ad.contiguous(), Cin, add=to_dev(addt, dtype), add_mask=mbits)
        assert rel(dxm, dx + addt * keep_t) < tol
    dWk, dbk = rt.gcn_wgrad(xd, dyd, Ad.contiguous())
    assert rel(dWk.reshape(K * Cout, Cin), dW) < tol
    assert rel(dbk, db) < tol
    # dA: the projected activations are rounded to the storage dtype inside the kernel (bf16 only)
    dAk = rt.gcn_agrad(xd, rt.pack_gcn_w(W4, K, Cin, dtype), bd, dyd)
    assert rel(dAk, dA) < tol


def fenced(t, cl=False):
    """The same device tensor with NaN on both sides of its memory: a read past either end -- even one that only meets
    zero weights -- turns the output into NaN.  cl: a [B,C,T,V] activation in channels-last layout."""
    src = t.permute(0, 2, 3, 1).contiguous() if cl else t.contiguous()
    n, pad = src.numel(), 4096
    buf = torch.full((n + 2 * pad,), float("nan"), dtype=t.dtype, device=t.device)
    buf[pad:pad + n] = src.reshape(-1)
    v = buf[pad:pad + n].view(src.shape)
    return v.permute(0, 3, 1, 2) if cl else v


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("lay,strat,B,Cin,Cout,T", [("openpose", "uniform", 1, 16, 8, 5), ("ntu-rgb+d", "spatial", 2, 48, 24, 3),
                                                    ("ntu-rgb+d", "spatial", 1, 32, 80, 4), ("openpose", "distance", 1, 256, 40, 2),
                                                    ("ntu-rgb+d", "spatial", 1, 64, 64, 1), ("ntu-rgb+d", "spatial", 1, 64, 64, 3),
                                                    ("openpose", "spatial", 1, 128, 128, 2), ("ntu-rgb+d", "spatial", 3, 64, 128, 23)])
def test_gcn_kernels_stay_inside_their_inputs(rt, lay, strat, B, Cin, Cout, T, dtype):
    """Channel counts that are multiples of one k-step but not of a 32-column tile, every input fenced with NaN:
    results finite and equal to the oracle's (a ragged tile's spare columns meet zero weights -- they must not be read).
    The 64- and 128-channel cases with one to three frames are for the frame prefetch, which runs one or two frames
    ahead of a wave's last frame and must re-read a frame of the tensor, not what lies behind it."""
    if dtype == torch.bfloat16 and Cout % 16:
        Cout += 8                                # the gradient kernels contract over whole k-steps of Cout (16 in bf16)
    A0 = O.build_graph(lay, strat)["A"]
    K, V = A0.shape[0], A0.shape[1]
    A = torch.tensor(A0) * t64(A0.shape, 61, 0.5, 1.5)
    x = q(t64((B, Cin, T, V), 62), dtype)
    W = q(t64((K * Cout, Cin), 63) * (1.5 / np.sqrt(Cin)), dtype)
    b = t64((K * Cout,), 64, -0.2, 0.2)
    dy = q(t64((B, Cout, T, V), 65), dtype)
    Aq = q(A, dtype)
    y = O.gcn_forward(x, Aq, W, b)
    dx, dW, db, dA = O.gcn_backward(x, Aq, W, b, dy)
    Ad = fenced(A.to(DEV).float())
    W4 = W.reshape(K * Cout, Cin, 1, 1).to(DEV).float()
    bd = fenced(b.to(DEV).float())
    xd, dyd = fenced(to_dev(x, dtype), cl=True), fenced(to_dev(dy, dtype), cl=True)
    assert xd.is_contiguous(memory_format=torch.channels_last)
    wp, wt = fenced(rt.pack_gcn_w(W4, K, Cin, dtype)), fenced(rt.pack_gcn_wt(W4, K, Cin, dtype))
    tol = TOL[dtype]
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
    got = dict(y=(rt.gcn_fused(xd, wp, Ad, Cout, bias=bd, stats=stats, adj_t=True), y),
               dx=(rt.gcn_fused(dyd, wt, Ad, Cin), dx))
    if Cin % (16 // xd.element_size()) == 0:
        addt = q(t64((B, Cin, T, V), 66), dtype)
        got["dx_add"] = (rt.gcn_fused(dyd, wt, Ad, Cin, add=fenced(to_dev(addt, dtype), cl=True)), dx + addt)
    dWk, dbk = rt.gcn_wgrad(xd, dyd, Ad)
    got.update(dW=(dWk.reshape(K * Cout, Cin), dW), db=(dbk, db), dA=(rt.gcn_agrad(xd, wp, bd, dyd), dA),
               dA_nobias=(rt.gcn_agrad(xd, wp, None, dyd), O.gcn_backward(x, Aq, W, None, dy)[3]))
    for k, (g, want) in got.items():
        assert bool(torch.isfinite(g).all()), k
        assert rel(g, want) < tol, k
    assert bool(torch.isfinite(stats).all())


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(1, 64, 8, 7, 18, 3, 1, 1, 1), (2, 64, 24, 9, 25, 9, 2, 4, 1), (1, 128, 40, 6, 25, 1, 1, 0, 1),
                                  (1, 64, 64, 11, 18, 3, 2, 2, 2), (2, 64, 64, 10, 25, 5, 3, 0, 1),
                                  (2, 64, 128, 13, 25, 1, 2, 0, 1), (1, 128, 64, 8, 18, 1, 2, 0, 1),
                                  # 9 taps, stride 2, several tiles per sample with a ragged last one: the weight gradient's
                                  # rows in flight are fetched from clamped addresses
                                  (2, 64, 64, 41, 25, 9, 2, 4, 1), (1, 128, 128, 30, 18, 9, 2, 4, 1)], ids=lambda c: "x".join(map(str, c)))
def test_conv_kernels_stay_inside_their_inputs(rt, case, dtype):
    """The temporal convolution at narrow / ragged output widths, strides up to 3 and dilation, inputs fenced with NaN."""
    B, Cin, Cout, T, V, kt, stride, pad, dil = case
    x = q(t64((B, Cin, T, V), 71), dtype)
    W = q(t64((Cout, Cin, kt), 72) * (1.5 / np.sqrt(Cin * kt)), dtype)
    b = t64((Cout,), 73, -0.2, 0.2)
    ref = O.tconv_forward(x, W, b, stride, pad, dil)
    To = ref.shape[2]
    do = q(t64((B, Cout, To, V), 74), dtype)
    dh, dW, _ = O.tconv_backward(x, W, do, stride, pad, dil)
    xd, dod = fenced(to_dev(x, dtype), cl=True), fenced(to_dev(do, dtype), cl=True)
    W4 = W[..., None].to(DEV).float()
    out = rt.conv_rows(xd, fenced(rt.pack_conv_w(W4, dtype)), Cout, To, bias=fenced(b.to(DEV).float()), ntaps=kt, stride=stride, pad=pad, dil=dil)
    assert bool(torch.isfinite(out).all()) and rel(out, ref) < TOL[dtype]
    gW = rt.conv_wgrad(xd, dod, kt, stride, pad, dil)
    assert bool(torch.isfinite(gW).all()) and rel(gW[..., 0], dW) < TOL[dtype]
    if Cout % 64 == 0:                       # the transposed form contracts over Cout: whole 128-byte chunks only
        got = rt.conv_rows(dod, fenced(rt.pack_conv_wt(W4, dtype)), Cin, T, ntaps=kt, stride=stride, pad=pad, dil=dil, transposed=True)
        assert bool(torch.isfinite(got).all()) and rel(got, dh) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("Cc,rows_shape", [(64, (2, 20, 25)), (128, (1, 19, 18)), (256, (2, 5, 25)), (96, (1, 7, 25)), (64, (3, 300, 25)),
                                            (64, (128, 300, 25))])       # the benchmarked step's largest activation: 960 000 rows
def test_block_out_and_bn_kernels(rt, Cc, rows_shape, dtype):
    B, T, V = rows_shape
    o = q(t64((B, Cc, T, V), 51), dtype)
    res = q(t64((B, Cc, T, V), 52), dtype)
    r = q(t64((B, Cc, T, V), 53), dtype)
    dout = q(t64((B, Cc, T, V), 54), dtype)
    g2, b2 = t64((Cc,), 55, 0.8, 1.2), t64((Cc,), 56, -0.2, 0.2)
    gr, br = t64((Cc,), 57, 0.8, 1.2), t64((Cc,), 58, -0.2, 0.2)
    rows = B * T * V
    # small shapes run with NaN on both sides of every activation: a read past a tensor's rows shows in the output
    td = (lambda t, dt: fenced(to_dev(t, dt), cl=True)) if rows < 100000 else to_dev
    f = lambda v: v.to(DEV).float().contiguous()
    sh4 = lambda v: v[None, :, None, None]
    zeros = torch.zeros(Cc, dtype=torch.float64)
    ones = torch.ones(Cc, dtype=torch.float64)

    def stats_of(t):
        s = torch.cat([t.sum(dim=(0, 2, 3)), (t ** 2).sum(dim=(0, 2, 3))]).to(DEV)
        return s

    for kind, p in (("identity", 0.0), ("conv", 0.0), ("none", 0.0), ("identity", 0.5)):
        seed = 1234
        bn2 = rt.bn_finalize(stats_of(o), rows, f(g2), f(b2), f(zeros), f(ones), True)
        q2, c2, _ = O.bn_forward(o, g2, b2, zeros, ones, True)
        keep, ks = None, 1.0
        if p > 0:
            keep = torch.tensor(O.dropout_keep_mask(o.numel(), seed, p).reshape(B, T, V, Cc)).permute(0, 3, 1, 2)
            ks = 1.0 / (1 - p)
            q2 = q2 * keep * ks
        bnr = None
        if kind == "identity":
            s = q2 + res
            resd = td(res, dtype)
        elif kind == "conv":
            bnr = rt.bn_finalize(stats_of(r), rows, f(gr), f(br), f(zeros), f(ones), True)
            rb, cr, _ = O.bn_forward(r, gr, br, zeros, ones, True)
            s = q2 + rb
            resd = td(r, dtype)
        else:
            s = q2
            resd = None
        out_ref = torch.clamp_min(s, 0)
        out, mask = rt.block_out_fwd(td(o, dtype), resd, bn2, bnr, p, seed, want_mask=True)
        tol = TOL[dtype]
        assert rel(out, out_ref) < tol, (kind, p)
        assert bool(torch.isfinite(out).all())
        assert torch.equal(out, rt.block_out_fwd(td(o, dtype), resd, bn2, bnr, p, seed))
        # the mask: 16 bits per 16-byte piece of the channels-last output, bit j = element j positive,
        # bit 8+j = element j kept by dropout
        vec = 16 // out.element_size()
        pos = (out.permute(0, 2, 3, 1).reshape(-1, vec) > 0).to(torch.int32)
        sh = torch.arange(vec, device=pos.device, dtype=torch.int32)
        want = (pos << sh).sum(1)
        kp = torch.ones_like(pos) if keep is None else keep.permute(0, 2, 3, 1).reshape(-1, vec).to(pos.device, torch.int32)
        want = want | ((kp << sh).sum(1) << 8)
        assert torch.equal(mask.to(torch.int32) & 0xFFFF, want)

        # backward, evaluated at the kernel's own output so the ReLU mask is identical
        outq = out.detach().to("cpu", torch.float64)
        ds = dout * (outq > 0)
        dq = ds if keep is None else ds * keep * ks
        do_ref, dg2_ref, db2_ref = O.bn_backward(o, g2, c2, dq, True)
        dO, dR, g, stats2, statsr, dbo, dbr = rt.block_out_bwd(
            td(dout, dtype), out, td(o, dtype), td(r, dtype) if kind == "conv" else None,
            bn2, f(g2), bnr, f(gr) if kind == "conv" else None, kind == "identity", p, seed, True)
        via_mask = rt.block_out_bwd(
            td(dout, dtype), None, td(o, dtype), td(r, dtype) if kind == "conv" else None,
            bn2, f(g2), bnr, f(gr) if kind == "conv" else None, kind == "identity", p, seed, True, relu_mask=mask)
        assert torch.equal(via_mask[0], dO) and (g is None or torch.equal(via_mask[2], g))
        assert rel(dO, do_ref) < tol * 2, (kind, p)
        assert bool(torch.isfinite(dO).all()) and bool(torch.isfinite(stats2).all())
        assert rel(stats2.cpu()[1], dg2_ref) < tol * 2 and rel(stats2.cpu()[0], db2_ref) < tol * 2
        if kind == "identity":
            assert rel(g, ds) < tol
        if kind == "conv":
            dr_ref, dgr_ref, dbr_ref = O.bn_backward(r, gr, cr, ds, True)
            assert rel(dR, dr_ref) < tol * 2
            assert rel(statsr.cpu()[1], dgr_ref) < tol * 2

    # BN1 backward apply
    y = o
    gin = dout
    _, c1, _ = O.bn_forward(y, g2, b2, zeros, ones, True)
    dy_ref, dg_ref, db_ref = O.bn_backward(y, g2, c1, gin, True)
    bn1 = rt.bn_finalize(stats_of(y), rows, f(g2), f(b2), f(zeros), f(ones), True)
    sums = torch.cat([db_ref, dg_ref]).to(DEV)
    dy = rt.bn_bwd_apply(td(gin, dtype), td(y, dtype), f(g2), bn1, sums, True)
    assert rel(dy, dy_ref) < TOL[dtype] * 2 and bool(torch.isfinite(dy).all())


def test_bn_finalize_running_stats(rt):
    Cc, n = 64, 1000
    x = t64((n, Cc), 61) * 3 + 0.5
    stats = torch.cat([x.sum(0), (x ** 2).sum(0)]).to(DEV)
    rm, rv = t64((Cc,), 62, -0.1, 0.1), t64((Cc,), 63, 0.5, 1.5)
    g, b = t64((Cc,), 64, 0.8, 1.2), t64((Cc,), 65, -0.1, 0.1)
    rmd, rvd = rm.to(DEV).float(), rv.to(DEV).float()
    bn = rt.bn_finalize(stats, n, g.to(DEV).float(), b.to(DEV).float(), rmd, rvd, True)
    _, (mean, rstd), (nrm, nrv) = O.bn_forward(x, g, b, rm, rv, True)
    assert rel(bn.mean, mean) < 1e-5 and rel(bn.rstd, rstd) < 1e-5
    assert rel(rmd, nrm) < 1e-5 and rel(rvd, nrv) < 1e-5
    assert rel(bn.scale, g * rstd) < 1e-5 and rel(bn.shift, b - mean * g * rstd) < 1e-4
    bn_e = rt.bn_finalize(None, n, g.to(DEV).float(), b.to(DEV).float(), rmd, rvd, False)
    assert rel(bn_e.mean, nrm) < 1e-6


def test_sgd_kernel(rt):
    n = 100003
    p, g, m = t64((n,), 71), t64((n,), 72), t64((n,), 73)
    pd, gd, md = p.to(DEV).float(), g.to(DEV).float(), m.to(DEV).float()
    rt.sgd_step(pd, gd, md, 0.1, 0.9, 1e-4, True, False)
    prm, grd, mom = {"w": p.clone()}, {"w": g}, {"w": m.clone()}
    O.sgd_nesterov_step(prm, grd, mom, 0.1)
    assert rel(pd, prm["w"]) < 1e-6 and rel(md, mom["w"]) < 1e-6
    pd2, md2 = p.to(DEV).float(), torch.zeros(n, device=DEV)
    rt.sgd_step(pd2, gd, md2, 0.1, 0.9, 1e-4, True, True)
    prm, mom = {"w": p.clone()}, {}
    O.sgd_nesterov_step(prm, grd, mom, 0.1)
    assert rel(pd2, prm["w"]) < 1e-6 and rel(md2, mom["w"]) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_block_out_bwd_pooled_gradient(rt, dtype):
    """The pooled-gradient form (a float32 [B,C] vector, constant over (t,v)) equals the stored-tensor form."""
    B, Cc, T, V = 3, 128, 7, 25
    o = q(t64((B, Cc, T, V), 81), dtype)
    out = torch.clamp_min(q(t64((B, Cc, T, V), 82), dtype), 0)
    dp = t64((B, Cc), 83)
    g2 = t64((Cc,), 84, 0.8, 1.2)
    f = lambda v: v.to(DEV).float().contiguous()
    stats = torch.cat([o.sum(dim=(0, 2, 3)), (o ** 2).sum(dim=(0, 2, 3))]).to(DEV)
    bn2 = rt.bn_finalize(stats, B * T * V, f(g2), f(torch.zeros(Cc)), f(torch.zeros(Cc)), f(torch.ones(Cc)), True)
    dense = dp[:, :, None, None].expand(B, Cc, T, V)
    _, c2, _ = O.bn_forward(o, g2, torch.zeros(Cc, dtype=torch.float64), torch.zeros(Cc, dtype=torch.float64), torch.ones(Cc, dtype=torch.float64), True)
    do_ref, dg_ref, db_ref = O.bn_backward(o, g2, c2, dense * (out > 0), True)
    dO, _, _, stats2, _, _, _ = rt.block_out_bwd(None, to_dev(out, dtype), to_dev(o, dtype), None, bn2, f(g2), None, None,
                                                 False, 0.0, 1, True, dpool=f(dp))
    tol = TOL[dtype]
    assert rel(dO, do_ref) < tol * 2
    assert rel(stats2.cpu()[1], dg_ref) < tol * 2 and rel(stats2.cpu()[0], db_ref) < tol * 2


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("Cc,V,T,stride", [(64, 25, 40, 1), (128, 18, 37, 2), (256, 25, 12, 1)], ids=lambda v: str(v))
def test_deferred_sums_equal_immediate_sums(rt, Cc, V, T, stride, dtype):
    """The _deferred entry points plus ONE stgcn_sum_list_run give the gradients of the two-stage entry points bit for
    bit (same partials, same order of additions), whatever else waits on the list; the float64 -> float32 copy that
    rides on the list equals .float(); a ninth entry is refused with status 5 and leaves the list as it was."""
    import ctypes
    from stgcn_rt import lib as L
    B, K = 3, 3
    To = (T - 1) // stride + 1
    x = to_dev(t64((B, Cc, T, V), 9100), dtype)
    dO = to_dev(t64((B, Cc, To, V), 9101), dtype)
    dy = to_dev(t64((B, Cc, T, V), 9102), dtype)
    sc = t64((Cc,), 9103, 0.5, 1.5).float().to(DEV)
    sh = t64((Cc,), 9104, -0.2, 0.2).float().to(DEV)
    adj = t64((K, V, V), 9105, 0.0, 0.3).float().to(DEV)
    Wg = t64((K * Cc, Cc, 1, 1), 9106, -0.1, 0.1).float().to(DEV)
    bg = t64((K * Cc,), 9107, -0.1, 0.1).float().to(DEV)
    wp = rt.pack_gcn_w(Wg, K, Cc, dtype)
    sums64 = torch.tensor(O.det_uniform((2, Cc), 9108, -1e3, 1e3), dtype=torch.float64, device=DEV)

    def run_all():
        dWt = rt.conv_wgrad(x, dO, 9, stride, 4, pro=(sc, sh))
        dWr = rt.conv_wgrad(x, dO, 1, stride, 0)
        dWp, dbp = rt.gcn_wgrad(x, dy, adj)
        dA = rt.gcn_agrad(x, wp, bg, dy)
        f = rt.to_float32(sums64)
        return [dWt, dWr, dWp, dbp, dA, f]

    want = run_all()
    with rt.deferred_sums(torch.device(DEV)) as d:
        got = run_all()
        assert d.count.value == (6 if dtype == torch.float32 else 5)       # the float32 graph-conv form sums W and b apart
        assert d.off > 0
    torch.cuda.synchronize()
    assert d.count.value == 0
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert torch.equal(want[5], sums64.float())

    # a full list refuses the next entry and keeps its eight
    buf = ctypes.create_string_buffer(L.SUM_LIST_BYTES)
    dst = torch.zeros((9, 2, Cc), dtype=torch.float32, device=DEV)
    for i in range(8):
        L.call("stgcn_sum_list_add_cast", ctypes.addressof(buf), sums64.data_ptr(), dst[i].data_ptr(), sums64.numel())
    rc = L.load().stgcn_sum_list_add_cast(ctypes.addressof(buf), sums64.data_ptr(), dst[8].data_ptr(), sums64.numel())
    assert rc == 5 and ctypes.c_int.from_buffer(buf).value == 8
    L.call("stgcn_sum_list_run", ctypes.addressof(buf), L.stream_ptr())
    torch.cuda.synchronize()
    assert ctypes.c_int.from_buffer(buf).value == 0
    for i in range(8):
        assert torch.equal(dst[i], sums64.float())
    assert not dst[8].any()


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("Cc,shape,with_r", [(64, (2, 20, 25), False), (128, (3, 19, 18), True), (256, (2, 5, 25), True), (96, (1, 7, 25), False)],
                         ids=lambda v: str(v))
def test_block_out_fwd_with_batchnorm_in_the_launch(rt, Cc, shape, with_r, dtype):
    """stgcn_block_out_fwd_bn (coefficients derived from the sums inside the launch) against stgcn_bn_finalize followed
    by stgcn_block_out_fwd: same output, same mask, same mean / rstd / scale / shift and running statistics, bit for bit."""
    B, T, V = shape
    rows = B * T * V
    o = to_dev(t64((B, Cc, T, V), 9301, -2, 2), dtype)
    res = to_dev(t64((B, Cc, T, V), 9302, -2, 2), dtype)
    f = lambda tag, lo, hi: t64((Cc,), tag, lo, hi).float().to(DEV)
    g2, b2, gr, br = f(9303, 0.5, 1.5), f(9304, -0.3, 0.3), f(9305, 0.5, 1.5), f(9306, -0.3, 0.3)

    def sums(x):
        x64 = x.double()
        return torch.stack([x64.sum(dim=(0, 2, 3)), (x64 * x64).sum(dim=(0, 2, 3))]).contiguous()

    st2, str_ = sums(o), sums(res)
    run = lambda: [f(9307, -0.1, 0.1), f(9308, 0.5, 1.5), f(9309, -0.1, 0.1), f(9310, 0.5, 1.5)]
    rm2, rv2, rmr, rvr = run()
    bn2 = rt.bn_finalize(st2.reshape(-1), rows, g2, b2, rm2, rv2, True)
    bnr = rt.bn_finalize(str_.reshape(-1), rows, gr, br, rmr, rvr, True) if with_r else None
    want, want_mask = rt.block_out_fwd(o, res, bn2, bnr, 0.5, 77, want_mask=True)
    qm2, qv2, qmr, qvr = run()
    got, got_mask, cn2, cnr = rt.block_out_fwd_bn(o, res, st2.reshape(-1), g2, b2, qm2, qv2,
                                                  str_.reshape(-1) if with_r else None, gr if with_r else None, br if with_r else None,
                                                  qmr if with_r else None, qvr if with_r else None, 0.5, 77)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got_mask, want_mask)
    for a, b in ((cn2, bn2), (cnr, bnr)):
        if b is not None:
            for name in ("mean", "rstd", "scale", "shift"):
                assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(qm2, rm2) and torch.equal(qv2, rv2)
    if with_r:
        assert torch.equal(qmr, rmr) and torch.equal(qvr, rvr)
    else:
        assert cnr is None
    # and the statistics themselves against float64
    mean64 = o.double().mean(dim=(0, 2, 3)).cpu()
    assert rel(cn2.mean, mean64) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(2, 64, 64, 20, 25, 9, 1), (3, 256, 256, 40, 18, 9, 1), (2, 128, 128, 37, 25, 9, 2), (1, 128, 64, 12, 18, 9, 1),
                                  (2, 64, 40, 33, 25, 9, 1), (2, 128, 256, 75, 25, 5, 1)], ids=lambda c: "x".join(map(str, c)))
def test_conv_rows_with_batchnorm_in_the_launch(rt, case, dtype):
    """stgcn_conv_rows_bn (prologue coefficients derived from the sums inside the launch, in each of the four forward
    convolution kernels) against stgcn_bn_finalize followed by stgcn_conv_rows: output, output statistics, mean / rstd /
    scale / shift and running statistics bit for bit."""
    B, Cin, Cout, T, V, kt, stride = case
    pad = (kt - 1) // 2
    To = O.tconv_out_len(T, kt, stride, pad)
    x = to_dev(t64((B, Cin, T, V), 9401, -2, 2), dtype)
    W = t64((Cout, Cin, kt), 9402) * (1.5 / np.sqrt(Cin * kt))
    wp = rt.pack_conv_w(W[..., None].to(DEV).float(), dtype)
    f = lambda tag, lo, hi: t64((Cin,), tag, lo, hi).float().to(DEV)
    gamma, beta, bias = f(9403, 0.5, 1.5), f(9404, -0.3, 0.3), t64((Cout,), 9405, -0.2, 0.2).float().to(DEV)
    x64 = x.double()
    sums = torch.cat([x64.sum(dim=(0, 2, 3)), (x64 * x64).sum(dim=(0, 2, 3))]).contiguous()
    rows = B * T * V
    for path in conv_paths(dtype, Cout):
        with path:
            rm, rv = f(9406, -0.1, 0.1), f(9407, 0.5, 1.5)
            bn = rt.bn_finalize(sums, rows, gamma, beta, rm, rv, True)
            st = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
            want = rt.conv_rows(x, wp, Cout, To, bias=bias, pro=(bn.scale, bn.shift), stats=st, ntaps=kt, stride=stride, pad=pad)
            qm, qv = f(9406, -0.1, 0.1), f(9407, 0.5, 1.5)
            st2 = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
            got, cn = rt.conv_rows_bn(x, wp, Cout, To, sums, gamma, beta, qm, qv, rows, bias=bias, stats=st2, ntaps=kt, stride=stride, pad=pad)
            torch.cuda.synchronize()
        assert torch.equal(got, want)
        for name in ("mean", "rstd", "scale", "shift"):
            assert torch.equal(getattr(cn, name), getattr(bn, name)), name
        assert torch.equal(qm, rm) and torch.equal(qv, rv)
        assert rel(st2, st) < 1e-12                       # the sums are exact up to the order of the double atomics


def test_mul_many_and_cross_entropy(rt):
    """stgcn_mul_many (A * importance of every block, and its backward) and stgcn_cross_entropy (loss and gradient of
    nn.CrossEntropyLoss() with its defaults) against float64 evaluations of the same formulas."""
    from stgcn_rt import functional as Fn
    a = t64((3, 25, 25), 9501).float().to(DEV)
    ins = [t64((3, 25, 25), 9510 + i).float().to(DEV) for i in range(10)]
    out = rt.mul_many(a, ins)
    assert torch.equal(out, torch.stack(ins) * a.unsqueeze(0))
    ins[4] = None
    out = rt.mul_many(a, ins)
    assert not out[4].any() and torch.equal(out[5], ins[5] * a)
    assert rt.mul_many(a, [a] * 17) is None and rt.mul_many(a, [a.double()]) is None

    for N, ncls in ((64, 60), (5, 400), (1, 3), (130, 1000)):
        logits = (t64((N, ncls), 9600 + N, -6, 6)).float().to(DEV).requires_grad_(True)
        labels = torch.tensor([(7 * i + 3) % ncls for i in range(N)], dtype=torch.int64, device=DEV)
        loss = Fn.cross_entropy(logits, labels)
        (loss * 3.0).backward()
        l64 = logits.detach().double().cpu().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(l64, labels.cpu())
        (ref * 3.0).backward()
        assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
        assert rel(logits.grad, l64.grad) < 1e-5
        again = Fn.cross_entropy(logits.detach(), labels)
        assert torch.equal(again, loss.detach())            # the same bits every run
