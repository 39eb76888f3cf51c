"""GPU: the allocation-free entry point stgcn_gcn_agrad_noalloc (partial sums in a static device array inside the
library) against stgcn_gcn_agrad (caller-provided workspace) through the C ABI: bit-identical results on two calls in
a row, the second with a smaller input (the array keeps the first call's contents), the oracle as the checker of
both, and the size limit's error."""
import pytest
import torch

from oracle import stgcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, T, V, Cin, Cout, K, dtype, tag):
    x = torch.tensor(O.det_uniform((B, Cin, T, V), tag)).to(dtype)
    dy = torch.tensor(O.det_uniform((B, Cout, T, V), tag + 1)).to(dtype)
    W = torch.tensor(O.det_uniform((K * Cout, Cin), tag + 2) * (1.5 / Cin ** 0.5)).to(dtype)
    b = torch.tensor(O.det_uniform((K * Cout,), tag + 3, -0.2, 0.2)).float()
    return x, dy, W, b


def _both(x, dy, W, b, K):
    """(dA through the static array, dA through a workspace, bytes torch allocated during the first call)."""
    from stgcn_rt import lib as L
    from stgcn_rt import ops as P
    B, Cin, T, V = x.shape
    Cout = dy.shape[1]
    xd, dyd = P.cl(x.to(DEV)), P.cl(dy.to(DEV))
    wp = W.reshape(K, Cout, Cin).contiguous().to(DEV)
    bd = b.to(DEV)
    ws = P.workspace(xd.device)
    dA0 = torch.full((K, V, V), float("nan"), device=DEV)
    dA1 = torch.full((K, V, V), float("nan"), device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    L.call("stgcn_gcn_agrad_noalloc", L.dtype_code(xd.dtype), L.ptr(xd), L.ptr(wp), L.ptr(bd), L.ptr(dyd), L.ptr(dA0),
           B * T, V, Cin, Cout, K, L.stream_ptr())
    grown = torch.cuda.memory_stats()["allocation.all.allocated"] - before
    L.call("stgcn_gcn_agrad", L.dtype_code(xd.dtype), L.ptr(xd), L.ptr(wp), L.ptr(bd), L.ptr(dyd), L.ptr(dA1),
           B * T, V, Cin, Cout, K, L.ptr(ws), ws.numel(), L.stream_ptr())
    torch.cuda.synchronize()
    # torch's statistics see only torch's own allocator; a hipMalloc inside the library would show in the device's
    # free memory instead.  Measured around a repeated call (the first one on a device caches the array's address and
    # may load the code object), with torch's own allocator quiet in between.
    dA2 = torch.full((K, V, V), float("nan"), device=DEV)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    L.call("stgcn_gcn_agrad_noalloc", L.dtype_code(xd.dtype), L.ptr(xd), L.ptr(wp), L.ptr(bd), L.ptr(dyd), L.ptr(dA2),
           B * T, V, Cin, Cout, K, L.stream_ptr())
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 == free0, (free0, free1)
    assert torch.equal(dA2, dA0)
    return dA0.cpu(), dA1.cpu(), grown


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_no_alloc_agrad_matches_workspace_entry_point(dtype):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    K = 3
    # first a large call (many resident workgroups' partials), then a smaller one over fewer frames and channels
    for (B, T, V, Cin, Cout, tag) in ((8, 150, 25, 64, 128, 9100), (2, 9, 18, 16, 32, 9200)):
        x, dy, W, b = _inputs(B, T, V, Cin, Cout, K, dtype, tag)
        got, ref_ws, grown = _both(x, dy, W, b, K)
        assert grown == 0                                  # nothing was allocated through torch either
        assert torch.isfinite(got).all()
        assert torch.equal(got, ref_ws), (got - ref_ws).abs().max()
        A = torch.zeros(K, V, V, dtype=torch.float64)
        _, _, _, dA = O.gcn_backward(x.double(), A, W.double(), b.double(), dy.double())
        tol = 1e-4 if dtype == torch.float32 else 2e-2
        assert float((got.double() - dA).abs().max() / dA.abs().max()) < tol


def test_no_alloc_agrad_rejects_what_exceeds_its_array():
    from stgcn_rt import lib as L
    from stgcn_rt import ops as P
    K, V, Cin, Cout, NF = 3, 32, 64, 64, 4096            # 2 workgroups per CU x 3 x 32 x 32 floats > 4 MiB on 256 CUs
    x = P.cl(torch.zeros(NF // 16, Cin, 16, V, device=DEV, dtype=torch.bfloat16))
    dy = P.cl(torch.zeros(NF // 16, Cout, 16, V, device=DEV, dtype=torch.bfloat16))
    wp = torch.zeros(K, Cout, Cin, device=DEV, dtype=torch.bfloat16)
    dA = torch.zeros(K, V, V, device=DEV)
    rc = L.load().stgcn_gcn_agrad_noalloc(L.BF16, L.ptr(x), L.ptr(wp), None, L.ptr(dy), L.ptr(dA), NF, V, Cin, Cout, K, L.stream_ptr())
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert rc == (4 if 2 * cus * K * V * V * 4 > (4 << 20) else 0), rc
    with pytest.raises(RuntimeError, match="static scratch"):
        if rc == 4:
            L.call("stgcn_gcn_agrad_noalloc", L.BF16, L.ptr(x), L.ptr(wp), None, L.ptr(dy), L.ptr(dA), NF, V, Cin, Cout, K, L.stream_ptr())
        else:
            raise RuntimeError("static scratch limit not reachable on this part")
