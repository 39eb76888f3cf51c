"""Single-kernel micro-benchmark at the shapes of the NTU training step (N=64 clips, M=2 -> B=128).

    python tools/kbench.py gcn_fused:64:64 [--iters 20]
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -- python3 tools/kbench.py gcn_fused:64:64 --iters 3

Case syntax  name:Cin:Cout[:stride[:T]]  with T derived from the channel width (64 -> 300, 128 -> 150, 256 -> 75) unless given.
Prints mean milliseconds per launch (events on the current stream).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn_amd"))
from stgcn_rt import ops as R  # noqa: E402
from stgcn_rt import lib as L  # noqa: E402

DEV = torch.device("cuda:0")
V, B, K = 25, 128, 3


def act(C, T, dtype):
    x = torch.randn((B, C, T, V), device=DEV, dtype=torch.float32).to(dtype)
    return x.contiguous(memory_format=torch.channels_last)


def make(case, dtype):
    parts = case.split(":")
    name, Cin, Cout = parts[0], int(parts[1]), int(parts[2])
    stride = int(parts[3]) if len(parts) > 3 else 1
    Tin = int(parts[4]) if len(parts) > 4 else {64: 300, 128: 150, 256: 75}.get(Cin, 300)
    if name in ("gcn_fused", "gcn_fused_ns", "gcn_fused_add", "gcn_agrad", "gcn_wgrad"):
        x = act(Cin, Tin, dtype)
        adj = torch.rand((K, V, V), device=DEV) * 0.2
        W = torch.randn((K * Cout, Cin, 1, 1), device=DEV) * (1.0 / Cin ** 0.5)
        bias = torch.randn((K * Cout,), device=DEV) * 0.1
        wp = R.pack_gcn_w(W, K, Cin, dtype)
        dy = act(Cout, Tin, dtype)
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
        if name == "gcn_fused":
            return lambda: R.gcn_fused(x, wp, adj, Cout, bias=bias, stats=stats)
        if name == "gcn_fused_ns":                # without the BatchNorm statistics
            return lambda: R.gcn_fused(x, wp, adj, Cout, bias=bias)
        if name == "gcn_fused_add":
            return lambda: R.gcn_fused(x, wp, adj, Cout, add=dy)
        if name == "gcn_agrad":
            return lambda: R.gcn_agrad(x, wp, bias, dy)
        return lambda: R.gcn_wgrad(x, dy, adj)
    if name in ("conv", "conv_np", "conv_ns", "conv_t", "conv_t_epi", "conv_wgrad", "conv1", "conv1_t", "conv1_wgrad"):
        kt = 1 if name.startswith("conv1") else 9
        pad = (kt - 1) // 2
        Tout = (Tin + 2 * pad - kt) // stride + 1
        x = act(Cin, Tin, dtype)
        W = torch.randn((Cout, Cin, kt, 1), device=DEV) * (1.0 / (Cin * kt) ** 0.5)
        bias = torch.randn((Cout,), device=DEV) * 0.1
        sc = torch.rand((Cin,), device=DEV) + 0.5
        sh = torch.randn((Cin,), device=DEV) * 0.1
        stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
        do = act(Cout, Tout, dtype)
        if name in ("conv", "conv1", "conv_np", "conv_ns"):    # _np: without the BatchNorm+ReLU prologue; _ns: without statistics
            wp = R.pack_conv_w(W, dtype)
            pro = (sc, sh) if (kt == 9 and name != "conv_np") else None
            st = None if name == "conv_ns" else stats
            return lambda: R.conv_rows(x, wp, Cout, Tout, bias=bias, pro=pro, stats=st, ntaps=kt, stride=stride, pad=pad)
        if name in ("conv_t", "conv1_t"):
            wt = R.pack_conv_wt(W, dtype)
            return lambda: R.conv_rows(do, wt, Cin, Tin, ntaps=kt, stride=stride, pad=pad, transposed=True)
        if name == "conv_t_epi":            # data gradient fused with the ReLU mask and BatchNorm-backward sums
            wt = R.pack_conv_wt(W, dtype)
            mean, rstd = torch.randn((Cin,), device=DEV) * 0.1, torch.rand((Cin,), device=DEV) + 0.5
            st = torch.zeros(2 * Cin, dtype=torch.float64, device=DEV)
            return lambda: R.conv_rows(do, wt, Cin, Tin, epilogue=L.EPI_RELU_BWD, aux=x, ep=(sc, sh, mean, rstd), stats=st,
                                       ntaps=kt, stride=stride, pad=pad, transposed=True)
        if name == "conv1_wgrad":
            return lambda: R.conv_wgrad(x, do, kt, stride, pad)
        return lambda: R.conv_wgrad(x, do, kt, stride, pad, pro=(sc, sh))
    if name in ("bob0", "bob1", "bob1g", "bob0r", "bob1r"):      # block_out_bwd: reduce / apply pass, identity or conv residual; bob1g also writes the masked gradient
        Cc, Tt = Cin, Tin
        o, dout = act(Cc, Tt, dtype), act(Cc, Tt, dtype)
        has_r = name.endswith("r")
        want_g = name.endswith("g")
        r = act(Cc, Tt, dtype) if has_r else None
        rows = o.shape[0] * Tt * V
        mk = lambda: (torch.rand((Cc,), device=DEV) + 0.5)
        bn2 = R.bn_finalize(torch.rand((2, Cc), dtype=torch.float64, device=DEV) * rows, rows, mk(), mk(), torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV), True)
        bnr = R.bn_finalize(torch.rand((2, Cc), dtype=torch.float64, device=DEV) * rows, rows, mk(), mk(), torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV), True)
        g2, gr = mk(), mk()
        mask = torch.randint(-32768, 32767, (rows * Cc * o.element_size() // 16,), dtype=torch.int16, device=DEV)
        acc = torch.zeros((4, Cc), dtype=torch.float64, device=DEV)
        dbf = torch.zeros((2, Cc), dtype=torch.float64, device=DEV)
        dO, dR, g = torch.empty_like(o), torch.empty_like(o), torch.empty_like(o)
        ps = 0 if name.startswith("bob0") else 1
        args = [L.dtype_code(o.dtype), ps, L.ptr(dout), None, Tt * V, 1.0, None, L.ptr(mask), L.ptr(o), L.ptr(r), L.ptr(bn2.mean), L.ptr(bn2.rstd), L.ptr(g2),
                L.ptr(bnr.mean) if has_r else None, L.ptr(bnr.rstd) if has_r else None, L.ptr(gr) if has_r else None,
                L.ptr(acc[0:2]), L.ptr(acc[2:4]) if has_r else None,
                L.ptr(dO) if ps else None, L.ptr(dR) if (ps and has_r) else None, L.ptr(g) if (ps and want_g) else None,
                L.ptr(dbf[0]), L.ptr(dbf[1]) if has_r else None, None, None, None, None,
                rows, Cc, 0.5, 1234, float(rows), 1, L.stream_ptr()]
        keepalive = (o, dout, r, mask, acc, dbf, dO, dR, g, bn2, bnr, g2, gr)
        key = (str(o.dtype), ps, rows, Cc, has_r) + ((want_g,) if ps else ())      # as ops.block_out_bwd names the launch
        return lambda: (R._launch(key, "stgcn_block_out_bwd", *args), keepalive)[0]
    if name == "bnbwd":                                  # bn_bwd_apply
        Cc, Tt = Cin, Tin
        g, y = act(Cc, Tt, dtype), act(Cc, Tt, dtype)
        rows = g.shape[0] * Tt * V
        mk = lambda: (torch.rand((Cc,), device=DEV) + 0.5)
        bn = R.bn_finalize(torch.rand((2, Cc), dtype=torch.float64, device=DEV) * rows, rows, mk(), mk(), torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV), True)
        st = torch.rand((2 * Cc,), dtype=torch.float64, device=DEV)
        gm = mk()
        out = torch.empty_like(g)
        return lambda: R.bn_bwd_apply(g, y, gm, bn, st, True, out=out)
    if name == "bofwd":                                  # block_out_fwd with identity residual and mask
        Cc, Tt = Cin, Tin
        o, res = act(Cc, Tt, dtype), act(Cc, Tt, dtype)
        rows = o.shape[0] * Tt * V
        mk = lambda: (torch.rand((Cc,), device=DEV) + 0.5)
        bn = R.bn_finalize(torch.rand((2, Cc), dtype=torch.float64, device=DEV) * rows, rows, mk(), mk(), torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV), True)
        return lambda: R.block_out_fwd(o, res, bn, None, 0.5, 1234, want_mask=True)
    raise SystemExit("unknown case " + case)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--lib", default=None, help="alternative build of libstgcn_hip.so (A/B timing on one device)")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many independent sets of tensors (a training step never "
                    "re-reads what the caches, the memory-side cache included, still hold from the launch before)")
    ap.add_argument("--batch", type=int, default=128, help="samples B = clips x persons (128: the 64-clip step; 16: the 8-clip step)")
    ap.add_argument("--interleave", action="store_true", help="cycle through the cases launch by launch (each launch between its own events): a "
                    "kernel as it runs between other kernels, not behind copies of itself")
    ap.add_argument("--print-key", action="store_true", help="also print the launch's timing key (entry point + shape), as bench.py's tables name it")
    a = ap.parse_args()
    global B
    B = a.batch
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.lib:
        L._LIB_PATH = os.path.abspath(a.lib)
    L.load()
    import time
    if a.interleave:
        sets = [[make(case, dtype) for _ in range(max(1, a.rotate))] for case in a.cases]
        evs = [[] for _ in a.cases]
        for it in range(a.iters + 10):
            for ci, fns in enumerate(sets):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fns[it % len(fns)]()
                e1.record()
                if it >= 10:
                    evs[ci].append((e0, e1))
        torch.cuda.synchronize()
        for case, ev in zip(a.cases, evs):
            ms = sorted(x.elapsed_time(y) for x, y in ev)
            print("%-28s %8.4f ms  (median %.4f, interleaved)" % (case, sum(ms) / len(ms), ms[len(ms) // 2]), flush=True)
        return
    first = True
    for case in a.cases:
        fns = [make(case, dtype) for _ in range(max(1, a.rotate))]
        fn = fns[0]
        if len(fns) > 1:
            state = {"i": 0}

            def fn(fns=fns, state=state):
                state["i"] = (state["i"] + 1) % len(fns)
                return fns[state["i"]]()
        if a.print_key:
            R.TIMING, R.TIMING_NAMES = {}, None
            fn()
            torch.cuda.synchronize()
            for k in R.TIMING:
                print("KEY %s %s" % (case, "|".join(map(str, k))), flush=True)
            R.TIMING = None
        # the first launches of a process run at a lower clock: keep the device busy for a while before timing
        t_end = time.time() + (0.5 if first else 0.05)
        first = False
        while True:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            if time.time() >= t_end:
                break
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%-28s %8.4f ms" % (case, e0.elapsed_time(e1) / a.iters), flush=True)


if __name__ == "__main__":
    main()
