#!/usr/bin/env python
"""Benchmark of the ST-GCN training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32] [--batch 64]

One "step" = REC_Processor.train's iteration (forward, mean cross-entropy, backward, gradient
all-reduce when N > 1, SGD-nesterov update) on synthetic NTU-RGB+D tensors already resident in
HBM: x [64, 3, 300, 25, 2] per GPU (weak scaling), 60 classes, dropout 0.5, lr 0.1, momentum 0.9,
weight decay 1e-4 (config/st_gcn/ntu-xsub/train.yaml of the reference).  Rank 0 prints ONE JSON
line.  `roofline` prices the kernel that takes the most time, with the launch durations measured by
events on the launch stream inside the timed region; `cpu_baseline` times the CPU oracle (a port of
the same step) on the host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

PEAK = {"bf16": 2500.0, "fp32": 157.3}      # dense MFMA TFLOP/s (MI355X)
HBM_PEAK = 8000.0                           # GB/s


def kernel_work(key):
    """Algorithmic FLOPs and compulsory bytes of one launch, from the timing key of stgcn_rt.ops."""
    name = key[0]
    es = 2 if "bfloat16" in str(key[1]) else 4
    if name == "stgcn_conv_rows":
        _, dt, tr, epi, B, Tin, Tout, V, Cin, Cout, ntaps, stride, pro = key
        rows_small = min(Tin, Tout) * V * B                    # every tap is applied once per low-rate row
        flops = 2.0 * Cin * Cout * ntaps * rows_small
        byts = (B * Tin * V * Cin + B * Tout * V * Cout) * es + ntaps * Cin * Cout * es
        if epi == 1:
            byts += B * Tout * V * Cout * es                   # the BN input re-read by the fused ReLU/BN backward
        return flops, byts, "mfma"
    if name == "stgcn_conv_wgrad":
        _, dt, B, Tin, Tout, V, Cin, Cout, ntaps, stride = key
        return 2.0 * Cin * Cout * ntaps * B * Tout * V, (B * Tin * V * Cin + B * Tout * V * Cout) * es + 4 * ntaps * Cin * Cout, "mfma"
    if name == "stgcn_gcn_fused":
        _, dt, NF, V, Cin, Nout, K, add = key
        return 2.0 * K * NF * V * (Cin * Nout + Nout * V), NF * V * (Cin + Nout * (2 if add else 1)) * es + K * Cin * Nout * es, "mfma"
    if name in ("stgcn_gcn_wgrad", "stgcn_gcn_agrad"):
        _, dt, NF, V, Cin, Cout, K = key
        return 2.0 * K * NF * V * (Cin * Cout + Cout * V), NF * V * (Cin + Cout) * es, "mfma"
    if name == "stgcn_block_out_fwd":
        _, dt, rows, C, has_res, _ = key
        return 0.0, rows * C * es * (3 if has_res else 2), "hbm"
    if name == "stgcn_block_out_bwd":
        rows, C, has_r = key[3], key[4], key[5]
        p = key[2]
        want_g = len(key) > 6 and key[6]
        # read: dout, o (+ r) and the ReLU/dropout bit mask (1/8 of a tensor); the apply pass also writes dO (+ dR) (+ g)
        n = (3 if has_r else 2) + 1.0 / 8 + (0 if p == 0 else (2 if has_r else 1) + (1 if want_g else 0))
        return 0.0, rows * C * es * n, "hbm"
    if name == "stgcn_bn_bwd_apply":
        _, dt, rows, C = key
        return 0.0, rows * C * es * 3, "hbm"
    return 0.0, 0.0, "hbm"


def recorded_traffic(key):
    """HBM-side bytes per launch of the kernel instance `key`, as measured with `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes; FETCH_SIZE doubled, the gfx950 correction for 16-byte-per-lane streaming
    reads) and recorded in bench_traffic.json next to this file; None when that instance was not measured."""
    try:
        with open(os.path.join(ROOT, "bench_traffic.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    ent = rec.get("|".join(map(str, key)))
    return None if ent is None else ent.get("traffic_bytes")


def cpu_baseline(n_clips, steps, threads):
    """The same training step on stock PyTorch CPU operators -- F.conv2d, einsum, F.batch_norm, F.dropout, autograd,
    optim.SGD: the operators the reference's CPU path runs (oracle/torch_step.py, pinned to the golden vectors) --
    on a bounded sample of the workload."""
    from oracle import stgcn_oracle as O
    from oracle import torch_step as TS
    torch.set_num_threads(threads)
    g = O.build_graph("ntu-rgb+d", "spatial")
    st = TS.Stepper(O.det_state(3, 60, g["A"], True, dtype=torch.float32), lr=0.1, dropout=0.5)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(n_clips, 3, 300, 25, 2, generator=gen)
    labels = torch.randint(0, 60, (n_clips,), generator=gen)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        st.step(x, labels)
        if it > 0:                                              # first iteration warms the allocator up
            times.append(time.perf_counter() - t0)
    t = sum(times) / len(times)
    return {"value": round(n_clips / t, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "%d timed steps (1 warm-up) of the stock-PyTorch-operator restatement at N=%d clips, T=300, V=25, M=2, float32, "
                      "dropout 0.5, SGD-nesterov; a bounded sample: the per-clip rate is taken to be independent of the "
                      "batch size (the GPU figure is at 64 clips)" % (steps, n_clips)}


def cpu_baseline_oracle(n_clips, steps, threads):
    """The explicit-formula CPU oracle's training step (float32 torch CPU tensor ops, tap loops), bounded sample."""
    from oracle import stgcn_oracle as O
    torch.set_num_threads(threads)
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, 60, g["A"], True, dtype=torch.float32)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(n_clips, 3, 300, 25, 2, generator=gen)
    labels = torch.randint(0, 60, (n_clips,), generator=gen)
    names = O.param_names(sd)
    moms = {}
    seeds = list(range(1, 11))
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        logits, cache, newb = O.model_forward(sd, x, True, dropout=0.5, seeds=seeds)
        loss, dlog = O.cross_entropy(logits, labels)
        grads = O.model_backward(sd, cache, dlog, True)
        prm = {k: sd[k] for k in names}
        O.sgd_nesterov_step(prm, grads, moms, 0.1)
        sd.update(prm)
        sd.update(newb)
        if it > 0:
            times.append(time.perf_counter() - t0)
    t = sum(times) / len(times)
    return {"value": round(n_clips / t, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "%d timed steps (1 warm-up) of the explicit-formula oracle at N=%d clips, T=300, V=25, M=2, float32, dropout 0.5, "
                      "SGD-nesterov" % (steps, n_clips)}


def kinetics_train_record(dev, weights_init, steps=10, warmup=3):
    """The reference's Kinetics training step (config/st_gcn/kinetics-skeleton/train.yaml:6-8, 17-33: openpose layout,
    400 classes, no dropout, a 150-frame window, 256 clips over 8 GPUs = 32 per GPU) in both precisions, outside the
    headline's timed region."""
    from net.st_gcn import Model
    from stgcn_rt.ddp import Trainer
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(32, 3, 150, 18, 2, generator=gen).to(dev)
    labels = torch.randint(0, 400, (32,), generator=gen).to(dev)
    rec = {"workload": "Kinetics-skeleton training step, 32 clips per GPU, T=150, V=18, M=2, 400 classes, spatial K=3, no dropout, "
                       "SGD-nesterov; HIP-graph replay", "steps": steps, "warmup": warmup}
    for name, dtype in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        torch.manual_seed(1234)
        model = Model(3, 400, dict(layout="openpose", strategy="spatial"), True, compute_dtype=dtype).to(dev)
        model.apply(weights_init)
        model.train()
        tr = Trainer(model, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True, graph=True)
        for _ in range(warmup):
            tr.step(x, labels)
        tr.prepare_graph(x, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(x, labels)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        rec[name] = {"ms_per_step": round(dt * 1e3, 3), "clips_per_s": round(32 / dt, 1), "hip_graph": tr.graph_ready(),
                     "final_loss": round(float(loss.item()), 4)}
        del tr, model
        torch.cuda.empty_cache()
    return rec


def infer_sweep_record(dev, weights_init, n_clips=1024, reps=3):
    """Large-batch bf16 inference: eval-mode forward of N = 1024 clips (T = 300, M = 2) for both skeleton layouts and
    the three partition strategies (K = 3 / 1 / 2: graph.py:85-119 of the reference), clips/s each."""
    from net.st_gcn import Model
    rec = {"workload": "eval-mode forward, bf16, N=%d clips, T=300, M=2, BatchNorm folded" % n_clips, "unit": "clips/s"}
    for lay, ncls, V in (("ntu-rgb+d", 60, 25), ("openpose", 400, 18)):
        x = torch.randn(n_clips, 3, 300, V, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        for strat in ("spatial", "uniform", "distance"):
            torch.manual_seed(1234)
            m = Model(3, ncls, dict(layout=lay, strategy=strat), True, compute_dtype=torch.bfloat16).to(dev)
            m.apply(weights_init)
            m.eval()
            with torch.no_grad():
                out = m(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = m(x)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            assert bool(torch.isfinite(out).all())
            rec["V%d_K%d_%s" % (V, m.A.shape[0], strat)] = round(n_clips / dt, 1)
            del m, out
        del x
        torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU (weak scaling) or in total (--strong)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the global batch (the reference's 64), split evenly over the GPUs")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=8)
    ap.add_argument("--single-dtype", action="store_true", help="time --dtype only (default: the other precision is timed too and reported as a nested record)")
    ap.add_argument("--no-graph", action="store_true", help="issue every step launch by launch instead of replaying the captured HIP graph")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the nested records of the other configurations (Kinetics training step, large-batch inference sweep)")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel time table to stderr")
    args = ap.parse_args()

    if os.environ.get("STGCN_BENCH_WATCHDOG"):               # debugging aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["STGCN_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ST-GCN HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local >= ndev:
        # rehearsal of the multi-process path on a box with fewer GPUs than ranks (STGCN_DIST_BACKEND=gloo)
        local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or (os.environ.get("STGCN_DDP_FORCE") == "1" and "RANK" in os.environ):   # FORCE: a one-rank RCCL group
        backend = os.environ.get("STGCN_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    from net.st_gcn import Model
    from stgcn_rt import ops as P
    from stgcn_rt.ddp import Trainer

    if args.strong:
        if args.batch % world != 0:
            raise SystemExit("--strong: the global batch %d does not split over %d GPUs" % (args.batch, world))
        args.batch //= world                                    # clips per GPU from here on
    gen = torch.Generator().manual_seed(rank)                   # synthetic data: x ~ N(0,1), labels uniform
    x = torch.randn(args.batch, 3, args.frames, 25, 2, generator=gen).to(dev)
    labels = torch.randint(0, 60, (args.batch,), generator=gen).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def weights_init(m):                                        # processor/recognition.py:21-33
        cn = m.__class__.__name__
        if cn.find("Conv1d") != -1 or cn.find("Conv2d") != -1:
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.fill_(0)
        elif cn.find("BatchNorm") != -1:
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)

    def run(dtype_name, steps, warmup):
        """W warm-up steps, then K timed steps of the training iteration in `dtype_name`; -> record (rank 0: dict)."""
        dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        torch.manual_seed(1234)
        model = Model(3, 60, dict(layout="ntu-rgb+d", strategy="spatial"), True, dropout=0.5, compute_dtype=dtype).to(dev)
        model.apply(weights_init)
        model.train()
        trainer = Trainer(model, lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True, graph=not args.no_graph)
        trainer.broadcast_from_rank0()

        # warm-up; the last warm-up step also records every kernel to find the one that dominates
        for i in range(warmup):
            if i == warmup - 1:
                P.TIMING, P.TIMING_NAMES = {}, None
                trainer.step(x, labels, eager=True)
            else:
                trainer.step(x, labels)
        table = P.timing_summary() if P.TIMING is not None else {}
        P.TIMING = None
        by_name = {}
        for k, (n, mean, tot) in table.items():
            by_name[k[0]] = by_name.get(k[0], 0.0) + tot
        if args.kernel_table and rank == 0:
            tot_all = sum(by_name.values())
            print("%s: per-kernel time of one step (ms), total %.2f" % (dtype_name, tot_all), file=sys.stderr)
            for k, (n, mean, tot) in sorted(table.items(), key=lambda kv: -kv[1][2]):
                fl, by, bound = kernel_work(k)
                print("  %7.3f ms x%-2d  %6.1f TFLOP/s %7.1f GB/s  %s" % (mean, n, fl / mean / 1e9, by / mean / 1e6, k), file=sys.stderr)
        # the kernel instance that takes the most time of a step; it is priced from launches bracketed by events on
        # the launch stream during the timed region (eager steps interleaved with the timed ones would distort the
        # step time, so with graph replay the instance is timed in extra eager steps right after the timed region)
        dom_key = max(table, key=lambda k: table[k][2]) if table else None

        graphed = trainer.prepare_graph(x, labels)               # captured here, outside the timed region
        if dom_key is not None and not graphed:
            P.TIMING, P.TIMING_NAMES = {}, {dom_key[0]}
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = trainer.step(x, labels)
        host_issue = time.perf_counter() - t0                    # the host's share: launching, autograd bookkeeping
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        if dom_key is not None and graphed:
            P.TIMING, P.TIMING_NAMES = {}, {dom_key[0]}
            for _ in range(min(steps, 5)):
                trainer.step(x, labels, eager=True)
        timed = P.timing_summary() if P.TIMING is not None else {}
        P.TIMING = None
        final_loss = float(loss.item())
        del trainer, model
        torch.cuda.empty_cache()
        if rank != 0:
            return None
        roof = None
        if timed:
            key, (n, mean, tot) = max(timed.items(), key=lambda kv: kv[1][2])
            fl, by, bound = kernel_work(key)
            if bound == "mfma":
                ach, peak, unit = fl / mean / 1e9, PEAK[dtype_name], "TFLOP/s"
            else:
                ach, peak, unit = by / mean / 1e6, HBM_PEAK, "GB/s"
            roof = {"bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                    "traffic": recorded_traffic(key), "kernel": key[0].replace("stgcn_", ""), "launch": list(map(str, key[1:])),
                    "avg_launch_ms": round(mean, 4), "launches_timed": n,
                    "timed_in": "eager steps after the timed region" if graphed else "the timed region"}
        return {"value": round(args.batch * world * steps / elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 3),
                "host_issue_ms_per_step": round(host_issue / steps * 1e3, 3), "steps": steps, "warmup": warmup,
                "hip_graph": bool(graphed), "final_loss": round(final_loss, 4), "roofline": roof}

    head = run(args.dtype, args.steps, args.warmup)
    other = None
    if not args.single_dtype:
        # the reference's own arithmetic (float32, processor/recognition.py:87-91) next to the bf16 headline -- or the
        # other way round --, same workload, same timed region, same invocation
        other_name = "fp32" if args.dtype == "bf16" else "bf16"
        other = run(other_name, args.steps, args.warmup)

    if rank == 0:
        cpu = cpu_o = None
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            cpu = cpu_baseline(args.cpu_clips, 3, threads)
            cpu_o = cpu_baseline_oracle(2, 1, threads)
        line = {
            "metric": "train clips/sec, ST-GCN NTU-60 (T=300,V=25,M=2)", "value": head["value"], "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "host_issue_ms_per_step": head["host_issue_ms_per_step"], "hip_graph": head["hip_graph"],
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "NTU-RGB+D xsub training step (forward + cross-entropy + backward + SGD-nesterov), "
                                   "%d clips per GPU, T=%d, V=25, M=2, 60 classes, spatial K=3, dropout 0.5" % (args.batch, args.frames),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "final_loss": head["final_loss"]},
            "roofline": head["roofline"], "cpu_baseline": cpu, "cpu_baseline_oracle": cpu_o,
        }
        if other is not None:
            line[other_name] = other
        # the reference computes in float32 (processor/recognition.py:87-91): that record's figures at top level too
        ref = head if args.dtype == "fp32" else other
        line["reference_dtype"] = "fp32"
        line["value_reference_dtype"] = None if ref is None else ref["value"]
        line["ms_per_step_reference_dtype"] = None if ref is None else ref["ms_per_step"]
        if world == 1 and not args.no_extra and args.batch == 64 and args.frames == 300:
            line["kinetics_train"] = kinetics_train_record(dev, weights_init)
            line["infer_n1024"] = infer_sweep_record(dev, weights_init)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
