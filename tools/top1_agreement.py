#!/usr/bin/env python
"""Top-1 agreement of the HIP model with the CPU restatement on a LOCAL checkpoint.

  python tools/top1_agreement.py --weights st_gcn.ntu-xsub.pt --data val_data.npy --label val_label.pkl \
         [--config config/st_gcn/ntu-xsub/test.yaml | --layout ntu-rgb+d --strategy spatial --num-class 60] \
         [--dtype bf16|fp32] [--batch 64] [--limit 256] [--impl hip,cpu]

The reference publishes top-1 accuracies for three pretrained checkpoints (README.md:141-144 of hensyou/st-gcn), which
its test phase reproduces (processor/recognition.py:110-143 with weights loaded by torchlight/torchlight/io.py:57-90).
Neither the checkpoints nor the datasets ship with the source, so this script takes local paths: it loads the
checkpoint into the HIP model (net.st_gcn.Model) and into the CPU restatement on stock PyTorch operators
(oracle/torch_step.py: test infrastructure, the checker here), runs both on the same clips in eval mode, and prints
each implementation's top-1 / top-5 against the labels and the fraction of clips on which their arg-max agrees.
When the checkpoint file does not exist it prints "checkpoint not present" and exits with status 3.
Prints one JSON line; `run()` returns the same record.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

NOT_PRESENT = 3


def _model_args(args):
    if args.config:
        import yaml
        with open(args.config) as f:
            cfg = yaml.safe_load(f)
        return dict(cfg["model_args"])
    return dict(in_channels=args.in_channels, num_class=args.num_class, edge_importance_weighting=True,
                graph_args=dict(layout=args.layout, strategy=args.strategy))


def _batches(feeder, limit, batch):
    n = min(len(feeder), limit) if limit > 0 else len(feeder)
    for s in range(0, n, batch):
        items = [feeder[i] for i in range(s, min(n, s + batch))]
        yield (torch.from_numpy(np.stack([np.asarray(c, dtype=np.float32) for c, _ in items])),
               torch.tensor([int(l) for _, l in items], dtype=torch.int64))


def _scores_hip(margs, weights, feeder, args):
    from net.st_gcn import Model
    from stgcn_rt import checkpoint as CK
    from stgcn_rt.evaluate import evaluate
    if not torch.cuda.is_available():
        raise SystemExit("the HIP implementation needs a GPU (there is no CPU path); use --impl cpu for the restatement alone")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    margs = dict(margs)
    margs.pop("dropout", None)                                   # the identity in eval mode
    model = Model(compute_dtype=dtype, **margs)
    CK.load_weights(model, weights)
    dev = torch.device("cuda", 0)
    model.to(dev)
    out = evaluate(model, _batches(feeder, args.limit, args.batch), dev, show_topk=(1, 5))
    return out["result"], out["label"]


def _scores_cpu(margs, weights, feeder, args):
    from oracle import torch_step as TS
    from stgcn_rt import checkpoint as CK
    raw = torch.load(weights, map_location="cpu", weights_only=True)
    sd = {CK._strip_load(k): (v.float() if v.is_floating_point() else v) for k, v in raw.items()}
    res, labels = [], []
    with torch.no_grad():
        for data, label in _batches(feeder, args.limit, min(args.batch, 16)):
            res.append(TS.forward(sd, data, False).numpy())
            labels.append(label.numpy())
    return np.concatenate(res), np.concatenate(labels)


def run(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--weights", required=True)
    ap.add_argument("--data", required=True, help="float32 .npy of clips (N, C, T, V, M)")
    ap.add_argument("--label", required=True, help="pickle of (sample names, labels)")
    ap.add_argument("--config", default=None, help="a processor YAML holding model_args")
    ap.add_argument("--layout", default="ntu-rgb+d")
    ap.add_argument("--strategy", default="spatial")
    ap.add_argument("--num-class", type=int, default=60)
    ap.add_argument("--in-channels", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--limit", type=int, default=0, help="first LIMIT clips only (0: all)")
    ap.add_argument("--impl", default="hip,cpu")
    args = ap.parse_args(argv)

    if not os.path.isfile(args.weights):
        print("checkpoint not present: %s (the pretrained files are fetched by the reference's tools/get_models.sh; "
              "place one at a local path)" % args.weights)
        return {"status": "checkpoint not present", "weights": args.weights}
    from feeder.feeder import Feeder
    from stgcn_rt.evaluate import topk_accuracy
    feeder = Feeder(args.data, args.label)
    margs = _model_args(args)
    rec = {"status": "ok", "weights": args.weights, "clips": min(len(feeder), args.limit) if args.limit > 0 else len(feeder),
           "dtype": args.dtype}
    scores = {}
    for impl in args.impl.split(","):
        res, labels = {"hip": _scores_hip, "cpu": _scores_cpu}[impl](margs, args.weights, feeder, args)
        scores[impl] = res
        k5 = min(5, res.shape[1])
        rec[impl] = {"top1": topk_accuracy(res, labels, 1), "top%d" % k5: topk_accuracy(res, labels, k5)}
        print("%-3s  top-1 %.4f  top-%d %.4f  (%d clips)" % (impl, rec[impl]["top1"], k5, rec[impl]["top%d" % k5], len(labels)))
    if "hip" in scores and "cpu" in scores:
        a, b = scores["hip"], scores["cpu"]
        rec["top1_agreement"] = float((a.argmax(1) == b.argmax(1)).mean())
        rec["max_logit_deviation"] = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
        print("top-1 agreement hip vs cpu: %.4f   largest logit deviation / max |logit|: %.2e" % (rec["top1_agreement"], rec["max_logit_deviation"]))
    print(json.dumps(rec))
    return rec


if __name__ == "__main__":
    sys.exit(NOT_PRESENT if run()["status"] != "ok" else 0)
