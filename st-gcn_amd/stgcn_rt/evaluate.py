"""Evaluation loop of the reference's recognition processor (processor/recognition.py:72-76, 110-143)."""
import numpy as np
import torch


def topk_accuracy(result, label, k):
    """Fraction of rows whose label is among the k highest scores (show_topk, recognition.py:72-76)."""
    rank = np.asarray(result).argsort()
    hits = [l in rank[i, -k:] for i, l in enumerate(label)]
    return sum(hits) * 1.0 / len(hits)


def evaluate(model, loader, device, show_topk=(1, 5), evaluation=True, loss_fn=None):
    """model.eval(); run every (data, label) batch of `loader` without gradients; gather the scores.

    Returns dict(result=[n, classes] float array, and with evaluation: label, mean_loss (mean of the per-batch
    mean losses, as the reference reports it) and topk {k: accuracy})."""
    model.eval()
    loss_fn = loss_fn or torch.nn.CrossEntropyLoss()
    losses, results, labels = [], [], []
    for data, label in loader:
        data = data.float().to(device)
        label = label.long().to(device)
        with torch.no_grad():
            output = model(data)
        results.append(output.detach().float().cpu().numpy())
        if evaluation:
            losses.append(loss_fn(output.float(), label).item())
            labels.append(label.cpu().numpy())
    out = {"result": np.concatenate(results)}
    if evaluation:
        out["label"] = np.concatenate(labels)
        out["mean_loss"] = float(np.mean(losses))
        out["topk"] = {k: topk_accuracy(out["result"], out["label"], k) for k in show_topk}
    return out
