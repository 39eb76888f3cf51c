"""Checkpoint and config interchange with the reference's processor.

Wire format (torchlight/torchlight/io.py:57-90 load, 101-107 save of hensyou/st-gcn): a flat OrderedDict
parameter-or-buffer name -> CPU tensor, written with torch.save; names carry no `module.` prefix (DataParallel's
wrapper name is stripped on both sides); the buffer `A` is part of it.  The module tree of net.st_gcn.Model keeps
the reference's names, so files move between the two implementations in both directions.
"""
import importlib
from collections import OrderedDict

import torch


def _strip_load(name):
    return name.split("module.")[-1]            # io.py:65: everything up to the last `module.` goes


def _strip_save(name):
    return "".join(name.split("module."))       # io.py:104: every `module.` goes


def load_weights(model, weights_path, ignore_weights=None, log=None):
    """Load a checkpoint into `model` the way IO.load_weights does (io.py:57-90).

    ignore_weights: a prefix or list of prefixes; matching entries of the file are dropped.  Entries the file
    lacks keep the model's current values (the reference's fallback after a failed strict load); entries of the
    file the model lacks, or shape mismatches, raise as torch's load_state_dict does.
    The file is read with weights_only=True: nothing in it is executed.
    """
    log = log or (lambda msg: None)
    if ignore_weights is None:
        ignore_weights = []
    if isinstance(ignore_weights, str):
        ignore_weights = [ignore_weights]
    log("Load weights from {}.".format(weights_path))
    raw = torch.load(weights_path, map_location="cpu", weights_only=True)
    weights = OrderedDict((_strip_load(k), v.cpu()) for k, v in raw.items())
    for prefix in ignore_weights:
        for name in [w for w in weights if w.startswith(prefix)]:
            weights.pop(name)
            log("Filter [{}] remove weights [{}].".format(prefix, name))
    for name in weights:
        log("Load weights [{}].".format(name))
    try:
        model.load_state_dict(weights)
    except (KeyError, RuntimeError):
        state = model.state_dict()
        for name in sorted(set(state.keys()) - set(weights.keys())):
            log("Can not find weights [{}].".format(name))
        state.update(weights)
        model.load_state_dict(state)
    return model


def save_model(model, path, log=None):
    """Write model.state_dict() as IO.save_model does (io.py:101-107): CPU tensors, `module.` removed."""
    weights = OrderedDict((_strip_save(k), v.detach().cpu()) for k, v in model.state_dict().items())
    torch.save(weights, path)
    if log:
        log("The model has been saved as {}.".format(path))
    return weights


def import_class(name):
    """Dotted path -> attribute, as torchlight's import_class (io.py:205-210)."""
    mod, _, attr = name.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def model_from_config(cfg, **overrides):
    """Build the model a processor config names: cfg is a mapping (or a YAML path) with `model` (dotted class path,
    e.g. net.st_gcn.Model) and `model_args` (in_channels, num_class, graph_args, edge_importance_weighting,
    dropout ...), as config/st_gcn/*/train.yaml hold them (processor/io.py:31-50, 82-85)."""
    if isinstance(cfg, str):
        import yaml
        with open(cfg) as f:
            cfg = yaml.safe_load(f)
    args = dict(cfg.get("model_args") or {})
    args.update(overrides)
    return import_class(cfg["model"])(**args)
