"""tools/top1_agreement.py: the harness behind "top-1 agreement on a local checkpoint" (the reference's published
accuracies, README.md:141-144, come from processor/recognition.py:110-143 on checkpoints and datasets that do not ship
with the source).  Exercised on a synthetic checkpoint written in the reference's wire format and a tiny .npy / pickle
pair in the feeder's format."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "st-gcn_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from oracle import stgcn_oracle as O        # noqa: E402
import top1_agreement as H                  # noqa: E402


def _files(tmp_path, n=6, T=24, ncls=60, prefix=""):
    g = O.build_graph("ntu-rgb+d", "spatial")
    sd = O.det_state(3, ncls, g["A"], True, dtype=torch.float32)
    from collections import OrderedDict
    from oracle import torch_step as TS
    x = O.det_input(n, 3, T, 25, 2, tag=4200).float().numpy()
    np.save(tmp_path / "d.npy", x)
    with torch.no_grad():
        # the closed-form weights send every clip to one class; centre the classifier on the clips' mean feature
        # (read out through an identity classifier) so that the arg-max differs from clip to clip
        probe = dict(sd)
        probe["fcn.weight"], probe["fcn.bias"] = torch.eye(256).reshape(256, 256, 1, 1), torch.zeros(256)
        feat = TS.forward(probe, torch.from_numpy(x), False)
        sd["fcn.bias"] = -(sd["fcn.weight"].reshape(ncls, 256) @ feat.mean(0))
        labels = TS.forward(dict(sd), torch.from_numpy(x), False).argmax(1).tolist()
    assert len(set(labels)) >= n // 2
    w = str(tmp_path / "w.pt")
    torch.save(OrderedDict((prefix + k, v) for k, v in sd.items()), w)
    labels[0] = (labels[0] + 1) % ncls                             # one clip "mislabelled": top-1 is (n - 1) / n
    with open(tmp_path / "l.pkl", "wb") as f:
        pickle.dump((["clip%d" % i for i in range(n)], labels), f)
    return w, str(tmp_path / "d.npy"), str(tmp_path / "l.pkl")


def test_checkpoint_not_present(tmp_path, capsys):
    rec = H.run(["--weights", str(tmp_path / "st_gcn.ntu-xsub.pt"), "--data", "x.npy", "--label", "y.pkl"])
    assert rec["status"] == "checkpoint not present"
    assert "checkpoint not present" in capsys.readouterr().out


def test_cpu_leg_on_a_synthetic_checkpoint(tmp_path, capsys):
    w, d, l = _files(tmp_path, prefix="module.")                   # DataParallel-style names, as old checkpoints have
    rec = H.run(["--weights", w, "--data", d, "--label", l, "--impl", "cpu"])
    assert rec["status"] == "ok" and rec["clips"] == 6
    assert rec["cpu"]["top1"] == pytest.approx(5 / 6) and rec["cpu"]["top5"] >= rec["cpu"]["top1"]
    assert "top-1" in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_hip_agrees_with_cpu_on_a_synthetic_checkpoint(tmp_path, dtype):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    w, d, l = _files(tmp_path, n=8)
    rec = H.run(["--weights", w, "--data", d, "--label", l, "--dtype", dtype, "--batch", "4"])
    assert rec["status"] == "ok" and rec["clips"] == 8
    # the centred logits are differences of the raw ones, so the raw tolerances (1e-4 / 2e-2) apply to a scale about
    # ten times the one this deviation is quoted on; one clip's margin is 1e-3 of that scale
    print("top-1 agreement %s: %.3f, largest logit deviation %.2e" % (dtype, rec["top1_agreement"], rec["max_logit_deviation"]))
    assert rec["max_logit_deviation"] < (1e-3 if dtype == "fp32" else 2e-1)
    assert rec["cpu"]["top1"] == pytest.approx(7 / 8)
    if dtype == "fp32":
        assert rec["top1_agreement"] == 1.0 and rec["hip"]["top1"] == rec["cpu"]["top1"]
    else:
        assert rec["top1_agreement"] >= 7 / 8
