"""CPU plumbing of the batched dataset runner (counterpart of the reference's src/scripts/amodel_dav2_inference.py:76-125):
file-name patterns, nearest-exact resize to 518, value conventions of the guide tensors, 16-bit PNG output and the metric
averaging -- with a stand-in model and the fp64 oracle as the evaluator (the product networks and metrics have no CPU path)."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from src.scripts import amodal_dav2_inference as R  # noqa: E402
from oracle import metrics_oracle as MO  # noqa: E402


def _make_tree(tmp, ids, rng):
    d = {k: tmp / k for k in ("occ", "whole", "obs", "gt")}
    for v in d.values():
        v.mkdir()
    for sid in ids:
        Image.fromarray((rng.random((64, 64, 3)) * 255).astype(np.uint8)).save(d["occ"] / f"{sid}_occlusion.png")
        m = np.zeros((64, 64), dtype=np.uint8); m[10:50, 8:40] = 255
        Image.fromarray(m).save(d["whole"] / f"{sid}_whole_mask.png")
        depth = (rng.uniform(0.2, 0.9, size=(32, 32)) * 65535).astype(np.uint16)
        Image.fromarray(depth).save(d["obs"] / f"{sid}_depth.png")
        Image.fromarray((rng.uniform(0.2, 0.9, size=(128, 128)) * 65535).astype(np.uint16)).save(d["gt"] / f"{sid}_depth.png")
    return d


class _FakeModel:
    def __init__(self):
        self.calls = []

    def __call__(self, x, guide_rgb=None, guide_mask=None, observation=None):
        self.calls.append((x.shape, float(guide_mask.min()), float(guide_mask.max()), float(observation.min()), float(observation.max())))
        return (observation + 1) / 2 * 0.5 + 0.25      # [B,1,518,518] in (0,1)


def _oracle_evaluate(pred, gt, mask):
    p, g, m = pred.numpy(), gt.numpy(), mask.numpy()
    al = np.stack([MO.align_depth_least_square(g[b], p[b], m[b])[0] for b in range(p.shape[0])])
    return MO.depth_metrics(np.clip(al, 1e-3, 1.0), g, m)


def test_runner_end_to_end(tmp_path):
    rng = np.random.default_rng(0)
    ids = ["101", "102", "103"]
    d = _make_tree(tmp_path, ids, rng)
    with open(tmp_path / "split.txt", "w") as f:
        f.write("\n".join(f"sa_{i}.jpg" for i in ids) + "\n")
    assert R.sample_ids(str(d["occ"]), str(tmp_path / "split.txt")) == ids
    assert R.sample_ids(str(d["occ"]), None) == ids

    def evaluate(pred, gt, mask):
        p, g, m = pred.numpy(), gt.numpy(), mask.numpy()
        al = np.stack([MO.align_depth_least_square(g[b], p[b], m[b])[0] for b in range(p.shape[0])])
        return MO.depth_metrics(np.clip(al, 1e-3, 1.0), g, m)

    model = _FakeModel()
    res = R.run(model, ids, str(d["occ"]), str(d["whole"]), str(d["obs"]), str(tmp_path / "out"), str(d["gt"]), batch_size=2, device="cpu", evaluate=evaluate)
    assert [c[0] for c in model.calls] == [(2, 3, 518, 518), (1, 3, 518, 518)]
    for c in model.calls:
        assert c[1] == -1.0 and c[2] == 1.0 and -1.0 <= c[3] < c[4] <= 1.0
    for sid in ids:
        out = np.asarray(Image.open(tmp_path / "out" / "amodal_depth" / f"{sid}_depth.png"))
        assert out.shape == (518, 518) and out.dtype == np.uint16 and out.min() >= 0.25 * 65535 - 1
    assert set(res) == set(MO.ALL) and 0 < res["abs_relative_difference"] < 2 and 0 <= res["delta1_acc"] <= 1


def test_load_sample_conventions(tmp_path):
    rng = np.random.default_rng(1)
    d = _make_tree(tmp_path, ["7"], rng)
    s = R.load_sample("7", str(d["occ"]), str(d["whole"]), str(d["obs"]), str(d["gt"]))
    assert s["image"].shape == (3, 518, 518) and 0 <= float(s["image"].min()) and float(s["image"].max()) <= 1
    assert s["whole_mask"].dtype == torch.bool and s["whole_mask"].shape == (1, 518, 518) and 0.2 < float(s["whole_mask"].float().mean()) < 0.5
    assert s["observation"].shape == (1, 518, 518) and s["gt_depth"].shape == (1, 518, 518)
    # nearest-exact: every output value is one of the source values
    src = np.asarray(Image.open(d["obs"] / "7_depth.png")).astype(np.float32) / 65535
    assert np.isin(s["observation"].numpy().ravel(), src.ravel()).all()
