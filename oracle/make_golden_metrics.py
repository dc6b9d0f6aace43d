"""Generates tests/golden/metrics/cases.npz by calling the REAL reference (``/root/reference/src/util/metric.py`` and
``alignment.py``) on seeded synthetic depth maps.  Run in the build container only (the reference does not travel):

    python oracle/make_golden_metrics.py

Inputs and the reference's outputs are stored; tests replay them against oracle/metrics_oracle.py (CPU) and the HIP path (GPU).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ADA_REFERENCE_ROOT", "/root/reference")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    # metric.py imports skimage's canny for the edge metrics (not part of these goldens): stand-in module
    sk = types.ModuleType("skimage"); skf = types.ModuleType("skimage.feature"); skf.canny = None
    sys.modules.setdefault("skimage", sk); sys.modules.setdefault("skimage.feature", skf)
    metric = _load(os.path.join(REF, "src", "util", "metric.py"), "_ref_metric")
    align = _load(os.path.join(REF, "src", "util", "alignment.py"), "_ref_alignment")
    rng = np.random.default_rng(20240607)
    out = {}
    names = ["abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log", "log10", "delta1_acc", "delta2_acc",
             "delta3_acc", "i_rmse", "silog_rmse"]
    cases = {"b1_small": (1, 37, 53), "b3_mid": (3, 74, 74)}
    for cname, (B, Hh, Ww) in cases.items():
        gt = rng.uniform(0.5, 10.0, size=(B, Hh, Ww)).astype(np.float32)
        pred = (gt * rng.uniform(0.7, 1.4, size=gt.shape) + rng.normal(0, 0.05, size=gt.shape)).astype(np.float32)
        pred = np.maximum(pred, 0.05)
        mask = rng.uniform(size=gt.shape) > 0.3
        out[f"{cname}.gt"], out[f"{cname}.pred"], out[f"{cname}.mask"] = gt, pred, mask
        for n in names:
            fn = getattr(metric, n)
            v = fn(torch.from_numpy(pred.copy()), torch.from_numpy(gt.copy()), torch.from_numpy(mask.copy()))
            out[f"{cname}.{n}"] = np.float64(float(v))
        # alignment: per image, relative prediction = affine-distorted gt + noise
        rel = ((gt - 1.3) / 2.7 + rng.normal(0, 0.01, size=gt.shape)).astype(np.float32)
        out[f"{cname}.rel"] = rel
        sc, sh = [], []
        for b in range(B):
            aligned, s, t = align.align_depth_least_square(gt[b], rel[b], mask[b], return_scale_shift=True)
            sc.append(float(np.asarray(s).reshape(-1)[0])); sh.append(float(np.asarray(t).reshape(-1)[0]))
            if b == 0 and cname == "b1_small":
                out[f"{cname}.aligned0"] = np.asarray(aligned, dtype=np.float64)
        out[f"{cname}.scale"], out[f"{cname}.shift"] = np.array(sc), np.array(sh)
    path = os.path.join(ROOT, "tests", "golden", "metrics", "cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
