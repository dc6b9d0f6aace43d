"""Depth metrics of the reference's evaluation (``src/util/metric.py:37-160``) on the device.

Same function names, arguments and reductions as the reference -- per-image masked mean, then the mean over the batch
(``log10``: mean over every valid pixel of the batch; ``silog_rmse``: sqrt of the batch mean, times 100) -- but every
function reads its per-image sums from ONE pass of ``ada_depth_eval_fwd`` over the maps instead of building a full-size
temporary per metric on the host.  ``depth_metrics`` returns all of them from a single pass.  Inputs are ``[B, H, W]``
(or ``[B, 1, H, W]``) tensors on a HIP device; there is no CPU fallback.

Not built: the edge metrics (``EdgeAcc`` / ``EdgeComp`` / ``soft_edge_error``, metric.py:221-328), which sit on skimage's
Canny detector.
"""
from typing import Dict, Optional

import torch

import hip_ext as H

__all__ = ["MetricTracker", "depth_metrics", "abs_relative_difference", "squared_relative_difference", "rmse_linear", "rmse_log",
           "log10", "threshold_percentage", "delta1_acc", "delta2_acc", "delta3_acc", "i_rmse", "silog_rmse"]


class MetricTracker:
    """Running averages per key (metric.py:13-34; same ``update`` / ``avg`` / ``result`` / ``reset`` behaviour, no pandas)."""

    def __init__(self, *keys, writer=None):
        self.writer = writer
        self._keys = list(keys)
        self.reset()

    def reset(self):
        self._total = {k: 0.0 for k in self._keys}
        self._counts = {k: 0 for k in self._keys}

    def update(self, key, value, n=1):
        if self.writer is not None:
            self.writer.add_scalar(key, value)
        if key not in self._total:
            raise KeyError(key)
        self._total[key] += float(value) * n
        self._counts[key] += n

    def avg(self, key):
        return self._total[key] / self._counts[key] if self._counts[key] else 0.0

    def result(self) -> Dict[str, float]:
        return {k: self.avg(k) for k in self._keys}


def _sums(output: torch.Tensor, target: torch.Tensor, valid_mask: Optional[torch.Tensor]) -> torch.Tensor:
    if output.shape != target.shape:
        raise ValueError(f"prediction {tuple(output.shape)} and target {tuple(target.shape)} differ in shape")
    if output.dim() < 2:
        raise ValueError("depth maps need at least [H, W]")
    hw = output.shape[-2:]
    o = output.reshape(-1, *hw).contiguous().float()
    t = target.reshape(-1, *hw).contiguous().float()
    m = None
    if valid_mask is not None:
        m = valid_mask.expand_as(output).reshape(-1, *hw).contiguous()
        if m.dtype not in (torch.bool, torch.uint8):
            m = m != 0
    return H.depth_eval(o, t, m)


def _per_image(sums: torch.Tensor, idx: int) -> torch.Tensor:
    return sums[:, idx] / sums[:, H.EVAL_N]


def _from_sums(s: torch.Tensor) -> Dict[str, torch.Tensor]:
    n = s[:, H.EVAL_N]
    first = s[:, H.EVAL_LOG_SQ] / n
    second = (s[:, H.EVAL_LOG] ** 2) / (n * n)
    return {
        "abs_relative_difference": _per_image(s, H.EVAL_ABS_REL).mean(),
        "squared_relative_difference": _per_image(s, H.EVAL_SQ_REL).mean(),
        "rmse_linear": torch.sqrt(_per_image(s, H.EVAL_SQ)).mean(),
        "rmse_log": torch.sqrt(first).mean(),
        "log10": s[:, H.EVAL_LOG10_ABS].sum() / n.sum(),
        "delta1_acc": _per_image(s, H.EVAL_D1).mean(),
        "delta2_acc": _per_image(s, H.EVAL_D2).mean(),
        "delta3_acc": _per_image(s, H.EVAL_D3).mean(),
        "i_rmse": torch.sqrt(_per_image(s, H.EVAL_INV_SQ)).mean(),
        "silog_rmse": torch.sqrt(torch.mean(first - second)) * 100,
    }


def depth_metrics(output, target, valid_mask=None) -> Dict[str, torch.Tensor]:
    """Every metric of this module from one pass over the maps (0-dim fp64 tensors on the device)."""
    return _from_sums(_sums(output, target, valid_mask))


def abs_relative_difference(output, target, valid_mask=None):   # metric.py:37-48
    return _per_image(_sums(output, target, valid_mask), H.EVAL_ABS_REL).mean()


def squared_relative_difference(output, target, valid_mask=None):   # metric.py:51-64
    return _per_image(_sums(output, target, valid_mask), H.EVAL_SQ_REL).mean()


def rmse_linear(output, target, valid_mask=None):   # metric.py:67-79
    return torch.sqrt(_per_image(_sums(output, target, valid_mask), H.EVAL_SQ)).mean()


def rmse_log(output, target, valid_mask=None):   # metric.py:82-92
    return torch.sqrt(_per_image(_sums(output, target, valid_mask), H.EVAL_LOG_SQ)).mean()


def log10(output, target, valid_mask=None):   # metric.py:95-102
    s = _sums(output, target, valid_mask)
    return s[:, H.EVAL_LOG10_ABS].sum() / s[:, H.EVAL_N].sum()


def threshold_percentage(output, target, threshold_val, valid_mask=None):   # metric.py:106-120
    idx = {1.25: H.EVAL_D1, 1.25 ** 2: H.EVAL_D2, 1.25 ** 3: H.EVAL_D3}.get(threshold_val)
    if idx is None:
        raise ValueError("threshold_percentage: the device pass counts the thresholds 1.25, 1.25**2 and 1.25**3")
    return _per_image(_sums(output, target, valid_mask), idx).mean()


def delta1_acc(pred, gt, valid_mask):   # metric.py:123-124
    return threshold_percentage(pred, gt, 1.25, valid_mask)


def delta2_acc(pred, gt, valid_mask):   # metric.py:127-128
    return threshold_percentage(pred, gt, 1.25 ** 2, valid_mask)


def delta3_acc(pred, gt, valid_mask):   # metric.py:131-132
    return threshold_percentage(pred, gt, 1.25 ** 3, valid_mask)


def i_rmse(output, target, valid_mask=None):   # metric.py:135-147
    return torch.sqrt(_per_image(_sums(output, target, valid_mask), H.EVAL_INV_SQ)).mean()


def silog_rmse(depth_pred, depth_gt, valid_mask=None):   # metric.py:150-163
    return _from_sums(_sums(depth_pred, depth_gt, valid_mask))["silog_rmse"]
