#!/usr/bin/env python
"""Batched dataset inference + evaluation: the counterpart of the reference's ``src/scripts/amodel_dav2_inference.py:76-125``.

The reference walks a hard-wired pix2gestalt tree one sample at a time: occlusion image, gt amodal ("whole") mask and the
occluded depth observation, each nearest-exact resized to 518x518, one forward per sample, depth written as a 16-bit PNG
``{id}_depth.png``.  This runner takes the directories as arguments, keeps the same file-name patterns and value conventions
(image / 255, depth PNGs / 65535, mask > 0, guide tensors mapped to [-1, 1]), batches the samples through the HIP forward and,
when a ground-truth depth directory is given, evaluates on the device: per-sample least-squares alignment inside the amodal mask
(src/util/alignment.py) and the depth metrics of src/util/metric.py from one pass per batch.

    python -m src.scripts.amodal_dav2_inference --trained_checkpoint DIR --occ_image_dir A --whole_mask_dir B \
        --observation_depth_dir C --output_dir OUT [--gt_depth_dir D] [--split_file val.txt] [--batch_size 32]

Multi-GPU: launch one process per GPU (``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m
src.scripts.amodal_dav2_inference ...``).  The sample list is sharded contiguously over the ranks (hip_ext.parallel.shard_range),
each rank writes its own PNGs, and the per-sample metric sums are combined with one all-reduce at the end.
"""
import argparse
import json
import os
import sys
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
if PKG not in sys.path:
    sys.path.insert(0, PKG)

RESIZE_HW = (518, 518)   # amodel_dav2_inference.py:73


def sample_ids(occ_image_dir: str, split_file: Optional[str]) -> List[str]:
    """Sample ids: from the split file (lines like ``sa_1234.jpg`` -> ``1234``, script :66-69,77) or every ``*_occlusion.png``."""
    if split_file:
        with open(split_file) as f:
            return [line.strip().split("_")[1].split(".")[0] for line in f if line.strip()]
    return sorted(n[: -len("_occlusion.png")] for n in os.listdir(occ_image_dir) if n.endswith("_occlusion.png"))


def _resize(t: torch.Tensor) -> torch.Tensor:   # InterpolationMode.NEAREST_EXACT (:74)
    return F.interpolate(t[None].float(), size=RESIZE_HW, mode="nearest-exact")[0]


def load_sample(sid: str, occ_image_dir: str, whole_mask_dir: str, observation_depth_dir: str, gt_depth_dir: Optional[str]) -> Dict[str, torch.Tensor]:
    img = np.asarray(Image.open(os.path.join(occ_image_dir, f"{sid}_occlusion.png")).convert("RGB"))
    image = _resize(torch.from_numpy(img.transpose(2, 0, 1).astype(np.float32)) / 255)                       # [3,518,518] in [0,1]
    obs = np.asarray(Image.open(os.path.join(observation_depth_dir, f"{sid}_depth.png"))).astype(np.float32) / 65535
    observation = _resize(torch.from_numpy(obs)[None])                                                        # [1,518,518] in [0,1]
    wm = np.asarray(Image.open(os.path.join(whole_mask_dir, f"{sid}_whole_mask.png")))
    if wm.ndim == 3:
        wm = wm[..., 0]
    whole_mask = _resize(torch.from_numpy(wm.astype(np.float32))[None]) > 0                                   # [1,518,518] bool
    out = {"image": image, "observation": observation, "whole_mask": whole_mask}
    if gt_depth_dir:
        gt = np.asarray(Image.open(os.path.join(gt_depth_dir, f"{sid}_depth.png"))).astype(np.float32) / 65535
        out["gt_depth"] = _resize(torch.from_numpy(gt)[None])
    return out


def run(model: Callable, ids: List[str], occ_image_dir: str, whole_mask_dir: str, observation_depth_dir: str, output_dir: str,
        gt_depth_dir: Optional[str] = None, batch_size: int = 32, device: str = "cuda", evaluate: Optional[Callable] = None,
        group=None) -> Dict[str, float]:
    """Runs the model over ``ids`` in batches, writes ``{output_dir}/amodal_depth/{id}_depth.png`` (uint16, depth * 65535, :124-125)
    and returns the per-sample metrics averaged over the samples when ground truth is available.  ``evaluate(pred, gt, mask) -> dict``
    (applied to one sample at a time) defaults to the device path."""
    out_depth = os.path.join(output_dir, "amodal_depth")
    os.makedirs(out_depth, exist_ok=True)
    # Metrics follow the reference's evaluation semantics -- one sample at a time (its loader runs batch size 1), then the mean over
    # the samples -- so the result does not depend on how the samples are batched or sharded (silog_rmse and log10 are not linear
    # in the batch).  ``evaluate(pred, gt, mask) -> dict`` is applied per sample; the default computes the per-image sums of the
    # whole batch in one device pass and derives each sample's metrics from its own row.
    per_sample = None
    if gt_depth_dir and evaluate is None:
        from src.util import alignment, metric

        def per_sample(pred, gt, mask):
            ss = alignment.scale_shift_least_square(gt, pred, mask)            # per-sample (scale, shift), fp64 [B,2]
            import hip_ext as H
            sums = H.depth_eval(pred.contiguous().float(), gt.contiguous().float(), mask.contiguous(), scale_shift=ss.float().contiguous(),
                                clip=(1e-3, 1.0))                              # aligned prediction clamped to the PNG depth range
            return [{k: float(v) for k, v in metric._from_sums(sums[b:b + 1]).items()} for b in range(pred.shape[0])]
    elif gt_depth_dir:
        def per_sample(pred, gt, mask):
            return [evaluate(pred[b:b + 1], gt[b:b + 1], mask[b:b + 1]) for b in range(pred.shape[0])]
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1:   # this rank's contiguous share of the sample list
        from hip_ext.parallel import shard_range
        lo, hi = shard_range(len(ids), dist.get_rank(group), world)
        ids = ids[lo:hi]
    totals: Dict[str, float] = {}
    count = 0
    for i in range(0, len(ids), batch_size):
        chunk = ids[i:i + batch_size]
        samples = [load_sample(s, occ_image_dir, whole_mask_dir, observation_depth_dir, gt_depth_dir) for s in chunk]
        image = torch.stack([s["image"] for s in samples]).to(device)
        mask = torch.stack([s["whole_mask"] for s in samples]).to(device)
        obs = torch.stack([s["observation"] for s in samples]).to(device)
        with torch.no_grad():   # both guides in [-1, 1] (:113-118)
            depth = model(image, guide_rgb=None, guide_mask=mask.float() * 2 - 1, observation=obs * 2 - 1)
        depth = depth.reshape(len(chunk), *RESIZE_HW)
        d16 = (depth.detach().float().cpu().numpy() * 65535.0).astype(np.uint16)
        for sid, arr in zip(chunk, d16):
            Image.fromarray(arr).save(os.path.join(out_depth, f"{sid}_depth.png"))
        if gt_depth_dir:
            gt = torch.stack([s["gt_depth"] for s in samples]).to(device).reshape(len(chunk), *RESIZE_HW)
            valid = mask.reshape(len(chunk), *RESIZE_HW) & (gt > 0)
            for res in per_sample(depth.float(), gt, valid):
                for k, v in res.items():
                    totals[k] = totals.get(k, 0.0) + v
            count += len(chunk)
    if world > 1 and gt_depth_dir:
        # metric names are fixed by the evaluator, so every rank (also one with an empty share) builds the same vector
        keys = sorted(totals) if totals else None
        names = [None] * world
        dist.all_gather_object(names, keys, group=group)          # control plane only: a list of metric names
        keys = next((k for k in names if k), [])
        vec = torch.tensor([totals.get(k, 0.0) for k in keys] + [float(count)], dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(vec, group=group)
        count = int(vec[-1].item())
        totals = {k: float(v) for k, v in zip(keys, vec[:-1].tolist())}
    return {k: v / count for k, v in totals.items()} if count else {}


def main(argv=None):
    ap = argparse.ArgumentParser(description="Batched amodal depth inference / evaluation (MI355X-native HIP path)")
    ap.add_argument("--trained_checkpoint", type=str, default=None, help="local directory with config.json + model.safetensors")
    ap.add_argument("--encoder", type=str, default="vitl")
    ap.add_argument("--guide_type", type=str, default="mask+observation")
    ap.add_argument("--loss_stategy", type=str, default="entire_target_object")
    ap.add_argument("--occ_image_dir", required=True)
    ap.add_argument("--whole_mask_dir", required=True)
    ap.add_argument("--observation_depth_dir", required=True)
    ap.add_argument("--gt_depth_dir", default=None)
    ap.add_argument("--split_file", default=None)
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:   # one process per GPU; the process group is created before anything touches the device
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.device.startswith("cuda"):
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            a.device = f"cuda:{local}"
            dist.init_process_group("nccl", device_id=torch.device(a.device))
        else:
            dist.init_process_group("gloo")
    from src.models import get_model
    model = get_model("AmodalDAv2", guide_type=a.guide_type, loss_stategy=a.loss_stategy, encoder=a.encoder, pretrained=False)
    if a.trained_checkpoint:
        model = model.from_pretrained(a.trained_checkpoint, strict=True)
    else:   # no checkpoints ship with this repository: deterministic synthetic weights keep the runner usable end to end
        from src.util.synth_weights import fill_state_dict_
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        fill_state_dict_(sd, 0)
        model.load_state_dict(sd, strict=True)
    model = model.eval().to(a.device)
    ids = sample_ids(a.occ_image_dir, a.split_file)
    metrics = run(model, ids, a.occ_image_dir, a.whole_mask_dir, a.observation_depth_dir, a.output_dir, a.gt_depth_dir, a.batch_size, a.device)
    if metrics and rank == 0:
        with open(os.path.join(a.output_dir, "metrics.json"), "w") as f:
            json.dump(metrics, f, indent=1)
        print(json.dumps(metrics))
    if rank == 0:
        print(f"wrote {len(ids)} depth maps to {os.path.join(a.output_dir, 'amodal_depth')}" + (f" ({world} ranks)" if world > 1 else ""))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
