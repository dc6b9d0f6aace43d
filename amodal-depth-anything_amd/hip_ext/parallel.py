"""Multi-GPU batched inference: one process per GPU, images sharded over ranks, ONE all-gather of the per-image
depth maps (RCCL over xGMI when the process group is 'nccl'; the same code runs on 'gloo' for CPU tests).

The forward pass has no cross-image operation (no BatchNorm; LayerNorm and attention are per image -- SURVEY.md §8e),
so weights are replicated and the only exchange is the output gather: [B_local, 1, H, W] fp32 per rank
(1.07 MB per 518x518 image).  The reference has no counterpart: its inference is single-process (infer.py:59-69).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of `total` items for `rank` (first `total % world` ranks get one extra)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_forward(forward: Callable[..., torch.Tensor], inputs: Sequence[Optional[torch.Tensor]], group=None,
                    gather: bool = True) -> torch.Tensor:
    """Runs ``forward(*inputs_local)`` on this rank's slice of the batch dimension and (optionally) all-gathers the
    outputs so that every rank returns the full ``[B, ...]`` result in the original order.

    ``inputs`` are the *global* batch tensors (``None`` entries are passed through).  Uneven batches are handled by
    padding the gathered buffers to the largest shard; ranks with an empty shard contribute nothing.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    batch = next(t.shape[0] for t in inputs if t is not None)
    lo, hi = shard_range(batch, rank, world)
    local = None
    if hi > lo:
        local = forward(*[None if t is None else t[lo:hi] for t in inputs])
    if world == 1 or not gather:
        return local
    # every rank must know the per-item shape/dtype/device even when its own shard is empty
    meta = [None] * world
    dist.all_gather_object(meta, None if local is None else (tuple(local.shape[1:]), str(local.dtype), str(local.device)), group=group)
    item_shape, dtype_s, dev_s = next(m for m in meta if m is not None)
    dtype = getattr(torch, dtype_s.split(".")[-1])
    device = local.device if local is not None else torch.device(dev_s if not dev_s.startswith("cuda") else f"cuda:{torch.cuda.current_device()}")
    sizes = [shard_range(batch, r, world) for r in range(world)]
    cap = max(h - l for l, h in sizes)
    send = torch.zeros((cap,) + item_shape, dtype=dtype, device=device)
    if local is not None:
        send[: hi - lo] = local
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    return torch.cat([recv[r][: h - l] for r, (l, h) in enumerate(sizes)], dim=0)
