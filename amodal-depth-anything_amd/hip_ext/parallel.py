"""Multi-GPU batched inference: one process per GPU, images sharded over ranks, ONE all-gather of the per-image
depth maps (RCCL over xGMI when the process group is 'nccl'; the same code runs on 'gloo' for the CPU tests).

The forward pass has no cross-image operation (no BatchNorm; LayerNorm and attention are per image -- SURVEY.md §8e),
so weights are replicated and the only exchange is the output gather: [B_local, 1, H, W] fp32 per rank
(1.07 MB per 518x518 image).  The reference has no counterpart: its inference is single-process (infer.py:59-69).

``DepthGather`` owns the pre-sized buffers of that exchange and is the ONE collective path of this package:
``sharded_forward`` (global batch in, global batch out), ``bench.py --gpus N`` (per-rank synthetic shards, asynchronous
gather riding under the next forward) go through it; ``src/scripts/amodal_dav2_inference.py`` shards its sample list with the same
``shard_range`` and combines per-sample metric sums with one all-reduce (it writes files, it gathers no maps).
No pickled metadata, no per-call allocation: shard sizes follow from (batch, world) alone, and the item shape / dtype are DECLARED
(every rank, also one with an empty shard, builds the same buffers) and checked against what the forward returned.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of `total` items for `rank` (first `total % world` ranks get one extra)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class DepthGather:
    """Pre-sized all-gather of per-item outputs over a process group.

    ``batch`` is the GLOBAL number of items; rank r owns ``shard_range(batch, r, world)``.  When the shards are even the
    collective writes straight into the ``[batch, *item_shape]`` result; otherwise every rank sends a buffer padded to the
    largest shard and the valid rows are compacted afterwards."""

    def __init__(self, batch: int, item_shape: Sequence[int], dtype: torch.dtype, device, group=None):
        self.group = group
        self.rank, self.world = _world(group)
        # an initialised process group always takes the collective path, also with ONE rank (the 1-rank `nccl` group of the GPU suite
        # pushes a real model output through RCCL this way)
        self._collective = dist.is_available() and dist.is_initialized()
        self.batch, self.item_shape = int(batch), tuple(int(s) for s in item_shape)
        self.spans: List[Tuple[int, int]] = [shard_range(self.batch, r, self.world) for r in range(self.world)]
        self.lo, self.hi = self.spans[self.rank]
        self.cap = max(h - l for l, h in self.spans) if self.spans else 0
        self.even = all(h - l == self.cap for l, h in self.spans)
        self.recv = torch.empty((self.world * self.cap,) + self.item_shape, dtype=dtype, device=device)
        self.send = None if self.even else torch.zeros((self.cap,) + self.item_shape, dtype=dtype, device=device)
        self._pending = None

    def start(self, local: Optional[torch.Tensor]):
        """Issues the collective for this rank's ``[hi - lo, *item_shape]`` output (``None`` for an empty shard) and returns
        at once; ``finish()`` completes it.  At most one gather is in flight per object."""
        self.finish_pending()
        n = self.hi - self.lo
        if n > 0:
            assert local is not None and tuple(local.shape) == (n,) + self.item_shape, \
                f"rank {self.rank}: expected {(n,) + self.item_shape}, got {None if local is None else tuple(local.shape)}"
        if n > 0:
            assert local.dtype == self.recv.dtype, f"rank {self.rank}: gather declared {self.recv.dtype}, got {local.dtype}"
        if not self._collective:   # no process group: the "gather" of one rank is a copy into the result buffer
            if n > 0:
                self.recv[:n].copy_(local)
            return
        if self.even:
            src = local.contiguous()
        else:
            src = self.send
            if n > 0:
                src[:n].copy_(local)
        self._pending = dist.all_gather_into_tensor(self.recv, src, group=self.group, async_op=True)

    def finish_pending(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def finish(self) -> torch.Tensor:
        """Waits for the gather in flight and returns the global ``[batch, *item_shape]`` result (a view of the receive
        buffer for even shards; it is overwritten by the next ``start``)."""
        self.finish_pending()
        if self.even:
            return self.recv[: self.batch]
        return torch.cat([self.recv[r * self.cap: r * self.cap + (h - l)] for r, (l, h) in enumerate(self.spans)], dim=0)

    def gather(self, local: Optional[torch.Tensor]) -> torch.Tensor:
        self.start(local)
        return self.finish()


_gathers = {}


def sharded_forward(forward: Callable[..., torch.Tensor], inputs: Sequence[Optional[torch.Tensor]], group=None,
                    gather: bool = True, item_shape: Optional[Sequence[int]] = None,
                    dtype: torch.dtype = torch.float32) -> Optional[torch.Tensor]:
    """Runs ``forward(*inputs_local)`` on this rank's slice of the batch dimension and (optionally) all-gathers the
    outputs so that every rank returns the full ``[B, ...]`` result in the original order.

    ``inputs`` are the *global* batch tensors (``None`` entries are passed through).  ``item_shape`` / ``dtype`` describe
    one output item; they default to the depth-map convention of this package, ``[1, H, W]`` fp32 with H, W the trailing
    dims of the first input -- every rank derives them locally, including a rank whose shard is empty.  A forward with another
    output convention (the raw model's ``[B, H, W]``, a half-precision output) must pass them: a mismatch raises on the rank
    that sees it instead of building different collective buffers on different ranks."""
    rank, world = _world(group)
    first = next(t for t in inputs if t is not None)
    batch = first.shape[0]
    lo, hi = shard_range(batch, rank, world)
    local = None
    if hi > lo:
        local = forward(*[None if t is None else t[lo:hi] for t in inputs])
    if not gather or not (dist.is_available() and dist.is_initialized()):
        return local
    if item_shape is None:
        item_shape = (1,) + tuple(first.shape[-2:])
    item_shape = tuple(int(v) for v in item_shape)
    if local is not None and (tuple(local.shape[1:]) != item_shape or local.dtype != dtype):
        raise ValueError(f"sharded_forward: the forward returned items of shape {tuple(local.shape[1:])} / {local.dtype} but "
                         f"{item_shape} / {dtype} was declared (pass item_shape= / dtype= for other output conventions)")
    key = (batch, tuple(item_shape), dtype, str(first.device), id(group))
    g = _gathers.get(key)
    if g is None:
        if len(_gathers) > 8:
            _gathers.clear()
        g = _gathers[key] = DepthGather(batch, item_shape, dtype, first.device, group)
    return g.gather(local).clone()
