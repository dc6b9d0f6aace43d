"""Tiled inference for inputs larger than the network's native 518 x 518.

The reference squashes every input to 518 x 518 (infer.py:17,84) and has no tiling of its own; this module is the
"tiled 518^2 inference with overlap-blend" the survey lists as the next step for arbitrary resolutions.  The image is cut
into ``tile x tile`` crops on a regular grid whose last row / column is aligned to the image edge, all crops of all images go
through the network as ONE batch (the forward is batch-invariant), and the per-tile predictions are merged on the device by
``ada_tile_blend_fwd`` with a separable linear feather over the overlap.  Everything stays in HBM.

``tile_origins`` and the weight definition are restated in ``oracle/tiling_oracle.py`` (CPU, test infrastructure), against which
``tests/test_gpu_tiling.py`` checks this path.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import HipExtError, tile_blend

PATCH = 14


def tile_origins(size: int, tile: int, overlap: int) -> List[int]:
    """Origins of the tiles along one axis: stride ``tile - overlap``, the last tile aligned to the end of the axis."""
    if size < tile:
        raise HipExtError(f"tiled inference needs at least {tile} pixels along every axis (got {size})")
    if not (0 <= overlap < tile):
        raise HipExtError("overlap must be in [0, tile)")
    stride = tile - overlap
    last = size - tile
    origins = list(range(0, last + 1, stride))
    if origins[-1] != last:
        origins.append(last)
    return origins


@torch.no_grad()
def tiled_apply(fn: Callable[..., torch.Tensor], inputs: Sequence[Optional[torch.Tensor]], tile: int = 518, overlap: int = 70,
                max_tiles_per_call: int = 64) -> torch.Tensor:
    """``fn(*crops) -> [n, 1, tile, tile]`` (or ``[n, tile, tile]``) applied to every ``tile x tile`` crop of ``inputs``
    (``[B, C, H, W]`` tensors on a HIP device, ``None`` entries passed through); returns the blended ``[B, H, W]`` map."""
    first = next(t for t in inputs if t is not None)
    if not first.is_cuda:
        raise HipExtError("tiled_apply: inputs must live on a HIP device (no CPU fallback)")
    if tile % PATCH:
        raise HipExtError(f"tile size {tile} must be a multiple of the {PATCH}-pixel patch")
    B, _, H, W = first.shape
    oys, oxs = tile_origins(H, tile, overlap), tile_origins(W, tile, overlap)
    origins: List[Tuple[int, int]] = [(y, x) for y in oys for x in oxs]
    T = len(origins)
    dev = first.device
    tiles = torch.empty(B * T, tile, tile, dtype=torch.float32, device=dev)
    oy = torch.tensor([o[0] for o in origins], dtype=torch.int32, device=dev)
    ox = torch.tensor([o[1] for o in origins], dtype=torch.int32, device=dev)
    # crop (b, t) = job b * T + t.  Every chunk of jobs is cut out of each input by ONE gather (advanced indexing with broadcast
    # row / column index tensors) and its predictions land in the tile stack by ONE copy: no per-tile device work is issued from Python.
    ar = torch.arange(tile, device=dev)
    rows = oy.long()[:, None] + ar          # [T, tile]
    cols = ox.long()[:, None] + ar
    jobs = torch.arange(B * T, device=dev)
    for i in range(0, B * T, max_tiles_per_call):
        j = jobs[i:i + max_tiles_per_call]
        jb, jt = j // T, j % T
        crops = [None if inp is None else
                 inp[jb[:, None, None, None], torch.arange(inp.shape[1], device=dev)[None, :, None, None], rows[jt][:, None, :, None], cols[jt][:, None, None, :]]
                 for inp in inputs]
        tiles[i:i + j.numel()] = fn(*crops).reshape(j.numel(), tile, tile)
    tiles = tiles.view(B, T, tile, tile)
    out = torch.empty(B, H, W, dtype=torch.float32, device=first.device)
    ramp = max(1, overlap)
    tile_blend(tiles, oy, ox, H, W, min(ramp, tile // 2), out)
    return out


def tiled_amodal_forward(model, x: torch.Tensor, guide_mask: torch.Tensor, observation: torch.Tensor, guide_rgb: Optional[torch.Tensor] = None,
                         tile: int = 518, overlap: int = 70) -> torch.Tensor:
    """AmodalDAv2 on an image larger than 518 x 518: ``[B, 1, H, W]`` like ``model.forward``."""
    def fn(xc, gr, gm, ob):
        return model(xc, guide_rgb=gr, guide_mask=gm, observation=ob)
    return tiled_apply(fn, [x, guide_rgb, guide_mask, observation], tile, overlap).unsqueeze(1)


def tiled_raw_forward(model, x_norm: torch.Tensor, tile: int = 518, overlap: int = 70) -> torch.Tensor:
    """Raw Depth-Anything-V2 (ImageNet-normalised input) on an image larger than 518 x 518: ``[B, H, W]``."""
    return tiled_apply(lambda xc: model(xc), [x_norm], tile, overlap)
