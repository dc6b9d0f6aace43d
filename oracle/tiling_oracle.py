"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the tiled-inference composition of ``hip_ext/tiling.py``: the same tile grid,
the same separable feather weights and the same normalised blend, in plain torch on the host, around any per-tile ``fn``
(the fp32 oracle forward in the tests).  The reference has no tiled path of its own (it squashes inputs to 518 x 518,
infer.py:17,84), so there is nothing of the reference to pin this against: it pins the DEVICE path
(gather kernel + tile bookkeeping) against an independent host formulation (scatter-accumulate)."""
import torch


def tile_origins(size, tile, overlap):
    stride = tile - overlap
    last = size - tile
    o = list(range(0, last + 1, stride))
    if o[-1] != last:
        o.append(last)
    return o


def feather(n, ramp):
    i = torch.arange(n, dtype=torch.float64)
    return torch.minimum(torch.minimum(i + 1, n - i), torch.tensor(float(ramp), dtype=torch.float64)) / ramp


def blend(tiles, origins, H, W, ramp):
    """tiles [B, T, th, tw] -> [B, H, W]: scatter-accumulate of weighted tiles, then one division."""
    B, T, th, tw = tiles.shape
    w2 = feather(th, ramp)[:, None] * feather(tw, ramp)[None, :]
    acc = torch.zeros(B, H, W, dtype=torch.float64)
    wsum = torch.zeros(H, W, dtype=torch.float64)
    for t, (y, x) in enumerate(origins):
        acc[:, y:y + th, x:x + tw] += tiles[:, t].double() * w2
        wsum[y:y + th, x:x + tw] += w2
    return (acc / wsum).float()


def tiled_apply(fn, inputs, tile=518, overlap=70):
    first = next(t for t in inputs if t is not None)
    B, _, H, W = first.shape
    origins = [(y, x) for y in tile_origins(H, tile, overlap) for x in tile_origins(W, tile, overlap)]
    tiles = torch.empty(B, len(origins), tile, tile)
    for b in range(B):
        for t, (y, x) in enumerate(origins):
            crops = [None if i is None else i[b:b + 1, :, y:y + tile, x:x + tile] for i in inputs]
            tiles[b, t] = fn(*crops).reshape(tile, tile)
    return blend(tiles, origins, H, W, min(max(1, overlap), tile // 2)), origins
