"""CPU: the bookkeeping of tiled inference (hip_ext.tiling.tile_origins -- pure Python) and the host restatement of the overlap
blend (oracle/tiling_oracle.py) that the GPU test compares the device kernel with."""
import pytest
import torch

from hip_ext import HipExtError
from hip_ext.tiling import tile_origins, tiled_apply
from oracle import tiling_oracle as TO


@pytest.mark.parametrize("size", [518, 519, 600, 966, 1036, 1500, 2072])
@pytest.mark.parametrize("overlap", [0, 14, 70, 140, 258])
def test_tile_origins_cover_every_pixel_with_the_requested_overlap(size, overlap):
    o = tile_origins(size, 518, overlap)
    assert o == TO.tile_origins(size, 518, overlap)
    assert o[0] == 0 and o[-1] == size - 518 and o == sorted(set(o))
    covered = torch.zeros(size, dtype=torch.int32)
    for y in o:
        covered[y:y + 518] += 1
    assert int(covered.min()) >= 1
    assert all(b - a <= 518 - overlap for a, b in zip(o, o[1:]))        # neighbouring tiles overlap by at least `overlap`


def test_tile_origins_reject_small_images_and_bad_overlap():
    with pytest.raises(HipExtError):
        tile_origins(500, 518, 70)
    with pytest.raises(HipExtError):
        tile_origins(600, 518, 518)
    with pytest.raises(HipExtError):     # the product path has no CPU fallback
        tiled_apply(lambda x: x, [torch.zeros(1, 3, 600, 600)])


def test_engine_rejects_images_beyond_the_row_limit_with_a_pointer_to_tiling():
    from hip_ext.engine import DepthEngine
    eng = DepthEngine.__new__(DepthEngine)
    assert eng.max_batch(518, 518) == 62 and eng.max_batch(4096, 4088) == 1
    with pytest.raises(HipExtError, match="tiled"):
        eng.max_batch(4102, 4102)


def test_host_blend_is_a_partition_of_unity_and_cross_fades_linearly():
    th, tw, H, W, ov = 20, 30, 33, 50, 10
    origins = [(y, x) for y in TO.tile_origins(H, th, ov) for x in TO.tile_origins(W, tw, ov)]
    const = torch.full((1, len(origins), th, tw), 0.625)
    assert torch.allclose(TO.blend(const, origins, H, W, ov), torch.full((1, H, W), 0.625), atol=1e-7)
    # two tiles side by side holding 0 and 1: inside the overlap the result ramps monotonically from the left tile's value to the right one's
    o2 = [(0, 0), (0, 20)]
    tiles = torch.stack([torch.zeros(th, tw), torch.ones(th, tw)])[None]
    out = TO.blend(tiles, o2, th, 50, ov)[0, 5]
    assert float(out[:20].abs().max()) == 0.0 and float((out[30:] - 1).abs().max()) == 0.0
    ramp = out[20:30]
    assert torch.all(ramp[1:] > ramp[:-1]) and 0.0 < float(ramp[0]) < float(ramp[-1]) < 1.0
