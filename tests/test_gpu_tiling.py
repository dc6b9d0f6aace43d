"""Tiled inference for inputs larger than 518 x 518: the device path (hip_ext.tiling: crops batched
through the HIP forward, ada_tile_blend_fwd) against the host composition of oracle/tiling_oracle.py around the fp32 oracle."""
import pytest
import torch

from _cases import build_product_model, oracle_forward, rel_l1, synth_state_dict
from src.util.synth_weights import make_inputs

pytestmark = pytest.mark.gpu


def test_tile_blend_kernel_matches_host_scatter_formulation(hip):
    from oracle import tiling_oracle as TO
    B, th, tw, H, W, ov = 2, 28, 42, 61, 100, 14
    origins = [(y, x) for y in TO.tile_origins(H, th, ov) for x in TO.tile_origins(W, tw, ov)]
    g = torch.Generator().manual_seed(3)
    tiles = torch.rand(B, len(origins), th, tw, generator=g)
    want = TO.blend(tiles, origins, H, W, ov)
    oy = torch.tensor([o[0] for o in origins], dtype=torch.int32, device="cuda")
    ox = torch.tensor([o[1] for o in origins], dtype=torch.int32, device="cuda")
    out = torch.full((B, H, W), float("nan"), device="cuda")
    hip.tile_blend(tiles.cuda(), oy, ox, H, W, ov, out)
    assert torch.allclose(out.cpu(), want, atol=2e-6), float((out.cpu() - want).abs().max())
    # a constant field stays constant (the weights are normalised), a single tile is returned unchanged
    ones = torch.full((1, len(origins), th, tw), 0.37, device="cuda")
    o1 = torch.empty(1, H, W, device="cuda")
    hip.tile_blend(ones, oy, ox, H, W, ov, o1)
    assert float((o1 - 0.37).abs().max()) < 1e-6
    single = torch.rand(1, 1, th, tw, generator=g).cuda()
    o2 = torch.empty(1, th, tw, device="cuda")
    z = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.tile_blend(single, z, z, th, tw, ov, o2)
    assert torch.allclose(o2, single[0], atol=1e-7)


def test_tile_origins_cover_the_image():
    from hip_ext.tiling import tile_origins
    from oracle import tiling_oracle as TO
    for size in (518, 519, 700, 966, 1036, 2000):
        for ov in (0, 70, 140):
            o = tile_origins(size, 518, ov)
            assert o == TO.tile_origins(size, 518, ov)
            assert o[0] == 0 and o[-1] == size - 518 and all(b - a <= 518 - ov for a, b in zip(o, o[1:]))


def test_tiled_amodal_forward_matches_oracle_composition(hip):
    """ViT-S amodal model on a 644 x 742 input: 2 x 2 tiles of 518 with at least 70 pixels of overlap."""
    from hip_ext.tiling import tiled_amodal_forward
    from oracle import tiling_oracle as TO
    case = dict(kind="amodal", encoder="vits", guide_type="mask+observation", loss="entire_target_object")
    model = build_product_model(case)
    sd = synth_state_dict(model)
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    x, _, mask, obs = make_inputs(1, 644, 742, seed=9)
    with torch.no_grad():
        got = tiled_amodal_forward(model, x.cuda(), mask.cuda(), obs.cuda()).cpu()
    assert list(got.shape) == [1, 1, 644, 742]

    def fn(xc, gr, gm, ob):
        return oracle_forward(sd, case, xc, None, gm, ob)

    want, origins = TO.tiled_apply(fn, [x, None, mask, obs])
    assert len(origins) == 4
    err = rel_l1(got[:, 0], want)
    print(f"tiled 644x742 ViT-S: rel-L1 vs oracle composition = {err:.3e}")
    assert err <= 1e-3
    # inside a region covered by a single tile the tiled result is that tile's plain forward
    with torch.no_grad():
        t0 = model(x[:, :, :518, :518].cuda(), guide_rgb=None, guide_mask=mask[:, :, :518, :518].cuda(), observation=obs[:, :, :518, :518].cuda()).cpu()
    assert torch.allclose(got[0, 0, :100, :200], t0[0, 0, :100, :200], atol=1e-6)
