"""CPU plumbing of the CLI pipeline (BASELINE.json config 1): infer_single_image runs end to end with the fp32 CPU
oracle standing in for both networks (the product networks have no CPU path), writes the two reference-named PNGs and
blends exactly as reference infer.py:30-44 / 94-103 describe."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infer  # noqa: E402
from _cases import build_product_model, synth_state_dict  # noqa: E402
from oracle import dav2_oracle as O  # noqa: E402


class _OracleRaw:
    def __init__(self):
        case = dict(kind="raw", encoder="vits", features=64, out_channels=[48, 96, 192, 384])
        self.sd = synth_state_dict(build_product_model(case))

    def __call__(self, x):
        return O.raw_forward(self.sd, "vits", x) + 0.5


class _OracleAmodal:
    def __init__(self):
        case = dict(kind="amodal", encoder="vits", guide_type="mask+observation", loss="entire_target_object")
        self.sd = synth_state_dict(build_product_model(case))
        self.calls = []

    def __call__(self, x, guide_rgb=None, guide_mask=None, observation=None):
        self.calls.append((x, guide_mask, observation))
        return O.amodal_forward(self.sd, "vits", "mask+observation", "entire_target_object", x, None, guide_mask, observation)


def test_infer_single_image_cpu_plumbing(tmp_path):
    rng = np.random.default_rng(0)
    img = (rng.random((60, 80, 3)) * 255).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "case.jpg")
    mask = np.zeros((32, 32), dtype=np.uint16)
    mask[8:20, 10:26] = 65535
    Image.fromarray(mask).save(tmp_path / "case_mask.png")
    amodal = _OracleAmodal()
    raw_out, agg_out = infer.infer_single_image(str(tmp_path / "case.jpg"), str(tmp_path / "case_mask.png"), str(tmp_path / "out"),
                                                _OracleRaw(), amodal, device="cpu")
    assert raw_out.shape == (60, 80, 3) and agg_out.shape == (60, 80, 3) and raw_out.dtype == np.uint8
    for suffix in ("raw_depth_rendered", "amodal_depth_rendered"):
        f = tmp_path / "out" / f"case_{suffix}.png"
        assert f.exists() and Image.open(f).size == (80, 60)
    x, gmask, obs = amodal.calls[0]
    assert x.shape == (1, 3, 518, 518) and float(x.min()) >= 0 and float(x.max()) <= 1
    assert set(torch.unique(gmask).tolist()) == {-1.0, 1.0}
    assert float(obs.min()) == -1.0 and float(obs.max()) == 1.0       # min-max normalised base depth mapped to [-1, 1]


def test_median_filter_blend_semantics():
    amodal = torch.full((8, 8), 0.9)
    base = torch.full((8, 8), 0.1)
    mask = np.zeros((8, 8))
    mask[2:6, 2:6] = 1
    out = infer.median_filter_blend(amodal, base, mask)
    assert abs(float(out[3, 3]) - 0.9) < 1e-6 and abs(float(out[0, 0]) - 0.1) < 1e-6
    assert 0.1 < float(out[2, 2]) < 0.9 and 0.1 < float(out[1, 1]) < 0.9     # 3x3 box blur on the mask border ring


def test_image_helpers_match_torch_references():
    from src.util import image_util as U
    a = np.arange(7 * 9, dtype=np.float32).reshape(7, 9)
    ref = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(torch.tensor(a)[None, None], (1, 1, 1, 1), mode="reflect"), 3, 1)[0, 0]
    assert np.allclose(U.box_blur(a, 3), ref.numpy(), atol=1e-5)
    col = U.colorize_depth_maps(np.linspace(0, 1, 12, dtype=np.float32).reshape(3, 4), 0, 1, cmap="Spectral_r")
    assert col.shape == (1, 3, 3, 4) and col.min() >= 0 and col.max() <= 1
    assert U.resize_nearest(np.arange(6).reshape(2, 3), 6, 4).shape == (4, 6)
