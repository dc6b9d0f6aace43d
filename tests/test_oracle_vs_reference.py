"""Build container only: the oracle agrees with the REAL reference (imported from /root/reference) tensor for tensor.
Skipped wherever the reference tree is absent (e.g. the GPU box); tests/test_oracle_golden.py covers that case."""
import pytest
import torch

from oracle import _refshim
from oracle import dav2_oracle as O
from src.util.synth_weights import fill_state_dict_, make_inputs

pytestmark = pytest.mark.skipif(not _refshim.reference_available(), reason="reference tree not present")


@pytest.mark.parametrize("guide_type,loss", [("mask+observation", "entire_target_object"), ("image+mask+observation", "x_ssi"), ("none", "y")])
def test_amodal_oracle_equals_reference(guide_type, loss):
    Amodal, _ = _refshim.load_reference()
    m = Amodal(guide_type=guide_type, loss_stategy=loss, encoder="vits", pretrained=False).eval()
    sd = m.state_dict()
    fill_state_dict_(sd, 3)
    m.load_state_dict(sd, strict=True)
    x, grgb, mask, obs = make_inputs(2, 98, 126, 4)
    with torch.no_grad():
        ref = m(x, guide_rgb=grgb, guide_mask=mask, observation=obs)
    mine = O.amodal_forward(sd, "vits", guide_type, loss, x, grgb, mask, obs)
    assert torch.allclose(mine, ref, atol=1e-6, rtol=1e-6), float((mine - ref).abs().max())


def test_raw_oracle_equals_reference():
    _, Raw = _refshim.load_reference()
    m = Raw(encoder="vits", features=64, out_channels=[48, 96, 192, 384]).eval()
    sd = m.state_dict()
    fill_state_dict_(sd, 5)
    m.load_state_dict(sd, strict=True)
    x = torch.randn(1, 3, 70, 112)
    with torch.no_grad():
        ref = m(x)
    mine = O.raw_forward(sd, "vits", x)
    assert mine.shape == ref.shape and torch.allclose(mine, ref, atol=1e-6, rtol=1e-6)
