"""CPU tests of the host side of the fp8 correction terms (hip_ext/engine.py: f8_weight_split, PackedWeights.f8 / tap_f8, DepthEngine._kdup) and of the
policy that decides where they are used (DA2/dpt.py::_f8_policy).  The kernels themselves: tests/test_gpu_f8.py."""
import pytest
import torch

from hip_ext import engine as E
from src.models.amodalsynthdrive.depth_anything_v2 import dpt as D
from src.models.amodalsynthdrive.depth_anything_v2_raw.dpt import DepthAnythingV2 as Raw

OP = torch.float16


def _decode(packed, word, K, taps):
    n = packed.shape[0]
    b = packed.contiguous().view(torch.uint8).reshape(n, taps, 4 * K)
    hi = b[..., :2 * K].contiguous().view(OP).double()
    hi8 = b[..., 2 * K:3 * K].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** (((word >> 8) & 255) - 127)
    lo8 = b[..., 3 * K:].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** (((word >> 24) & 255) - 127)
    return hi, hi8, lo8


@pytest.mark.parametrize("taps,K,scale", [(1, 128, 0.05), (9, 256, 3e-3), (1, 384, 40.0)])
def test_weight_split_round_trip(taps, K, scale):
    g = torch.Generator().manual_seed(5)
    w = torch.randn(24, taps * K, generator=g) * scale
    w[0, 0] = 0.0
    packed, word = E.f8_weight_split(w, OP, taps=taps)
    assert packed.dtype == OP and packed.shape == (24, taps * 2 * K)
    assert (word & 255) == 117 and ((word >> 16) & 255) == 127       # the activation's two scale bytes: 2^-10 for lo8, 1 for hi8
    hi, hi8, lo8 = _decode(packed, word, K, taps)
    wd = w.double().reshape(24, taps, K)
    assert torch.equal(hi, wd.float().to(OP).double())
    top = float(wd.abs().max())
    assert float((hi8 - hi).abs().max()) <= top * 2.0 ** -4 * 1.01                       # e4m3: three mantissa bits, values under max * 2^-15 flush
    assert float((hi + lo8 - wd).abs().max()) <= top * 2.0 ** -11 * 2.0 ** -4 * 1.01   # the residual after the fp8 copy of the fp16 rounding error
    with pytest.raises(AssertionError):
        E.f8_weight_split(w[:, :taps * K - 64].contiguous(), OP, taps=1)               # 128-element granularity


def test_policy_and_packed_forms(monkeypatch):
    monkeypatch.setattr(E, "OC1_COMMUTE", False)     # (the commuted output_conv1 and the sub-pixel merge compose their weights on the device)
    monkeypatch.setattr(E, "SUBPIXEL", False)
    assert D._f8_policy("relu") == "both" and D._f8_policy("sigmoid") == "none" and D._f8_policy("none") == "none" and D._LADDER_F8 == "head"
    sd = Raw(encoder="vitb", features=128, out_channels=[96, 192, 384, 768]).state_dict()
    raw = E.PackedWeights(sd, "vitb", guided=False, amodal_head=False, split_head=True, enc_split_blocks=4, f8="both")
    assert raw.enc_f8 and raw.blocks[0]["qkv_f8"] and not raw.blocks[4]["qkv_f8"] and raw.blocks[0]["qkv_w"].shape[1] == 2 * 768
    # a group goes to the fp8 pipe iff its operand width is a multiple of 128 (192 channels: the three fp16 terms stay)
    assert {"proj", "rcu0", "out3", "rn0", "rs3", "oc1"} <= raw.f8_groups and not ({"rn1", "rs1", "oc2"} & raw.f8_groups)
    assert getattr(raw.rn_w[0], "f8_scales", 0) and not getattr(raw.rn_w[1], "f8_scales", 0)
    none = E.PackedWeights(sd, "vitb", guided=False, amodal_head=False, split_head=True, enc_split_blocks=4, f8="none")
    assert not none.enc_f8 and not none.f8_groups and none.blocks[0]["qkv_w"].shape[1] == 3 * 768 and not none.tap_f8
    # first rung of the sigmoid ladder: its own products on the fp16 pipe, the taps already in the second rung's form
    first = E.PackedWeights(sd, "vitb", guided=False, amodal_head=False, split_head=("out1", "out2", "out3"), tap_split=True, f8="none", tap_f8=True)
    assert not first.f8_groups and first.tap_f8
    second = E.PackedWeights(sd, "vitb", guided=False, amodal_head=False, split_head=True, head_only=True, f8="head")
    assert "proj" in second.f8_groups and second.tap_f8 and not second.blocks
    with pytest.raises(E.HipExtError):
        E.PackedWeights(sd, "vitb", guided=False, amodal_head=False, f8="all")


def test_kdup_recognises_the_fp8_form():
    w = torch.randn(16, 9 * 256)
    packed, word = E.f8_weight_split(w, OP, taps=9)
    packed.f8_scales = word
    kd = E.DepthEngine._kdup(512, packed, taps=9)
    assert kd == dict(K=18 * 256, lda=512, f8_from=256, f8_mid=384, f8_scales=word)
    lin, wl = E.f8_weight_split(torch.randn(8, 384), OP)
    lin.f8_scales = wl
    assert E.DepthEngine._kdup(768, lin, a_seg=-384) == dict(K=768, lda=768, f8_from=384, f8_mid=576, f8_scales=wl)
    with pytest.raises(E.HipExtError):
        E.DepthEngine._kdup(768, lin, a_seg=384)            # fp8-form weights against [hi | lo] taps
    triple = torch.zeros(8, 3 * 384, dtype=OP)
    with pytest.raises(E.HipExtError):
        E.DepthEngine._kdup(768, triple, a_seg=-384)        # three-term fp16 weights against [hi | lo8 | hi8] taps
    single = torch.zeros(8, 384, dtype=OP)
    assert E.DepthEngine._kdup(768, single, a_seg=-384) == dict(K=384, lda=768, a_dup_seg=0)     # the hi half only
