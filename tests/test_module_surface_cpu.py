"""CPU: the nn.Module surface is a drop-in for the reference's (constructor args, error behaviour, state_dict schema,
hub-mixin persistence).  No forward runs here -- the product path has no CPU fallback, and that is asserted too."""
import json
import os

import pytest
import torch

from _cases import GOLDEN_DIR
from src.models import get_model, model_name_class_dict
from src.models.amodalsynthdrive.dav2 import AmodalDAv2
from src.models.amodalsynthdrive.depth_anything_v2_raw.dpt import DepthAnythingV2 as RawDepthAnythingV2

SCHEMA = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_schema.json")))


def _shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


@pytest.mark.parametrize("key", sorted(k for k in SCHEMA if k.startswith("amodal/")))
def test_amodal_state_dict_matches_reference_schema(key):
    _, enc, gt = key.split("/")
    with torch.device("meta"):                       # shapes only: no parameter initialisation
        m = get_model("AmodalDAv2", guide_type=gt, loss_stategy="entire_target_object", encoder=enc, pretrained=False)
    assert _shapes(m) == SCHEMA[key]
    assert list(m.state_dict().keys()) == list(SCHEMA[key].keys()), "key order differs"
    assert "pixel_mean" not in m.state_dict() and m.pixel_mean.shape == (3, 1, 1)


@pytest.mark.parametrize("enc,feat,oc", [("vits", 64, [48, 96, 192, 384]), ("vitb", 128, [96, 192, 384, 768]), ("vitl", 256, [256, 512, 1024, 1024]), ("vitg", 384, [1536] * 4)])
def test_raw_state_dict_matches_reference_schema(enc, feat, oc):
    with torch.device("meta"):
        m = RawDepthAnythingV2(encoder=enc, features=feat, out_channels=oc)
    assert _shapes(m) == SCHEMA[f"raw/{enc}"]


def test_raw_state_dict_with_class_token_readout_matches_reference_schema():
    with torch.device("meta"):
        m = RawDepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384], use_clstoken=True)
    assert _shapes(m) == SCHEMA["raw/vits/clstoken"]


def test_raw_state_dict_with_batchnorm_fusion_blocks_matches_reference_schema():
    m = RawDepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384], use_bn=True)
    assert _shapes(m) == SCHEMA["raw/vits/bn"]
    assert list(m.state_dict().keys()) == list(SCHEMA["raw/vits/bn"].keys())
    rcu = m.depth_head.scratch.refinenet2.resConfUnit1
    with pytest.raises(NotImplementedError):      # batch statistics (training mode) are not built
        rcu.train().folded()
    # the folded conv equals conv + inference BatchNorm
    torch.manual_seed(0)
    rcu.eval()
    for bn in (rcu.bn1, rcu.bn2):
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.normal_(1, 0.1); bn.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 64, 9, 11)
    w1, b1, w2, b2 = rcu.folded()
    want = rcu.bn1(rcu.conv1(x))
    got = torch.nn.functional.conv2d(x, w1, b1, padding=1)
    assert float((want - got).abs().max()) < 1e-5


def test_registry_and_error_behaviour():
    assert "AmodalDAv2" in model_name_class_dict
    with pytest.raises(NotImplementedError):
        get_model("NoSuchModel")
    with pytest.raises(NotImplementedError):           # reference dinov2.py:124-125
        AmodalDAv2(guide_type="banana", encoder="vits")
    with pytest.raises(KeyError):                      # reference dav2.py:22,31-34: default encoder 'vitg' has no config
        AmodalDAv2(guide_type="mask")


def test_guidance_embed_starts_at_zero_and_guide_concat_order():
    m = AmodalDAv2(guide_type="image+mask+observation", encoder="vits", pretrained=False)
    proj = m.encoder.pretrained.patch_embed_guidance.proj
    assert float(proj.weight.abs().max()) == 0.0 and float(proj.bias.abs().max()) == 0.0 and proj.weight.shape[1] == 5
    rgb, mask, obs = torch.rand(1, 3, 4, 4), torch.rand(1, 1, 4, 4), torch.rand(1, 1, 4, 4)
    g = m.build_guide(rgb, mask, obs)
    assert torch.equal(g, torch.cat([rgb, mask, obs], 1))
    assert AmodalDAv2(guide_type="none", encoder="vits").build_guide(rgb, mask, obs) is None
    assert torch.equal(AmodalDAv2(guide_type="observation", encoder="vits").build_guide(rgb, mask, obs), obs)


def test_save_and_from_pretrained_roundtrip(tmp_path):
    m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder="vits", pretrained=False)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01)
    m.save_pretrained(tmp_path)
    assert (tmp_path / "model.safetensors").exists() and (tmp_path / "config.json").exists()
    cfg = json.load(open(tmp_path / "config.json"))
    assert cfg["guide_type"] == "mask+observation" and cfg["encoder"] == "vits"
    m2 = AmodalDAv2.from_pretrained(str(tmp_path), strict=True)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_product_path_refuses_cpu_tensors():
    import hip_ext
    m = get_model("AmodalDAv2", guide_type="mask", loss_stategy="x", encoder="vits", pretrained=False).eval()
    x = torch.rand(1, 3, 28, 28)
    with pytest.raises(hip_ext.HipExtError, match="HIP device"):
        m(x, guide_mask=torch.ones(1, 1, 28, 28))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "amodal-depth-anything_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(root, f)
    for f in ("infer.py",):
        src = open(os.path.join(os.path.dirname(GOLDEN_DIR), "..", f)).read()
        assert "import oracle" not in src and "from oracle" not in src


def test_encoder_precision_policy_table(monkeypatch):
    """Which models run the linear layers of their leading transformer blocks in split precision by default ("auto"): every unbounded head -- the
    raw model's ReLU, the 'ssi' heads (no activation) -- because nothing compresses the encoder's operand noise there: every block on ViT-S (round 5), 4 on ViT-B,
    8 on ViT-L / ViT-G (profiles/r04_e_raw_vitg_precision.txt, r04_q_unbounded_heads_encoder_precision.txt); the benchmarked sigmoid models none."""
    from src.models.amodalsynthdrive.depth_anything_v2.dpt import _encoder_split_policy
    monkeypatch.delenv("ADA_ENC_SPLIT", raising=False)
    for enc in ("vits", "vitb", "vitl"):
        assert _encoder_split_policy("auto", enc, "sigmoid") == 0
    for act in ("relu", "none"):
        assert [_encoder_split_policy("auto", enc, act) for enc in ("vits", "vitb", "vitl", "vitg")] == [12, 4, 8, 8]   # (round 5: every block of ViT-S)
    assert _encoder_split_policy(5, "vitl", "sigmoid") == 5
    monkeypatch.setenv("ADA_ENC_SPLIT", "3")
    assert _encoder_split_policy("auto", "vitl", "sigmoid") == 3 and _encoder_split_policy(0, "vitl", "none") == 0


def test_engine_stamp_sees_a_parameter_replaced_on_a_sub_module():
    """The packed weights are stamped by the parameter tensors' (address, version) pairs taken from a cached list; every call checks that each slot
    of the module tree still holds the object that was stamped, so a Parameter REPLACED on a sub-module (attribute assignment, load_state_dict with
    assign=True on a child, pruning utilities) is picked up by the very next forward -- round 5 noticed it only at its next periodic re-walk."""
    m = get_model("AmodalDAv2", guide_type="mask", loss_stategy="x", encoder="vits", pretrained=False).eval()
    enc = m.encoder
    first = enc._param_tensors()
    assert enc._engine_pnames == list(enc.state_dict().keys()) and all(a is b for a, b in zip(first, enc.state_dict(keep_vars=True).values()))
    assert enc._param_tensors() is first            # nothing changed: the cached list itself
    blk = enc.pretrained.blocks[3]
    old = blk.attn.proj.weight
    blk.attn.proj.weight = torch.nn.Parameter(old.detach().clone() * 2)
    second = enc._param_tensors()
    i = enc._engine_pnames.index("pretrained.blocks.3.attn.proj.weight")
    assert second is not first and second[i] is blk.attn.proj.weight and first[i] is old
    sd = {k: v.clone() + 1 for k, v in enc.depth_head.state_dict().items()}
    enc.depth_head.load_state_dict(sd, assign=True)       # replaces every Parameter object of the child
    third = enc._param_tensors()
    j = enc._engine_pnames.index("depth_head.scratch.output_conv2.2.bias")
    assert third[j] is enc.depth_head.scratch.output_conv2[2].bias and third[j] is not second[j]
