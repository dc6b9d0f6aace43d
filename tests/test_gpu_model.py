"""GPU parity of the whole forward pass: HIP path vs (a) the reference's golden outputs, (b) the fp32 oracle.
Bar (BASELINE.json north_star): relative L1 of the depth map <= 1e-3, mean|a-b| / mean|b|."""
import pytest
import torch

from _cases import build_product_model, case_inputs, golden_names, load_golden, oracle_forward, rel_l1, synth_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-3            # north-star bar on the predicted depth map -- ONE tolerance for every head (sigmoid, 'ssi' logits, raw ReLU)


def _tol(case):
    return TOL


def _run_product(model, case, x, grgb, mask, obs):
    model = model.cuda()
    with torch.no_grad():
        if case["kind"] == "raw":
            return model(x.cuda()).cpu()
        return model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda()).cpu()


# fixtures run at a batch where ada_igemm's heuristic must select the benchmarked kernel: the 256x256 tile (tile code 3; 103 when
# the phased main loop is selected) for the encoder's linear layers
BATCHED = {"vitl_518_b8": 8 * 1370, "vitb_518_b8": 8 * 1370}


@pytest.mark.parametrize("name", golden_names())
def test_hip_forward_matches_reference_golden(hip, name):
    if name == "raw_vitg_1022":
        pytest.skip("checked by test_raw_vitg_1022_batch8_config5 on the same model object (one 1.1 G-parameter synthetic fill instead of two)")
    gold, meta = load_golden(name)
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    x, grgb, mask, obs = case_inputs(case)
    log = []
    hip.set_tile_log(log)
    try:
        out = _run_product(model, case, x, grgb, mask, obs)
    finally:
        hip.set_tile_log(None)
    assert list(out.shape) == meta["out_shape"]
    assert torch.isfinite(out).all()
    st = case["stride"]
    sub = out[..., ::st, ::st]
    err = rel_l1(sub, gold)
    per_image = [rel_l1(sub[i], gold[i]) for i in range(gold.shape[0])]
    print(f"{name}: rel-L1 vs reference golden = {err:.3e}  per image max {max(per_image):.3e}")
    assert err <= _tol(case), f"{name}: rel-L1 {err:.3e} > {_tol(case)}"
    assert max(per_image) <= _tol(case), f"{name}: worst image rel-L1 {max(per_image):.3e} > {_tol(case)}"
    if name in BATCHED:
        rows = BATCHED[name]
        enc = [(m, n, k, code) for (m, n, k, code) in log if m == rows]
        assert len(enc) >= 4 * 12 and all(code == 3 for (_, _, _, code) in enc), \
            f"{name}: encoder GEMMs did not all run on the 256x256 tile: {sorted(set(enc))[:8]}"


@pytest.mark.parametrize("variant,attn", [(8, 5), (4, 0), (8, 3)])
def test_hip_forward_batch8_other_kernel_variants(hip, variant, attn):
    """ViT-B at batch 8 through the round-1 main loop / attention kernel combinations: every shipped variant meets the bar."""
    gold, meta = load_golden("vitb_518_b8")
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    x, grgb, mask, obs = case_inputs(case)
    hip.debug_set_variant(variant)
    hip.debug_set_attention_variant(attn)
    try:
        out = _run_product(model, case, x, grgb, mask, obs)
    finally:
        hip.debug_set_variant(4)
        hip.debug_set_attention_variant(5)
    st = case["stride"]
    err = rel_l1(out[..., ::st, ::st], gold)
    print(f"vitb_518_b8 variant {variant} attention {attn}: rel-L1 = {err:.3e}")
    assert err <= TOL


def test_hip_forward_matches_oracle_full_map_and_batch_invariance(hip):
    """Full-resolution comparison against the oracle run on the box, B=3, plus bs-invariance of the HIP path."""
    _, meta = load_golden("vits_518")
    case = dict(meta["case"], B=3)
    model = build_product_model(case)
    sd = synth_state_dict(model, meta)
    model.load_state_dict(sd, strict=True)
    x, grgb, mask, obs = case_inputs(case)
    ref = oracle_forward(sd, case, x, grgb, mask, obs)
    out = _run_product(model, case, x, grgb, mask, obs)
    err = rel_l1(out, ref)
    print(f"vits B=3 full map: rel-L1 vs oracle = {err:.3e}")
    assert err <= TOL
    with torch.no_grad():
        one = model(x[1:2].cuda(), guide_rgb=None, guide_mask=mask[1:2].cuda(), observation=obs[1:2].cuda()).cpu()
    assert torch.equal(one[0], out[1]), "HIP path is not batch invariant"


def test_fused_tail_on_and_off(hip, monkeypatch):
    """ada_dpt_tail_fwd (resize + output_conv2 in one kernel; off by default) against the reference golden and the two-launch tail."""
    from hip_ext import engine as E
    gold, meta = load_golden("vitb_518")
    case = meta["case"]
    x, grgb, mask, obs = case_inputs(case)
    st = case["stride"]
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(E, "FUSED_TAIL", fused)
        model = build_product_model(case)
        model.load_state_dict(synth_state_dict(model, meta), strict=True)
        outs[fused] = _run_product(model, case, x, grgb, mask, obs)
        err = rel_l1(outs[fused][..., ::st, ::st], gold)
        print(f"vitb_518 fused_tail={fused}: rel-L1 = {err:.3e}")
        assert err <= TOL
    assert rel_l1(outs[True], outs[False]) < 2e-4


def test_layernorm_folding_on_and_off(hip):
    """The block LayerNorms folded into qkv / fc1 (default) and as stand-alone launches: both meet the bar on ViT-B at batch 8, and the
    folded forward issues 2 * depth - 1 fewer LayerNorm launches."""
    from hip_ext import engine as E
    gold, meta = load_golden("vitb_518_b8")
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    x, grgb, mask, obs = case_inputs(case)
    counts = {}
    real_ln = E.k_layernorm
    st = case["stride"]
    try:
        for fold in (True, False):
            n = [0]

            def counting(*a, _n=n, **k):
                _n[0] += 1
                return real_ln(*a, **k)
            E.k_layernorm = counting
            model.encoder.fold_layernorm = fold
            out = _run_product(model, case, x, grgb, mask, obs)
            counts[fold] = n[0]
            err = rel_l1(out[..., ::st, ::st], gold)
            print(f"vitb_518_b8 fold_layernorm={fold}: rel-L1 = {err:.3e}, {n[0]} LayerNorm launches")
            assert err <= TOL
    finally:
        E.k_layernorm = real_ln
    assert counts[False] - counts[True] == 2 * 12 - 1


def test_raw_vitg_1022_batch8_config5(hip):
    """BASELINE config 5 at its full size: raw ViT-G, 8 x 1022 x 1022 (73 x 73 patches, N = 5330 tokens, bicubic pos-embed).  Size-
    independent properties: shape, finiteness, non-negativity (ReLU head), batch invariance against single-image runs of two
    of the images, and the B = 1 reference golden (tests/golden/raw_vitg_1022.npz) reproduced by the same model object."""
    gold, meta = load_golden("raw_vitg_1022")
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    model = model.cuda()
    x8, _, _, _ = case_inputs(dict(case, B=8, seed=11))
    with torch.no_grad():
        out = model(x8.cuda())
        assert list(out.shape) == [8, 1022, 1022]
        assert torch.isfinite(out).all() and float(out.min()) >= 0.0
        assert float(out.std()) > 0.1
        for b in (0, 5):
            one = model(x8[b:b + 1].cuda())
            err = rel_l1(one[0], out[b])
            assert err < 1e-6, f"image {b}: batch of 8 differs from the single-image run by {err:.2e}"
        x1, _, _, _ = case_inputs(case)
        o1 = model(x1.cuda()).cpu()
    st = case["stride"]
    err = rel_l1(o1[..., ::st, ::st], gold)
    print(f"raw_vitg_1022 (after the batch-8 run): rel-L1 vs reference golden = {err:.3e}")
    assert err <= TOL


def test_graph_replay_is_bit_identical_and_follows_inputs_and_weights(hip, monkeypatch):
    """Single-image calls replay a captured HIP graph (hip_ext/engine.py::_GraphedForward): same bits as plain launches, for every new input
    of the captured shape, and a parameter update re-packs the weights and re-captures."""
    from hip_ext import engine as E
    _, meta = load_golden("vits_518")
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    ins = [case_inputs(case, seed=s) for s in (0, 1)]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setattr(E, "GRAPH_MODE", mode)
        outs[mode] = [_run_product(model, case, *i) for i in (ins[0], ins[1], ins[0])]
    eng = model.encoder._engine()
    assert len(eng._graphs) == 1 and all(g is not False for g in eng._graphs.values()), "the forward was not captured"
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
    assert torch.equal(outs["1"][0], outs["1"][2]) and not torch.equal(outs["1"][0], outs["1"][1])
    with torch.no_grad():
        model.encoder.depth_head.scratch.output_conv2[2].bias.add_(0.25)
    shifted = _run_product(model, case, *ins[0])
    assert model.encoder._engine() is not eng and not torch.equal(shifted, outs["1"][0])
    monkeypatch.setattr(E, "GRAPH_MODE", "0")
    assert torch.equal(shifted, _run_product(model, case, *ins[0]))


def test_state_dict_reload_repacks(hip):
    """load_state_dict after a forward must invalidate the packed operand copies."""
    _, meta = load_golden("vits_g_mask")
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    x, grgb, mask, obs = case_inputs(case)
    a = _run_product(model, case, x, grgb, mask, obs)
    model.load_state_dict({k: v.cuda() for k, v in synth_state_dict(model, meta, seed=5).items()}, strict=True)
    b = _run_product(model, case, x, grgb, mask, obs)
    assert rel_l1(a, b) > 1e-2
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    c = _run_product(model, case, x, grgb, mask, obs)
    assert torch.equal(a, c)


def test_cpu_tensors_are_rejected(hip):
    _, meta = load_golden("vits_g_mask")
    case = meta["case"]
    model = build_product_model(case).cuda()
    x, grgb, mask, obs = case_inputs(case)
    with pytest.raises(hip.HipExtError):
        model(x, guide_rgb=None, guide_mask=mask, observation=obs)
    with pytest.raises(AssertionError, match="not a multiple of patch"):
        model(torch.zeros(1, 3, 100, 98).cuda(), guide_rgb=None, guide_mask=torch.zeros(1, 1, 100, 98).cuda(), observation=None)


def test_module_by_module_path_matches_oracle(hip):
    """The stand-alone L1 modules (PatchEmbed, Block, Attention, Mlp, DPTHead, FeatureFusionBlock ...) composed the way the
    reference composes them (get_intermediate_layers -> DPTHead.forward) agree with the oracle too."""
    _, meta = load_golden("vits_g_image_mask_observation")
    case = meta["case"]
    model = build_product_model(case)
    sd = synth_state_dict(model, meta)
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    x, grgb, mask, obs = case_inputs(case)
    ref = oracle_forward(sd, case, x, grgb, mask, obs)
    with torch.no_grad():
        guide = model.build_guide(grgb.cuda(), mask.cuda(), obs.cuda())
        out = model.encoder.forward_modular(x.cuda(), guide).cpu()
        fused = model(x.cuda(), guide_rgb=grgb.cuda(), guide_mask=mask.cuda(), observation=obs.cuda()).cpu()
    e1, e2 = rel_l1(out, ref), rel_l1(fused, ref)
    print(f"modular path rel-L1 = {e1:.3e}; fused engine rel-L1 = {e2:.3e}")
    assert e1 <= TOL and e2 <= TOL


def test_class_token_readout_modular_path(hip):
    """use_clstoken=True through the module-by-module path (DPTHead.forward with readout_projects) against the oracle; the fused engine
    path of the same model is covered by the raw_vits_clstoken reference golden."""
    gold, meta = load_golden("raw_vits_clstoken")
    case = meta["case"]
    model = build_product_model(case)
    sd = synth_state_dict(model, meta)
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    x, grgb, mask, obs = case_inputs(case)
    with torch.no_grad():
        out = model.forward_modular(x.cuda()).cpu()
    err = rel_l1(out, gold)
    print(f"raw_vits_clstoken modular path: rel-L1 vs reference golden = {err:.3e}")
    assert err <= TOL


def test_batchnorm_fusion_blocks_modular_path(hip):
    """use_bn=True through the module-by-module path (ResidualConvUnit.folded) against the reference golden; the fused engine path of the
    same model is covered by the raw_vits_bn golden in test_model_matches_reference_golden."""
    gold, meta = load_golden("raw_vits_bn")
    case = meta["case"]
    model = build_product_model(case)
    sd = synth_state_dict(model, meta)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    x, grgb, mask, obs = case_inputs(case)
    with torch.no_grad():
        out = model.forward_modular(x.cuda()).cpu()
    err = rel_l1(out, gold)
    print(f"raw_vits_bn modular path: rel-L1 vs reference golden = {err:.3e}")
    assert err <= TOL


def test_raw_swiglu_module_path(hip):
    """ViT-G style SwiGLU block through the module-level functional path vs torch."""
    import torch.nn.functional as F
    from src.models.amodalsynthdrive.depth_anything_v2.dinov2_layers import SwiGLUFFNFused
    torch.manual_seed(0)
    ffn = SwiGLUFFNFused(128, 384).cuda()   # fused flavour: hidden = (int(384*2/3)+7)//8*8 = 256
    x = torch.randn(2, 50, 128).cuda()
    with torch.no_grad():
        got = ffn(x).cpu()
        x12 = F.linear(x.cpu(), ffn.w12.weight.cpu(), ffn.w12.bias.cpu())
        a, b = x12.chunk(2, -1)
        ref = F.linear(F.silu(a) * b, ffn.w3.weight.cpu(), ffn.w3.bias.cpu())
    assert rel_l1(got, ref) < 3e-3


def test_infer_cli_end_to_end_on_gpu(hip, tmp_path):
    """python infer.py ... on the GPU with (synthetic-weight) ViT-S models: both reference-named PNGs are written, and their
    pixels agree with the same CLI pipeline driven by the fp32 CPU oracle (same weights, infer.infer_single_image with
    device='cpu').  The renders go through a 256-entry colour LUT (Spectral_r) whose neighbouring entries differ by up to 4 grey
    levels, so a depth difference of 3e-4 that moves a pixel to the next LUT entry shows as a 2-3 level jump: the bar is therefore
    every pixel within ONE LUT step (<= 4 levels), >= 98.5 % of the pixels within 1 level and a mean difference below 0.1 level."""
    import os
    import subprocess
    import sys

    import numpy as np
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import infer
    from oracle import dav2_oracle as O
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:90, 0:120]
    img = np.stack([(np.sin(xx / 17.0) * 0.5 + 0.5) * 255, (np.cos(yy / 11.0) * 0.5 + 0.5) * 255, (xx + yy) / 210.0 * 255], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "img.png")
    m = np.zeros((64, 64), dtype=np.uint8)
    m[20:50, 10:40] = 255
    Image.fromarray(m).save(tmp_path / "img_mask.png")
    r = subprocess.run([sys.executable, os.path.join(root, "infer.py"), "--input_image_path", str(tmp_path / "img.png"),
                        "--input_mask_path", str(tmp_path / "img_mask.png"), "--output_folder", str(tmp_path / "out"),
                        "--raw_encoder", "vits", "--amodal_encoder", "vits"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]

    # the same pipeline with the CPU oracle standing in for both networks (synthetic weights = what the CLI falls back to)
    raw_case = dict(kind="raw", encoder="vits", features=64, out_channels=[48, 96, 192, 384])
    am_case = dict(kind="amodal", encoder="vits", guide_type="mask+observation", loss="entire_target_object")
    raw_sd = synth_state_dict(build_product_model(raw_case))
    am_sd = synth_state_dict(build_product_model(am_case))

    def oracle_raw(x):
        return O.raw_forward(raw_sd, "vits", x)

    def oracle_amodal(x, guide_rgb=None, guide_mask=None, observation=None):
        return O.amodal_forward(am_sd, "vits", "mask+observation", "entire_target_object", x, None, guide_mask, observation)

    infer.infer_single_image(str(tmp_path / "img.png"), str(tmp_path / "img_mask.png"), str(tmp_path / "ref"), oracle_raw, oracle_amodal, device="cpu")
    for suffix in ("raw_depth_rendered", "amodal_depth_rendered"):
        got = np.asarray(Image.open(tmp_path / "out" / f"img_{suffix}.png")).astype(np.int32)
        want = np.asarray(Image.open(tmp_path / "ref" / f"img_{suffix}.png")).astype(np.int32)
        assert got.shape == want.shape == (90, 120, 3)
        diff = np.abs(got - want)
        close = (diff.max(-1) <= 1).mean()
        print(f"{suffix}: {100 * close:.2f} % of the pixels within 1 grey level of the oracle-driven render (max diff {diff.max()}, mean {diff.mean():.3f})")
        assert close >= 0.985, f"{suffix}: only {100 * close:.2f} % of the pixels within 1 grey level"
        assert diff.max() <= 4 and diff.mean() < 0.1, f"{suffix}: max diff {diff.max()}, mean {diff.mean():.3f}"


def test_on_device_pipeline_matches_host_composition(hip):
    """hip_ext.pipeline (both nets + normalise + blend on the GPU) == the host-side composition of reference infer.py."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import infer
    from hip_ext.pipeline import amodal_depth_pipeline
    raw_case = dict(kind="raw", encoder="vits", features=64, out_channels=[48, 96, 192, 384])
    am_case = dict(kind="amodal", encoder="vits", guide_type="mask+observation", loss="entire_target_object")
    raw = build_product_model(raw_case)
    raw.load_state_dict(synth_state_dict(raw), strict=True)
    am = build_product_model(am_case)
    am.load_state_dict(synth_state_dict(am), strict=True)
    raw, am = raw.cuda(), am.cuda()
    g = torch.Generator().manual_seed(7)
    rgb = torch.rand(2, 3, 126, 154, generator=g).cuda()
    mask = torch.zeros(2, 1, 126, 154)
    mask[0, :, 20:80, 30:100] = 1
    mask[1, :, :40, :] = 1
    base_norm, pred, out = amodal_depth_pipeline(raw, am, rgb, mask.cuda())
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1).cuda()
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1).cuda()
    with torch.no_grad():
        base = raw((rgb - mean) / std).cpu()
    for b in range(2):
        bn = (base[b] - base[b].min()) / (base[b].max() - base[b].min())
        assert torch.allclose(base_norm[b].cpu(), bn, atol=1e-6)
        with torch.no_grad():
            p = am(rgb[b:b + 1], guide_rgb=None, guide_mask=mask[b:b + 1].cuda() * 2 - 1, observation=(bn[None, None].cuda() * 2 - 1)).cpu()[0, 0]
        assert torch.allclose(pred[b].cpu(), p, atol=1e-5)
        want = infer.median_filter_blend(p, bn.clone(), mask[b, 0].numpy())
        assert torch.allclose(out[b].cpu(), want, atol=1e-5)


def test_large_batches_are_chunked_and_repeatable(hip, monkeypatch):
    """Batches whose pixel count exceeds the kernels' row-index limit are split internally; results equal the unsplit run
    bit for bit, and repeated calls are deterministic (no atomics / split-K anywhere on the path)."""
    from hip_ext import engine as E
    _, meta = load_golden("vits_g_observation")
    case = dict(meta["case"], B=3)
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    x, grgb, mask, obs = case_inputs(case)
    full = _run_product(model, case, x, grgb, mask, obs)
    again = _run_product(model, case, x, grgb, mask, obs)
    assert torch.equal(full, again)
    monkeypatch.setattr(E, "MAX_ROWS", case["H"] * case["W"] * 2)     # forces chunks of 2 + 1 images
    split = _run_product(model, case, x, grgb, mask, obs)
    assert torch.equal(full, split)


def test_runs_on_a_side_stream(hip):
    """Kernels are launched on torch's *current* stream: the forward works (and is ordered) on a non-default stream."""
    _, meta = load_golden("vits_g_mask")
    case = meta["case"]
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model, meta), strict=True)
    x, grgb, mask, obs = case_inputs(case)
    ref = _run_product(model, case, x, grgb, mask, obs)
    s = torch.cuda.Stream()
    xc, mc = x.cuda(), mask.cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(s), torch.no_grad():
        out = model(xc, guide_rgb=None, guide_mask=mc, observation=None)
    s.synchronize()
    assert torch.equal(out.cpu(), ref)
