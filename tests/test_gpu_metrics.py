"""GPU: device-side depth metrics / least-squares alignment (ada_depth_eval_fwd behind src/util/metric.py and alignment.py of the
product package) against the reference-pinned fp64 oracle and the reference's own outputs (tests/golden/metrics/cases.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics", "cases.npz")
# fp32 per-pixel terms (log / divide on the device) summed in fp64: relative agreement with the fp64 oracle
RTOL = 2e-6


@pytest.fixture(scope="module")
def mods(hip):
    from src.util import alignment, metric
    return metric, alignment


def _rand(B, H, W, seed):
    rng = np.random.default_rng(seed)
    gt = rng.uniform(0.5, 10.0, size=(B, H, W)).astype(np.float32)
    pred = np.maximum(gt * rng.uniform(0.7, 1.4, size=gt.shape) + rng.normal(0, 0.05, size=gt.shape), 0.05).astype(np.float32)
    mask = rng.uniform(size=gt.shape) > 0.3
    return pred, gt, mask


@pytest.mark.parametrize("case", ["b1_small", "b3_mid"])
def test_metrics_match_reference_goldens(mods, case):
    metric, _ = mods
    g = np.load(GOLD)
    pred, gt, mask = (torch.from_numpy(g[f"{case}.{k}"]).cuda() for k in ("pred", "gt", "mask"))
    got = metric.depth_metrics(pred, gt, mask)
    for name in MO.ALL:
        assert float(got[name]) == pytest.approx(float(g[f"{case}.{name}"]), rel=2e-5, abs=1e-7), name
        assert float(getattr(metric, name)(pred, gt, mask)) == pytest.approx(float(got[name]), rel=1e-12), name


@pytest.mark.parametrize("shape", [(1, 518, 518), (4, 518, 518), (2, 266, 518), (3, 1, 7)])
@pytest.mark.parametrize("use_mask", [True, False])
def test_metrics_match_oracle(mods, shape, use_mask):
    metric, _ = mods
    pred, gt, mask = _rand(*shape, seed=sum(shape))
    m = mask if use_mask else None
    want = MO.depth_metrics(pred, gt, m)
    got = metric.depth_metrics(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), None if m is None else torch.from_numpy(m).cuda())
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(v, rel=RTOL, abs=1e-9), k


def test_metrics_accept_channel_dim_and_uint8_mask(mods):
    metric, _ = mods
    pred, gt, mask = _rand(2, 37, 41, seed=5)
    want = MO.depth_metrics(pred, gt, mask)
    got = metric.depth_metrics(torch.from_numpy(pred).cuda()[:, None], torch.from_numpy(gt).cuda()[:, None],
                               torch.from_numpy(mask.astype(np.uint8)).cuda()[:, None])
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(v, rel=RTOL, abs=1e-9), k


def test_metric_tracker(mods):
    metric, _ = mods
    t = metric.MetricTracker("a", "b")
    t.update("a", 2.0); t.update("a", 4.0, n=3); t.update("b", 1.0)
    assert t.avg("a") == pytest.approx(3.5) and t.result() == {"a": pytest.approx(3.5), "b": 1.0}
    t.reset()
    assert t.result() == {"a": 0.0, "b": 0.0}


@pytest.mark.parametrize("case", ["b1_small", "b3_mid"])
def test_alignment_matches_reference_goldens(mods, case):
    _, alignment = mods
    g = np.load(GOLD)
    gt, rel, mask = g[f"{case}.gt"], g[f"{case}.rel"], g[f"{case}.mask"]
    ss = alignment.scale_shift_least_square(torch.from_numpy(gt).cuda(), torch.from_numpy(rel).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    np.testing.assert_allclose(ss[:, 0], g[f"{case}.scale"], rtol=1e-5)
    np.testing.assert_allclose(ss[:, 1], g[f"{case}.shift"], rtol=1e-5, atol=1e-6)
    # reference-shaped call: numpy in, numpy out, one image
    aligned, s, t = alignment.align_depth_least_square(gt[0], rel[0], mask[0])
    assert isinstance(aligned, np.ndarray) and aligned.shape == rel[0].shape
    assert float(s[0]) == pytest.approx(float(g[f"{case}.scale"][0]), rel=1e-5)
    if f"{case}.aligned0" in g.files:
        np.testing.assert_allclose(aligned, g[f"{case}.aligned0"], rtol=1e-5, atol=1e-5)


def test_alignment_recovers_affine_map_at_full_size(mods):
    _, alignment = mods
    rng = np.random.default_rng(3)
    gt = rng.uniform(1, 20, size=(518, 518)).astype(np.float32)
    rel = ((gt - 2.5) / 7.0).astype(np.float32)
    mask = rng.uniform(size=gt.shape) > 0.5
    aligned, s, t = alignment.align_depth_least_square(torch.from_numpy(gt).cuda()[None], torch.from_numpy(rel).cuda()[None], torch.from_numpy(mask).cuda()[None])
    assert aligned.shape == (1, 518, 518) and aligned.is_cuda
    assert float(s) == pytest.approx(7.0, rel=1e-5) and float(t) == pytest.approx(2.5, rel=1e-4)
    a2, s2, t2 = alignment.align_depth_least_square(gt, rel, mask, max_resolution=128)
    assert float(s2[0]) == pytest.approx(7.0, rel=1e-4)


def test_eval_scale_shift_and_clip_inside_the_kernel(hip):
    import hip_ext as H
    pred, gt, mask = _rand(2, 64, 64, seed=9)
    ss = np.array([[1.2, 0.1], [0.9, -0.05]], dtype=np.float32)
    p2 = np.clip(pred * ss[:, 0, None, None] + ss[:, 1, None, None], 0.6, 9.0).astype(np.float32)
    want = MO.depth_metrics(p2, gt, mask)
    s = H.depth_eval(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda(),
                     scale_shift=torch.from_numpy(ss).cuda(), clip=(0.6, 9.0))
    from src.util.metric import _from_sums
    got = _from_sums(s)
    for k, v in want.items():
        assert float(got[k]) == pytest.approx(v, rel=5e-6, abs=1e-9), k


def test_depth2disparity(mods):
    _, alignment = mods
    d = torch.tensor([[0.0, 2.0], [-1.0, 4.0]]).cuda()
    inv, m = alignment.depth2disparity(d, return_mask=True)
    assert inv.cpu().tolist() == [[0.0, 0.5], [0.0, 0.25]] and m.cpu().tolist() == [[False, True], [False, True]]
    assert alignment.disparity2depth(np.array([0.0, 0.5])).tolist() == [0.0, 2.0]


def test_dataset_runner_on_device_matches_oracle_evaluation(hip, tmp_path):
    """Batched runner end to end on the HIP path (ViT-S, synthetic weights): the device evaluation (alignment + metrics from
    ada_depth_eval_fwd) equals the fp64 oracle evaluation of the same predictions."""
    from PIL import Image
    from _cases import build_product_model, synth_state_dict
    from src.scripts import amodal_dav2_inference as R
    rng = np.random.default_rng(0)
    ids = ["11", "12", "13"]
    d = {k: tmp_path / k for k in ("occ", "whole", "obs", "gt")}
    for v in d.values():
        v.mkdir()
    for sid in ids:
        Image.fromarray((rng.random((64, 64, 3)) * 255).astype(np.uint8)).save(d["occ"] / f"{sid}_occlusion.png")
        m = np.zeros((64, 64), dtype=np.uint8); m[10:50, 8:40] = 255
        Image.fromarray(m).save(d["whole"] / f"{sid}_whole_mask.png")
        Image.fromarray((rng.uniform(0.2, 0.9, size=(32, 32)) * 65535).astype(np.uint16)).save(d["obs"] / f"{sid}_depth.png")
        Image.fromarray((rng.uniform(0.2, 0.9, size=(128, 128)) * 65535).astype(np.uint16)).save(d["gt"] / f"{sid}_depth.png")
    case = dict(kind="amodal", encoder="vits", guide_type="mask+observation", loss="entire_target_object")
    model = build_product_model(case)
    model.load_state_dict(synth_state_dict(model), strict=True)
    model = model.eval().cuda()

    def oracle_eval(pred, gt, mask):
        p, g, m = pred.double().cpu().numpy(), gt.double().cpu().numpy(), mask.cpu().numpy()
        al = np.stack([MO.align_depth_least_square(g[b], p[b], m[b])[0] for b in range(p.shape[0])])
        return MO.depth_metrics(np.clip(al, 1e-3, 1.0), g, m)

    args = (model, ids, str(d["occ"]), str(d["whole"]), str(d["obs"]))
    got = R.run(*args, str(tmp_path / "o1"), str(d["gt"]), batch_size=2)
    want = R.run(*args, str(tmp_path / "o2"), str(d["gt"]), batch_size=2, evaluate=oracle_eval)
    assert set(got) == set(MO.ALL)
    for k in MO.ALL:
        assert got[k] == pytest.approx(want[k], rel=2e-5, abs=1e-7), k
    out = np.asarray(Image.open(tmp_path / "o1" / "amodal_depth" / "12_depth.png"))
    assert out.shape == (518, 518) and out.dtype == np.uint16 and out.std() > 0
