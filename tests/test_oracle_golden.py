"""CPU: the oracle restatement reproduces the REAL reference's outputs stored in tests/golden/ (generated
by oracle/make_golden.py from /root/reference).  This is what pins the oracle; it runs anywhere."""
import pytest
import torch

from _cases import case_inputs, golden_names, load_golden, oracle_forward, schema_state_dict

FAST = [n for n in golden_names() if n not in ("vitl_518", "vitb_518", "raw_vitg_224")]
SLOW = [n for n in golden_names() if n in ("vitl_518", "vitb_518", "raw_vitg_224")]


@pytest.mark.parametrize("name", FAST + SLOW)
def test_oracle_matches_reference_golden(name):
    gold, meta = load_golden(name)
    case = meta["case"]
    sd = schema_state_dict(case, meta)         # reference key/shape schema fixture + deterministic fill
    x, grgb, mask, obs = case_inputs(case)
    out = oracle_forward(sd, case, x, grgb, mask, obs)
    assert list(out.shape) == meta["out_shape"]
    st = case["stride"]
    sub = out[..., ::st, ::st]
    # same torch build => bit-identical in practice; allow a few ulps for other CPU kernels / thread counts
    assert torch.allclose(sub, gold, atol=2e-5, rtol=1e-5), float((sub - gold).abs().max())
    assert abs(float(out.mean()) - meta["out_mean"]) < 1e-5
