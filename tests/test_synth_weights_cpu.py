"""CPU: the synthetic fill is a fixed function of (key, shape, seed, tail) -- every reference fixture in tests/golden was generated on it, so its values
are pinned bit for bit (digests of whole state dicts, recorded from the fill that produced the fixtures before it was rewritten to draw in place)."""
import hashlib
import json
import os

import pytest
import torch

from src.util.synth_weights import fill_state_dict_

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIGESTS = {
    "amodal/vits/mask+observation|0|normal": "9a8212d47c54d053cad4d2cb7408e1c9d859460242203ddc73705c1defdf7f76",
    "amodal/vits/mask+observation|3|heavy": "b57cfa029ec75d8c089b7ee559542e6a97a9c43b74e7250bccd06c03006f93b4",
    "amodal/vitb/mask+observation|0|normal": "19a8fbd793ca54def64fd65cfb71e3edb0a7d6aa96bb736b32eb09f065fa85e1",
    "amodal/vitb/mask+observation|3|heavy": "f2916e689453eebe79e7b3510730cd91bfe56f6f986764a52b098b840b7dec4a",
    "raw/vits/bn|0|normal": "3c5435d0f2f9d25134440022b2eed27cbbd622b5886f2c3ce0a5a039ef59be09",
    "raw/vits/bn|3|heavy": "489f7a2f41fedb9792e9935e685fe34373769fdff84ccd0bee5a09534de4b8c1",
    "raw/vits/clstoken|0|normal": "10b8d1de9e7225df47623ed02f9f714bc9537ce281a7bc276fac36382830d718",
    "raw/vits/clstoken|3|heavy": "c7fabeecf3a83e7d2860ab8041edf773273eb87a7274d00bd39492837093099c"
}


@pytest.mark.parametrize("name", sorted(DIGESTS))
def test_fill_is_bit_identical_to_the_fill_the_fixtures_were_generated_with(name):
    key, seed, tail = name.split("|")
    schema = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_schema.json")))[key]
    sd = {k: torch.zeros(shape, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, shape in schema.items()}
    fill_state_dict_(sd, int(seed), tail=tail)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].numpy().tobytes())
    assert h.hexdigest() == DIGESTS[name]


def test_fill_into_non_contiguous_destinations_matches_the_in_place_path():
    """The in-place fast path (contiguous fp32 destination) and the staging path (anything else) produce the same values."""
    a = {"blocks.0.attn.qkv.weight": torch.zeros(96, 32), "blocks.0.ls1.gamma": torch.zeros(32), "norm.weight": torch.zeros(32)}
    b = {k: torch.zeros(v.shape[::-1]).t() if v.ndim == 2 else torch.zeros(2 * v.numel())[::2] for k, v in a.items()}
    fill_state_dict_(a, 7, tail="heavy")
    fill_state_dict_(b, 7, tail="heavy")
    for k in a:
        assert not b[k].is_contiguous() and torch.equal(a[k], b[k])
