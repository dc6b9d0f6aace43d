"""TEST INFRASTRUCTURE / numerics study (not imported by the product): which contraction injects the fp16-operand noise
that the unbounded heads ('ssi' losses, raw ReLU model) show?  Every contraction of the oracle forward is numbered in call
order; the forward is then re-run with operand rounding enabled for ONE contraction at a time (only-one analysis) and the
relative L1 of the output against the fp32 forward is reported, largest first.

    python oracle/study_operand_noise.py vits_ssi_image_mask [fp16|bf16]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "amodal-depth-anything_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import dav2_oracle as O  # noqa: E402
from _cases import case_inputs, load_golden, schema_state_dict  # noqa: E402


class Selective(O._Numerics):
    def __init__(self, dt, only=None, skip=None):
        super().__init__(dt)
        self.only, self.skip, self.n, self.names = only, skip, 0, []

    def _on(self, kind, shape):
        i = self.n
        self.n += 1
        self.names.append(f"{kind}{tuple(shape)}")
        if self.only is not None:
            return i in self.only
        if self.skip is not None:
            return i not in self.skip
        return True

    def _q(self, on, t):
        return t.to(self.dt).to(torch.float32) if on else t

    def linear(self, x, w, b=None):
        on = self._on("linear", w.shape)
        return torch.nn.functional.linear(self._q(on, x), self._q(on, w), b)

    def conv(self, x, w, b=None, stride=1, padding=0):
        on = self._on("conv", w.shape)
        return torch.nn.functional.conv2d(self._q(on, x), self._q(on, w), b, stride=stride, padding=padding)

    def convT(self, x, w, b, stride):
        on = self._on("convT", w.shape)
        return torch.nn.functional.conv_transpose2d(self._q(on, x), self._q(on, w), b, stride=stride)

    def matmul(self, a, b):
        on = self._on("matmul", a.shape[-2:] + b.shape[-1:])
        return self._q(on, a) @ self._q(on, b)


def forward(case, sd, inputs, nm):
    x, grgb, mask, obs = inputs
    real = O._Numerics
    O._Numerics = lambda dt=None: nm      # the forwards construct their numerics object: hand them ours
    try:
        if case["kind"] == "raw":
            return O.raw_forward(sd, case["encoder"], x)
        return O.amodal_forward(sd, case["encoder"], case["guide_type"], case["loss"], x, grgb, mask, obs)
    finally:
        O._Numerics = real


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vits_ssi_image_mask"
    dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float16
    _, meta = load_golden(name)
    case = meta["case"]
    sd = schema_state_dict(case, meta)
    inputs = case_inputs(case)
    with torch.no_grad():
        ref = forward(case, sd, inputs, Selective(torch.float32))
        allq = Selective(dt)
        full = forward(case, sd, inputs, allq)
        print(f"{name}: all {allq.n} contractions rounded to {dt}: rel-L1 = {O.rel_l1(full, ref):.3e}")
        rows = []
        for i in range(allq.n):
            out = forward(case, sd, inputs, Selective(dt, only={i}))
            rows.append((O.rel_l1(out, ref), i, allq.names[i]))
        rows.sort(reverse=True)
        tot = sum(r[0] ** 2 for r in rows) ** 0.5
        print(f"root-sum-square of the single contributions: {tot:.3e}")
        for e, i, n in rows[:25]:
            print(f"  #{i:3d} {n:40s} {e:.3e}")


if __name__ == "__main__":
    main()
