#!/usr/bin/env python
"""Oracle study (CPU, test infrastructure -- never imported by the product): what bounds the EVERYTHING-SPLIT engine (the ladder's third rung) on raw ViT-G?
With every contraction's operands carried as hi + lo, what is still rounded to fp16 is (A) the attention core's operands -- q, k, v as the qkv linear wrote
them, and P -- and (B) the activations that exist in the operand type only: the attention output feeding proj, the SwiGLU hidden feeding w3.  Each is emulated
alone on the fp32 oracle (round-to-nearest fp16 of exactly those tensors, everything else fp32) and the output compared with the fp32 run.
    python oracle/study_rung3_floor.py [fixture ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from _cases import build_product_model, case_inputs, load_golden, synth_state_dict  # noqa: E402
from oracle import dav2_oracle as O  # noqa: E402


class Sel(O._Numerics):
    """fp32 everywhere except the tensors named in `what`: "attn" (q, k, v, P), "act" (inputs of proj / w3 / fc2), "w" (every linear weight), "ln" (inputs of
    qkv / w12 / fc1), "head" (every conv of the DPT head, operands and weights)."""

    def __init__(self, what, D, hidden):
        super().__init__(torch.float32)
        self.what, self.D, self.hidden = set(what), D, hidden

    DT = {"fp16": torch.float16, "bf16": torch.bfloat16}[os.environ.get("DT", "fp16")]      # DT=bf16: the same question for the operand type the north star names

    @classmethod
    def r(cls, t):
        return t.to(cls.DT).to(torch.float32)

    def linear(self, x, w, b=None):
        n, k = w.shape
        from_act = (n == self.D and k in (self.D, self.hidden))          # proj, w3 / fc2
        if "w" in self.what:
            w = self.r(w)
        if ("act" in self.what and from_act) or ("ln" in self.what and not from_act):
            x = self.r(x)
        return F.linear(x, w, b)

    def matmul(self, a, b):
        return (self.r(a) @ self.r(b)) if "attn" in self.what else a @ b

    def conv(self, x, w, b=None, stride=1, padding=0):
        if "head" in self.what and x.shape[1] != 3:
            x, w = self.r(x), self.r(w)
        return F.conv2d(x, w, b, stride=stride, padding=padding)

    def convT(self, x, w, b, stride):
        if "head" in self.what:
            x, w = self.r(x), self.r(w)
        return F.conv_transpose2d(x, w, b, stride=stride)


def main():
    names = sys.argv[1:] or ["raw_vitg_126x154_unc", "raw_vitg_224"]
    configs = [("attention core only (q, k, v, P)", ["attn"]), ("proj / w3 input activations only", ["act"]), ("attention core + those activations (= what the third rung leaves)", ["attn", "act"]),
               ("linear weights only", ["w"]), ("LayerNorm outputs feeding qkv / w12 only", ["ln"]), ("head convs only", ["head"]), ("everything (single precision everywhere)", ["attn", "act", "w", "ln", "head"])]
    for name in names:
        gold, meta = load_golden(name)
        case = meta["case"]
        model = build_product_model(case)
        sd = {k: v.float() for k, v in synth_state_dict(model, meta).items()}
        x = case_inputs(case)[0]
        enc = case["encoder"]
        D = O.VIT[enc]["dim"]
        hidden = sd["pretrained.blocks.0.mlp.w3.weight"].shape[1] if O.VIT[enc]["ffn"] == "swiglu" else sd["pretrained.blocks.0.mlp.fc2.weight"].shape[1]
        real = O._Numerics
        with torch.no_grad():
            tr = {}
            ref = O.raw_forward(sd, enc, x, trace=tr)
            zr = tr["logits"]
            print(f"# {name}: raw {enc} {case['H']}x{case['W']}; reference map mean {float(ref.mean()):.3f}, zeros {float((ref == 0).float().mean()):.2f}, "
                  f"r = N+ / sum out = {float((ref > 0).sum() / ref.sum()):.3f}")
            for label, what in configs:
                O._Numerics = lambda dt=None, _w=what: Sel(_w, D, hidden)
                try:
                    tr2 = {}
                    out = O.raw_forward(sd, enc, x, trace=tr2)
                finally:
                    O._Numerics = real
                e = float((out - ref).abs().mean() / ref.abs().mean())
                dz = float((tr2["logits"] - zr).abs().mean())
                t3 = float((tr2["tap3"] - tr["tap3"]).abs().mean() / tr["tap3"].abs().mean())
                print(f"  {os.environ.get('DT', 'fp16')} rounding of {label:72s}: output rel-L1 {e:.3e}   mean |d logit| {dz:.3e}   last tap rel-L1 {t3:.3e}")


if __name__ == "__main__":
    main()
