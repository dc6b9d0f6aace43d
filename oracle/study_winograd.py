"""TEST INFRASTRUCTURE / numerics study (not imported by the product): would Winograd F(2x2, 3x3) on fp16 operands keep the 1e-3 bar?

The 3x3 stride-1 convolutions of the DPT head's ResidualConvUnits (DA2/util/blocks.py:49-76) are ~5 TFLOP of the 36.5 the ViT-L bs=32 step
executes; F(2x2, 3x3) does them in 16 instead of 36 multiplies per output pair (2.25x fewer MACs).  The price is numerical: the transformed
input tile V = B^T d B (sums of up to four activations) and the transformed filter U = G g G^T have to be rounded to fp16 before the matrix cores
see them, and the output transform A^T M A adds products of mixed sign.  This script runs the oracle forward with EVERY contraction's operands
rounded to fp16 (the product's single-precision path, to first order) and then again with the ResidualConvUnit convolutions evaluated by an
emulated fp16-operand / fp32-accumulate Winograd, and reports both relative L1 errors against the fp32 forward.

    python oracle/study_winograd.py vitb_518 [vitl_518 ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "amodal-depth-anything_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import dav2_oracle as O  # noqa: E402
from oracle.study_operand_noise import forward  # noqa: E402
from _cases import case_inputs, load_golden, schema_state_dict  # noqa: E402

BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def winograd_conv3x3(x, w, b, dt, split_v=False):
    """conv2d(x, w, b, padding=1) by F(2x2, 3x3); x and w are already operand-rounded; U and V are rounded to `dt`, sums in fp32 (fp64 here)."""
    B, C, H, W = x.shape
    Co = w.shape[0]
    nh, nw = (H + 1) // 2, (W + 1) // 2
    xp = torch.nn.functional.pad(x.double(), (1, 1 + 2 * nw - W, 1, 1 + 2 * nh - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [B, C, nh, nw, 4, 4]
    bt = BT.double()
    V = torch.einsum("ij,bcxyjk,lk->bcxyil", bt, d, bt)          # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", G.double(), w.double(), G.double())
    q = lambda t: t.to(torch.float32).to(dt).to(torch.float64)  # noqa: E731
    Uq = q(U)
    if split_v:      # V as hi + lo (two operand-typed terms): what a split-precision variant would contract
        Vh = q(V)
        Vq = Vh + q(V - Vh)
    else:
        Vq = q(V)
    M = torch.einsum("ocil,bcxyil->boxyil", Uq, Vq)
    at = AT.double()
    Y = torch.einsum("pi,boxyil,ql->boxpyq", at, M, at)          # A^T M A -> [B, Co, nh, 2, nw, 2]
    y = Y.reshape(B, Co, 2 * nh, 2 * nw)[:, :, :H, :W]
    if b is not None:
        y = y + b.double().view(1, -1, 1, 1)
    return y.to(torch.float32)


class WinogradRCU(O._Numerics):
    """All contractions operand-rounded to dt; 3x3 / stride 1 / pad 1 convolutions with Cin == Cout (the ResidualConvUnit convs) by Winograd when on."""

    def __init__(self, dt, winograd, split_v=False):
        super().__init__(dt)
        self.winograd, self.split_v, self.count, self.macs = winograd, split_v, 0, 0

    def _q(self, t):
        return t.to(self.dt).to(torch.float32)

    def linear(self, x, w, b=None):
        return torch.nn.functional.linear(self._q(x), self._q(w), b)

    def conv(self, x, w, b=None, stride=1, padding=0):
        if self.winograd and w.shape[2:] == (3, 3) and stride == 1 and padding == 1 and w.shape[0] == w.shape[1] and x.shape[2] * x.shape[3] >= 16:
            self.count += 1
            self.macs += x.shape[0] * x.shape[2] * x.shape[3] * w.shape[0] * w.shape[1] * 9
            return winograd_conv3x3(self._q(x), self._q(w), b, self.dt, self.split_v)
        return torch.nn.functional.conv2d(self._q(x), self._q(w), b, stride=stride, padding=padding)

    def convT(self, x, w, b, stride):
        return torch.nn.functional.conv_transpose2d(self._q(x), self._q(w), b, stride=stride)

    def matmul(self, a, b):
        return self._q(a) @ self._q(b)


def main():
    names = sys.argv[1:] or ["vitb_518"]
    dt = torch.float16
    for name in names:
        _, meta = load_golden(name)
        case = dict(meta["case"])
        if "take" in case:       # one image is enough for a noise study
            case["take"] = case["take"][:1]
        sd = schema_state_dict(case, meta)
        inputs = case_inputs(case)
        with torch.no_grad():
            ref = forward(case, sd, inputs, O._Numerics(torch.float32))
            direct = forward(case, sd, inputs, WinogradRCU(dt, False))
            nm = WinogradRCU(dt, True)
            wino = forward(case, sd, inputs, nm)
            nm2 = WinogradRCU(dt, True, split_v=True)
            wino2 = forward(case, sd, inputs, nm2)
        print(f"{name}: every contraction on fp16 operands, direct convolutions      rel-L1 = {O.rel_l1(direct, ref):.3e}")
        print(f"{name}: + {nm.count} ResidualConvUnit convs by Winograd F(2x2,3x3), fp16 U and V rel-L1 = {O.rel_l1(wino, ref):.3e}   ({nm.macs / 1e9:.1f} GMAC direct -> {nm.macs / 2.25e9:.1f})")
        print(f"{name}: + the same with V = hi + lo (two fp16 terms, 2x the transformed MACs) rel-L1 = {O.rel_l1(wino2, ref):.3e}", flush=True)


if __name__ == "__main__":
    main()
