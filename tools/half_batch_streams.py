#!/usr/bin/env python
"""Experiment: a small batch whose launches fill half a round of the 256 CUs, run as TWO half batches on two HIP streams (second
stream staggered) against the plain forward.  A throw-away measurement: two module copies so that each half has its own workspace.
    ENCODER=vitb B=8 python tools/half_batch_streams.py"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hip_ext import engine as E  # noqa: E402
from src.models import get_model  # noqa: E402
from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs  # noqa: E402

ENC, B = os.environ.get("ENCODER", "vitb"), int(os.environ.get("B", "8"))


def main():
    m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder=ENC, pretrained=False).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    fill_state_dict_(sd, 0)
    cb = centred_final_bias(ENC, ROOT)
    if cb:
        sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
    m.load_state_dict(sd)
    m = m.cuda()
    m2 = copy.deepcopy(m)
    for mm in (m, m2):
        mm.encoder.precision_ladder = False      # no host read between the launches: the streams must be free to overlap
    x, _, mask, obs = make_inputs(B, 518, 518, 0, device="cuda")
    h = B // 2
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def whole():
        return m(x, guide_mask=mask, observation=obs)

    def halves():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = m(x[:h], guide_mask=mask[:h], observation=obs[:h])
        with torch.cuda.stream(s2):
            b = m2(x[h:], guide_mask=mask[h:], observation=obs[h:])
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        return torch.cat([a, b], 0)

    def timeit(fn, n=20):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    for mode in ("0", "1"):
        E.GRAPH_MODE = mode
        tw, th = timeit(whole), timeit(halves)
        with torch.no_grad():
            same = torch.equal(whole(), halves())
        print(f"{ENC} {B} x 518^2, graph replay {'on' if mode == '1' else 'off'}: whole batch {tw * 1e3:.3f} ms = {B / tw:.0f} images/s; two half batches on two streams "
              f"{th * 1e3:.3f} ms = {B / th:.0f} images/s ({(tw / th - 1) * 100:+.1f} %); outputs bit-identical: {same}")


if __name__ == "__main__":
    main()
