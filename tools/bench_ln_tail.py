#!/usr/bin/env python
"""Where does the LayerNorm tail's time go?  proj / fc2 shapes at ViT-L bs=32 with the fp32 residual epilogue: (1) no tail, (2) tail armed but the
tickets poisoned so that no tile is ever last (write-through stores + drain + ticket only), (3) the full tail, (4) the stand-alone LayerNorm launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ext as H  # noqa: E402

op = H.operand_dtype()
dev = "cuda"
torch.manual_seed(0)
T, reps = 43840, int(os.environ.get("REPS", "30"))


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, K in (("tiny-K", 64), ("proj", 1024), ("fc2", 4096)):
    N = 1024
    A = torch.randn(T, K, device=dev).to(op)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(op)
    bias, gamma = torch.randn(N, device=dev), torch.rand(N, device=dev) * 0.01
    lw, lb = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    x = torch.randn(T, N, device=dev)
    y = torch.empty(T, N, dtype=op, device=dev)
    cnt = torch.zeros(T // 128 + 2, dtype=torch.int32, device=dev)
    kw = dict(M=T, N=N, K=K, A=A, lda=K, W=W, bias=bias, gamma=gamma, res=x, ldr=N, out_f32=x, ldo_f32=N, flags=H.EP_BIAS | H.EP_GAMMA | H.EP_RESIDUAL)
    tail = dict(ln_weight=lw, ln_bias=lb, ln_eps=1e-6, ln_out=y, ld_ln=N, ln_counter=cnt)
    t_plain = timeit(lambda: H.igemm(**kw))
    cnt.fill_(-(1 << 30))      # nobody ever draws tiles_n - 1
    t_pub = timeit(lambda: H.igemm(**kw, **tail))
    cnt.zero_()
    t_tail = timeit(lambda: H.igemm(**kw, **tail))
    t_ln = timeit(lambda: H.layernorm(x, N, T, N, lw, lb, 1e-6, out_op=y, ld_op=N))
    print(f"{name} M={T} N={N} K={K}: GEMM {t_plain:6.1f} us | + write-through stores, drain, ticket {t_pub:6.1f} us | + LayerNorm by the last arriver {t_tail:6.1f} us | "
          f"stand-alone LayerNorm {t_ln:5.1f} us -> fused {t_tail:6.1f} vs separate {t_plain + t_ln:6.1f} us")
