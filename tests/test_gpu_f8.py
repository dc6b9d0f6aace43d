"""GPU tests of the fp8 correction terms of split-precision products (include/ada_hip.h: ada_igemm_args.f8_from, split_seg < 0):
    x w  ~  x_hi w_hi (fp16 matrix pipe)  +  2^-10 x_lo8 w_hi8  +  x_hi8 w_lo8   (v_mfma_scale_f32_16x16x128_f8f6f4, twice the rate)
Producers write [hi | lo8 | hi8] rows, the packer [w_hi | w_hi8 | w_lo8]; each piece is checked against the decoded bytes (what the kernel must compute
exactly, up to fp32 summation) and against the exact product (what the scheme is for)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _e5m2(t):
    return t.float().clamp(-57344.0, 57344.0).to(torch.float8_e5m2)


def _a_f8(x, op):
    """[..., K] fp32 -> [..., 2 K] operand-typed storage [hi | lo8 | hi8] (what a producer writes for split_seg = -K) + the decoded pieces."""
    hi = x.to(op)
    lo8 = _e5m2((x - hi.float()) * 1024.0)
    hi8 = _e5m2(x)
    K = x.shape[-1]
    packed = torch.cat([hi.contiguous().view(torch.uint8).reshape(*x.shape[:-1], 2 * K), lo8.view(torch.uint8), hi8.view(torch.uint8)], dim=-1)
    return packed.contiguous().view(op), hi.double(), lo8.double() / 1024.0, hi8.double()


def _w_decode(packed, word, K, taps, op):
    """packed [N, taps * 2 K] op-typed from engine.f8_weight_split -> (w_hi, w_hi8 2^-s_hi, w_lo8 2^-s_lo) as float64 [N, taps, K]"""
    n = packed.shape[0]
    b = packed.contiguous().view(torch.uint8).reshape(n, taps, 4 * K)
    hi = b[..., :2 * K].contiguous().view(op).double()
    hi8 = b[..., 2 * K:3 * K].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** (((word >> 8) & 255) - 127)
    lo8 = b[..., 3 * K:].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** (((word >> 24) & 255) - 127)
    assert (word & 255) == 117 and ((word >> 16) & 255) == 127
    return hi, hi8, lo8


def _need_f16(hip):
    if hip.operand_dtype() != torch.float16:
        pytest.skip("the engine uses the fp8 correction terms with fp16 operands only")


@pytest.mark.parametrize("M,N,K,cfg", [(300, 200, 128, -1), (2740, 1152, 384, -1), (1000, 512, 1024, 3), (515, 256, 256, 4), (700, 96, 256, 1), (900, 384, 768, 2),
                                       (4096, 1024, 1536, -1)])
def test_igemm_f8_corrections(hip, forced_tile, M, N, K, cfg):
    from hip_ext import engine as eng
    _need_f16(hip)
    op = torch.float16
    x = _rand(M, K, seed=61) * 2.0
    w = _rand(N, K, seed=62) * K ** -0.5
    b = _rand(N, seed=63)
    a_packed, a_hi, a_lo, a_h8 = _a_f8(x, op)
    w_packed, word = eng.f8_weight_split(w, op)
    w_hi, w_h8, w_l8 = (t[:, 0] for t in _w_decode(w_packed, word, K, 1, op))
    out = torch.zeros(M, N, device=DEV)
    if cfg >= 0:
        forced_tile(cfg, 0)
    hip.igemm(M=M, N=N, K=2 * K, A=a_packed.to(DEV), lda=2 * K, W=w_packed.to(DEV), bias=b.to(DEV), flags=hip.EP_BIAS, out_f32=out, ldo_f32=N,
              f8_from=K, f8_mid=K + K // 2, f8_scales=word)
    decoded = a_hi @ w_hi.T + a_lo @ w_h8.T + a_h8 @ w_l8.T + b.double()
    exact = x.double() @ w.double().T + b.double()
    got = out.cpu().double()
    e_dec = float((got - decoded).abs().mean() / decoded.abs().mean())
    e_f8 = float((got - exact).abs().mean() / exact.abs().mean())
    e_single = float((a_hi @ w_hi.T + b.double() - exact).abs().mean() / exact.abs().mean())
    print(f"fp8 corrections {M}x{N}x{K} tile {cfg}: against the decoded bytes {e_dec:.2e}; against the exact product {e_f8:.2e} (single operands {e_single:.2e})")
    assert e_dec < 2e-6, "the kernel does not contract the bytes it was given"
    assert e_f8 < e_single / 5


@pytest.mark.parametrize("cfg", [-1, 3, 4, 1])
def test_conv3x3_f8_corrections(hip, forced_tile, cfg):
    from hip_ext import engine as eng
    _need_f16(hip)
    op = torch.float16
    B, C, H, W_, Co = 2, 128, 13, 17, 192
    x = _rand(B, C, H, W_, seed=71)
    w = _rand(Co, C, 3, 3, seed=72) * (9 * C) ** -0.5
    xin = torch.zeros(B, H + 2, W_ + 2, 2 * C, dtype=op)
    xin[:, 1:-1, 1:-1] = _a_f8(x.permute(0, 2, 3, 1).contiguous(), op)[0]
    wp, word = eng.f8_weight_split(w.permute(0, 2, 3, 1).reshape(Co, 9 * C), op, taps=9)
    of = torch.zeros(B * H * W_, Co, device=DEV)
    if cfg >= 0:
        forced_tile(cfg, 0)
    hip.igemm(M=B * H * W_, N=Co, K=18 * C, A=xin.to(DEV), lda=2 * C, W=wp.to(DEV), a_mode=hip.A_CONV3, conv=(H, W_, H + 2, W_ + 2, 1), out_f32=of, ldo_f32=Co,
              f8_from=C, f8_mid=C + C // 2, f8_scales=word)
    ref = F.conv2d(x.double(), w.double(), padding=1).permute(0, 2, 3, 1)
    single = F.conv2d(x.to(op).double(), w.to(op).double(), padding=1).permute(0, 2, 3, 1)
    e_f8 = float((of.cpu().double().reshape(B, H, W_, Co) - ref).abs().mean() / ref.abs().mean())
    e_single = float((single - ref).abs().mean() / ref.abs().mean())
    print(f"conv3x3 with fp8 corrections, tile {cfg}: rel-L1 {e_f8:.2e} (single operands {e_single:.2e})")
    assert e_f8 < e_single / 5


def _check_f8_rows(buf, v, C, seg, what):
    """buf [..., 2 seg] fp16 storage written for split_seg = -seg; v [..., C] the fp32 values the producer held"""
    b = buf.cpu().contiguous().view(torch.uint8).reshape(*buf.shape[:-1], 4 * seg)
    hi = b[..., :2 * seg].contiguous().view(torch.float16)[..., :C]
    lo8 = b[..., 2 * seg:2 * seg + C]
    hi8 = b[..., 3 * seg:3 * seg + C]
    v = v.float().cpu()
    want_hi = v.clamp(-65504.0, 65504.0).to(torch.float16)
    want_lo8 = _e5m2((v - want_hi.float()) * 1024.0).view(torch.uint8)
    want_hi8 = _e5m2(v).view(torch.uint8)
    ok = (hi == want_hi) & (lo8 == want_lo8) & (hi8 == want_hi8)
    frac = float(ok.float().mean())
    assert frac > 0.999, f"{what}: only {frac:.4f} of the (hi, lo8, hi8) triples are the roundings of the fp32 value"
    # decoded, the three pieces give the value back to ~2^-14
    back = hi.float() + lo8.view(torch.float8_e5m2).float() / 1024.0
    assert float((back - v).abs().max() / v.abs().max()) < 2e-4, what
    for a, z in ((2 * C, 2 * seg), (2 * seg + C, 3 * seg), (3 * seg + C, 4 * seg)):
        if z > a:
            assert int(b[..., a:z].max()) == 0, what + ": pad bytes written"


@pytest.mark.parametrize("cfg", [-1, 3, 4, 1])
def test_f8_split_stores(hip, forced_tile, cfg):
    _need_f16(hip)
    op = torch.float16
    if cfg >= 0:
        forced_tile(cfg, 4)
    # LayerNorm (both lane mappings: 16 and 64 lanes per row)
    for rows, D, seg in ((40, 96, 128), (300, 1024, 1024), (77, 384, 384)):
        xs = _rand(rows, D, seed=304) * 2
        wln, bln = _rand(D, seed=305), _rand(D, seed=306)
        o = torch.zeros(rows, 2 * seg, dtype=op, device=DEV)
        of = torch.zeros(rows, D, device=DEV)
        hip.layernorm(xs.to(DEV), D, rows, D, wln.to(DEV), bln.to(DEV), 1e-6, out_op=o, ld_op=2 * seg, out_f32=of, ld_f32=D, split_seg=-seg)
        _check_f8_rows(o, of, D, seg, f"layernorm {rows}x{D}")
    # bilinear (identity resample = a cast) and a real resample into a padded grid
    B, C, H, W_ = 2, 48, 9, 11
    x = _rand(B * H * W_, C, seed=301) * 3
    buf = torch.zeros(B * H * W_, 2 * 64, dtype=op, device=DEV)
    hip.bilinear(x.to(DEV), C, B, H, W_, H, W_, C, out_op=buf, ld_op=2 * 64, map_op=hip.MAP_PLAIN, split_seg=-64)
    _check_f8_rows(buf, x, C, 64, "bilinear cast")
    for Cw, hi_, wi_, ho_, wo_ in ((64, 5, 7, 10, 14), (128, 19, 19, 37, 37)):
        xin = _rand(B, Cw, hi_, wi_, seed=303)
        o = torch.zeros(B, ho_ + 2, wo_ + 2, 2 * Cw, dtype=op, device=DEV)
        of = torch.zeros(B * ho_ * wo_, Cw, device=DEV)
        hip.bilinear(xin.permute(0, 2, 3, 1).reshape(-1, Cw).contiguous().to(DEV), Cw, B, hi_, wi_, ho_, wo_, Cw, out_f32=of, ld_f32=Cw,
                     out_op=o, ld_op=2 * Cw, map_op=hip.MAP_PAD, relu=True, split_seg=-Cw)
        _check_f8_rows(o[:, 1:-1, 1:-1], of.reshape(B, ho_, wo_, Cw).clamp_min(0), Cw, Cw, f"bilinear pad C={Cw}")
        border = o.clone()
        border[:, 1:-1, 1:-1] = 0
        assert float(border.abs().max()) == 0.0
    # GEMM epilogues: fp32 + operand copy (4-column path), operand only (8-column path), padded grid
    M, N, K, seg = 700, 192, 128, 256
    A = _rand(M, K, seed=311).to(op).to(DEV)
    Wt = (_rand(N, K, seed=312) * K ** -0.5).to(op).to(DEV)
    b = _rand(N, seed=313).to(DEV)
    res = _rand(M, N, seed=314).to(DEV)
    o = torch.zeros(M, 2 * seg, dtype=op, device=DEV)
    of = torch.zeros(M, N, device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=Wt, bias=b, res=res, ldr=N, flags=hip.EP_BIAS | hip.EP_RESIDUAL, out_f32=of, ldo_f32=N, out_op=o, ldo_op=2 * seg, split_seg=-seg)
    _check_f8_rows(o, of, N, seg, f"igemm fp32 + operand copy, tile {cfg}")
    o2 = torch.zeros(M, 2 * seg, dtype=op, device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=Wt, bias=b, flags=hip.EP_BIAS | hip.EP_RELU_OP, out_op=o2, ldo_op=2 * seg, split_seg=-seg)
    of2 = torch.zeros(M, N, device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=Wt, bias=b, flags=hip.EP_BIAS, out_f32=of2, ldo_f32=N)
    _check_f8_rows(o2, of2.clamp_min(0), N, seg, f"igemm operand only, tile {cfg}")
    Bc, Cc, Hc, Wc = 2, 64, 13, 17
    xc = _rand(Bc, Hc + 2, Wc + 2, Cc, seed=315).to(op)
    xc[:, 0] = 0; xc[:, -1] = 0; xc[:, :, 0] = 0; xc[:, :, -1] = 0
    wc = (_rand(Cc, 9 * Cc, seed=316) * (9 * Cc) ** -0.5).to(op)
    oc = torch.zeros(Bc, Hc + 2, Wc + 2, 2 * Cc, dtype=op, device=DEV)
    ofc = torch.zeros(Bc * Hc * Wc, Cc, device=DEV)
    hip.igemm(M=Bc * Hc * Wc, N=Cc, K=9 * Cc, A=xc.to(DEV), lda=Cc, W=wc.to(DEV), a_mode=hip.A_CONV3, conv=(Hc, Wc, Hc + 2, Wc + 2, 1),
              flags=hip.EP_RELU_OP, out_f32=ofc, ldo_f32=Cc, out_op=oc, ldo_op=2 * Cc, map_op=hip.MAP_PAD, map_h=Hc, map_w=Wc, split_seg=-Cc)
    _check_f8_rows(oc[:, 1:-1, 1:-1], ofc.reshape(Bc, Hc, Wc, Cc).clamp_min(0), Cc, Cc, f"igemm padded grid, tile {cfg}")
    border = oc.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0


def test_f8_store_behind_a_sub_pixel_shuffle_and_contraction_into_the_tail(hip):
    """ConvTranspose k = s = 2 writing the fp8 form (the MAP_SHUFFLE epilogue); a 3x3 convolution reading it into the fused ReLU -> 1x1 -> activation tail"""
    from hip_ext import engine as eng
    _need_f16(hip)
    op = torch.float16
    B, s_, Co, Ci, segc = 2, 2, 128, 64, 128
    xs = _rand(B, Ci, 5, 6, seed=317).to(op).float()
    wt = (_rand(Ci, Co, s_, s_, seed=318) * Ci ** -0.5).to(op).float()
    A2 = xs.permute(0, 2, 3, 1).reshape(-1, Ci).to(op).contiguous()
    Wp = wt.permute(2, 3, 1, 0).reshape(s_ * s_ * Co, Ci).to(op).contiguous()
    o = torch.zeros(B, 12, 14, 2 * segc, dtype=op, device=DEV)
    hip.igemm(M=B * 30, N=s_ * s_ * Co, K=Ci, A=A2.to(DEV), lda=Ci, W=Wp.to(DEV), out_op=o, ldo_op=2 * segc, map_op=hip.MAP_SHUFFLE,
              map_h=5, map_w=6, shuffle_s=s_, shuffle_c=Co, split_seg=-segc)
    ref = F.conv_transpose2d(xs, wt, stride=s_).permute(0, 2, 3, 1)
    b = o[:, 1:-1, 1:-1].cpu().contiguous().view(torch.uint8).reshape(B, 10, 12, 4 * segc)
    hi = b[..., :2 * segc].contiguous().view(op).float()
    lo = b[..., 2 * segc:3 * segc].contiguous().view(torch.float8_e5m2).float() / 1024.0
    h8 = b[..., 3 * segc:].contiguous().view(torch.float8_e5m2).float()
    assert float((hi + lo - ref).abs().max() / ref.abs().max()) < 2e-4
    assert float((h8 - ref).abs().max() / ref.abs().max()) < 0.13       # two mantissa bits
    border = o.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0
    with pytest.raises(hip.HipExtError):     # 4-column groups are not offered behind a shuffle
        hip.igemm(M=B * 30, N=s_ * s_ * 12, K=Ci, A=A2.to(DEV), lda=Ci, W=Wp[:48].contiguous().to(DEV), out_op=o, ldo_op=2 * segc, map_op=hip.MAP_SHUFFLE,
                  map_h=5, map_w=6, shuffle_s=s_, shuffle_c=12, split_seg=-segc)
    # 3x3 conv (N = 32) + ReLU + 1x1 + sigmoid from that tensor (output_conv2's shape)
    w3 = _rand(32, Co, 3, 3, seed=319) * (9 * Co) ** -0.5
    b3, tw = _rand(32, seed=320) * 0.1, _rand(32, seed=321) * 0.3
    wp, word = eng.f8_weight_split(w3.permute(0, 2, 3, 1).reshape(32, 9 * Co), op, taps=9)
    out = torch.zeros(B * 10 * 12, device=DEV)
    hip.igemm(M=B * 120, N=32, K=18 * Co, A=o, lda=2 * segc, W=wp.to(DEV), a_mode=hip.A_CONV3, conv=(10, 12, 12, 14, 1), bias=b3.to(DEV), flags=hip.EP_BIAS | hip.EP_TAIL,
              out_f32=out, ldo_f32=1, tail_w=tw.to(DEV), tail_b=0.05, tail_act=hip.ACT_SIGMOID, f8_from=Co, f8_mid=Co + Co // 2, f8_scales=word)
    y = F.relu(F.conv2d(ref.permute(0, 3, 1, 2).double(), w3.double(), b3.double(), padding=1))
    want = torch.sigmoid((y * tw.double()[None, :, None, None]).sum(1) + 0.05)
    err = float((out.cpu().double().reshape(B, 10, 12) - want).abs().mean() / want.abs().mean())
    assert err < 3e-5, err


def test_f8_arguments_are_validated(hip):
    _need_f16(hip)
    op = torch.float16
    A = torch.zeros(256, 512, dtype=op, device=DEV)
    W = torch.zeros(64, 512, dtype=op, device=DEV)
    out = torch.zeros(256, 64, device=DEV)
    for bad in (dict(f8_from=100, f8_mid=384), dict(f8_from=256, f8_mid=128), dict(f8_from=256, f8_mid=576), dict(f8_from=512, f8_mid=512)):
        with pytest.raises(hip.HipExtError):
            hip.igemm(M=256, N=64, K=512, A=A, lda=512, W=W, out_f32=out, ldo_f32=64, f8_scales=0x7f7f7f7f, **bad)
    with pytest.raises(hip.HipExtError):
        hip.igemm(M=256, N=64, K=512, A=A, lda=256, a_wrap=256, W=W, out_f32=out, ldo_f32=64, f8_from=256, f8_mid=384, f8_scales=0x7f7f7f7f)


def _check_bytes_against_hi_lo(lo8, hi8, hi, lo, what):
    """The byte segments of a [hi | lo8 | hi8] row against the [hi | lo] form of the same launch: lo8 = e5m2((v - hi) 2^10), hi8 = e5m2(v) with v the producer's fp32
    value, of which hi + lo is all the test can see.  e5m2 keeps 2 mantissa bits (relative error <= 2^-3); the fp16 lo is itself rounded -- to a 6e-8 grid where it is
    subnormal, which it is for the small values these producers write -- so the bytes are compared to that tolerance, not bit for bit (the producer rounds the fp32
    residual directly, which is the better value)."""
    v = hi + lo
    lo_dec = lo8.contiguous().view(torch.float8_e5m2).float() / 1024.0
    hi_dec = hi8.contiguous().view(torch.float8_e5m2).float()
    ok_lo = (lo_dec - lo).abs() <= 0.13 * lo.abs() + 1.2e-7
    ok_hi = (hi_dec - v).abs() <= 0.13 * v.abs() + 1.6e-5      # (e5m2's smallest subnormal is 2^-16)
    frac = float((ok_lo & ok_hi).float().mean())
    assert frac > 0.9995, f"{what}: only {frac:.5f} of the (lo8, hi8) pairs decode to the value's residual / the value within e5m2's rounding"
    back = hi + lo_dec
    assert float((back - v).abs().max() / v.abs().max()) < 2e-4, what


# ---- round 6: the activations that exist in the operand type only -- attention output (feeds attn.proj), SwiGLU hidden (feeds mlp.w3) -- in the split forms ----
@pytest.mark.parametrize("B,N,heads,ld", [(2, 1370, 2, 256), (1, 65, 3, 384), (3, 200, 6, 1024)])
def test_attention_split_output_forms(hip, B, N, heads, ld):
    """ada_attention_ex: the plain row with a stride, [hi | lo] (split_seg > 0) and [hi | lo8 | hi8] (split_seg < 0).  The hi segment is the plain output bit for
    bit; hi + lo is the kernel's fp32 value (so it is nearer the fp32 softmax(QK^T)V than hi alone, and |lo| <= ulp(hi) / 2); the byte segments are the e5m2
    roundings of that value's residual x 2^10 and of the value."""
    _need_f16(hip)
    op = torch.float16
    LOG2E = 1.4426950408889634
    D = heads * 64
    seg = ld // 2
    qkv = _rand(B * N, 3 * D, seed=633)
    qkv[:, :D] *= 0.125 * LOG2E
    qkv = qkv.to(op).to(DEV)
    plain = torch.zeros(B * N, D, dtype=op, device=DEV)
    hip.attention(qkv, plain, B, N, heads)
    strided = torch.zeros(B * N, ld, dtype=op, device=DEV)
    hip.attention(qkv, strided, B, N, heads, ld_out=ld)
    assert torch.equal(strided[:, :D], plain) and float(strided[:, D:].abs().max()) == 0.0
    two = torch.zeros(B * N, ld, dtype=op, device=DEV)
    hip.attention(qkv, two, B, N, heads, ld_out=ld, split_seg=seg)
    assert torch.equal(two[:, :D], plain)
    hi, lo = two[:, :D].float().cpu(), two[:, seg:seg + D].float().cpu()
    assert float(lo.abs().max()) > 0 and bool((lo.abs() <= hi.abs() * 2.0 ** -11 + 6.0e-8).all())
    t = qkv.float().cpu().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (((t[0] @ t[1].transpose(-2, -1)) / LOG2E).softmax(-1) @ t[2]).transpose(1, 2).reshape(B * N, D)
    e_hi, e_two = float((hi - ref).abs().mean()), float((hi + lo - ref).abs().mean())
    print(f"attention B={B} N={N} heads={heads}: mean |err| of hi {e_hi:.3e}, of hi + lo {e_two:.3e}")
    assert e_two <= e_hi
    f8 = torch.zeros(B * N, ld, dtype=op, device=DEV)
    hip.attention(qkv, f8, B, N, heads, ld_out=ld, split_seg=-seg)
    b = f8.cpu().contiguous().view(torch.uint8).reshape(B * N, 2 * ld)
    assert torch.equal(b[:, :2 * D].contiguous().view(op), plain.cpu())
    _check_bytes_against_hi_lo(b[:, 2 * seg:2 * seg + D], b[:, 3 * seg:3 * seg + D], hi, lo, "attention")
    for a, z in ((2 * D, 2 * seg), (2 * seg + D, 3 * seg), (3 * seg + D, 4 * seg)):
        if z > a:
            assert int(b[:, a:z].max()) == 0, "pad bytes written"


@pytest.mark.parametrize("cfg", [-1, 3, 4])
def test_swiglu_split_output_forms_and_the_w3_contraction(hip, forced_tile, cfg):
    """EP_SWIGLU with split_seg: the gated hidden silu(x1) * x2 (swiglu_ffn.py:31-32) written [hi | lo] and [hi | lo8 | hi8], then contracted by a w3-shaped
    launch (f8_from) -- against the exact product of the fp32 hidden with the fp32 weights: ~1e-5 where the plain fp16 hidden leaves ~3e-4."""
    _need_f16(hip)
    from hip_ext.engine import f8_weight_split
    op = torch.float16
    if cfg >= 0:
        forced_tile(cfg, 4)
    M, K, Hd, N = 1000, 128, 256, 192
    A = _rand(M, K, seed=640).to(op).to(DEV)
    w12 = (_rand(2 * Hd, K, seed=641) * K ** -0.5).to(op)
    b12 = _rand(2 * Hd, seed=642)
    idx = torch.arange(Hd).reshape(-1, 32)
    order = torch.stack([idx, idx + Hd], dim=1).reshape(-1)
    x12 = A.double().cpu() @ w12.double().T + b12.double()
    ref = (F.silu(x12[:, :Hd]) * x12[:, Hd:]).float()
    kw = dict(M=M, N=2 * Hd, K=K, A=A, lda=K, W=w12[order].contiguous().to(DEV), bias=b12[order].contiguous().to(DEV), flags=hip.EP_BIAS | hip.EP_SWIGLU)
    plain = torch.zeros(M, Hd, dtype=op, device=DEV)
    hip.igemm(out_op=plain, ldo_op=Hd, **kw)
    two = torch.zeros(M, 2 * Hd, dtype=op, device=DEV)
    hip.igemm(out_op=two, ldo_op=2 * Hd, split_seg=Hd, **kw)
    assert torch.equal(two[:, :Hd], plain)
    rec = two[:, :Hd].float().cpu() + two[:, Hd:].float().cpu()
    assert float((rec - ref).abs().max() / ref.abs().max()) < 2e-5
    f8 = torch.zeros(M, 2 * Hd, dtype=op, device=DEV)
    hip.igemm(out_op=f8, ldo_op=2 * Hd, split_seg=-Hd, **kw)
    b = f8.cpu().contiguous().view(torch.uint8).reshape(M, 4 * Hd)
    assert torch.equal(b[:, :2 * Hd].contiguous().view(op), plain.cpu())
    _check_bytes_against_hi_lo(b[:, 2 * Hd:3 * Hd], b[:, 3 * Hd:], two[:, :Hd].float().cpu(), two[:, Hd:].float().cpu(), f"swiglu tile {cfg}")
    # the consumer: mlp.w3 over the split hidden, fp8 correction terms
    w3 = _rand(N, Hd, seed=643) * Hd ** -0.5
    packed, word = f8_weight_split(w3.to(DEV), op)
    out = torch.zeros(M, N, device=DEV)
    hip.igemm(M=M, N=N, K=2 * Hd, A=f8, lda=2 * Hd, W=packed, f8_from=Hd, f8_mid=Hd + Hd // 2, f8_scales=word, out_f32=out, ldo_f32=N)
    single = torch.zeros(M, N, device=DEV)
    hip.igemm(M=M, N=N, K=Hd, A=plain, lda=Hd, W=w3.to(op).to(DEV), out_f32=single, ldo_f32=N)
    exact = (ref.double() @ w3.double().T).float()
    e8 = float((out.cpu() - exact).abs().mean() / exact.abs().mean())
    e1 = float((single.cpu() - exact).abs().mean() / exact.abs().mean())
    print(f"w3 over the split hidden (tile {cfg}): fp8 correction terms {e8:.2e} against the exact product, single precision {e1:.2e}")
    assert e8 < 4e-5 and e8 < e1 / 4


@pytest.mark.parametrize("cfg", [-1, 3, 4])
@pytest.mark.parametrize("s_", [2, 4])
def test_sub_pixel_convolution_with_f8_corrections(hip, forced_tile, cfg, s_):
    """Round 6: the merged transposed-conv + 3x3 conv of a split-precision level (ada_igemm_args.tap_cols / tap_mask: the k-walk of an N-tile visits the union of its
    phases' taps only) over a [hi | lo8 | hi8] patch-grid tensor against per-tap [w_hi | w_hi8 | w_lo8] weights -- the masked tap walk and the fp8 k-steps together.
    Against conv3x3(conv_transpose(x)) in float64: the scheme's error, far below single fp16 operands'."""
    from hip_ext import engine as eng
    from hip_ext.functional import subpixel_merge
    _need_f16(hip)
    op = torch.float16
    B, C, H, W_, Cm, Co = 2, 128, 9, 11, 32, 64
    x = _rand(B, C, H, W_, seed=81)
    wt = _rand(C, Cm, s_, s_, seed=82) * C ** -0.5
    bt = _rand(Cm, seed=83) * 0.1
    w3 = _rand(Co, Cm, 3, 3, seed=84) * (9 * Cm) ** -0.5
    b3 = _rand(Co, seed=85) * 0.1
    wm, bias, tapb, masks = subpixel_merge(wt.to(DEV), bt.to(DEV), w3.to(DEV), b3.to(DEV), s_)
    wq, word = eng.f8_weight_split(wm.reshape(wm.shape[0], -1), op, taps=9)
    xin = torch.zeros(B, H + 2, W_ + 2, 2 * C, dtype=op)
    xin[:, 1:-1, 1:-1] = _a_f8(x.permute(0, 2, 3, 1).contiguous(), op)[0]
    ncol = s_ * s_ * Co
    of = torch.zeros(B * H * W_, ncol, device=DEV)
    if cfg >= 0:
        forced_tile(cfg, 0)
    hip.igemm(M=B * H * W_, N=ncol, K=18 * C, A=xin.to(DEV), lda=2 * C, W=wq, a_mode=hip.A_CONV3, conv=(H, W_, H + 2, W_ + 2, 1), bias=bias, flags=hip.EP_BIAS,
              out_f32=of, ldo_f32=ncol, tap_cols=Co, tap_mask=masks, f8_from=C, f8_mid=C + C // 2, f8_scales=word)
    # reference: the fine map, then the 3x3 conv over it with zero padding -- the merged conv adds the transposed conv's bias under every tap, also those that fall into the
    # padding of the fine grid (the LayerNorm behind it takes them out through `tapb`): compare on the interior phases' pixels, where no tap is padded
    fine = F.conv_transpose2d(x.double(), wt.double(), bt.double(), stride=s_)
    ref = F.conv2d(fine, w3.double(), b3.double(), padding=1)                    # [B, Co, s H, s W]
    got = of.cpu().double().reshape(B, H, W_, s_, s_, Co).permute(0, 5, 1, 3, 2, 4).reshape(B, Co, s_ * H, s_ * W_)
    inner = (slice(None), slice(None), slice(1, s_ * H - 1), slice(1, s_ * W_ - 1))
    e_f8 = float((got[inner] - ref[inner]).abs().mean() / ref[inner].abs().mean())
    fine1 = F.conv_transpose2d(x.to(op).double(), wt.to(op).double(), bt.double(), stride=s_)
    single = F.conv2d(fine1.to(op).double(), w3.to(op).double(), b3.double(), padding=1)
    e_single = float((single[inner] - ref[inner]).abs().mean() / ref[inner].abs().mean())
    print(f"sub-pixel convolution s={s_} with fp8 corrections, tile {cfg}: rel-L1 {e_f8:.2e} (two single-precision launches {e_single:.2e})")
    assert e_f8 < 5e-5 and e_f8 < e_single / 5
