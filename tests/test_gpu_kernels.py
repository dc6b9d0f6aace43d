"""GPU parity tests of each libada_hip entry point against a plain PyTorch fp32 computation of the same
op on the same operand-rounded inputs (so the only differences are fp32 summation order)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOG2E = 1.4426950408889634


def _op(hip):
    return hip.operand_dtype()


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _close(got, ref, atol, rtol=2e-3, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = err > lim
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})"


def test_selftest_fragment_layouts(hip):
    assert hip.selftest() == 0, hip.load().ada_last_error()


@pytest.mark.parametrize("M,N,K", [(300, 200, 128), (128, 128, 64), (1, 4, 64), (2740, 1152, 384), (257, 48, 192), (515, 32, 128)])
def test_igemm_plain_bias(hip, M, N, K):
    op = _op(hip)
    A = _rand(M, K, seed=1).to(op).to(DEV)
    W = _rand(N, K, scale=K ** -0.5, seed=2).to(op).to(DEV)
    b = _rand(N, seed=3).to(DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, flags=hip.EP_BIAS, out_f32=out, ldo_f32=N)
    ref = A.float().cpu() @ W.float().cpu().T + b.cpu()
    _close(out, ref, 2e-4, what="igemm bias")


def test_igemm_gelu_operand_out(hip):
    op = _op(hip)
    M, N, K = 700, 384, 256
    A = _rand(M, K, seed=4).to(op).to(DEV)
    W = _rand(N, K, scale=K ** -0.5, seed=5).to(op).to(DEV)
    b = _rand(N, seed=6).to(DEV)
    out = torch.zeros(M, N, dtype=op, device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, flags=hip.EP_BIAS | hip.EP_GELU, out_op=out, ldo_op=N)
    ref = F.gelu(A.float().cpu() @ W.float().cpu().T + b.cpu())
    _close(out, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="igemm gelu")


def test_igemm_layerscale_residual_inplace(hip):
    op = _op(hip)
    M, N, K = 1370, 384, 384
    A = _rand(M, K, seed=7).to(op).to(DEV)
    W = _rand(N, K, scale=K ** -0.5, seed=8).to(op).to(DEV)
    b, g = _rand(N, seed=9).to(DEV), _rand(N, seed=10).to(DEV)
    x = _rand(M, N, seed=11).to(DEV)
    ref = x.cpu() + (A.float().cpu() @ W.float().cpu().T + b.cpu()) * g.cpu()
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, gamma=g, res=x, ldr=N,
              flags=hip.EP_BIAS | hip.EP_GAMMA | hip.EP_RESIDUAL, out_f32=x, ldo_f32=N)
    _close(x, ref, 3e-4, what="igemm ls+res")


def test_igemm_token_map_with_pos(hip):
    """patch-embed epilogue: rows skip the cls slot, pos_embed added with the row index modulo Np."""
    op = _op(hip)
    B, Np, D, K = 3, 50, 128, 192
    A = _rand(B * Np, K, seed=12).to(op).to(DEV)
    W = _rand(D, K, scale=K ** -0.5, seed=13).to(op).to(DEV)
    b = _rand(D, seed=14).to(DEV)
    pos = _rand(Np + 1, D, seed=15).to(DEV)
    x = torch.zeros(B * (Np + 1), D, device=DEV)
    hip.igemm(M=B * Np, N=D, K=K, A=A, lda=K, W=W, bias=b, res=pos, ldr=D, res_row_mod=Np, res_row_off=1,
              flags=hip.EP_BIAS | hip.EP_RESIDUAL, out_f32=x, ldo_f32=D, map_f32=hip.MAP_TOKEN, map_h=Np)
    ref = (A.float().cpu() @ W.float().cpu().T + b.cpu()).reshape(B, Np, D) + pos.cpu()[1:]
    got = x.reshape(B, Np + 1, D).cpu()
    _close(got[:, 1:], ref, 3e-4, what="token map")
    assert float(got[:, 0].abs().max()) == 0.0


def _pad_nhwc(x_nchw, cp, op):
    B, C, H, W = x_nchw.shape
    buf = torch.zeros(B, H + 2, W + 2, cp, dtype=op)
    buf[:, 1:-1, 1:-1, :C] = x_nchw.permute(0, 2, 3, 1).to(op)
    return buf


def _pack3(w, cp, op):
    co, ci = w.shape[:2]
    p = torch.zeros(co, 3, 3, cp)
    p[..., :ci] = w.permute(0, 2, 3, 1)
    return p.reshape(co, 9 * cp).to(op)


@pytest.mark.parametrize("B,C,H,W,Co,stride", [(2, 64, 19, 19, 64, 1), (1, 48, 37, 37, 96, 1), (2, 128, 37, 37, 128, 2), (1, 64, 30, 23, 200, 1)])
def test_igemm_conv3x3(hip, B, C, H, W, Co, stride):
    op = _op(hip)
    cp = (C + 63) // 64 * 64
    x = _rand(B, C, H, W, seed=16).to(op).float()
    w = (_rand(Co, C, 3, 3, seed=17) * (9 * C) ** -0.5).to(op).float()
    b = _rand(Co, seed=18)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.zeros(B * Ho * Wo, Co, device=DEV)
    xin = _pad_nhwc(x, cp, op).to(DEV)
    hip.igemm(M=B * Ho * Wo, N=Co, K=9 * cp, A=xin, lda=cp, W=_pack3(w, cp, op).to(DEV), a_mode=hip.A_CONV3,
              conv=(Ho, Wo, H + 2, W + 2, stride), bias=b.to(DEV), flags=hip.EP_BIAS, out_f32=out, ldo_f32=Co)
    ref = F.conv2d(x, w, b, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    _close(out, ref, 3e-4, what="conv3x3")


def test_igemm_conv3x3_relu_padded_output_and_residual(hip):
    op = _op(hip)
    B, C, H, W = 2, 64, 21, 17
    x = _rand(B, C, H, W, seed=19).to(op).float()
    w = (_rand(C, C, 3, 3, seed=20) * (9 * C) ** -0.5).to(op).float()
    b = _rand(C, seed=21)
    res = _rand(B * H * W, C, seed=22)
    out_f = torch.zeros(B * H * W, C, device=DEV)
    out_p = torch.zeros(B, H + 2, W + 2, C, dtype=op, device=DEV)
    hip.igemm(M=B * H * W, N=C, K=9 * C, A=_pad_nhwc(x, C, op).to(DEV), lda=C, W=_pack3(w, C, op).to(DEV), a_mode=hip.A_CONV3,
              conv=(H, W, H + 2, W + 2, 1), bias=b.to(DEV), res=res.to(DEV), ldr=C, flags=hip.EP_BIAS | hip.EP_RESIDUAL | hip.EP_RELU_OP,
              out_f32=out_f, ldo_f32=C, out_op=out_p, ldo_op=C, map_op=hip.MAP_PAD, map_h=H, map_w=W)
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, C) + res
    _close(out_f, ref, 3e-4, what="conv+res f32")
    _close(out_p[:, 1:-1, 1:-1].reshape(-1, C), ref.clamp_min(0), 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="conv relu padded")
    border = out_p.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0


@pytest.mark.parametrize("s,C,Ci", [(4, 48, 48), (2, 96, 96), (4, 256, 256)])
def test_igemm_conv_transpose_shuffle(hip, s, C, Ci):
    op = _op(hip)
    B, H, W = 2, 7, 5
    cp = (Ci + 63) // 64 * 64
    x = _rand(B, Ci, H, W, seed=23).to(op).float()
    w = (_rand(Ci, C, s, s, seed=24) * Ci ** -0.5).to(op).float()
    b = _rand(C, seed=25)
    A = torch.zeros(B * H * W, cp, dtype=op)
    A[:, :Ci] = x.permute(0, 2, 3, 1).reshape(-1, Ci).to(op)
    Wp = torch.zeros(s * s * C, cp)
    Wp[:, :Ci] = w.permute(2, 3, 1, 0).reshape(s * s * C, Ci)
    out = torch.zeros(B, s * H + 2, s * W + 2, C, dtype=op, device=DEV)
    hip.igemm(M=B * H * W, N=s * s * C, K=cp, A=A.to(DEV), lda=cp, W=Wp.to(op).to(DEV), bias=b.repeat(s * s).to(DEV), flags=hip.EP_BIAS,
              out_op=out, ldo_op=C, map_op=hip.MAP_SHUFFLE, map_h=H, map_w=W, shuffle_s=s, shuffle_c=C)
    ref = F.conv_transpose2d(x, w, b, stride=s).permute(0, 2, 3, 1)
    _close(out[:, 1:-1, 1:-1], ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="convT")


# Sub-pixel convolution: ConvTranspose2d(k = s, stride s) followed by Conv2d(3x3, padding 1) with nothing in between (reference DA2/dpt.py
# resize_layers[0/1] -> input_projection[i][0]) as ONE masked-tap 3x3 convolution over the coarse grid + the LayerNorm that reads it
@pytest.mark.parametrize("s,Ci,Cm,Co,H,W,cfg", [(4, 48, 48, 48, 7, 5, -1), (2, 96, 96, 96, 9, 6, -1), (4, 256, 256, 256, 12, 11, 3), (2, 512, 128, 512, 10, 13, 3),
                                               (4, 64, 64, 64, 1, 1, -1), (2, 64, 64, 192, 1, 3, 4), (4, 96, 64, 96, 20, 17, 3), (4, 128, 128, 128, 16, 16, 1)])
def test_subpixel_conv_matches_conv_of_conv_transpose(hip, forced_tile, s, Ci, Cm, Co, H, W, cfg):
    from hip_ext.functional import subpixel_merge
    op = _op(hip)
    B = 2
    cp = (Ci + 63) // 64 * 64
    x = _rand(B, Ci, H, W, seed=31).to(op).float()
    wt = _rand(Ci, Cm, s, s, seed=32) * Ci ** -0.5
    bt = _rand(Cm, seed=33)
    w3 = _rand(Co, Cm, 3, 3, seed=34) * (9 * Cm) ** -0.5
    b3 = _rand(Co, seed=35)
    ref = F.conv2d(F.conv_transpose2d(x, wt, bt, stride=s), w3, b3, padding=1)          # fp32, [B, Co, sH, sW]
    wm, bias, tapb, masks = subpixel_merge(wt.to(DEV), bt.to(DEV), w3.to(DEV), b3.to(DEV), s)     # composed on the device (functional.compose_f32)
    wm, bias, tapb = wm.cpu(), bias.cpu(), tapb.cpu()
    assert sum(bin(m).count("1") for m in masks) == (36 if s == 4 else 16)
    wp = torch.zeros(s * s * Co, 9, cp)
    wp[..., :Ci] = wm
    N = s * s * Co
    P = B * H * W
    out = torch.full((P, N), float("nan"), device=DEV)
    if cfg >= 0:
        forced_tile(cfg, 0)
    hip.igemm(M=P, N=N, K=9 * cp, A=_pad_nhwc(x, cp, op).to(DEV), lda=cp, W=wp.reshape(N, 9 * cp).to(op).to(DEV), a_mode=hip.A_CONV3,
              conv=(H, W, H + 2, W + 2, 1), bias=bias.to(DEV), flags=hip.EP_BIAS, out_f32=out, ldo_f32=N, tap_cols=Co, tap_mask=masks)
    if cfg >= 0:
        _check_tile(hip, cfg, 0)
    # (a) the raw [coarse pixel, phase * Co] output, un-shuffled on the host, with the ring correction applied on the host
    got = out.cpu().view(B, H, W, s, s, Co).permute(0, 5, 1, 3, 2, 4).reshape(B, Co, s * H, s * W).clone()
    tb = tapb.view(s, s, Co, 3, 3)
    Y, X = torch.arange(s * H), torch.arange(s * W)
    for dy in range(3):
        for dx in range(3):
            miss_y = (Y // s + dy - 1 < 0) | (Y // s + dy - 1 >= H)
            miss_x = (X // s + dx - 1 < 0) | (X // s + dx - 1 >= W)
            miss = (miss_y[:, None] | miss_x[None, :]).float()                            # [sH, sW]
            t = tb[Y % s][:, X % s][:, :, :, dy, dx].permute(2, 0, 1)                     # [Co, sH, sW]
            got -= (miss[None] * t)[None]
    wide = 8.0 if op == torch.bfloat16 else 1.0      # (the bf16 measurement library: eight times the operand rounding)
    _close(got, ref, 2e-3 * wide, rtol=3e-3 * wide, what="sub-pixel conv (host un-shuffle)")
    # (b) the LayerNorm that consumes it in place: channels-first LN + ReLU into a zero-bordered operand tensor (DA2/dpt.py:153-159)
    g, beta = 1.0 + 0.1 * _rand(Co, seed=36), 0.1 * _rand(Co, seed=37)
    outp = torch.zeros(B, s * H + 2, s * W + 2, Co, dtype=op, device=DEV)
    hip.layernorm(out, N, B * s * H * s * W, Co, g.to(DEV), beta.to(DEV), 1e-6, out_op=outp, ld_op=Co, map_op=hip.MAP_PAD, map_h=s * H, map_w=s * W,
                  relu=True, unshuffle_s=s, tap_bias=tapb.to(DEV))
    ln_ref = F.relu(F.layer_norm(ref.permute(0, 2, 3, 1), (Co,), g, beta, 1e-6))
    _close(outp[:, 1:-1, 1:-1], ln_ref, 6e-3, rtol=1e-2 if op == torch.bfloat16 else 4e-3, what="LayerNorm over the sub-pixel output")
    border = outp.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0
    # (c) the raw head's consumer is not a LayerNorm: the same kernel as a re-layout pass (identity), fp32 pre-activation + ReLU'd operand copy
    outp2 = torch.zeros(B, s * H + 2, s * W + 2, Co, dtype=op, device=DEV)
    outf2 = torch.full((B * s * H * s * W, Co), float("nan"), device=DEV)
    hip.layernorm(out, N, B * s * H * s * W, Co, None, None, 1e-6, identity=True, relu=2, out_f32=outf2, ld_f32=Co, out_op=outp2, ld_op=Co, map_op=hip.MAP_PAD,
                  map_h=s * H, map_w=s * W, unshuffle_s=s, tap_bias=tapb.to(DEV))
    _close(outf2.view(B, s * H, s * W, Co), ref.permute(0, 2, 3, 1), 2e-3, rtol=3e-3, what="re-layout pass, fp32 copy")
    _close(outp2[:, 1:-1, 1:-1], F.relu(ref.permute(0, 2, 3, 1)), 4e-3, rtol=1e-2 if op == torch.bfloat16 else 4e-3, what="re-layout pass, ReLU'd operand copy")


@pytest.mark.parametrize("M,N,K,cfg", [(300, 200, 128, -1), (2740, 1152, 384, -1), (1000, 512, 1024, 3), (515, 256, 256, 4)])
def test_igemm_weight_only_split(hip, forced_tile, M, N, K, cfg):
    """ada_igemm_args.a_wrap: a plain operand-typed activation walked twice against [w_hi | w_lo] weights = x w to fp32 weight accuracy (the
    activation's own rounding is the caller's: here x is exactly representable).  Used for proj / fc2 / w3 of the split-precision blocks."""
    op = _op(hip)
    x = _rand(M, K, seed=51).to(op)
    w = _rand(N, K, seed=52) * K ** -0.5
    b = _rand(N, seed=53)
    w_hi = w.to(op)
    wp = torch.cat([w_hi, (w - w_hi.float()).to(op)], dim=1).contiguous()
    out = torch.zeros(M, N, device=DEV)
    if cfg >= 0:
        forced_tile(cfg, 0)
    hip.igemm(M=M, N=N, K=2 * K, A=x.to(DEV), lda=K, a_wrap=K, W=wp.to(DEV), bias=b.to(DEV), flags=hip.EP_BIAS, out_f32=out, ldo_f32=N)
    ref = x.double() @ w.double().T + b.double()
    single = x.float() @ w_hi.float().T + b
    e_split = float((out.cpu().double() - ref).abs().mean() / ref.abs().mean())
    e_single = float((single.double() - ref).abs().mean() / ref.abs().mean())
    print(f"weight-only split {M}x{N}x{K}: rel-L1 {e_split:.2e} (single operands: {e_single:.2e})")
    assert e_split < (2e-6 if op == torch.float16 else 2e-5) and e_split < 0.05 * e_single


@pytest.mark.parametrize("B,C,Cin,hi,wi,ho,wo", [(2, 128, 256, 37, 37, 74, 74), (1, 64, 128, 20, 30, 40, 60), (3, 32, 64, 9, 11, 18, 22), (1, 128, 64, 10, 10, 25, 33),
                                                 (1, 128, 256, 148, 148, 296, 296), (2, 64, 64, 1, 1, 2, 2), (1, 128, 128, 5, 7, 16, 15)])
@pytest.mark.parametrize("tmap", ["f32", "op"])
def test_tapsum_resize_is_conv3x3_of_the_upsampled_map(hip, B, C, Cin, hi, wi, ho, wo, tmap):
    """ada_tapsum_resize_fwd on the nine coarse tap maps (W_t W_out) u + W_t b_out (one GEMM) == conv3x3(bilinear_ac(out_conv(u))) in fp32."""
    op = _op(hip)
    u = _rand(B, Cin, hi, wi, seed=61).to(op).float()
    w_out = _rand(Cin, Cin, 1, 1, seed=62) * Cin ** -0.5
    b_out = _rand(Cin, seed=63)
    w1 = _rand(C, Cin, 3, 3, seed=64) * (9 * Cin) ** -0.5
    b1 = _rand(C, seed=65)
    ref = F.conv2d(F.interpolate(F.conv2d(u, w_out, b_out), size=(ho, wo), mode="bilinear", align_corners=True), w1, b1, padding=1)
    wt = w1.double().permute(2, 3, 0, 1).reshape(9 * C, Cin)
    wc = (wt @ w_out.double().reshape(Cin, Cin)).float()
    bc = (wt @ b_out.double()).float()
    A = u.permute(0, 2, 3, 1).reshape(-1, Cin).to(op).contiguous().to(DEV)
    T = torch.zeros(B * hi * wi, 9 * C, dtype=op if tmap == "op" else torch.float32, device=DEV)
    tout = dict(out_op=T, ldo_op=9 * C) if tmap == "op" else dict(out_f32=T, ldo_f32=9 * C)
    hip.igemm(M=B * hi * wi, N=9 * C, K=Cin, A=A, lda=Cin, W=wc.to(op).to(DEV), bias=bc.to(DEV), flags=hip.EP_BIAS, **tout)
    out = torch.full((B * ho * wo, C), float("nan"), device=DEV)
    hip.tapsum_resize(T, 9 * C, B, hi, wi, ho, wo, C, b1.to(DEV), out, C)
    got = out.view(B, ho, wo, C).permute(0, 3, 1, 2)
    # operand rounding of the composed weights and of the nine tap maps: a few 1e-3 absolute on outputs of magnitude ~1
    _close(got, ref, 4e-2 if op == torch.bfloat16 else 5e-3, rtol=2e-2 if op == torch.bfloat16 else 4e-3, what="tap-sum resize")
    err = float((got.cpu() - ref).abs().mean() / ref.abs().mean())
    print(f"tap-sum resize {B}x{C}x{hi}x{wi}->{ho}x{wo}: rel-L1 {err:.2e}")
    assert err < (8e-3 if op == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("wo", [17, 33, 47, 50, 65, 81, 97, 113])
def test_tapsum_resize_right_edge_halo_column_reads_staged_data(hip, wo):
    """Widths that are not a multiple of the 16-pixel tile: the last tile of a row starts at tx0 > 0 (patch origin px0 > 0) and its halo holds the masked column
    X == wo.  The compute loop issues that column's LDS reads (weight 0): their offsets must stay inside the staged patch -- with a source index of 0 they
    were (0 - px0) * PIXB < 0, bytes of another slab or of the static tables, and 0 * Inf/NaN bit patterns reached the edge pixel x = wo - 1."""
    op = _op(hip)
    B, C, Cin, hi, ho = 1, 64, 64, 9, 18
    wi = (wo + 1) // 2
    u = (_rand(B, Cin, hi, wi, seed=66) * 200.0).to(op).float()       # large tap-map values: half-precision bit patterns near the exponent top
    w1 = _rand(C, Cin, 3, 3, seed=67) * (9 * Cin) ** -0.5
    b1 = _rand(C, seed=68)
    ref = F.conv2d(F.interpolate(u, size=(ho, wo), mode="bilinear", align_corners=True), w1, b1, padding=1)
    wt = w1.permute(2, 3, 0, 1).reshape(9 * C, Cin)
    A = u.permute(0, 2, 3, 1).reshape(-1, Cin).to(op).contiguous().to(DEV)
    T = torch.zeros(B * hi * wi, 9 * C, dtype=torch.float32, device=DEV)
    hip.igemm(M=B * hi * wi, N=9 * C, K=Cin, A=A, lda=Cin, W=wt.to(op).to(DEV), flags=0, out_f32=T, ldo_f32=9 * C)
    for tdt in (torch.float32, op):
        out = torch.full((B * ho * wo, C), float("nan"), device=DEV)
        hip.tapsum_resize(T.to(tdt), 9 * C, B, hi, wi, ho, wo, C, b1.to(DEV), out, C)
        got = out.view(B, ho, wo, C).permute(0, 3, 1, 2).cpu()
        assert torch.isfinite(got).all(), f"wo={wo}: non-finite output at the right edge"
        err = float((got[..., -2:] - ref[..., -2:]).abs().mean() / ref[..., -2:].abs().mean())
        assert err < 3e-3, f"wo={wo} {tdt}: last two columns rel-L1 {err:.2e}"


@pytest.mark.parametrize("B,H,W", [(3, 518, 518), (1, 14, 14), (2, 126, 154)])
def test_depth_stats_per_image_moments(hip, B, H, W):
    """ada_depth_stats_fwd: per-image (sum s, sum s (1 - s)) in fixed-order chunks -- the precision ladder's first trigger."""
    s_ = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(5)).to(DEV)
    sums = torch.full((B, 8, 2), float("nan"), device=DEV)
    hip.depth_stats(s_, sums)
    got = sums.cpu().double().sum(1)
    ref = torch.stack([s_.cpu().double().flatten(1).sum(1), (s_.cpu().double() * (1 - s_.cpu().double())).flatten(1).sum(1)], dim=1)
    assert torch.allclose(got, ref, rtol=2e-6), (got, ref)
    again = torch.empty_like(sums)
    hip.depth_stats(s_, again)
    assert torch.equal(sums, again)       # no atomics: bit-reproducible
    # the pairs of the heads without a sigmoid (ABI 8): ReLU (sum out, number of positive outputs), none (sum |out|, number of outputs)
    z = (s_ - 0.7) * 3.0
    zr = z.clamp_min(0)
    hip.depth_stats(zr, sums, hip.ACT_RELU)
    got = sums.cpu().double().sum(1)
    ref = torch.stack([zr.cpu().double().flatten(1).sum(1), (zr.cpu() > 0).double().flatten(1).sum(1)], dim=1)
    assert torch.allclose(got, ref, rtol=2e-6), (got, ref)
    hip.depth_stats(z, sums, hip.ACT_NONE)
    got = sums.cpu().double().sum(1)
    ref = torch.stack([z.cpu().double().abs().flatten(1).sum(1), torch.full((B,), float(H * W), dtype=torch.float64)], dim=1)
    assert torch.allclose(got, ref, rtol=2e-6), (got, ref)


@pytest.mark.parametrize("B,Np,D,ld", [(2, 1369, 768, 768), (3, 99, 1024, 2048), (1, 1, 384, 384), (2, 37, 100, 128)])
def test_token_diversity_of_a_tap(hip, B, Np, D, ld):
    """ada_token_diversity_fwd: sum of the columns' variances over an image's tokens and sum of their mean squares, per 64-column chunk -- ~0 for an image
    whose tokens are all alike (the ladder's second trigger), the [hi | lo] lo half of a split tap ignored (ld > dim)."""
    op = _op(hip)
    t = _rand(B, Np, ld, seed=81).to(op)
    if B > 1:
        t[1, :, :D] = t[1, :1, :D].clone()       # image 1: every token equal
    t = t.reshape(B * Np, ld).contiguous().to(DEV)
    G = (D + 63) // 64
    sums = torch.full((B, G, 2), float("nan"), device=DEV)
    hip.token_diversity(t, ld, B, Np, D, sums)
    got = sums.cpu().double().sum(1)
    x = t.cpu().double().view(B, Np, ld)[:, :, :D]
    ref = torch.stack([x.var(dim=1, unbiased=False).sum(-1), (x * x).mean(dim=1).sum(-1)], dim=1)
    assert torch.allclose(got, ref, rtol=1e-3, atol=1e-3 * float(ref[:, 1].max())), (got, ref)
    if B > 1 and Np > 1:
        assert float(got[1, 0] / got[1, 1]) < 1e-3 < float(got[0, 0] / got[0, 1])


@pytest.mark.parametrize("M,N,K", [(9 * 256, 16 * 256, 256), (9 * 96, 4 * 96, 96), (1152, 256, 256), (7, 5, 3)])
def test_compose_f32_weight_products(hip, M, N, K):
    """functional.compose_f32 (weight composition at pack time: sub-pixel merges, output_conv1 o out_conv) on the library's own split-precision GEMM
    against the fp64 product: ~fp32 accuracy, any M / N / K (padded internally)."""
    from hip_ext.functional import compose_f32
    a, b = _rand(M, K, seed=301) * K ** -0.5, _rand(N, K, seed=302) * K ** -0.5
    got = compose_f32(a.to(DEV), b.to(DEV)).cpu()
    ref = a.double() @ b.double().T
    assert got.shape == (M, N)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    print(f"compose_f32 {M}x{N}x{K}: max abs error / max |ref| = {err:.2e}")
    assert err < (2e-6 if _op(hip) == torch.float16 else 1e-4)     # dropped lo x lo terms and the lo halves' own rounding: ~2^-22 each


@pytest.mark.parametrize("G,rows,N,K,cfg,gelu", [(3, 50, 64, 128, -1, True), (5, 1369, 384, 384, -1, True), (2, 300, 256, 192, 3, False), (4, 77, 128, 64, 4, True), (7, 9, 96, 128, 1, False)])
def test_igemm_bias_per_row_group(hip, forced_tile, G, rows, N, K, cfg, gelu):
    """ada_igemm_args.bias_row_mod: one bias vector per group of rows (the class-token read-out's per-image bias) in ONE launch."""
    op = _op(hip)
    M = G * rows
    x = _rand(M, K, seed=71).to(op).float()
    w = (_rand(N, K, seed=72) * K ** -0.5).to(op).float()
    b = _rand(G, N, seed=73)
    out = torch.zeros(M, 2 * N, dtype=op, device=DEV)
    if cfg >= 0:
        forced_tile(cfg, 0)
    hip.igemm(M=M, N=N, K=K, A=x.to(op).to(DEV), lda=K, W=w.to(op).to(DEV), bias=b.to(DEV), bias_row_mod=rows, flags=hip.EP_BIAS | (hip.EP_GELU if gelu else 0),
              out_op=out, ldo_op=2 * N, split_seg=N)
    ref = x @ w.T + b.repeat_interleave(rows, dim=0)
    if gelu:
        ref = F.gelu(ref)
    _close(out[:, :N].float() + out[:, N:].float(), ref, 2e-5 if op == torch.float16 else 3e-4, rtol=2e-5 if op == torch.float16 else 3e-4, what="per-group bias (hi + lo)")


def test_layernorm_second_output_drops_cls_rows(hip):
    """One pass, two normalised outputs of the same rows: all rows with (gain, bias) 1 -> the next block's LN1; the rows of every group of N
    but the first with (gain, bias) 2, compacted -> the tap LayerNorm (DA2/dinov2.py:337-340)."""
    op = _op(hip)
    B, N, D = 3, 11, 384
    x = _rand(B * N, D, seed=41) * 2 + 0.3
    g1, b1, g2, b2 = 1 + 0.1 * _rand(D, seed=42), 0.1 * _rand(D, seed=43), 1 + 0.1 * _rand(D, seed=44), 0.1 * _rand(D, seed=45)
    o1 = torch.zeros(B * N, D, dtype=op, device=DEV)
    o2 = torch.zeros(B * (N - 1), 2 * D, dtype=op, device=DEV)
    hip.layernorm(x.to(DEV), D, B * N, D, g1.to(DEV), b1.to(DEV), 1e-6, out_op=o1, ld_op=D, weight2=g2.to(DEV), bias2=b2.to(DEV), out2_op=o2, ld2_op=2 * D,
                  out2_group=N, out2_skip=1, split_seg2=D)
    r1 = F.layer_norm(x, (D,), g1, b1, 1e-6)
    r2 = F.layer_norm(x.view(B, N, D)[:, 1:].reshape(-1, D), (D,), g2, b2, 1e-6)
    tol = dict(atol=2e-2, rtol=1e-2) if op == torch.bfloat16 else dict(atol=2e-3, rtol=2e-3)
    _close(o1, r1, tol["atol"], rtol=tol["rtol"], what="LN main output")
    hi, lo = o2[:, :D].float().cpu(), o2[:, D:].float().cpu()
    _close(hi, r2, tol["atol"], rtol=tol["rtol"], what="LN second output (hi)")
    _close(hi + lo, r2, 2e-5 if op == torch.float16 else 2e-4, rtol=2e-5 if op == torch.float16 else 2e-4, what="LN second output (hi + lo)")


@pytest.mark.parametrize("act", ["sigmoid", "relu", "none"])
def test_igemm_tail(hip, act):
    op = _op(hip)
    B, C, H, W, Co = 1, 64, 28, 42, 32
    x = _rand(B, C, H, W, seed=26).to(op).float()
    w = (_rand(Co, C, 3, 3, seed=27) * (9 * C) ** -0.5).to(op).float()
    b = _rand(Co, seed=28)
    tw, tb = _rand(Co, seed=29), 0.25
    out = torch.zeros(B, 1, H, W, device=DEV)
    code = {"sigmoid": hip.ACT_SIGMOID, "relu": hip.ACT_RELU, "none": hip.ACT_NONE}[act]
    hip.igemm(M=B * H * W, N=Co, K=9 * C, A=_pad_nhwc(x, C, op).to(DEV), lda=C, W=_pack3(w, C, op).to(DEV), a_mode=hip.A_CONV3,
              conv=(H, W, H + 2, W + 2, 1), bias=b.to(DEV), flags=hip.EP_BIAS | hip.EP_TAIL, out_f32=out, ldo_f32=1,
              tail_w=tw.to(DEV), tail_b=tb, tail_act=code)
    d = (F.relu(F.conv2d(x, w, b, padding=1)) * tw.view(1, -1, 1, 1)).sum(1, keepdim=True) + tb
    ref = {"sigmoid": torch.sigmoid, "relu": F.relu, "none": lambda t: t}[act](d)
    _close(out, ref, 3e-4, what="tail")


def test_igemm_swiglu(hip):
    op = _op(hip)
    M, K, Hd = 333, 128, 192
    A = _rand(M, K, seed=30).to(op).to(DEV)
    w12 = (_rand(2 * Hd, K, seed=31) * K ** -0.5).to(op)
    b12 = _rand(2 * Hd, seed=32)
    idx = torch.arange(Hd).reshape(-1, 32)
    order = torch.stack([idx, idx + Hd], dim=1).reshape(-1)
    out = torch.zeros(M, Hd, dtype=op, device=DEV)
    hip.igemm(M=M, N=2 * Hd, K=K, A=A, lda=K, W=w12[order].contiguous().to(DEV), bias=b12[order].contiguous().to(DEV),
              flags=hip.EP_BIAS | hip.EP_SWIGLU, out_op=out, ldo_op=Hd)
    x12 = A.float().cpu() @ w12.float().T + b12
    ref = F.silu(x12[:, :Hd]) * x12[:, Hd:]
    _close(out, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="swiglu")


@pytest.mark.parametrize("B,N,heads", [(2, 1370, 2), (1, 64, 1), (1, 65, 3), (1, 1, 1), (3, 200, 6)])
def test_attention(hip, B, N, heads):
    op = _op(hip)
    D = heads * 64
    qkv = _rand(B * N, 3 * D, seed=33).to(op)
    qkv[:, :D] *= 0.125 * LOG2E  # the packer folds head_dim**-0.5 * log2(e) into q (base-2 softmax in the kernel)
    qkv = qkv.to(op)
    out = torch.zeros(B * N, D, dtype=op, device=DEV)
    hip.attention(qkv.to(DEV), out, B, N, heads)
    t = qkv.float().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    p = ((t[0] @ t[1].transpose(-2, -1)) / LOG2E).softmax(-1)
    ref = (p @ t[2]).transpose(1, 2).reshape(B * N, D)
    _close(out, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 3e-3, what="attention")


def test_attention_forces_online_rescale(hip):
    """One key in a late tile dominates: the running max must jump and rescale the accumulated O (T13 hazard)."""
    op = _op(hip)
    B, N, heads = 1, 300, 1
    qkv = _rand(B * N, 192, seed=34) * 0.3
    qkv[:, 64:128][250] = qkv[:, :64][7] * 40.0   # k_250 aligned with q_7
    qkv = qkv.to(op)
    out = torch.zeros(N, 64, dtype=op, device=DEV)
    hip.attention(qkv.to(DEV), out, B, N, heads)
    t = qkv.float()
    p = ((t[:, :64] @ t[:, 64:128].T) / LOG2E).softmax(-1)
    _close(out, p @ t[:, 128:], 2e-3, rtol=1e-2 if op == torch.bfloat16 else 3e-3, what="attention rescale")


@pytest.mark.parametrize("rows,dim", [(1370, 384), (77, 1024), (5, 1536), (100, 48)])
def test_layernorm_plain(hip, rows, dim):
    op = _op(hip)
    x = _rand(rows, dim, scale=3.0, seed=35) + 0.5
    w, b = _rand(dim, seed=36), _rand(dim, seed=37)
    o_op = torch.zeros(rows, dim, dtype=op, device=DEV)
    o_f = torch.zeros(rows, dim, device=DEV)
    hip.layernorm(x.to(DEV), dim, rows, dim, w.to(DEV), b.to(DEV), 1e-6, out_op=o_op, ld_op=dim, out_f32=o_f, ld_f32=dim)
    ref = F.layer_norm(x, (dim,), w, b, 1e-6)
    _close(o_f, ref, 2e-5, rtol=1e-5, what="ln f32")
    _close(o_op, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="ln op")


def test_layernorm_drop_cls_and_padded_relu(hip):
    op = _op(hip)
    B, N, D = 3, 21, 128   # tokens: 1 cls + 4x5 grid
    x = _rand(B * N, D, seed=38)
    w, b = _rand(D, seed=39), _rand(D, seed=40)
    ref = F.layer_norm(x, (D,), w, b, 1e-6).reshape(B, N, D)[:, 1:]
    o = torch.zeros(B * (N - 1), D, dtype=op, device=DEV)
    hip.layernorm(x.to(DEV), D, B * (N - 1), D, w.to(DEV), b.to(DEV), 1e-6, group_in=N, skip=1, out_op=o, ld_op=D)
    _close(o, ref.reshape(-1, D), 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="ln drop cls")
    xp = _rand(B * 20, D, seed=41)
    o2 = torch.zeros(B, 6, 7, D, dtype=op, device=DEV)
    hip.layernorm(xp.to(DEV), D, B * 20, D, w.to(DEV), b.to(DEV), 1e-6, out_op=o2, ld_op=D, map_op=hip.MAP_PAD, map_h=4, map_w=5, relu=True)
    ref2 = F.relu(F.layer_norm(xp, (D,), w, b, 1e-6)).reshape(B, 4, 5, D)
    _close(o2[:, 1:-1, 1:-1], ref2, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="ln pad relu")


@pytest.mark.parametrize("cg", [0, 2, 5])
def test_patchify(hip, cg):
    op = _op(hip)
    B, H, W = 2, 42, 56
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(42))
    g = torch.rand(B, max(cg, 1), H, W, generator=torch.Generator().manual_seed(43)) * 2 - 1
    K = (3 + cg) * 196
    ld = (K + 63) // 64 * 64
    out = torch.full((B * 12, ld), 7.0, dtype=op, device=DEV)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    hip.patchify(x.to(DEV), g.to(DEV) if cg else None, B, cg, H, W, mean, tuple(1 / s for s in std), out, ld)
    xn = (x - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    full = torch.cat([xn, g], 1) if cg else xn
    ref = F.unfold(full, 14, stride=14).transpose(1, 2).reshape(B * 12, K)
    _close(out[:, :K], ref, 1e-3, rtol=1e-2 if op == torch.bfloat16 else 1e-3, what="patchify")
    assert float(out[:, K:].float().abs().max()) == 0.0 if ld > K else True


def test_write_cls(hip):
    B, N, D = 3, 10, 64
    t = torch.zeros(B * N, D, device=DEV)
    cls, pos = _rand(D, seed=44), _rand(N, D, seed=45)
    hip.write_cls(t, B, N, D, cls.to(DEV), pos.to(DEV))
    got = t.reshape(B, N, D).cpu()
    assert torch.equal(got[:, 0], (cls + pos[0]).expand(B, D))
    assert float(got[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("hi,wi,ho,wo", [(19, 19, 37, 37), (37, 37, 74, 74), (8, 5, 16, 10), (20, 30, 35, 49), (3, 3, 1, 1)])
def test_bilinear_align_corners(hip, hi, wi, ho, wo):
    op = _op(hip)
    B, C = 2, 64
    x = _rand(B, C, hi, wi, seed=46)
    add = _rand(B * ho * wo, C, seed=47)
    of = torch.zeros(B * ho * wo, C, device=DEV)
    oo = torch.zeros(B, ho + 2, wo + 2, C, dtype=op, device=DEV)
    hip.bilinear(x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV), C, B, hi, wi, ho, wo, C, add=add.to(DEV), ld_add=C,
                 out_f32=of, ld_f32=C, out_op=oo, ld_op=C, map_op=hip.MAP_PAD, relu=True)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).reshape(-1, C) + add
    _close(of, ref, 2e-5, rtol=1e-5, what="bilinear f32")
    _close(oo[:, 1:-1, 1:-1].reshape(-1, C), ref.clamp_min(0), 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="bilinear op")


@pytest.mark.parametrize("C,hi,wi,ho,wo", [(128, 19, 19, 37, 37), (256, 37, 37, 74, 74), (128, 20, 30, 35, 49), (256, 30, 41, 60, 82), (128, 74, 74, 129, 129)])
@pytest.mark.parametrize("mode", ["all", "f32_only", "op_plain"])
def test_bilinear_tiled_wide_channels(hip, C, hi, wi, ho, wo, mode):
    """128 / 256 channels take the LDS-tiled kernel (8x32 output tiles, source patch staged once): partial tiles at the right and
    bottom edges, non-square maps, a non-integer scale, every output combination."""
    op = _op(hip)
    B = 2
    x = _rand(B, C, hi, wi, seed=146)
    xin = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).reshape(-1, C)
    if mode == "all":
        add = _rand(B * ho * wo, C, seed=147)
        of = torch.zeros(B * ho * wo, C, device=DEV)
        oo = torch.zeros(B, ho + 2, wo + 2, C, dtype=op, device=DEV)
        hip.bilinear(xin, C, B, hi, wi, ho, wo, C, add=add.to(DEV), ld_add=C, out_f32=of, ld_f32=C, out_op=oo, ld_op=C, map_op=hip.MAP_PAD, relu=True)
        ref = ref + add
        _close(of, ref, 2e-5, rtol=1e-5, what="tiled bilinear f32")
        _close(oo[:, 1:-1, 1:-1].reshape(-1, C), ref.clamp_min(0), 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="tiled bilinear op")
        border = oo.clone()
        border[:, 1:-1, 1:-1] = 0
        assert float(border.abs().max()) == 0.0
    elif mode == "f32_only":
        of = torch.zeros(B * ho * wo, C, device=DEV)
        hip.bilinear(xin, C, B, hi, wi, ho, wo, C, out_f32=of, ld_f32=C)
        _close(of, ref, 2e-5, rtol=1e-5, what="tiled bilinear f32 only")
    else:
        oo = torch.zeros(B * ho * wo, C, dtype=op, device=DEV)
        hip.bilinear(xin, C, B, hi, wi, ho, wo, C, out_op=oo, ld_op=C, map_op=hip.MAP_PLAIN)
        _close(oo, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="tiled bilinear op plain")


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 40, 23), (3, 128, 9, 5), (1, 256, 33, 70)])
def test_igemm_conv3x3_padded_output_interior_and_edge_tiles(hip, B, C, H, W):
    """Zero-bordered NHWC output over tiles that are fully interior (streamlined epilogue, row walk without divisions), tiles that
    cross image boundaries, the ragged last tile, and a grid narrower than the walk step (general path)."""
    op = _op(hip)
    x = _rand(B, C, H, W, seed=119).to(op).float()
    w = (_rand(C, C, 3, 3, seed=120) * (9 * C) ** -0.5).to(op).float()
    b = _rand(C, seed=121)
    out_p = torch.zeros(B, H + 2, W + 2, C, dtype=op, device=DEV)
    hip.igemm(M=B * H * W, N=C, K=9 * C, A=_pad_nhwc(x, C, op).to(DEV), lda=C, W=_pack3(w, C, op).to(DEV), a_mode=hip.A_CONV3,
              conv=(H, W, H + 2, W + 2, 1), bias=b.to(DEV), flags=hip.EP_BIAS | hip.EP_RELU_OP,
              out_op=out_p, ldo_op=C, map_op=hip.MAP_PAD, map_h=H, map_w=W)
    ref = F.conv2d(x, w, b, padding=1).clamp_min(0).permute(0, 2, 3, 1).reshape(-1, C)
    _close(out_p[:, 1:-1, 1:-1].reshape(-1, C), ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="conv relu padded")
    border = out_p.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0


def test_errors_are_reported_not_thrown(hip):
    op = _op(hip)
    A = torch.zeros(64, 64, dtype=op, device=DEV)
    out = torch.zeros(64, 64, device=DEV)
    with pytest.raises(hip.HipExtError, match="multiple of 64"):
        hip.igemm(M=64, N=64, K=48, A=A, lda=64, W=A, out_f32=out, ldo_f32=64)
    with pytest.raises(hip.HipExtError, match="HIP device"):
        hip.igemm(M=64, N=64, K=64, A=A.cpu(), lda=64, W=A, out_f32=out, ldo_f32=64)
    with pytest.raises(hip.HipExtError, match="multiple of the 14-pixel patch"):
        hip.patchify(torch.zeros(1, 3, 30, 28, device=DEV), None, 1, 0, 30, 28, None, None, torch.zeros(4, 640, dtype=op, device=DEV), 640)


def test_pipeline_glue_kernels(hip):
    """min/max, normalise, paste + border blur against the host restatement used by infer.py."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import infer
    B, H, W = 3, 37, 53
    d = _rand(B, H, W, seed=50).abs() * 3
    mm = torch.zeros(B, 2, device=DEV)
    hip.minmax(d.to(DEV), mm)
    assert torch.equal(mm.cpu()[:, 0], d.reshape(B, -1).min(1).values) and torch.equal(mm.cpu()[:, 1], d.reshape(B, -1).max(1).values)
    norm, obs = torch.zeros(B, H, W, device=DEV), torch.zeros(B, 1, H, W, device=DEV)
    hip.normalize(d.to(DEV), mm, norm=norm, obs=obs)
    ref = (d - d.reshape(B, -1).min(1).values.view(B, 1, 1)) / (d.reshape(B, -1).max(1).values - d.reshape(B, -1).min(1).values).view(B, 1, 1)
    assert torch.allclose(norm.cpu(), ref, atol=1e-6) and torch.allclose(obs.cpu()[:, 0], ref * 2 - 1, atol=1e-6)
    am = torch.rand(B, H, W, generator=torch.Generator().manual_seed(51))
    mask = torch.zeros(B, H, W)
    mask[0, 5:20, 10:30] = 1
    mask[1, :, :7] = 1          # touches the image border: exercises reflect-101
    mask[2, 30:, 40:] = 1
    out = torch.zeros(B, H, W, device=DEV)
    hip.blend(am.to(DEV), ref.contiguous().to(DEV), mask.to(DEV), out)
    for b in range(B):
        want = infer.median_filter_blend(am[b], ref[b].clone(), mask[b].numpy())
        assert torch.allclose(out[b].cpu(), want, atol=1e-6), float((out[b].cpu() - want).abs().max())


def test_patchify_split_precision(hip):
    """[hi | lo] layout: hi + lo reproduces the fp32 pixel to ~2^-22, and a GEMM over the k segments (hi, lo, hi) against [w_hi | w_hi | w_lo]
    (a_dup_seg: the third segment re-reads the first) matches fp32."""
    op = _op(hip)
    B, H, W, cg = 1, 28, 42, 2
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(60))
    g = torch.rand(B, cg, H, W, generator=torch.Generator().manual_seed(61)) * 2 - 1
    K = (3 + cg) * 196
    seg = (K + 63) // 64 * 64
    out = torch.zeros(B * 6, 2 * seg, dtype=op, device=DEV)
    hip.patchify(x.to(DEV), g.to(DEV), B, cg, H, W, None, None, out, 2 * seg, split=True)
    ref = F.unfold(torch.cat([x, g], 1), 14, stride=14).transpose(1, 2).reshape(B * 6, K)
    o = out.float().cpu()
    rec = o[:, :K] + o[:, seg:seg + K]
    tol = 1e-6 if op == torch.float16 else 1e-4
    assert float((rec - ref).abs().max()) < tol
    w = _rand(64, K, seed=62) * K ** -0.5
    wp = torch.zeros(64, seg)
    wp[:, :K] = w
    w_hi = wp.to(op)
    w_lo = (wp - w_hi.float()).to(op)
    Wcat = torch.cat([w_hi, w_hi, w_lo], 1).contiguous().to(DEV)
    y = torch.zeros(B * 6, 64, device=DEV)
    hip.igemm(M=B * 6, N=64, K=3 * seg, A=out, lda=2 * seg, a_dup_seg=seg, W=Wcat, out_f32=y, ldo_f32=64)
    err = (y.cpu() - ref @ w.T).abs().max()
    assert float(err) < (2e-5 if op == torch.float16 else 2e-3), float(err)


# =====================================================================================================================
# Every tile configuration x every epilogue: the heuristic of ada_igemm picks the 256x256 tile only
# for problems with hundreds of tiles, so the small shapes above never reach the kernel the benchmark spends 70 % of its
# time in.  These tests force each tile (ada_debug_set_tile) and both main loops of the 256x256 tile
# (ada_debug_set_variant: 16 = hand-scheduled 4-wave loop, 4 = single-barrier 8-wave loop) on problems that span >= 3 tile rows, end in
# a ragged tile, and (for the wide ones) engage the column-group tile order.
# =====================================================================================================================
TILE_CASES = [(0, 4), (1, 4), (2, 4), (3, 4), (3, 16), (4, 4)]   # (tile cfg, main-loop variant: 4 single barrier, 8 phased)


def _check_tile(hip, cfg, variant):
    code = hip.debug_last_tile()
    assert code % 100 == cfg, f"forced tile {cfg} but the launch used {code}"
    if cfg == 3:
        assert code // 100 == {16: 2}.get(variant, 0), f"variant {variant} but tile code {code}"


@pytest.mark.parametrize("cfg,variant", TILE_CASES)
@pytest.mark.parametrize("epi", ["std_f32_res", "gelu", "gamma_op", "relu_op_and_f32", "plain_op"])
def test_igemm_forced_tile_epilogues(hip, forced_tile, cfg, variant, epi):
    op = _op(hip)
    M, N, K = 3 * 256 + 77, 2 * 256 + 64, 320     # 4 ragged tile rows of 256, partial last N tile, odd number of k-tiles (5)
    A = _rand(M, K, seed=201).to(op).to(DEV)
    W = _rand(N, K, scale=K ** -0.5, seed=202).to(op).to(DEV)
    b, g = _rand(N, seed=203).to(DEV), (_rand(N, seed=204) * 0.5 + 1).to(DEV)
    lin = A.float().cpu() @ W.float().cpu().T + b.cpu()
    forced_tile(cfg, variant)
    loose = dict(atol=2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3)
    if epi == "std_f32_res":
        x = _rand(M, N, seed=205).to(DEV)
        ref = x.cpu() + lin * g.cpu()
        hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, gamma=g, res=x, ldr=N, flags=hip.EP_BIAS | hip.EP_GAMMA | hip.EP_RESIDUAL, out_f32=x, ldo_f32=N)
        _check_tile(hip, cfg, variant)
        _close(x, ref, 3e-4, what=f"tile {cfg}/{variant} ls+res")
    elif epi == "gelu":
        out = torch.zeros(M, N, dtype=op, device=DEV)
        hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, flags=hip.EP_BIAS | hip.EP_GELU, out_op=out, ldo_op=N)
        _check_tile(hip, cfg, variant)
        _close(out, F.gelu(lin), what=f"tile {cfg}/{variant} gelu", **loose)
    elif epi == "gamma_op":
        out = torch.zeros(M, N, dtype=op, device=DEV)
        hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, gamma=g, flags=hip.EP_BIAS | hip.EP_GAMMA, out_op=out, ldo_op=N)
        _check_tile(hip, cfg, variant)
        _close(out, lin * g.cpu(), what=f"tile {cfg}/{variant} gamma", **loose)
    elif epi == "relu_op_and_f32":
        of = torch.zeros(M, N, device=DEV)
        oo = torch.zeros(M, N, dtype=op, device=DEV)
        hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, flags=hip.EP_BIAS | hip.EP_RELU_OP, out_f32=of, ldo_f32=N, out_op=oo, ldo_op=N)
        _check_tile(hip, cfg, variant)
        _close(of, lin, 3e-4, what=f"tile {cfg}/{variant} f32")
        _close(oo, lin.clamp_min(0), what=f"tile {cfg}/{variant} relu op", **loose)
    else:
        out = torch.zeros(M, N, dtype=op, device=DEV)
        hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, out_op=out, ldo_op=N)
        _check_tile(hip, cfg, variant)
        _close(out, lin - b.cpu(), what=f"tile {cfg}/{variant} plain", **loose)


@pytest.mark.parametrize("cfg,variant", [(3, 8), (3, 4), (2, 8), (4, 8)])
def test_igemm_forced_tile_column_groups_long_k(hip, forced_tile, cfg, variant):
    """N = 2304 (9 column tiles of 256) and K = 4096: the weight panel exceeds the L2 model's budget so the launcher walks
    the tiles in column groups (group_n < tiles_n); also run with the group width forced to 2 and to 1."""
    op = _op(hip)
    M, N, K = 1100, 2304, 4096
    A = _rand(M, K, seed=211).to(op).to(DEV)
    W = _rand(N, K, scale=K ** -0.5, seed=212).to(op).to(DEV)
    b = _rand(N, seed=213).to(DEV)
    ref = A.float().cpu() @ W.float().cpu().T + b.cpu()
    forced_tile(cfg, variant)
    for group in (0, 2, 1):
        hip.debug_set_group(group)
        out = torch.full((M, N), float("nan"), device=DEV)
        hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W, bias=b, flags=hip.EP_BIAS, out_f32=out, ldo_f32=N)
        _check_tile(hip, cfg, variant)
        _close(out, ref, 5e-4, what=f"tile {cfg}/{variant} group {group}")


@pytest.mark.parametrize("cfg,variant", [(3, 8), (3, 4), (4, 8), (2, 8), (1, 8)])
def test_igemm_forced_tile_conv3x3_pad_residual(hip, forced_tile, cfg, variant):
    """3x3 implicit GEMM (9 taps x 2 k-tiles), stride 1, fp32 + residual and ReLU'd zero-bordered NHWC outputs."""
    op = _op(hip)
    B, C, H, W_ = 2, 128, 23, 31          # M = 1426: 6 tile rows of 256 with a ragged end
    x = _rand(B, C, H, W_, seed=221).to(op).float()
    w = (_rand(C, C, 3, 3, seed=222) * (9 * C) ** -0.5).to(op).float()
    b = _rand(C, seed=223)
    res = _rand(B * H * W_, C, seed=224)
    out_f = torch.zeros(B * H * W_, C, device=DEV)
    out_p = torch.zeros(B, H + 2, W_ + 2, C, dtype=op, device=DEV)
    forced_tile(cfg, variant)
    hip.igemm(M=B * H * W_, N=C, K=9 * C, A=_pad_nhwc(x, C, op).to(DEV), lda=C, W=_pack3(w, C, op).to(DEV), a_mode=hip.A_CONV3,
              conv=(H, W_, H + 2, W_ + 2, 1), bias=b.to(DEV), res=res.to(DEV), ldr=C, flags=hip.EP_BIAS | hip.EP_RESIDUAL | hip.EP_RELU_OP,
              out_f32=out_f, ldo_f32=C, out_op=out_p, ldo_op=C, map_op=hip.MAP_PAD, map_h=H, map_w=W_)
    _check_tile(hip, cfg, variant)
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, C) + res
    _close(out_f, ref, 3e-4, what=f"tile {cfg}/{variant} conv f32")
    _close(out_p[:, 1:-1, 1:-1].reshape(-1, C), ref.clamp_min(0), 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what="conv relu padded")
    border = out_p.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0


@pytest.mark.parametrize("cfg,variant", [(3, 8), (3, 4), (4, 8)])
def test_igemm_forced_tile_shuffle_and_swiglu(hip, forced_tile, cfg, variant):
    op = _op(hip)
    # ConvTranspose2d k = s = 2 as GEMM + pixel shuffle: N = 4 * 256 = 1024
    s_, C, Ci, B, H, W_ = 2, 256, 128, 2, 19, 23
    x = _rand(B, Ci, H, W_, seed=231).to(op).float()
    w = (_rand(Ci, C, s_, s_, seed=232) * Ci ** -0.5).to(op).float()
    b = _rand(C, seed=233)
    A = x.permute(0, 2, 3, 1).reshape(-1, Ci).to(op).contiguous()
    Wp = w.permute(2, 3, 1, 0).reshape(s_ * s_ * C, Ci).to(op).contiguous()
    out = torch.zeros(B, s_ * H + 2, s_ * W_ + 2, C, dtype=op, device=DEV)
    forced_tile(cfg, variant)
    hip.igemm(M=B * H * W_, N=s_ * s_ * C, K=Ci, A=A.to(DEV), lda=Ci, W=Wp.to(DEV), bias=b.repeat(s_ * s_).to(DEV), flags=hip.EP_BIAS,
              out_op=out, ldo_op=C, map_op=hip.MAP_SHUFFLE, map_h=H, map_w=W_, shuffle_s=s_, shuffle_c=C)
    _check_tile(hip, cfg, variant)
    ref = F.conv_transpose2d(x, w, b, stride=s_).permute(0, 2, 3, 1)
    _close(out[:, 1:-1, 1:-1], ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what=f"tile {cfg}/{variant} convT")
    # SwiGLU (x1 | x2 interleaved in 32-column groups)
    M, K, Hd = 900, 192, 512
    A2 = _rand(M, K, seed=234).to(op).to(DEV)
    w12 = (_rand(2 * Hd, K, seed=235) * K ** -0.5).to(op)
    b12 = _rand(2 * Hd, seed=236)
    idx = torch.arange(Hd).reshape(-1, 32)
    order = torch.stack([idx, idx + Hd], dim=1).reshape(-1)
    o2 = torch.zeros(M, Hd, dtype=op, device=DEV)
    hip.igemm(M=M, N=2 * Hd, K=K, A=A2, lda=K, W=w12[order].contiguous().to(DEV), bias=b12[order].contiguous().to(DEV),
              flags=hip.EP_BIAS | hip.EP_SWIGLU, out_op=o2, ldo_op=Hd)
    _check_tile(hip, cfg, variant)
    x12 = A2.float().cpu() @ w12.float().T + b12
    _close(o2, F.silu(x12[:, :Hd]) * x12[:, Hd:], 2e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what=f"tile {cfg}/{variant} swiglu")


@pytest.mark.parametrize("variant", [3, 5])
@pytest.mark.parametrize("B,N,heads", [(2, 1370, 2), (1, 64, 1), (1, 65, 3), (1, 1, 1), (3, 200, 6), (1, 129, 1), (1, 1409, 1), (1, 5330, 1)])
def test_attention_variants(hip, variant, B, N, heads):
    """Both attention kernels (5: the shipped mixed-stream kernel, 3: the round-1 4-wave kernel) on
    sequence lengths that exercise: one tile only, an odd tile count, a single valid key in the last tile, a
    query block with inactive waves, and the ViT-G 1022^2 length (N = 5330)."""
    op = _op(hip)
    D = heads * 64
    qkv = _rand(B * N, 3 * D, seed=33).to(op)
    qkv[:, :D] *= 0.125 * LOG2E
    qkv = qkv.to(op)
    out = torch.full((B * N, D), float("nan"), dtype=op, device=DEV)
    hip.debug_set_attention_variant(variant)
    try:
        hip.attention(qkv.to(DEV), out, B, N, heads)
    finally:
        hip.debug_set_attention_variant(5)
    t = qkv.float().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    p = ((t[0] @ t[1].transpose(-2, -1)) / LOG2E).softmax(-1)
    ref = (p @ t[2]).transpose(1, 2).reshape(B * N, D)
    _close(out, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 3e-3, what=f"attention variant {variant}")


@pytest.mark.parametrize("variant", [3, 5])
@pytest.mark.parametrize("spike_key", [250, 200, 5, 130])
def test_attention_variants_force_online_rescale(hip, variant, spike_key):
    """A dominating key in a late tile of either key half (tile 3 = odd half, tile 3 again via key 200, tile 0): the running
    max of that half must jump, rescale its O and l, and the merge of the two halves must weight them correctly."""
    op = _op(hip)
    N = 300
    qkv = _rand(N, 192, seed=34) * 0.3
    qkv[:, 64:128][spike_key] = qkv[:, :64][7] * 40.0
    qkv = qkv.to(op)
    out = torch.zeros(N, 64, dtype=op, device=DEV)
    hip.debug_set_attention_variant(variant)
    try:
        hip.attention(qkv.to(DEV), out, 1, N, 1)
    finally:
        hip.debug_set_attention_variant(5)
    t = qkv.float()
    p = ((t[:, :64] @ t[:, 64:128].T) / LOG2E).softmax(-1)
    _close(out, p @ t[:, 128:], 2e-3, rtol=1e-2 if op == torch.bfloat16 else 3e-3, what="attention rescale")


def test_attention_last_batch_does_not_read_past_the_buffer(hip):
    """The ping-pong kernel runs its copies two tiles ahead of the sequence end and relies on the buffer bounds check to
    zero-fill them: the qkv tensor here ends exactly at the end of an allocation-sized block of its own."""
    op = _op(hip)
    B, N, heads = 2, 77, 1
    qkv = (_rand(B * N, 192, seed=35)).to(op).to(DEV).clone()
    out = torch.zeros(B * N, 64, dtype=op, device=DEV)
    hip.attention(qkv, out, B, N, heads)
    torch.cuda.synchronize()
    t = qkv.float().cpu().reshape(B, N, 3, 1, 64).permute(2, 0, 3, 1, 4)
    p = ((t[0] @ t[1].transpose(-2, -1)) / LOG2E).softmax(-1)
    ref = (p @ t[2]).transpose(1, 2).reshape(B * N, 64)
    _close(out, ref, 2e-3, rtol=1e-2 if op == torch.bfloat16 else 3e-3, what="attention tail batch")


# =====================================================================================================================
# Split-precision operand stores (split_seg): [hi | lo] column segments, contracted as (hi, lo, hi) (a_dup_seg) with weights [w_hi | w_hi | w_lo]
# =====================================================================================================================
def _triple_w(w, op):
    hi = w.to(op)
    lo = (w - hi.float()).to(op)
    return torch.cat([hi, hi, lo], dim=-1).contiguous()


def _check_split(buf, ref, C, seg, op, what, exact=True):
    """buf [..., 2*seg] op-typed; ref [..., C] fp32: hi = round(ref), lo = round(ref - hi), pads zero.
    exact=False: ref was computed on the host (differs from the device's fp32 value by ulps), so only hi + lo ~ ref is checked."""
    b = buf.float().cpu()
    assert b.shape[-1] == 2 * seg
    hi, lo = b[..., :C], b[..., seg:seg + C]
    want_hi = ref.to(op).float()
    want_lo = (ref - want_hi).to(op).float()
    tol = 1e-2 if op == torch.bfloat16 else 2e-3
    _close(hi, ref, 2e-3, rtol=tol, what=what + " hi")
    resid = (hi + lo - ref).abs().max() / ref.abs().max()
    lim = (3e-5 if op == torch.bfloat16 else 1e-6) if exact else (1e-4 if op == torch.bfloat16 else 3e-6)
    assert float(resid) < lim, f"{what}: hi + lo misses the fp32 value by {float(resid):.2e} (relative)"
    if exact:
        frac_exact = float(((hi == want_hi) & (lo == want_lo)).float().mean())
        assert frac_exact > 0.99, f"{what}: only {frac_exact:.3f} of the (hi, lo) pairs are the roundings of the fp32 value"
    for a, z in ((C, seg), (seg + C, 2 * seg)):
        if z > a:
            assert float(b[..., a:z].abs().max()) == 0.0, what + ": pad columns written"


def test_split_store_bilinear_layernorm_and_gemm_accuracy(hip):
    op = _op(hip)
    B, C, H, W_ = 2, 48, 9, 11
    seg = 64
    x = _rand(B * H * W_, C, seed=301) * 3
    # identity resample (hi == ho, wi == wo) = a split cast
    buf = torch.zeros(B * H * W_, 2 * seg, dtype=op, device=DEV)
    hip.bilinear(x.to(DEV), C, B, H, W_, H, W_, C, out_op=buf, ld_op=2 * seg, map_op=hip.MAP_PLAIN, split_seg=seg)
    _check_split(buf, x, C, seg, op, "bilinear split")
    # GEMM over the triple K against triple weights ~ fp32 product
    N = 96
    w = _rand(N, C, seed=302) * C ** -0.5
    wp = torch.zeros(N, seg)
    wp[:, :C] = w
    out = torch.zeros(B * H * W_, N, device=DEV)
    hip.igemm(M=B * H * W_, N=N, K=3 * seg, A=buf, lda=2 * seg, a_dup_seg=seg, W=_triple_w(wp, op).to(DEV), out_f32=out, ldo_f32=N)
    ref = x @ w.T
    err = float((out.cpu() - ref).abs().max() / ref.abs().max())
    single = float((x.to(op).float() @ w.to(op).float().T - ref).abs().max() / ref.abs().max())
    print(f"split GEMM max rel err {err:.2e} (single-precision operands: {single:.2e})")
    assert err < (2e-4 if op == torch.bfloat16 else 3e-6) and err < single / 20
    # padded NHWC + ReLU through the bilinear kernels (per-pixel and LDS-tiled)
    for Cw, hi_, wi_, ho_, wo_ in ((64, 5, 7, 10, 14), (128, 19, 19, 37, 37)):
        xin = _rand(B, Cw, hi_, wi_, seed=303)
        o = torch.zeros(B, ho_ + 2, wo_ + 2, 2 * Cw, dtype=op, device=DEV)
        of = torch.zeros(B * ho_ * wo_, Cw, device=DEV)
        hip.bilinear(xin.permute(0, 2, 3, 1).reshape(-1, Cw).contiguous().to(DEV), Cw, B, hi_, wi_, ho_, wo_, Cw, out_f32=of, ld_f32=Cw,
                     out_op=o, ld_op=2 * Cw, map_op=hip.MAP_PAD, relu=True, split_seg=Cw)
        r = F.interpolate(xin, size=(ho_, wo_), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        _close(of.reshape(B, ho_, wo_, Cw), r, 2e-5, rtol=1e-5, what="bilinear f32")
        _check_split(o[:, 1:-1, 1:-1], of.cpu().reshape(B, ho_, wo_, Cw).clamp_min(0), Cw, Cw, op, f"bilinear pad split C={Cw}")
        border = o.clone()
        border[:, 1:-1, 1:-1] = 0
        assert float(border.abs().max()) == 0.0
    # LayerNorm
    D = 96
    xs = _rand(40, D, seed=304)
    wln, bln = _rand(D, seed=305), _rand(D, seed=306)
    o = torch.zeros(40, 2 * 128, dtype=op, device=DEV)
    of = torch.zeros(40, D, device=DEV)
    hip.layernorm(xs.to(DEV), D, 40, D, wln.to(DEV), bln.to(DEV), 1e-6, out_op=o, ld_op=2 * 128, out_f32=of, ld_f32=D, split_seg=128)
    _close(of, F.layer_norm(xs, (D,), wln, bln, 1e-6), 2e-5, rtol=1e-5, what="ln f32")
    _check_split(o, of.cpu(), D, 128, op, "layernorm split")


@pytest.mark.parametrize("cfg", [-1, 3, 4, 1])
def test_igemm_split_output_plain_pad_shuffle(hip, forced_tile, cfg):
    op = _op(hip)
    if cfg >= 0:
        forced_tile(cfg, 4)
    # PLAIN + residual (fp32 path) and PLAIN op-only (8-column path)
    M, N, K, seg = 700, 192, 128, 192
    A = _rand(M, K, seed=311).to(op).to(DEV)
    Wt = (_rand(N, K, seed=312) * K ** -0.5).to(op).to(DEV)
    b = _rand(N, seed=313).to(DEV)
    res = _rand(M, N, seed=314)
    lin = A.float().cpu() @ Wt.float().cpu().T + b.cpu()
    o = torch.zeros(M, 2 * seg, dtype=op, device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=Wt, bias=b, res=res.to(DEV), ldr=N, flags=hip.EP_BIAS | hip.EP_RESIDUAL, out_op=o, ldo_op=2 * seg, split_seg=seg)
    _check_split(o, lin + res, N, seg, op, f"igemm split plain+res cfg {cfg}", exact=False)
    o.zero_()
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=Wt, bias=b, flags=hip.EP_BIAS | hip.EP_RELU_OP, out_op=o, ldo_op=2 * seg, split_seg=seg)
    _check_split(o, lin.clamp_min(0), N, seg, op, f"igemm split plain relu cfg {cfg}", exact=False)
    # PAD (conv3x3 reading a split input, writing a split output): the whole split-precision conv
    B, C, H, W_ = 2, 64, 13, 17
    x = _rand(B, C, H, W_, seed=315)
    w = _rand(C, C, 3, 3, seed=316) * (9 * C) ** -0.5
    xin = torch.zeros(B, H + 2, W_ + 2, 2 * C, dtype=op)
    xh = x.permute(0, 2, 3, 1).to(op)
    xin[:, 1:-1, 1:-1, :C] = xh
    xin[:, 1:-1, 1:-1, C:] = (x.permute(0, 2, 3, 1) - xh.float()).to(op)
    wp = _triple_w(w.permute(0, 2, 3, 1).contiguous(), op).reshape(C, 27 * C)
    o = torch.zeros(B, H + 2, W_ + 2, 2 * C, dtype=op, device=DEV)
    of = torch.zeros(B * H * W_, C, device=DEV)
    hip.igemm(M=B * H * W_, N=C, K=27 * C, A=xin.to(DEV), lda=2 * C, a_dup_seg=C, W=wp.to(DEV), a_mode=hip.A_CONV3, conv=(H, W_, H + 2, W_ + 2, 1),
              flags=hip.EP_RELU_OP, out_f32=of, ldo_f32=C, out_op=o, ldo_op=2 * C, map_op=hip.MAP_PAD, map_h=H, map_w=W_, split_seg=C)
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1)
    err = float((of.cpu().reshape(B, H, W_, C) - ref).abs().max() / ref.abs().max())
    assert err < (2e-4 if op == torch.bfloat16 else 3e-6), f"split conv3x3 error {err:.2e}"
    _check_split(o[:, 1:-1, 1:-1], of.cpu().reshape(B, H, W_, C).clamp_min(0), C, C, op, f"igemm split pad cfg {cfg}")
    border = o.clone()
    border[:, 1:-1, 1:-1] = 0
    assert float(border.abs().max()) == 0.0
    # SHUFFLE (ConvTranspose k = s = 2)
    s_, Co, Ci = 2, 48, 64
    xs = _rand(B, Ci, 5, 6, seed=317).to(op).float()
    wt = (_rand(Ci, Co, s_, s_, seed=318) * Ci ** -0.5).to(op).float()
    A2 = xs.permute(0, 2, 3, 1).reshape(-1, Ci).to(op).contiguous()
    Wp = wt.permute(2, 3, 1, 0).reshape(s_ * s_ * Co, Ci).to(op).contiguous()
    segc = 64
    o = torch.zeros(B, 12, 14, 2 * segc, dtype=op, device=DEV)
    hip.igemm(M=B * 30, N=s_ * s_ * Co, K=Ci, A=A2.to(DEV), lda=Ci, W=Wp.to(DEV), out_op=o, ldo_op=2 * segc, map_op=hip.MAP_SHUFFLE,
              map_h=5, map_w=6, shuffle_s=s_, shuffle_c=Co, split_seg=segc)
    ref = F.conv_transpose2d(xs, wt, stride=s_).permute(0, 2, 3, 1)
    _check_split(o[:, 1:-1, 1:-1], ref, Co, segc, op, f"igemm split shuffle cfg {cfg}", exact=False)


# =====================================================================================================================
# Fused DPT tail: bilinear resize + 3x3 conv (-> 32) + ReLU + 1x1 (-> 1) + activation in one kernel (ada_dpt_tail_fwd)
# =====================================================================================================================
@pytest.mark.parametrize("B,C,hi,wi,ho,wo", [(2, 64, 9, 11, 16, 19), (1, 128, 20, 30, 35, 52), (2, 128, 17, 17, 30, 30), (1, 64, 40, 37, 70, 65),
                                             (1, 128, 8, 40, 14, 41), (3, 128, 96, 120, 168, 210), (5, 64, 64, 72, 112, 126)])
@pytest.mark.parametrize("act", ["sigmoid", "relu", "none"])
def test_dpt_tail_fused(hip, B, C, hi, wi, ho, wo, act):
    op = _op(hip)
    x = _rand(B, C, hi, wi, seed=501)
    w = (_rand(32, C, 3, 3, seed=502) * (9 * C) ** -0.5).to(op).float()
    b = _rand(32, seed=503)
    tw, tb = _rand(32, seed=504), 0.2
    code = {"sigmoid": hip.ACT_SIGMOID, "relu": hip.ACT_RELU, "none": hip.ACT_NONE}[act]
    xin = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
    wp = _pack3(w, C, op).to(DEV)
    out = torch.full((B, ho, wo), float("nan"), device=DEV)
    hip.dpt_tail(xin, C, B, hi, wi, ho, wo, C, wp, b.to(DEV), tw.to(DEV), tb, code, out)
    up = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True).to(op).float()
    d = (F.relu(F.conv2d(up, w, b, padding=1)) * tw.view(1, -1, 1, 1)).sum(1) + tb
    ref = {"sigmoid": torch.sigmoid, "relu": F.relu, "none": lambda t: t}[act](d)
    # atol: one operand-ulp flip of an interpolated value (fp16: 2^-11 relative) times a weight, out of 9 C products; the 100 k-pixel cases
    # reach 6e-4 on a handful of pixels against outputs of magnitude 10
    _close(out, ref, 2e-3 if op == torch.bfloat16 else 8e-4, rtol=1e-2 if op == torch.bfloat16 else 1e-3, what=f"fused tail {act}")
    # and against the two-launch path of the same library (resize kernel -> padded operand map -> tail GEMM)
    fin = torch.zeros(B, ho + 2, wo + 2, C, dtype=op, device=DEV)
    hip.bilinear(xin, C, B, hi, wi, ho, wo, C, out_op=fin, ld_op=C, map_op=hip.MAP_PAD)
    out2 = torch.zeros(B, 1, ho, wo, device=DEV)
    hip.igemm(M=B * ho * wo, N=32, K=9 * C, A=fin, lda=C, W=wp, a_mode=hip.A_CONV3, conv=(ho, wo, ho + 2, wo + 2, 1), bias=b.to(DEV),
              flags=hip.EP_BIAS | hip.EP_TAIL, out_f32=out2, ldo_f32=1, tail_w=tw.to(DEV), tail_b=tb, tail_act=code)
    # same arithmetic, but the interpolated operand is rounded from differently contracted fp32 expressions: a few fp16 ulps flip
    wide = 8.0 if _op(hip) == torch.bfloat16 else 1.0      # (the two paths round the interpolated map to the operand type at different points)
    _close(out, out2[:, 0], 6e-4 * wide, rtol=1e-3 * wide, what="fused tail vs two-launch path")


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_dpt_tail_fused_many_tiles_per_workgroup(hip, seed):
    """More tiles than CUs and two channel passes: every persistent workgroup walks several (tile, pass) units through its two halo buffers
    while its producer waves run a unit ahead of its consumer waves.  Repeated with fresh data: a hand-over race shows up as O(0.1) errors."""
    op = _op(hip)
    B, C, hi, wi, ho, wo = 4, 128, 96, 120, 168, 210
    x = _rand(B, C, hi, wi, seed=700 + seed)
    x[:, :64] += 3.0          # the two passes and the images carry different levels: stale data from another unit cannot hide
    x[1:] -= 5.0
    w = (_rand(32, C, 3, 3, seed=710 + seed) * (9 * C) ** -0.5).to(op).float()
    b, tw = _rand(32, seed=720 + seed), _rand(32, seed=730 + seed)
    xin = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
    wp = _pack3(w, C, op).to(DEV)
    up = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True).to(op).float()
    ref = (F.relu(F.conv2d(up, w, b, padding=1)) * tw.view(1, -1, 1, 1)).sum(1) + 0.1
    for rep in range(3):
        out = torch.full((B, ho, wo), float("nan"), device=DEV)
        hip.dpt_tail(xin, C, B, hi, wi, ho, wo, C, wp, b.to(DEV), tw.to(DEV), 0.1, hip.ACT_NONE, out)
        _close(out, ref, 8e-3 if op == torch.bfloat16 else 3e-3, rtol=1e-2 if op == torch.bfloat16 else 2e-3, what=f"fused tail, many tiles (rep {rep})")


def test_dpt_tail_fused_stress_full_size_50_seeds(hip):
    """The model's own tail shape (296 x 296 x 128 -> 518 x 518, here 8 images: ~9400 tiles, 36 per persistent workgroup, two channel passes),
    50 fresh inputs, every output element compared with the two-launch path of the same library (resize kernel -> padded operand map -> tail
    GEMM: different kernels, same arithmetic up to a few operand ulps).  The round-3 failure -- a source row copied while its fetch was still
    in flight (profiles/r04_a_tail_inflight_register_root_cause.txt) -- showed as ~300 of 105 k outputs off by up to 0.13, timing dependent:
    a concurrent stream keeps the memory system busy on every other seed so that fetch latencies vary between runs."""
    op = _op(hip)
    B, C, hi, wi, ho, wo = 8, 128, 296, 296, 518, 518
    g = torch.Generator(device=DEV)
    g.manual_seed(4242)
    w = (_rand(32, C, 3, 3, seed=811) * (9 * C) ** -0.5).to(op).float()
    wp = _pack3(w, C, op).to(DEV)
    b, tw = _rand(32, seed=812).to(DEV), _rand(32, seed=813).to(DEV)
    fin = torch.zeros(B, ho + 2, wo + 2, C, dtype=op, device=DEV)
    out2 = torch.zeros(B, 1, ho, wo, device=DEV)
    side = torch.cuda.Stream()
    noise_a = torch.randn(1 << 26, device=DEV)
    worst = 0.0
    for seed in range(50):
        xin = torch.randn(B * hi * wi, C, device=DEV, generator=g)
        xin[:, :64] += 3.0 * ((seed % 3) - 1)      # the two channel passes carry different levels: stale data from another unit cannot hide
        out = torch.full((B, ho, wo), float("nan"), device=DEV)
        if seed & 1:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    noise_a.mul_(1.0001)
        hip.dpt_tail(xin, C, B, hi, wi, ho, wo, C, wp, b, tw, 0.1, hip.ACT_NONE, out)
        hip.bilinear(xin, C, B, hi, wi, ho, wo, C, out_op=fin, ld_op=C, map_op=hip.MAP_PAD)
        hip.igemm(M=B * ho * wo, N=32, K=9 * C, A=fin, lda=C, W=wp, a_mode=hip.A_CONV3, conv=(ho, wo, ho + 2, wo + 2, 1), bias=b,
                  flags=hip.EP_BIAS | hip.EP_TAIL, out_f32=out2, ldo_f32=1, tail_w=tw, tail_b=0.1, tail_act=hip.ACT_NONE)
        torch.cuda.current_stream().wait_stream(side)
        diff = (out - out2[:, 0]).abs()
        assert torch.isfinite(out).all()
        err = float(diff.max())
        worst = max(worst, err)
        nbad = int((diff > 4e-3 + 1e-3 * out2[:, 0].abs()).sum())     # a few operand-ulp flips reach ~5e-4; the failure mode was 1e-1
        assert nbad == 0, f"seed {seed}: {nbad} of {out.numel()} outputs differ from the two-launch path by more than 4e-3 + 1e-3 |ref| (max {err:.3e})"
    print(f"fused tail stress: 50 seeds x {B * ho * wo} outputs, worst |fused - two-launch| = {worst:.2e}")


def test_dpt_tail_fused_rejects_what_it_cannot_hold(hip):
    """The producers keep 8 source rows per 10-row halo tile in registers and the weights of at most two 64-channel passes in LDS:
    a vertical scale below 1.5 or more than 128 channels is ADA_EUNSUPPORTED (the engine then takes the two-launch tail), never a wrong map."""
    op = _op(hip)
    for C, hi, ho in [(128, 8, 8), (64, 20, 25), (192, 16, 28)]:
        xin = torch.zeros(hi * 12, C, device=DEV)
        wp = torch.zeros(32, 9 * C, dtype=op, device=DEV)
        out = torch.zeros(1, ho, 21, device=DEV)
        with pytest.raises(hip.HipExtError):
            hip.dpt_tail(xin, C, 1, hi, 12, ho, 21, C, wp, torch.zeros(32, device=DEV), torch.zeros(32, device=DEV), 0.0, hip.ACT_NONE, out)


@pytest.mark.parametrize("ph,pw,dim", [(19, 23, 384), (73, 73, 64), (16, 16, 128), (37, 50, 32), (9, 11, 1024)])
def test_pos_embed_bicubic_resize_matches_aten(hip, ph, pw, dim):
    """ada_pos_embed_resize against F.interpolate(mode='bicubic', scale_factor=((ph+0.1)/37, (pw+0.1)/37)) -- reference DA2/dinov2.py:219-225."""
    sq = 37
    pos = _rand(1 + sq * sq, dim, seed=601)
    out = torch.full((1 + ph * pw, dim), float("nan"), device=DEV)
    hip.pos_embed_resize(pos.to(DEV), sq, dim, ph, pw, (ph + 0.1) / sq, (pw + 0.1) / sq, out)
    grid = pos[1:].reshape(1, sq, sq, dim).permute(0, 3, 1, 2)
    ref = F.interpolate(grid, scale_factor=((ph + 0.1) / sq, (pw + 0.1) / sq), mode="bicubic", antialias=False)
    assert ref.shape[-2:] == (ph, pw)
    ref = torch.cat([pos[:1], ref.permute(0, 2, 3, 1).reshape(ph * pw, dim)], 0)
    _close(out, ref, 3e-6, rtol=1e-5, what="bicubic pos embed")


def test_operand_store_saturates_and_is_counted(hip):
    """fp32 -> fp16 operand stores clamp to +-65504 instead of overflowing to inf (csrc/ada_common.h to_op), and the saturation probe
    (ada_debug_count_saturated) finds exactly the clamped elements.  bf16 builds have fp32's range: nothing clamps, nothing is counted."""
    op = _op(hip)
    M, N, K = 300, 128, 64
    A = torch.ones(M, K).to(op).to(DEV)
    W = torch.zeros(N, K)
    W[:40] = 2000.0        # 64 * 2000 = 128000 > 65504
    W[40:80] = -2000.0
    W[80:] = 3.0
    out = torch.zeros(M, N, dtype=op, device=DEV)
    hip.igemm(M=M, N=N, K=K, A=A, lda=K, W=W.to(op).to(DEV), flags=0, out_op=out, ldo_op=N)
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    hip.count_saturated(out, counter)
    got = out.float().cpu()
    assert torch.isfinite(got).all() or op == torch.bfloat16
    assert torch.equal(got[:, 80:], torch.full((M, 48), 192.0))
    if op == torch.float16:
        assert torch.equal(got[:, :40], torch.full((M, 40), 65504.0)) and torch.equal(got[:, 40:80], torch.full((M, 40), -65504.0))
        assert int(counter.item()) == M * 80
    else:
        assert int(counter.item()) == 0
    # odd length / tail elements of the sweep
    counter.zero_()
    flat = out.reshape(-1)[: M * N - 3]
    hip.count_saturated(flat, counter)
    assert int(counter.item()) == (M * 80 if op == torch.float16 else 0)
