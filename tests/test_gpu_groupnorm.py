"""Fused GroupNorm(+SiLU) HIP kernels (include/gip_nn.h) against torch's fp32 GroupNorm + SiLU, forward and dL/dx,
at every channel width / resolution the SD1.5 U-Net, ControlNet and VAE encoder use.  Tolerance: fp16 storage —
outputs are rounded to half once, so |err| <= 2^-10 relative to the output scale plus statistics rounding (2e-3 abs on
unit-variance data)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(8, 320, 64, 64), (8, 640, 32, 32), (8, 1280, 16, 16), (8, 1280, 8, 8), (8, 2560, 8, 8), (8, 1920, 32, 32),
          (8, 960, 64, 64), (4, 128, 512, 512), (4, 256, 256, 256), (4, 512, 64, 64), (1, 32, 4, 4), (2, 64, 7, 5),
          # round 6: samples of <= 256 rows take the one-launch kernel (gn_small_fwd_kernel): groups that straddle the 8-channel
          # chunks (1920 / 32 = 60, 640 / 32 = 20), the widest concatenations, the batch of a 1-view shard
          (12, 1920, 16, 16), (3, 2560, 16, 16), (3, 640, 16, 16), (12, 2560, 8, 8), (3, 1280, 8, 8)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("act", [True, False])
def test_fused_groupnorm_matches_torch(shape, act):
    from gaussianip_amd.guidance.fused import GroupNormAct
    N, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(C + H)
    x32 = torch.randn(shape, device="cuda", generator=g) * 1.7 + 0.3
    m = GroupNormAct(32, C, eps=1e-5, act=act).cuda().half()
    with torch.no_grad():
        m.weight.copy_(torch.randn(C, device="cuda", generator=g) * 0.5 + 1.0)
        m.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.2)
    m.requires_grad_(False)
    x = x32.half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = m(x)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.float16
    dy32 = torch.randn(shape, device="cuda", generator=g)
    dy = dy32.half().contiguous(memory_format=torch.channels_last)
    (dx,) = torch.autograd.grad(y, x, dy)

    xr = x.detach().float().contiguous().requires_grad_(True)
    yr = F.group_norm(xr, 32, m.weight.float(), m.bias.float(), 1e-5)
    if act:
        yr = F.silu(yr)
    (dxr,) = torch.autograd.grad(yr, xr, dy.float().contiguous())
    assert float((y.float() - yr).abs().max()) < 4e-3 * max(1.0, float(yr.abs().max()))
    assert float((dx.float() - dxr).abs().max()) < 4e-3 * max(1.0, float(dxr.abs().max()))
    # same call twice: bitwise identical (fixed-order reductions, no float atomics)
    y2 = m(x)
    assert torch.equal(y2, y)


def test_fused_path_refuses_silently_falling_back_on_gpu():
    """The fused op must be the one that runs for fp16 NHWC inputs on the GPU."""
    from gaussianip_amd.guidance import fused
    calls = []
    orig = fused._FusedGN.apply
    try:
        fused._FusedGN.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
        m = fused.GroupNormAct(32, 320, act=True).cuda().half().requires_grad_(False)
        x = torch.randn(2, 320, 16, 16, device="cuda").half().contiguous(memory_format=torch.channels_last)
        m(x)
    finally:
        fused._FusedGN.apply = orig
    assert calls


def test_groupnorm_addend_and_pointwise_companions():
    """addend path (conv bias + time embedding folded into the norm), add_bias_residual and geglu vs plain torch."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(3)
    N, C, H, W = 6, 640, 32, 32
    x = (torch.randn(N, C, H, W, device="cuda", generator=g)).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    big = torch.randn(N, 3 * C, device="cuda", generator=g).half()
    for addend in (big[:, C:2 * C], torch.randn(C, device="cuda", generator=g).half()):   # strided [N, C] view, shared [C]
        m = fused.GroupNormAct(32, C, act=True).cuda().half().requires_grad_(False)
        y = m(x, addend)
        dy = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
        (dx,) = torch.autograd.grad(y, x, dy)
        xr = x.detach().float().contiguous().requires_grad_(True)
        ad = addend.float().reshape(-1 if addend.dim() == 2 else 1, C, 1, 1)
        yr = F.silu(F.group_norm(xr + ad, 32, m.weight.float(), m.bias.float(), 1e-5))
        (dxr,) = torch.autograd.grad(yr, xr, dy.float().contiguous())
        assert float((y.float() - yr).abs().max()) < 4e-3 * max(1.0, float(yr.abs().max()))
        assert float((dx.float() - dxr).abs().max()) < 4e-3 * max(1.0, float(dxr.abs().max()))
    a = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bias = torch.randn(C, device="cuda", generator=g).half()
    out = fused.add_bias_residual(a, b, bias)
    ref = (a.float() + b.float() + bias.float().reshape(1, C, 1, 1))
    assert float((out.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    ga, gb = torch.autograd.grad(out, [a, b], torch.ones_like(out))
    assert torch.equal(ga, torch.ones_like(a)) and torch.equal(gb, ga)
    with torch.no_grad():
        z = torch.randn(12, 1024, 2 * 1280, device="cuda", generator=g).half()
        o = fused.geglu(z)
        v, gate = z.float().chunk(2, dim=-1)
        r = v * F.gelu(gate)
        assert o.shape == (12, 1024, 1280) and float((o.float() - r).abs().max()) <= 2e-3 * float(r.abs().max())


def test_resblock_fused_equals_unfused():
    """The fused ResnetBlock2D path (addend + one-pass residual) against the module's own plain-torch path in fp32."""
    from gaussianip_amd.guidance import networks as nw
    torch.manual_seed(0)
    for cin, cout in ((320, 320), (640, 320)):
        blk = nw.init_for_benchmark(nw.ResBlock(cin, cout)).cuda()
        with torch.no_grad():
            for p in blk.parameters():
                if p.ndim == 1:
                    p.add_(torch.randn_like(p) * 0.1)
        ref = blk.float()
        x = torch.randn(3, cin, 32, 32, device="cuda")
        temb = torch.randn(3, 1280, device="cuda")
        want = ref(x, temb)
        import copy
        h = copy.deepcopy(ref).half().requires_grad_(False).to(memory_format=torch.channels_last)
        got = h(x.half().contiguous(memory_format=torch.channels_last), temb.half())
        assert float((got.float() - want).abs().max()) < 2e-2 * float(want.abs().max())


def test_mfma_linear_residual_and_geglu_epilogues():
    """gip_linear_f16 (conv3x3.hip, TAPS = 1) against fp32 F.linear: bias + residual epilogue, GEGLU epilogue, ragged M."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for M, K, N in ((1000, 320, 320), (4096, 640, 1280), (130, 64, 72)):
            x = torch.randn(M, K, device="cuda", generator=g).half()
            w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
            b = torch.randn(N, device="cuda", generator=g).half()
            r = torch.randn(M, N, device="cuda", generator=g).half()
            assert fused.linear_supported(x, w)
            ref = F.linear(x.float(), w.float(), b.float()) + r.float()
            got = fused.linear(x, w, b, r)
            assert float((got.float() - ref).abs().max()) <= 1.5e-3 * float(ref.abs().max())
            got2 = fused.linear(x.view(1, M, K), w)            # leading dims, no bias / residual
            assert got2.shape == (1, M, N)
            assert float((got2.float() - F.linear(x.float(), w.float())).abs().max()) <= 1.5e-3 * float(ref.abs().max())
        for M, K, D in ((777, 320, 1280), (2048, 640, 2560)):
            x = torch.randn(M, K, device="cuda", generator=g).half()
            w = (torch.randn(2 * D, K, device="cuda", generator=g) / K ** 0.5).half()
            b = torch.randn(2 * D, device="cuda", generator=g).half()
            v, gate = F.linear(x.float(), w.float(), b.float()).chunk(2, -1)
            ref = v * F.gelu(gate)
            got = fused.linear(x, w, b, None, True)
            assert got.shape == (M, D) and float((got.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(4096 * 3, 320), (1024 * 2 + 5, 640), (777, 1280), (64, 512), (33, 2048), (19, 8), (1, 136)])
def test_layernorm_rows_match_torch_fp32(rows, C):
    """csrc/groupnorm.hip layernorm_kernel (BasicTransformerBlock.norm1/2/3) against F.layer_norm in fp32 on the same
    half inputs; tolerance = one half-precision rounding of the output."""
    import torch.nn.functional as F
    from gaussianip_amd.guidance import fused
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(rows + C)
    x = (torch.randn(rows, C, device=dev, generator=g) * 2.0 + 0.7).half()
    ln = fused.LayerNorm(C).to(dev).half().requires_grad_(False)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, device=dev, generator=g) * 0.5 + 1.0)
        ln.bias.copy_(torch.randn(C, device=dev, generator=g) * 0.3)
        y = ln(x.view(1, rows, C)).view(rows, C)
        with fused.disabled():
            assert ln(x).dtype == torch.float16                           # the library path still works
    ref = F.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps)
    err = (y.float() - ref).abs()
    assert float(err.max()) <= 1e-3 * float(ref.abs().max()) + 1e-3, float(err.max())
    # training-mode inputs take the autograd path
    xg = x.clone().requires_grad_(True)
    ln(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


# ---- round 3: the statistics pass taken by the producing kernel's epilogue (gip_conv3x3_stats_nhwc_f16 /
# gip_linear_stats_f16 -> gip_gn_silu_forward_stats) ----
# N, Cin, Cout, H, W; the last three run split-K (statistics from the reduce kernel)
STAT_SHAPES = [(4, 128, 128, 64, 64), (2, 64, 320, 64, 64), (3, 128, 256, 32, 48), (2, 192, 640, 16, 24), (8, 1280, 1280, 16, 16)]


@pytest.mark.parametrize("shape", STAT_SHAPES)
@pytest.mark.parametrize("residual", [False, True])
def test_conv_epilogue_statistics_and_the_groupnorm_that_uses_them(shape, residual, monkeypatch):
    from gaussianip_amd.guidance import fused
    monkeypatch.setattr(fused, "_MIN_CONV_TILES", 0)
    monkeypatch.setattr(fused, "stats_wanted", lambda N, H, W, c: (H * W) % 128 == 0 and c % 8 == 0)   # also for small tile counts
    N, ci, co, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(ci + co + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, ci, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(co, ci, 3, 3, device="cuda", generator=g) / (3 * ci ** 0.5)).half().contiguous(**cl)
    b = torch.randn(co, device="cuda", generator=g).half()
    r = torch.randn(N, co, H, W, device="cuda", generator=g).half().contiguous(**cl) if residual else None
    out = fused.conv3x3(x, w, b, r, gn_next=True)
    st = fused.producer_stats(out)
    R = 128 if (H * W) % 128 == 0 else 64
    assert st is not None and st.shape == (N * H * W // R, co, 2), "the epilogue statistics are missing"
    # 1. they are the per-(R-pixel block, channel) sums of the tensor the kernel wrote
    rows = out.permute(0, 2, 3, 1).reshape(-1, R, co).double()
    want = torch.stack([rows.sum(1), (rows * rows).sum(1)], dim=-1)
    assert float((st.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # 2. the same output as the convolution without statistics (which may run split-K at these tile counts: another
    #    summation order, so one half-precision ulp of slack)
    plain = fused.conv3x3(x, w, b, r)
    assert float((out.float() - plain.float()).abs().max()) <= 2e-3 * max(1.0, float(plain.float().abs().max()))
    # 3. GroupNorm fed by them == GroupNorm with its own statistics pass (mean / rstd differ by summation order only)
    gn = fused.GroupNormAct(32, co, eps=1e-5, act=True).cuda().half().requires_grad_(False)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(co, device="cuda", generator=g) * 0.5 + 1.0)
        gn.bias.copy_(torch.randn(co, device="cuda", generator=g) * 0.2)
    addend = (torch.randn(N, co, device="cuda", generator=g) * 0.5).half()
    for ad in (None, addend, addend[0]):
        seen = []
        orig = fused._FusedGN.apply
        monkeypatch.setattr(fused._FusedGN, "apply", staticmethod(lambda *a: (seen.append(a[-1]), orig(*a))[1]))
        y_fused = gn(out, ad)
        monkeypatch.setattr(fused._FusedGN, "apply", orig)
        assert seen and seen[0] is st, "the GroupNorm did not take the producer's statistics"
        y_plain = gn(out.clone(memory_format=torch.channels_last), ad)      # a copy carries no statistics
        assert float((y_fused.float() - y_plain.float()).abs().max()) <= 2e-3 * max(1.0, float(y_plain.float().abs().max()))
        xa = out.float() if ad is None else out.float() + ad.float().reshape(-1 if ad.dim() == 2 else 1, co, 1, 1)
        ref = F.silu(F.group_norm(xa, 32, gn.weight.float(), gn.bias.float(), 1e-5))
        assert float((y_fused.float() - ref).abs().max()) < 4e-3 * max(1.0, float(ref.abs().max()))


def test_linear_epilogue_statistics_feed_the_next_groupnorm():
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, W, C = 2, 32, 32, 320
    t = torch.randn(B, H * W, C, device="cuda", generator=g).half()
    wt = (torch.randn(C, C, device="cuda", generator=g) / C ** 0.5).half()
    bias = torch.randn(C, device="cuda", generator=g).half()
    res = torch.randn(B, H * W, C, device="cuda", generator=g).half()
    holder = []
    with torch.no_grad():
        o = fused.linear(t, wt, bias, res, stats=holder)
    assert holder, "gip_linear_stats_f16 did not run"
    assert torch.equal(o, fused.linear(t, wt, bias, res))
    x = fused.attach_stats(o.reshape(B, H, W, C).permute(0, 3, 1, 2), holder[0])
    assert fused.producer_stats(x) is holder[0]
    gn = fused.GroupNormAct(32, C, act=True).cuda().half().requires_grad_(False)
    ref = F.silu(F.group_norm(x.float(), 32, gn.weight.float(), gn.bias.float(), 1e-5))
    assert float((gn(x).float() - ref).abs().max()) < 4e-3 * max(1.0, float(ref.abs().max()))


# ---- the U-Net decoder's skip concatenation: skip + ControlNet residual, torch.cat and norm1's statistics in one pass
# (gip_cat2_stats_f16) ----
@pytest.mark.parametrize("shape", [(8, 1280, 1280, 16, 16), (8, 1280, 640, 32, 32), (8, 640, 320, 64, 64), (2, 320, 320, 64, 64),
                                   (8, 1280, 1280, 8, 8), (1, 64, 128, 8, 16), (1, 64, 128, 5, 7), (3, 128, 64, 3, 1)])     # the last two: ragged row counts
@pytest.mark.parametrize("with_residual", [True, False])
def test_skip_concatenation_with_residual_and_statistics(shape, with_residual):
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    N, Ca, Cb, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(11)
    mk = lambda c: torch.randn(N, c, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)  # noqa: E731
    h, s, r = mk(Ca), mk(Cb), (mk(Cb) if with_residual else None)
    before = _lib.call_counts.get("gip_cat2_stats_f16", 0)
    with torch.no_grad():
        out = fused.cat_skip(h, s, r)
    assert _lib.call_counts.get("gip_cat2_stats_f16", 0) == before + 1, "the HIP concatenation did not run"
    ref = torch.cat([h, s if r is None else s + r], dim=1)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, ref)        # bit-exact: same roundings
    st = fused.producer_stats(out)
    if (H * W) % 128:
        assert st is None
        return
    assert st is not None and st.shape == (N * H * W // 128, Ca + Cb, 2)
    rows = ref.permute(0, 2, 3, 1).reshape(N * H * W // 128, 128, Ca + Cb).double()
    assert float((st[..., 0].double() - rows.sum(1)).abs().max()) < 1e-3
    assert float((st[..., 1].double() - (rows * rows).sum(1)).abs().max()) < 1e-2
    gn = fused.GroupNormAct(32, Ca + Cb, act=True).cuda().half().requires_grad_(False)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(Ca + Cb, generator=torch.Generator().manual_seed(1)) * 0.5 + 1)
        gn.bias.copy_(torch.randn(Ca + Cb, generator=torch.Generator().manual_seed(2)) * 0.1)
        calls = _lib.call_counts.get("gip_gn_silu_forward_stats", 0)
        y = gn(out)
        assert _lib.call_counts.get("gip_gn_silu_forward_stats", 0) == calls + 1, "the GroupNorm did not take the producer's statistics"
        y2 = gn(ref)                                                                               # own statistics pass
    want = F.silu(F.group_norm(ref.float(), 32, gn.weight.float(), gn.bias.float(), 1e-5))
    tol = 4e-3 * max(1.0, float(want.abs().max()))
    assert float((y.float() - want).abs().max()) < tol and float((y.float() - y2.float()).abs().max()) < tol


def test_skip_concatenation_falls_back_for_shapes_the_kernel_does_not_take():
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(3)
    h = torch.randn(2, 40, 8, 8, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    s = torch.randn(2, 24, 8, 8, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    assert torch.equal(fused.cat_skip(h, s, s), torch.cat([h, s + s], dim=1))


@pytest.mark.parametrize("grad", [False, True])
def test_stride2_convolution_leaves_the_next_groupnorm_its_statistics(grad):
    """gip_conv3x3s2_stats_nhwc_f16 (the VAE's Downsample2D, F.pad(x, (0, 1, 0, 1)) + stride 2): same output as the kernel without
    statistics, and per-(128-pixel block, channel) sums of exactly that output attached to it; also through the autograd node."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(11)
    cl = dict(memory_format=torch.channels_last)
    N, C, H, W, co = 4, 128, 256, 256, 128                                     # 4 x 128^2 output pixels = 512 tiles
    x = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(co, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)).half().contiguous(**cl)
    b = torch.randn(co, device="cuda", generator=g).half()
    xin = x.clone(**cl).requires_grad_(True) if grad else x
    out = fused.downsample_asym(xin, w, b)
    st = fused.producer_stats(out)
    assert st is not None and st.shape == (N * (H // 2) * (W // 2) // 128, co, 2)
    rows = out.detach().permute(0, 2, 3, 1).reshape(-1, 128, co).double()
    want = torch.stack([rows.sum(1), (rows * rows).sum(1)], dim=-1)
    assert float((st.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
    assert float((out.detach().float() - ref).abs().max()) <= 3e-3 * float(ref.abs().max())
    import os
    os.environ["GIP_CONV_S2_STATS"] = "0"
    try:
        plain = fused.downsample_asym(x, w, b)
    finally:
        del os.environ["GIP_CONV_S2_STATS"]
    assert fused.producer_stats(plain) is None and torch.equal(plain, out.detach())
    if grad:
        out.float().square().sum().backward()
        assert xin.grad is not None and torch.isfinite(xin.grad).all()


def test_vae_conv_in_leaves_the_first_groupnorm_its_statistics():
    """gip_conv3x3_c3_fwd_stats_nhwc_f16 (3 -> 128 channels, the VAE encoder's conv_in): identical output to the kernel without
    statistics, per-(16 x 8 half tile, channel) sums of that output attached to it, and its image gradient unchanged."""
    import os

    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(12)
    cl = dict(memory_format=torch.channels_last)
    N, H, W = 2, 128, 256
    x = torch.randn(N, 3, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(128, 3, 3, 3, device="cuda", generator=g) / 5.0).half().contiguous(**cl)
    b = torch.randn(128, device="cuda", generator=g).half()
    a = x.clone(**cl).requires_grad_(True)
    out = fused.conv3x3_few_inputs(a, w, b)
    st = fused.producer_stats(out)
    assert st is not None and st.shape == (N * H * W // 128, 128, 2)
    # block (n, tile, half) = rows 8 half .. 8 half + 7 of the 16 x 16 tile
    t = out.detach().permute(0, 2, 3, 1).reshape(N, H // 16, 2, 8, W // 16, 16, 128).permute(0, 1, 4, 2, 3, 5, 6).reshape(-1, 128, 128).double()
    want = torch.stack([t.sum(1), (t * t).sum(1)], dim=-1)
    assert float((st.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    os.environ["GIP_CONV_S2_STATS"] = "0"
    try:
        c = x.clone(**cl).requires_grad_(True)
        plain = fused.conv3x3_few_inputs(c, w, b)
    finally:
        del os.environ["GIP_CONV_S2_STATS"]
    assert fused.producer_stats(plain) is None and torch.equal(plain, out)
    up = torch.randn(out.shape, device="cuda", generator=g).half().contiguous(**cl)
    out.backward(up)
    plain.backward(up)
    assert torch.equal(a.grad, c.grad)
