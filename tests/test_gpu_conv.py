"""gip_conv3x3_nhwc_f16 (MFMA implicit GEMM, csrc/conv3x3.hip) against torch's fp32 convolution of the same fp16
operands: forward with bias / residual epilogue, data gradient (same kernel on the flipped-transposed weight), image
borders, ragged M (N*H*W not a multiple of the 128-pixel tile), both channel-tile widths (128, 160).
Tolerance: one rounding of the fp32 accumulator to half => |err| <= 2^-11 * |out| + accumulation-order noise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # N, Cin, Cout, H, W   (tile counts below 256 take the split-K path)
    (2, 64, 64, 9, 7), (1, 64, 128, 1, 1), (3, 128, 320, 17, 5), (12, 320, 320, 64, 64), (12, 640, 320, 64, 64),
    (12, 1280, 640, 32, 32), (4, 128, 128, 96, 80), (4, 512, 512, 64, 64), (2, 192, 72, 33, 31), (12, 1280, 1280, 16, 16), (12, 1280, 1280, 8, 8), (12, 2560, 1280, 8, 8)]


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_forward_backward(shape, monkeypatch):
    from gaussianip_amd.guidance import fused
    monkeypatch.setattr(fused, "_MIN_CONV_TILES", 0)          # always take the HIP kernel, also for tiny shapes
    N, ci, co, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(ci * 7 + co + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, ci, H, W, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    w = (torch.randn(co, ci, 3, 3, device="cuda", generator=g) / (3 * ci ** 0.5)).half().contiguous(**cl)
    b = torch.randn(co, device="cuda", generator=g).half()
    r = torch.randn(N, co, H, W, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    calls = []
    orig = fused._conv_call
    monkeypatch.setattr(fused, "_conv_call", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    out = fused.conv3x3(x, w, b, r)
    assert calls, "the HIP convolution did not run"
    ref = F.conv2d(x.detach().float(), w.float(), b.float(), padding=1) + r.detach().float()
    scale = float(ref.abs().max())
    assert float((out.float() - ref).abs().max()) <= 1.5e-3 * scale
    plain = fused.conv3x3(x, w)
    ref_p = F.conv2d(x.detach().float(), w.float(), None, padding=1)
    assert float((plain.float() - ref_p).abs().max()) <= 1.5e-3 * float(ref_p.abs().max())
    if co % 64 == 0:
        dy = torch.randn(N, co, H, W, device="cuda", generator=g).half().contiguous(**cl)
        n0 = len(calls)
        dx, dr = torch.autograd.grad(out, [x, r], dy)
        assert len(calls) > n0 and torch.equal(dr, dy)
        dx_ref = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), padding=1)
        assert float((dx.float() - dx_ref).abs().max()) <= 1.5e-3 * float(dx_ref.abs().max())
    # bitwise reproducible
    assert torch.equal(fused.conv3x3(x, w, b, r), out)


def test_single_tile_problems_stay_on_miopen_and_small_ones_do_not():
    """Only single-tile problems stay on the library (round 4: the 20- and 30-tile layers of a sharded step — batch 3 / 6 at the
    8 x 8 level — were on MIOpen's atomic split-K kernels, which made the sharded denoise irreproducible)."""
    from gaussianip_amd.guidance import fused
    x = torch.randn(1, 1280, 8, 8, device="cuda").half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 1280, 3, 3, device="cuda").half().contiguous(memory_format=torch.channels_last) * 0.01
    assert fused._conv_tiles(1, 8, 8, 128) < fused._MIN_CONV_TILES
    out = fused.conv3x3(x, w)
    assert torch.allclose(out.float(), F.conv2d(x, w, padding=1).float(), atol=2e-2)
    # the 8 x 8 level at batch 3 (one view of a sharded step): 20 tiles -> the MFMA kernel with split-K, bitwise reproducible
    x = torch.randn(3, 1280, 8, 8, device="cuda").half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(1280, 1280, 3, 3, device="cuda").half().contiguous(memory_format=torch.channels_last) * 0.01
    assert fused._MIN_CONV_TILES <= fused._conv_tiles(3, 8, 8, 1280) < 32
    calls = []
    real = F.conv2d
    try:
        F.conv2d = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        outs = [fused.conv3x3(x, w) for _ in range(4)]
    finally:
        F.conv2d = real
    assert not calls, "a 20-tile layer reached F.conv2d"
    assert all(torch.equal(o, outs[0]) for o in outs)
    ref = real(x.float(), w.float(), padding=1)
    assert float((outs[0].float() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    x16 = torch.randn(3, 1280, 16, 16, device="cuda").half().contiguous(memory_format=torch.channels_last)
    d = [fused.downsample_sym(x16, w, None) for _ in range(4)]
    assert all(torch.equal(o, d[0]) for o in d)
    ref = real(x16.float(), w.float(), stride=2, padding=1)
    assert float((d[0].float() - ref).abs().max()) < 2e-3 * float(ref.abs().max())


def test_vae_downsample_gradient_through_the_dilated_convolution():
    """Data gradient of pad(0,1,0,1) + 3x3/stride-2 via the stride-1 MFMA convolution of the zero-dilated gradient."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(2)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(2, 128, 64, 48, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    w = (torch.randn(128, 128, 3, 3, device="cuda", generator=g) / 34.0).half().contiguous(**cl)
    b = torch.randn(128, device="cuda", generator=g).half()
    y = fused.downsample_asym(x, w, b)
    assert y.grad_fn is not None and "DownsampleAsym" in type(y.grad_fn).__name__
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (0, 1, 0, 1)), w.float(), b.float(), stride=2)
    assert float((y.float() - yr).abs().max()) <= 2e-3 * float(yr.abs().max())
    dy = torch.randn(y.shape, device="cuda", generator=g).half().contiguous(**cl)
    (dx,) = torch.autograd.grad(y, x, dy)
    (dxr,) = torch.autograd.grad(yr, xr, dy.float())
    assert float((dx.float() - dxr).abs().max()) <= 2e-3 * float(dxr.abs().max())


def test_conv_in_gradient_through_the_padded_mfma_convolution():
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(4)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(2, 3, 160, 96, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    w = (torch.randn(128, 3, 3, 3, device="cuda", generator=g) / 5.0).half().contiguous(**cl)
    b = torch.randn(128, device="cuda", generator=g).half()
    y = fused.conv3x3_few_inputs(x, w, b)
    assert "ConvFewInputChannels" in type(y.grad_fn).__name__
    dy = torch.randn(y.shape, device="cuda", generator=g).half().contiguous(**cl)
    (dx,) = torch.autograd.grad(y, x, dy)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), padding=1)
    assert dx.shape == x.shape and float((dx.float() - dxr).abs().max()) <= 2e-3 * float(dxr.abs().max())


@pytest.mark.parametrize("N,C,Co,H,W,pad", [(12, 320, 320, 64, 64, 1), (12, 1280, 1280, 16, 16, 1), (4, 128, 128, 64, 48, 0),
                                            (2, 256, 256, 34, 18, 0), (3, 64, 72, 10, 6, 1)])
def test_stride2_convolution(N, C, Co, H, W, pad, monkeypatch):
    """gip_conv3x3s2_nhwc_f16: symmetric padding=1 (U-Net Downsample2D) and the VAE's pad(0,1,0,1) form."""
    from gaussianip_amd.guidance import fused
    monkeypatch.setattr(fused, "_MIN_CONV_TILES", 0)
    g = torch.Generator(device="cuda").manual_seed(C + H + pad)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(Co, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)).half().contiguous(**cl)
    b = torch.randn(Co, device="cuda", generator=g).half()
    with torch.no_grad():
        got = fused.downsample_sym(x, w, b) if pad else fused.downsample_asym(x, w, b)
    xin = x.float() if pad else F.pad(x.float(), (0, 1, 0, 1))
    ref = F.conv2d(xin, w.float(), b.float(), stride=2, padding=pad)
    assert got.shape == ref.shape and float((got.float() - ref).abs().max()) <= 1.5e-3 * float(ref.abs().max())
    if not pad and Co % 64 == 0:          # differentiable VAE form: forward through the kernel, gradient checked too
        xg = x.clone().requires_grad_(True)
        y = fused.downsample_asym(xg, w, b)
        dy = torch.randn(y.shape, device="cuda", generator=g).half().contiguous(**cl)
        (dx,) = torch.autograd.grad(y, xg, dy)
        xr = x.float().requires_grad_(True)
        (dxr,) = torch.autograd.grad(F.conv2d(F.pad(xr, (0, 1, 0, 1)), w.float(), b.float(), stride=2), xr, dy.float())
        assert float((y.float() - ref).abs().max()) <= 1.5e-3 * float(ref.abs().max())
        assert float((dx.float() - dxr).abs().max()) <= 2e-3 * float(dxr.abs().max())


def test_transposed_weight_cache_is_never_stale(monkeypatch):
    """The data-gradient convolution caches the flipped / transposed copy of a frozen weight.  A weight that is freed and
    whose address the allocator hands to a NEW weight of the same shape (a rebuilt network, a checkpoint loaded later)
    must not be served the old copy: the cache pins its source tensor, so the address cannot be reused while the entry
    lives."""
    from gaussianip_amd.guidance import fused
    monkeypatch.setattr(fused, "_MIN_CONV_TILES", 0)
    fused._wt_cache.clear()
    cl = dict(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2, 64, 16, 16, device="cuda", generator=g).half().contiguous(**cl)
    dy = torch.randn(2, 64, 16, 16, device="cuda", generator=g).half().contiguous(**cl)
    ptrs = set()
    for trial in range(6):
        w = (torch.randn(64, 64, 3, 3, device="cuda", generator=g) / 24.0).half().contiguous(**cl)
        ptrs.add(w.data_ptr())
        xi = x.clone().requires_grad_(True)
        fused.conv3x3(xi, w).backward(dy)
        ref = torch.nn.grad.conv2d_input(xi.shape, w.float(), dy.float(), padding=1)
        assert float((xi.grad.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), trial
        del w
    assert len(ptrs) == 6 and len(fused._wt_cache) == 6          # every weight kept its own address while cached
    fused._wt_cache.clear()


# ---- round 3: the 256 x 256 tile 8-wave kernel (conv_big_kernel) ----
@pytest.mark.parametrize("shape", [(4, 256, 256, 128, 128), (1, 64, 256, 241, 239), (2, 128, 512, 160, 192)])
def test_conv_big_tile_kernel(shape):
    """Shapes whose 256-pixel x 256-channel tiles fill the chip take conv_big_kernel; forced on and off through the debug
    knob, both against the fp32 convolution of the same operands — bias, residual, ragged M (57 599 pixels), the epilogue
    statistics of both 128-row blocks of a tile."""
    import ctypes
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    N, ci, co, H, W = shape
    knob = ctypes.c_int.in_dll(_lib.nn_lib()._lib, "gip_dbg_conv_big")
    g = torch.Generator(device="cuda").manual_seed(ci + co + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, ci, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(co, ci, 3, 3, device="cuda", generator=g) / (3 * ci ** 0.5)).half().contiguous(**cl)
    b = torch.randn(co, device="cuda", generator=g).half()
    r = torch.randn(N, co, H, W, device="cuda", generator=g).half().contiguous(**cl)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    outs = {}
    try:
        for big in (0, 1):
            knob.value = big
            outs[big] = (fused._conv_call(x, w, co, b), fused._conv_call(x, w, co, b, r))
            if (H * W) % 128 == 0:
                holder = []
                ys = fused._conv_call(x, w, co, b, r, holder)
                assert holder and torch.equal(ys, outs[big][1])
                if ci == 128 and not big and H % 8 == 0 and W % 16 == 0:
                    # the halo-resident kernel (Cin = 128): a statistics block is a 16 x 8 image block, not 128 consecutive pixels
                    rows = ys.reshape(N, co, H // 8, 8, W // 16, 16).permute(0, 2, 4, 3, 5, 1).reshape(-1, 128, co).double()
                else:
                    rows = ys.permute(0, 2, 3, 1).reshape(-1, 128, co).double()
                want = torch.stack([rows.sum(1), (rows * rows).sum(1)], dim=-1)
                assert float((holder[0].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    finally:
        knob.value = -1
    for big in (0, 1):
        assert float((outs[big][0].float() - ref).abs().max()) <= 1.5e-3 * float(ref.abs().max())
        assert float((outs[big][1].float() - (ref + r.float())).abs().max()) <= 1.5e-3 * float((ref + r.float()).abs().max())
    assert torch.equal(outs[0][0], outs[1][0])       # same summation order: bit-identical


@pytest.mark.parametrize("cin,cout", [(128, 128), (128, 256)])
@pytest.mark.parametrize("sums", [False, True])
def test_resblock_as_one_autograd_node(cin, cout, sums, monkeypatch):
    """The differentiable ResnetBlock2D of the VAE encoder as ONE autograd node (fused._ResBlockNode: the shortcut's
    gradient rides in the GroupNorm backward's apply pass) against the same block in fp32 PyTorch ops, forward and
    dL/dx, and against the five-node path."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    from gaussianip_amd.guidance.networks import ResBlock, init_for_benchmark
    # sums: the GroupNorm backward's two reductions come out of the data-gradient convolutions' epilogues
    # (gip_conv3x3_gnbwd_nhwc_f16 -> gip_gn_silu_backward_sums) instead of their own pass
    monkeypatch.setenv("GIP_GN_BWD_SUMS", "1" if sums else "0")
    monkeypatch.setattr(fused, "_GN_SUMS_MIN_TILES", 0)
    before = _lib.call_counts.get("gip_gn_silu_backward_sums", 0)
    torch.manual_seed(0)
    blk = init_for_benchmark(ResBlock(cin, cout, temb_dim=0, eps=1e-6), seed=3)
    with torch.no_grad():
        for m in (blk.norm1, blk.norm2):
            m.weight.add_(torch.randn_like(m.weight) * 0.2)
            m.bias.add_(torch.randn_like(m.bias) * 0.2)
        blk.conv1.bias.add_(torch.randn_like(blk.conv1.bias) * 0.3)
        blk.conv2.bias.add_(torch.randn_like(blk.conv2.bias) * 0.3)
    ref_blk = blk.cuda().float()
    import copy
    blk = copy.deepcopy(ref_blk).half().to(memory_format=torch.channels_last).requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2, cin, 64, 64, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(2, cout, 64, 64, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)

    def run(node):
        monkeypatch.setenv("GIP_RESBLOCK_NODE", "1" if node else "0")
        xi = x.clone(memory_format=torch.channels_last).requires_grad_(True)
        seen = []
        orig = fused._ResBlockNode.apply
        monkeypatch.setattr(fused._ResBlockNode, "apply", staticmethod(lambda *a: (seen.append(1), orig(*a))[1]))
        y = blk(xi)
        monkeypatch.setattr(fused._ResBlockNode, "apply", orig)
        assert bool(seen) == node
        (dx,) = torch.autograd.grad(y, xi, dy)
        return y.detach(), dx

    y1, dx1 = run(True)
    assert (_lib.call_counts.get("gip_gn_silu_backward_sums", 0) - before == 2) == sums      # opt-in path (measured neutral)
    y0, dx0 = run(False)
    xr = x.float().contiguous().requires_grad_(True)
    ref_blk.requires_grad_(False)
    yr = ref_blk(xr)
    (dxr,) = torch.autograd.grad(yr, xr, dy.float().contiguous())
    for got_y, got_dx in ((y1, dx1), (y0, dx0)):
        assert float((got_y.float() - yr).abs().max()) < 6e-3 * max(1.0, float(yr.abs().max()))
        assert float((got_dx.float() - dxr).abs().max()) < 8e-3 * max(1.0, float(dxr.abs().max()))
    assert float((dx1.float() - dx0.float()).abs().max()) < 4e-3 * max(1.0, float(dx0.float().abs().max()))


@pytest.mark.parametrize("shape", [(2, 64, 48), (4, 512, 512), (1, 40, 56)])
def test_vae_conv_in_and_its_data_gradient(shape):
    """conv_in of the VAE encoder (3 -> 128) on csrc/conv_small.hip, forward (bias, borders) and dL/dx (the gradient that
    leaves the VAE), against fp32 PyTorch on the same fp16 operands; (1, 40, 56) is not a multiple of the 16-pixel tile and
    takes the library / 128-wide routes."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    N, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, 3, H, W, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    w = (torch.randn(128, 3, 3, 3, device="cuda", generator=g) / 5.0).half().contiguous(**cl)
    b = torch.randn(128, device="cuda", generator=g).half()
    dy = torch.randn(N, 128, H, W, device="cuda", generator=g).half().contiguous(**cl)
    fwd = lambda: _lib.call_counts.get("gip_conv3x3_c3_fwd_nhwc_f16", 0) + _lib.call_counts.get("gip_conv3x3_c3_fwd_stats_nhwc_f16", 0)  # noqa: E731
    before = (fwd(), _lib.call_counts.get("gip_conv3x3_c3_dgrad_nhwc_f16", 0))
    y = fused.conv3x3_few_inputs(x, w, b)
    (dx,) = torch.autograd.grad(y, x, dy)
    ran = (fwd() - before[0], _lib.call_counts.get("gip_conv3x3_c3_dgrad_nhwc_f16", 0) - before[1])
    assert ran == ((1, 1) if H % 16 == 0 and W % 16 == 0 else (0, 0))
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, w.float(), b.float(), padding=1)
    (dxr,) = torch.autograd.grad(yr, xr, dy.float())
    assert y.shape == yr.shape and y.is_contiguous(**cl)
    assert float((y.float() - yr).abs().max()) <= 1.5e-3 * float(yr.abs().max())
    assert float((dx.float() - dxr).abs().max()) <= 2e-3 * float(dxr.abs().max())


# ---- the ControlNet's conditioning stem: few-channel 3x3 convolutions (stride 1 / 2) with bias + SiLU in the epilogue
# (gip_conv3x3_fewch_nhwc_f16, csrc/conv_small.hip) ----
@pytest.mark.parametrize("cfg", [(3, 16, 1, 64, 48), (3, 128, 1, 32, 32), (16, 16, 1, 64, 48), (16, 32, 2, 64, 96), (32, 32, 1, 24, 32),
                                 (32, 96, 2, 48, 64), (96, 96, 1, 16, 48), (96, 256, 2, 24, 64)])
@pytest.mark.parametrize("act", [True, False])
def test_few_channel_convolutions_of_the_controlnet_stem(cfg, act):
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    cin, cout, stride, H, W = cfg
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + cout)
    x = torch.randn(3, cin, H, W, device="cuda", generator=g).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (9 * cin) ** 0.5).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", generator=g).half()
    before = _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0)
    with torch.no_grad():
        out = fused.conv3x3_fewch(x, w, b, stride, act)
    assert _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0) == before + 1, "the HIP kernel did not run"
    assert out.shape == (3, cout, H // stride, W // stride) and out.is_contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.float(), w.float(), b.float(), stride=stride, padding=1)
    if act:
        ref = F.silu(ref.half().float())          # torch's own order: the convolution result is rounded to half first
    scale = max(1.0, float(ref.abs().max()))
    # tolerance: one (two with the activation) half roundings of an O(1) value + fp32 accumulation order
    assert float((out.float() - ref).abs().max()) <= 2.5e-3 * scale
    # borders: the first / last rows and columns see the zero padding
    edge = torch.cat([(out.float() - ref)[:, :, 0].flatten(), (out.float() - ref)[:, :, -1].flatten(),
                      (out.float() - ref)[:, :, :, 0].flatten(), (out.float() - ref)[:, :, :, -1].flatten()])
    assert float(edge.abs().max()) <= 2.5e-3 * scale


def test_controlnet_stem_runs_on_the_few_channel_kernels():
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    from gaussianip_amd.guidance.networks import ControlNet, init_for_benchmark
    torch.manual_seed(0)
    net = init_for_benchmark(ControlNet()).cuda().half().to(memory_format=torch.channels_last).requires_grad_(False)
    cond = torch.rand(2, 3, 128, 128, device="cuda").half().contiguous(memory_format=torch.channels_last)
    before = _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0)
    with torch.no_grad():
        emb = net.embed_condition(cond)
        assert _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0) - before == 7, "a stem layer took the library route"
        with fused.disabled():
            ref = net.embed_condition(cond)
    assert emb.shape == (2, 320, 16, 16)
    assert float((emb.float() - ref.float()).abs().max()) <= 4e-3 * max(1.0, float(ref.float().abs().max()))


# ---- data gradient of the VAE's stride-2 convolution as four parity classes (gip_conv3x3s2_dgrad_nhwc_f16) ----
@pytest.mark.parametrize("N,C,Co,H,W", [(2, 128, 128, 64, 48), (1, 256, 256, 34, 18), (2, 72, 128, 20, 12), (1, 512, 512, 16, 32), (1, 320, 64, 8, 8)])
def test_stride2_data_gradient_by_parity_classes(N, C, Co, H, W, monkeypatch):
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    monkeypatch.setattr(fused, "_MIN_CONV_TILES", 0)
    g = torch.Generator(device="cuda").manual_seed(C + Co + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    w = (torch.randn(Co, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)).half().contiguous(**cl)
    b = torch.randn(Co, device="cuda", generator=g).half()
    before = _lib.call_counts.get("gip_conv3x3s2_dgrad_nhwc_f16", 0)
    y = fused.downsample_asym(x, w, b)
    dy = torch.randn(y.shape, device="cuda", generator=g).half().contiguous(**cl)
    (dx,) = torch.autograd.grad(y, x, dy)
    assert _lib.call_counts.get("gip_conv3x3s2_dgrad_nhwc_f16", 0) == before + 1, "the parity-class kernel did not run"
    xr = x.detach().float().requires_grad_(True)
    (dxr,) = torch.autograd.grad(F.conv2d(F.pad(xr, (0, 1, 0, 1)), w.float(), b.float(), stride=2), xr, dy.float())
    assert dx.shape == x.shape and dx.is_contiguous(**cl)
    err = (dx.float() - dxr).abs()
    assert float(err.max()) <= 2e-3 * float(dxr.abs().max())
    # every parity class and the image borders (first / last rows and columns) individually
    for pi in range(2):
        for pj in range(2):
            assert float(err[:, :, pi::2, pj::2].max()) <= 2e-3 * float(dxr.abs().max())
    assert float(torch.cat([err[:, :, 0].flatten(), err[:, :, -1].flatten(), err[:, :, :, 0].flatten(), err[:, :, :, -1].flatten()]).max()) \
        <= 2e-3 * float(dxr.abs().max())


# ---- nearest 2x upsampling + 3x3 convolution as four 2 x 2-tap parity classes (gip_upsample2x_conv3x3_nhwc_f16) ----
@pytest.mark.parametrize("N,C,Co,H,W", [(8, 1280, 1280, 16, 16), (8, 640, 640, 32, 32), (2, 64, 72, 7, 5), (1, 128, 320, 16, 24)])
def test_upsample_then_convolution_by_parity_classes(N, C, Co, H, W, monkeypatch):
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    monkeypatch.setattr(fused, "_UPCONV_MIN_TILES", 0)
    g = torch.Generator(device="cuda").manual_seed(C + Co + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(Co, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)).half().contiguous(**cl)
    b = torch.randn(Co, device="cuda", generator=g).half()
    before = _lib.call_counts.get("gip_upsample2x_conv3x3_nhwc_f16", 0)
    with torch.no_grad():
        out = fused.upsample2x_conv3x3(x, w, b)
    assert _lib.call_counts.get("gip_upsample2x_conv3x3_nhwc_f16", 0) == before + 1, "the parity-class kernel did not run"
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    assert out.shape == ref.shape and out.is_contiguous(**cl)
    err = (out.float() - ref).abs()
    # tolerance: half output rounding + the summed weights' one extra half rounding (2^-11 relative per weight, random signs)
    tol = 2.5e-3 * float(ref.abs().max())
    assert float(err.max()) <= tol
    for pi in range(2):
        for pj in range(2):
            assert float(err[:, :, pi::2, pj::2].max()) <= tol
    assert float(torch.cat([err[:, :, 0].flatten(), err[:, :, -1].flatten(), err[:, :, :, 0].flatten(), err[:, :, :, -1].flatten()]).max()) <= tol
    # the two-step form on the same kernels agrees to the same tolerance
    with torch.no_grad():
        two = fused.conv3x3(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b)
    assert float((out.float() - two.float()).abs().max()) <= tol


@pytest.mark.parametrize("cfg", [(320, 4, 3, 64, 64), (512, 8, 2, 64, 48), (320, 4, 1, 8, 16)])
def test_narrow_output_convolutions(cfg):
    """conv_out of the U-Net (320 -> 4) and of the VAE encoder (512 -> 8, differentiable) on the halo-in-LDS kernel."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    cin, cout, N, H, W = cfg
    g = torch.Generator(device="cuda").manual_seed(cin + H)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, cin, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3 * cin ** 0.5)).half().contiguous(**cl)
    b = torch.randn(cout, device="cuda", generator=g).half()
    before = _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0)
    with torch.no_grad():
        out = fused.conv3x3_narrow_out(x, w, b)
    assert _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0) == before + 1, "the HIP kernel did not run"
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    assert out.shape == ref.shape and float((out.float() - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))
    # differentiable form: forward on the kernel, dL/dx from the library
    xg = x.clone(**cl).requires_grad_(True)
    y = fused.conv3x3_narrow_out(xg, w, b)
    dy = torch.randn(y.shape, device="cuda", generator=g).half().contiguous(**cl)
    (dx,) = torch.autograd.grad(y, xg, dy)
    dxr = torch.nn.grad.conv2d_input(x.shape, w.float(), dy.float(), padding=1)
    assert torch.equal(y, out) and float((dx.float() - dxr).abs().max()) <= 2e-3 * max(1.0, float(dxr.abs().max()))


# ---- Winograd F(2x2, 3x3) at the 16 x 16 level (gip_winograd_input_f16 / _output_f16 around one batched GEMM) ----
@pytest.mark.parametrize("cfg", [(12, 1280, 1280, True), (12, 2560, 1280, False), (8, 1920, 1280, True), (8, 1280, 640, False),
                                 (12, 640, 1280, True, 16), (6, 2560, 1280, False, 16), (4, 960, 640, True, 32), (3, 1920, 640, False, 32)])
def test_winograd_convolution_at_the_16x16_level(cfg, monkeypatch):
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    monkeypatch.setenv("GIP_WINOGRAD", "1")
    N, cin, cout, with_res = cfg[:4]
    H = W = cfg[4] if len(cfg) > 4 else 16
    g = torch.Generator(device="cuda").manual_seed(cin + cout)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, cin, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3 * cin ** 0.5)).half().contiguous(**cl)
    b = torch.randn(cout, device="cuda", generator=g).half()
    r = torch.randn(N, cout, H, W, device="cuda", generator=g).half().contiguous(**cl) if with_res else None
    ran = lambda: _lib.call_counts.get("gip_winograd_output_f16", 0) + _lib.call_counts.get("gip_winograd_output_stats_f16", 0)  # noqa: E731
    before = ran()
    with torch.no_grad():
        out = fused.conv3x3(x, w, b, r)
        monkeypatch.setenv("GIP_WINOGRAD", "0")
        direct = fused.conv3x3(x, w, b, r)
    assert ran() == before + 1, "the Winograd path did not run"
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + (0 if r is None else r.float())
    scale = max(1.0, float(ref.abs().max()))
    e_w, e_d = float((out.float() - ref).abs().max()), float((direct.float() - ref).abs().max())
    # fp16 Winograd rounds the transformed input and the sixteen products once more: allow 3x the implicit GEMM's bound
    assert e_d <= 1.5e-3 * scale and e_w <= 4.5e-3 * scale, (e_w / scale, e_d / scale)
    rms_w = float((out.float() - ref).pow(2).mean().sqrt())
    rms_d = float((direct.float() - ref).pow(2).mean().sqrt())
    assert rms_w <= 4.0 * rms_d + 1e-4, (rms_w, rms_d)
    # borders: the zero padding enters through the input transform's out-of-image taps
    err = (out.float() - ref).abs()
    edge = torch.cat([err[:, :, 0].flatten(), err[:, :, -1].flatten(), err[:, :, :, 0].flatten(), err[:, :, :, -1].flatten()])
    assert float(edge.max()) <= 4.5e-3 * scale
    # with the following GroupNorm's statistics taken by the output transform: same tensor, sums of what was written
    monkeypatch.setenv("GIP_WINOGRAD", "1")
    with torch.no_grad():
        out2 = fused.conv3x3(x, w, b, r, gn_next=True)
    st = fused.producer_stats(out2)
    assert torch.equal(out2, out) and st is not None and st.shape == (N * H * W // 128, cout, 2)
    rows = out2.permute(0, 2, 3, 1).reshape(-1, 128, cout).double()
    want = torch.stack([rows.sum(1), (rows * rows).sum(1)], dim=-1)
    assert float((st.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


# N, Cout, H, W (Cin = 128).  (Cout % 256 == 0 at >= 224 tiles of 256 pixels goes to the 256 x 256 streaming kernel instead.)
HALO_SHAPES = [(2, 128, 128, 128), (3, 512, 48, 64), (1, 128, 8, 4096), (4, 128, 64, 128), (4, 128, 256, 256), (2, 384, 128, 256)]


@pytest.mark.parametrize("shape", HALO_SHAPES)
@pytest.mark.parametrize("mode", ["plain", "bias_residual_stats"])
def test_halo_resident_tile_of_the_128_channel_convolution(shape, mode):
    """Cin = 128 (the VAE encoder's first level): the pixel tile is a 16 x 8 image block whose halo sits in LDS once and the K
    loop streams only weights (conv3x3_kernel<..., HALO>).  Against F.conv2d in fp32, bitwise against the streaming kernel (same
    K order: tap-major, then channel block — same MFMA sequence), with the per-channel statistics of its epilogue; image borders,
    several samples, two channel tiles."""
    import ctypes
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    N, co, H, W = shape
    assert fused._conv_tiles(N, H, W, co) >= 256 and H % 8 == 0 and W % 16 == 0
    # (the debug knob below also switches the 8-wave halo form off: `base` is the streaming 128-pixel kernel for every shape)
    g = torch.Generator(device="cuda").manual_seed(co + H + W)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, 128, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(co, 128, 3, 3, device="cuda", generator=g) / 34.0).half().contiguous(**cl)
    full = mode != "plain"
    b = torch.randn(co, device="cuda", generator=g).half() if full else None
    r = torch.randn(N, co, H, W, device="cuda", generator=g).half().contiguous(**cl) if full else None
    holder = [] if full else None
    out = fused._conv_call(x, w, co, b, r, holder)
    knob = ctypes.c_int.in_dll(_lib.nn_lib()._lib, "gip_dbg_conv_epilogue")
    knob.value = 0                                   # per-lane epilogue: the halo path needs the LDS one, so this is the streaming kernel
    try:
        base = fused._conv_call(x, w, co, b, r, None)
    finally:
        knob.value = -1
    ref = F.conv2d(x.float(), w.float(), None if b is None else b.float(), padding=1)
    if full:
        ref = ref.half().float() + r.float()        # fp16(fp16(conv + bias) + residual)
    err = float((out.float() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-3, err
    assert torch.equal(out, base), "halo and streaming kernels differ: %g" % float((out.float() - base.float()).abs().max())
    if full:
        st = holder[0]
        rows = out.permute(0, 2, 3, 1).float()
        tot = torch.stack([rows.reshape(N, -1, co).double().sum(1), (rows.reshape(N, -1, co).double() ** 2).sum(1)], dim=-1)   # per sample
        got = st.reshape(N, -1, co, 2).double().sum(1)
        assert float((got - tot).abs().max()) <= 2e-5 * float(tot.abs().max())
        # a statistics block = the 128 pixels of ONE 16 x 8 image block of one sample; block order: (sample, block row, block column)
        # on the 4-wave kernel, (sample, 16-row tile, tile column, upper / lower half) on the 8-wave one
        big = False       # (the 8-wave 16 x 16 form was measured slower and is not dispatched: tools/experiments/conv_big_halo_16x16.diff.txt)
        if big:
            blk = out.float().reshape(N, co, H // 16, 2, 8, W // 16, 16).permute(0, 2, 5, 3, 1, 4, 6).reshape(-1, co, 128)
        else:
            blk = out.float().reshape(N, co, H // 8, 8, W // 16, 16).permute(0, 2, 4, 1, 3, 5).reshape(N * (H // 8) * (W // 16), co, 128)
        want = torch.stack([blk.double().sum(-1), (blk.double() ** 2).sum(-1)], dim=-1)
        assert float((st.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_halo_kernel_as_data_gradient_with_groupnorm_backward_sums():
    """The same kernel as the data gradient of conv1 / conv2 of the VAE's first ResnetBlock2D, with the GroupNorm-backward
    reductions in its epilogue (gip_conv3x3_gnbwd_nhwc_f16): output and sums against the streaming kernel."""
    import ctypes
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    N, C, H, W = 2, 128, 128, 128
    g = torch.Generator(device="cuda").manual_seed(9)
    cl = dict(memory_format=torch.channels_last)
    dy = (torch.randn(N, C, H, W, device="cuda", generator=g) * 0.1).half().contiguous(**cl)
    xg = torch.randn(N, C, H, W, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / 34.0).half().contiguous(**cl)
    gn = fused.GroupNormAct(32, C, eps=1e-6, act=True).cuda().half().requires_grad_(False)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, device="cuda", generator=g) * 0.5 + 1.0)
        gn.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.2)
    _, mean, rstd = fused._gn_fwd_raw(xg, gn, None, None)
    out, sums = fused._dgrad_with_gn_sums(dy, w, xg, gn, mean, rstd, None)
    assert sums is not None
    dx = fused._gn_bwd_raw(xg, out, gn, mean, rstd, None, chan_sums=sums)
    plain = fused._conv_call(dy, fused._transposed_weight(w), C)
    assert torch.equal(out, plain)
    dx_ref = fused._gn_bwd_raw(xg, plain, gn, mean, rstd, None)          # its own reduction pass
    assert float((dx.float() - dx_ref.float()).abs().max()) <= 2e-3 * float(dx_ref.float().abs().max())


# ---- round 5: GroupNorm + SiLU inside the halo-resident convolution, own kernels for the last library convolutions ----
@pytest.mark.parametrize("cout", [128, 192])
def test_groupnorm_inside_the_halo_convolution_equals_the_apply_pass_path(cout, monkeypatch):
    """fused._ResBlockNode with GIP_CONV_GNIN=1 (gip_gn_stats_from_partials + gip_conv3x3_gnin_nhwc_f16: the normalised tensor is
    never materialised) against the same node with the separate GroupNorm apply pass: the kernel repeats the apply pass'
    arithmetic on its LDS halo, so the forward output agrees to one half ulp (in a few elements) — and the fp32 block agrees to fp16
    tolerance."""
    import copy
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    from gaussianip_amd.guidance.networks import ResBlock, init_for_benchmark
    torch.manual_seed(0)
    blk = init_for_benchmark(ResBlock(128, cout, temb_dim=0, eps=1e-6), seed=5)
    with torch.no_grad():
        for m in (blk.norm1, blk.norm2):
            m.weight.add_(torch.randn_like(m.weight) * 0.2)
            m.bias.add_(torch.randn_like(m.bias) * 0.2)
        blk.conv1.bias.add_(torch.randn_like(blk.conv1.bias) * 0.3)
    ref_blk = blk.cuda().float().requires_grad_(False)
    blk = copy.deepcopy(ref_blk).half().to(memory_format=torch.channels_last).requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(2)
    cl = dict(memory_format=torch.channels_last)
    N, H, W = 2, 128, 144                                         # 288 / 432 output tiles, image borders on all four sides of many tiles
    x0 = (torch.randn(N, 128, H, W, device="cuda", generator=g) * 1.5 + 0.3).half().contiguous(**cl)
    dy = torch.randn(N, cout, H, W, device="cuda", generator=g).half().contiguous(**cl)
    # x must carry its producer's statistics (the VAE's conv_in / the previous block leave them): make them with the statistics kernel
    w_id = torch.zeros(128, 128, 3, 3, device="cuda").half()
    w_id[torch.arange(128), torch.arange(128), 1, 1] = 1.0
    w_id = w_id.contiguous(**cl)

    def run(gnin):
        monkeypatch.setenv("GIP_CONV_GNIN", "1" if gnin else "0")
        x = fused.conv3x3(x0, w_id, gn_next=True)                # identity convolution: x0 with chan_stats attached
        assert fused.producer_stats(x) is not None and torch.equal(x, x0)
        st = fused.producer_stats(x)
        xi = x.detach().requires_grad_(True)
        fused.attach_stats(xi, st)
        before = _lib.call_counts.get("gip_conv3x3_gnin_nhwc_f16", 0)
        y = blk(xi)
        n_calls = _lib.call_counts.get("gip_conv3x3_gnin_nhwc_f16", 0) - before
        (dx,) = torch.autograd.grad(y, xi, dy)
        return y.detach(), dx, n_calls, fused.producer_stats(y)

    y1, dx1, n1, st1 = run(True)
    y0, dx0, n0, st0 = run(False)
    assert n1 == (2 if cout == 128 else 1) and n0 == 0, (n1, n0)      # conv2 of the 128 -> 192 block has 192 input channels: apply pass
    # the same fp32 formula and one rounding to half in both paths; the compiler is free to round `half(y * sigmoid(y))` from the
    # exact product in one kernel (v_fma_mixlo_f16) and from the float32 product in the other, so: equal to ONE half ulp, and
    # unequal at all in few elements (measured: bit-identical with the round-5 first build, <= 1 ulp in 0.0x % after a refactor)
    # (what differs by one half ulp is the NORMALISED INPUT of a few convolution taps; the block output then differs by a few ulp
    # of its own scale in the few elements those taps reach)
    dy_ = (y1.float() - y0.float()).abs()
    assert float(dy_.max()) <= 2e-3 * float(y0.float().abs().max()) and float((dy_ > 0).float().mean()) < 0.02, \
        (float(dy_.max()), float((dy_ > 0).float().mean()))
    assert float((dx1.float() - dx0.float()).abs().max()) <= 2e-3 * float(dx0.float().abs().max())
    assert st1 is not None and st0 is not None
    assert float((st1.double() - st0.double()).abs().max()) <= 1e-4 * float(st0.double().abs().max())
    xr = x0.float().contiguous().requires_grad_(True)
    yr = ref_blk(xr)
    (dxr,) = torch.autograd.grad(yr, xr, dy.float().contiguous())
    assert float((y1.float() - yr).abs().max()) < 6e-3 * max(1.0, float(yr.abs().max()))
    assert float((dx1.float() - dxr).abs().max()) < 8e-3 * max(1.0, float(dxr.abs().max()))


def test_latent_conv_in_and_the_folded_quant_conv_run_on_own_kernels():
    """conv_in of the U-Net / ControlNet (4 latent channels -> 320) and the VAE encoder's conv_out with its 1x1 quant_conv composed
    in (512 -> 8, forward and data gradient): the last convolutions of the training step that ran on MIOpen (round 5)."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(9)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(12, 4, 64, 64, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(320, 4, 3, 3, device="cuda", generator=g) / 6.0).half().contiguous(**cl)
    b = torch.randn(320, device="cuda", generator=g).half()
    before = _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0)
    with torch.no_grad():
        out = fused.conv3x3_latent_in(x, w, b)
        out4 = fused.conv3x3_latent_in(x[:4], w, b)               # the shared prefix of a replicated batch
    assert _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0) == before + 2
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    assert float((out.float() - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max())) and torch.equal(out4, out[:4])
    # conv_out . quant_conv of the VAE encoder
    conv_out = torch.nn.Conv2d(512, 8, 3, padding=1).cuda().half().to(**cl).requires_grad_(False)
    quant = torch.nn.Conv2d(8, 8, 1).cuda().half().to(**cl).requires_grad_(False)
    with torch.no_grad():
        quant.weight.add_(torch.eye(8, device="cuda").half().reshape(8, 8, 1, 1))
    h = torch.randn(4, 512, 64, 64, device="cuda", generator=g).half().contiguous(**cl).requires_grad_(True)
    w2, b2 = fused.folded_quant_conv(conv_out, quant)
    assert fused.folded_quant_conv(conv_out, quant)[0] is w2      # cached
    y = fused.conv3x3_narrow_out(h, w2, b2)
    dy = torch.randn(y.shape, device="cuda", generator=g).half().contiguous(**cl)
    c0 = _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0)
    (dh,) = torch.autograd.grad(y, h, dy)
    assert _lib.call_counts.get("gip_conv3x3_fewch_nhwc_f16", 0) == c0 + 1, "the data gradient did not run on the few-channel kernel"
    hr = h.detach().float().requires_grad_(True)
    yr = F.conv2d(F.conv2d(hr, conv_out.weight.float(), conv_out.bias.float(), padding=1), quant.weight.float(), quant.bias.float())
    (dhr,) = torch.autograd.grad(yr, hr, dy.float())
    assert float((y.float() - yr).abs().max()) <= 3e-3 * max(1.0, float(yr.abs().max()))
    assert float((dh.float() - dhr).abs().max()) <= 3e-3 * max(1.0, float(dhr.abs().max()))


def test_batched_own_gemm_equals_sixteen_products():
    """gip_linear_batched_f16 (blockIdx.y = product): the sixteen GEMMs of a Winograd convolution in one launch of the own MFMA
    kernel == torch.bmm in fp32 of the same half operands; both channel-tile widths, a ragged row count."""
    import ctypes
    from gaussianip_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(4)
    for T, K, Nout in ((768, 1280, 1280), (200, 640, 320), (3072, 960, 640)):
        V = (torch.randn(16, T, K, device="cuda", generator=g) * 0.5).half()
        U = (torch.randn(16, Nout, K, device="cuda", generator=g) * 0.05).half()
        M = torch.empty(16, T, Nout, device="cuda", dtype=torch.float16)
        rc = _lib.nn_lib().gip_linear_batched_f16(ctypes.c_void_p(V.data_ptr()), ctypes.c_void_p(U.data_ptr()), ctypes.c_void_p(M.data_ptr()), 16, T, K, Nout,
                                                  T * K, Nout * K, T * Nout, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        ref = torch.bmm(V.float(), U.float().transpose(1, 2))
        assert float((M.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), (T, K, Nout)


def test_winograd_convolution_with_the_own_batched_gemm(monkeypatch):
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    monkeypatch.setenv("GIP_WINOGRAD_GEMM", "own")
    g = torch.Generator(device="cuda").manual_seed(21)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(12, 1280, 16, 16, device="cuda", generator=g).half().contiguous(**cl)
    w = (torch.randn(1280, 1280, 3, 3, device="cuda", generator=g) / (3 * 1280 ** 0.5)).half().contiguous(**cl)
    b = torch.randn(1280, device="cuda", generator=g).half()
    before = _lib.call_counts.get("gip_linear_batched_f16", 0)
    with torch.no_grad():
        assert fused._winograd_applies(x, w, None)
        out = fused.conv3x3(x, w, b)
    assert _lib.call_counts.get("gip_linear_batched_f16", 0) == before + 1
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    assert float((out.float() - ref).abs().max()) <= 6e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cfg", [(12, 640, 640, 1280, 16, "rows"), (8, 1280, 640, 1280, 16, None), (4, 640, 320, 640, 32, "one"),
                                 (12, 1280, 1280, 1280, 16, "rows")])
def test_groupnorm_inside_the_winograd_input_transform(cfg, monkeypatch):
    """fused.conv3x3_gn: GroupNorm + SiLU applied while the Winograd input transform loads its patches (gip_winograd_input_gn_f16,
    statistics from the producer's partial sums) against the apply pass followed by the plain transform: the same fp32 formula with
    one rounding of the normalised value to half, so the outputs agree to the rounding of a few taps (see the halo test above);
    groups that straddle the kernel's 8-channel chunks (1920 / 32 = 60 channels per group), per-sample / shared / no addend."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    N, ca, cb, cout, H, ad_kind = cfg
    C = ca + cb
    g = torch.Generator(device="cuda").manual_seed(C + H)
    cl = dict(memory_format=torch.channels_last)
    a = (torch.randn(N, ca, H, H, device="cuda", generator=g) * 1.3 + 0.2).half().contiguous(**cl)
    b = (torch.randn(N, cb, H, H, device="cuda", generator=g) * 0.7 - 0.4).half().contiguous(**cl)
    w = (torch.randn(cout, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)).half().contiguous(**cl)
    bias = torch.randn(cout, device="cuda", generator=g).half()
    res = torch.randn(N, cout, H, H, device="cuda", generator=g).half().contiguous(**cl)
    gn = fused.GroupNormAct(32, C, eps=1e-5, act=True).cuda().half().requires_grad_(False)
    with torch.no_grad():
        gn.weight.add_(torch.randn(C, device="cuda", generator=g).half() * 0.2)
        gn.bias.add_(torch.randn(C, device="cuda", generator=g).half() * 0.2)
    addend = {None: None, "one": torch.randn(C, device="cuda", generator=g).half() * 0.3,
              "rows": torch.randn(N, C, device="cuda", generator=g).half() * 0.3}[ad_kind]

    def run(on):
        monkeypatch.setenv("GIP_WINOGRAD_GN", "1" if on else "0")
        x = fused.cat_skip(a, b)
        assert fused.producer_stats(x) is not None
        before = _lib.call_counts.get("gip_winograd_input_gn_f16", 0)
        with torch.no_grad():
            assert fused._winograd_applies(x, w, res)
            y = fused.conv3x3_gn(x, gn, addend, w, bias, res, gn_next=True)
        return y, _lib.call_counts.get("gip_winograd_input_gn_f16", 0) - before, x

    y1, n1, x = run(True)
    y0, n0, _ = run(False)
    assert (n1, n0) == (1, 0)
    d = (y1.float() - y0.float()).abs()
    assert float(d.max()) <= 2e-3 * float(y0.float().abs().max()) and float((d > 0).float().mean()) < 0.05, \
        (float(d.max()), float((d > 0).float().mean()))
    st1, st0 = fused.producer_stats(y1), fused.producer_stats(y0)
    assert st1 is not None and float((st1.double() - st0.double()).abs().max()) <= 1e-4 * float(st0.double().abs().max())
    xf = x.float()
    if addend is not None:
        xf = xf + addend.float().reshape(-1 if addend.dim() == 2 else 1, C, 1, 1)
    ref = F.conv2d(F.silu(F.group_norm(xf, 32, gn.weight.float(), gn.bias.float(), 1e-5)), w.float(), bias.float(), padding=1) + res.float()
    assert float((y1.float() - ref).abs().max()) <= 6e-3 * max(1.0, float(ref.abs().max()))
    # without the producer's statistics the apply-pass path runs (nothing to normalise with)
    monkeypatch.setenv("GIP_WINOGRAD_GN", "1")
    before = _lib.call_counts.get("gip_winograd_input_gn_f16", 0)
    with torch.no_grad():
        y2 = fused.conv3x3_gn(x.clone(memory_format=torch.channels_last), gn, addend, w, bias, res)
    # (its GroupNorm then takes its own two-pass statistics: mean / rstd differ from the partial-sum ones in the last float bits)
    assert _lib.call_counts.get("gip_winograd_input_gn_f16", 0) == before
    assert float((y2.float() - y0.float()).abs().max()) <= 2e-3 * float(y0.float().abs().max())
