"""gip_attention_fwd_f16 (csrc/attention.hip) against an fp32 softmax(QK^T/sqrt(D))V of the same fp16 operands.
Tolerance: probabilities and the output are rounded to half once each => ~1e-3 relative to max|out|."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,Nq,Nkv,D", [(1, 1, 128, 64, 40), (2, 8, 256, 256, 40), (2, 3, 1024, 1024, 64), (12, 8, 4096, 4096, 40),
                                          (1, 2, 384, 1152, 40), (2, 8, 512, 77, 40), (2, 8, 256, 81, 40), (1, 2, 128, 1, 64),
                                          (1, 2, 128, 130, 40), (1, 2, 2048, 16384, 40),
                                          (2, 8, 1024, 1024, 80), (1, 3, 256, 77, 80), (2, 8, 4096, 8192, 80),
                                          # round 3: head dim 160 (the 16x16 / 8x8 levels) and query counts that are multiples of 32 only
                                          (12, 8, 256, 256, 160), (12, 8, 64, 64, 160), (3, 8, 256, 81, 160), (2, 8, 64, 77, 160), (1, 2, 96, 200, 40),
                                          (2, 4, 32, 5, 80)])
@pytest.mark.parametrize("spread", [1.0, 6.0])
def test_attention_matches_fp32_reference(B, H, Nq, Nkv, D, spread):
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(Nq + D)
    q = (torch.randn(B, Nq, H * D, device="cuda", generator=g) * spread).half()
    k = (torch.randn(B, Nkv, H * D, device="cuda", generator=g) * spread).half()
    v = torch.randn(B, Nkv, H * D, device="cuda", generator=g).half()
    assert fused.attention_supported(q, k, H)
    with torch.no_grad():
        o = fused.attention(q, k, v, H)
        sp = lambda t, n: t.float().view(B, n, H, D).transpose(1, 2)  # noqa: E731
        ref = F.scaled_dot_product_attention(sp(q, Nq), sp(k, Nkv), sp(v, Nkv)).transpose(1, 2).reshape(B, Nq, H * D)
    err = float((o.float() - ref).abs().max())
    assert err <= 3e-3 * max(1.0, float(ref.abs().max())), err
    assert torch.equal(fused.attention(q, k, v, H), o)


def test_attention_module_uses_the_hip_kernel(monkeypatch):
    from gaussianip_amd.guidance import fused, networks as nw
    calls = []
    orig = fused.attention
    monkeypatch.setattr(fused, "attention", lambda *a: (calls.append(1), orig(*a))[1])
    att = nw.init_for_benchmark(nw.Attention(320, None, 8)).cuda().half().requires_grad_(False)
    x = torch.randn(2, 1024, 320, device="cuda").half()
    with torch.no_grad():
        got = att(x)
        q, k, v = att.to_q(x), att.to_k(x), att.to_v(x)
        ref = att.to_out(F.scaled_dot_product_attention(att._split(q), att._split(k), att._split(v)).transpose(1, 2).reshape(2, 1024, 320))
    assert calls and float((got.float() - ref.float()).abs().max()) < 5e-3 * max(1.0, float(ref.abs().max()))


def test_single_wide_head_attention_as_dense_gemms():
    """The VAE mid-block attention (one head of 512 channels) runs as softmax(QK^T)V through three GEMMs; forward and
    input gradient against an fp32 SDPA reference."""
    from gaussianip_amd.guidance import networks as nw
    att = nw.init_for_benchmark(nw.Attention(512, None, heads=1), 1)
    att.to_q, att.to_k, att.to_v = torch.nn.Linear(512, 512), torch.nn.Linear(512, 512), torch.nn.Linear(512, 512)
    att = nw.init_for_benchmark(att, 1).cuda()
    ref = att.float()
    x = torch.randn(2, 1024, 512, device="cuda")
    xr = x.clone().requires_grad_(True)
    q, k, v = ref.to_q(xr), ref.to_k(xr), ref.to_v(xr)
    want = ref.to_out(F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0])
    (gw,) = torch.autograd.grad(want, xr, torch.ones_like(want))
    import copy
    h = copy.deepcopy(ref).half().requires_grad_(False)
    xh = x.half().requires_grad_(True)
    got = h(xh)
    (gg,) = torch.autograd.grad(got, xh, torch.ones_like(got))
    assert float((got.detach().float() - want.detach()).abs().max()) < 1e-2 * max(1.0, float(want.detach().abs().max()))
    assert float((gg.float() - gw).abs().max()) < 2e-2 * max(1.0, float(gw.abs().max()))


def test_decoupled_cross_attention_two_key_sets():
    """text keys (77) + image-prompt keys (4), separate softmaxes, hidden = text + 0.5 * ip (LoRAIPAttnProcessor2_0)."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(9)
    for D in (40, 80):
        _two_sets(g, D)


def _two_sets(g, D):
    from gaussianip_amd.guidance import fused
    B, H, N = 3, 8, 1024
    q = torch.randn(B, N, H * D, device="cuda", generator=g).half()
    k1, v1 = [torch.randn(B, 77, H * D, device="cuda", generator=g).half() for _ in range(2)]
    k2, v2 = [torch.randn(B, 4, H * D, device="cuda", generator=g).half() for _ in range(2)]
    with torch.no_grad():
        o = fused.attention(q, k1, v1, H, k2, v2, 0.5)
        sp = lambda t: t.float().view(B, t.shape[1], H, D).transpose(1, 2)  # noqa: E731
        ref = F.scaled_dot_product_attention(sp(q), sp(k1), sp(v1)) + 0.5 * F.scaled_dot_product_attention(sp(q), sp(k2), sp(v2))
        ref = ref.transpose(1, 2).reshape(B, N, H * D)
    assert float((o.float() - ref).abs().max()) <= 3e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("fold", [False, True])
def test_reference_processor_fixtures_through_the_hip_attention(monkeypatch, fold):
    """The outputs of the REFERENCE's LoRAAttnProcessor2_0 / LoRAIPAttnProcessor2_0 (tests/golden/attention_processors.npz,
    attention_processor_faceid.py:211-523) reproduced by networks.Attention in fp16 with the MFMA attention kernel:
    256-token cases run the HIP kernel (asserted), the 64-token ones its fallback.  fp16 tolerance: 1e-2 of max|out|."""
    import os
    import numpy as np
    import fixture_inputs as fx
    from test_golden_guidance import GOLD, make_attention
    from gaussianip_amd.guidance import fused
    d = np.load(os.path.join(GOLD, "attention_processors.npz"))
    dim, heads, rank, B = int(d["dim"]), int(d["heads"]), int(d["rank"]), int(d["batch"])
    calls = []
    orig = fused.attention
    monkeypatch.setattr(fused, "attention", lambda *a: (calls.append(len(a)), orig(*a))[1])
    dev = dict(device="cuda", dtype=torch.float16)
    a_self = make_attention(fx.attn_weights(11, dim, None, rank), dim, None, heads, rank, False, fold=fold, **dev)
    a_cross = make_attention(fx.attn_weights(12, dim, 768, rank, ip=True), dim, 768, heads, rank, True, float(d["ip_scale"]),
                             fold=fold, **dev)
    with torch.no_grad():
        for n_tok, nb in ((64, B), (256, 1)):
            before = len(calls)
            x = torch.from_numpy(fx.attn_tokens(21 + n_tok, nb, n_tok, dim)).cuda().half()
            want = d["self_normal_%d" % n_tok]
            got = a_self(x).float().cpu().numpy()
            assert np.abs(got - want).max() < 1e-2 * np.abs(want).max(), ("self", n_tok)
            x = torch.from_numpy(fx.attn_tokens(31 + n_tok, nb, n_tok, dim)).cuda().half()
            ctx = torch.from_numpy(fx.attn_tokens(41, nb, 81, 768)).cuda().half()
            want = d["cross_ip_%d" % n_tok]
            got = a_cross(x, ctx).float().cpu().numpy()
            assert np.abs(got - want).max() < 1e-2 * np.abs(want).max(), ("cross", n_tok)
            if n_tok == 256:
                assert calls[before:] == [4, 7], calls[before:]     # one plain call, one two-key-set call of the HIP kernel


def test_keys_as_column_ranges_of_a_wide_projection_matrix():
    """gip_attention_fwd_strided_f16: k / v (and k2 / v2) read in place from one wide row-major matrix == packed copies."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(21)
    B, H, D, N = 3, 8, 40, 1024
    C = H * D
    q = torch.randn(B, N, C, device="cuda", generator=g).half()
    wide = torch.randn(B, 77, 5 * C + 64, device="cuda", generator=g).half()
    wide2 = torch.randn(B, 4, 3 * C, device="cuda", generator=g).half()
    k, v = wide[:, :, C:2 * C], wide[:, :, 3 * C + 64:4 * C + 64]
    k2, v2 = wide2[:, :, :C], wide2[:, :, 2 * C:]
    assert not k.is_contiguous() and fused.attention_supported(q, k, H)
    with torch.no_grad():
        assert torch.equal(fused.attention(q, k, v, H), fused.attention(q, k.contiguous(), v.contiguous(), H))
        assert torch.equal(fused.attention(q, k, v, H, k2, v2, 0.5),
                           fused.attention(q, k.contiguous(), v.contiguous(), H, k2.contiguous(), v2.contiguous(), 0.5))
        # a pair with different row strides, or a misaligned slice, is copied instead of misread
        odd = wide[:, :, 4:C + 4]
        assert fused._kv_rows(odd) is None or odd.data_ptr() % 16 == 0
        assert torch.equal(fused.attention(q, k, wide2[:, :1].expand(B, 77, 3 * C)[:, :, :C], H),
                           fused.attention(q, k.contiguous(), wide2[:, :1].expand(B, 77, 3 * C)[:, :, :C].contiguous(), H))


def test_staged_context_projections_equal_the_per_layer_projections():
    """_Encoder.stage_context: the prompt tokens projected for all cross-attention layers by one GEMM per network
    (U-Net: text + image-prompt tokens; ControlNet: all 81 tokens) give the same denoise output as 4 small GEMMs per layer."""
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
    assert gd.unet._ctx_pack is not None and gd.controlnet._ctx_pack is not None
    g = torch.Generator(device="cuda").manual_seed(4)
    B = 2
    lat = torch.randn(B, 4, 32, 32, device="cuda", generator=g)
    ctrl = torch.rand(B, 3, 256, 256, device="cuda", generator=g)
    emb = (torch.randn(3 * B, 81, 768, device="cuda", generator=g) * 0.1).half()
    tt = torch.tensor([100, 700], device="cuda")
    x3, c3, t3 = torch.cat([lat] * 3), torch.cat([ctrl] * 3), torch.cat([tt] * 3)
    with torch.no_grad():
        staged = gd.forward_unet(x3, c3, t3, emb, True)
        packs = gd.unet._ctx_pack, gd.controlnet._ctx_pack
        gd.unet._ctx_pack = gd.controlnet._ctx_pack = None
        plain = gd.forward_unet(x3, c3, t3, emb, True)
        gd.unet._ctx_pack, gd.controlnet._ctx_pack = packs
    assert float((staged.float() - plain.float()).abs().max()) < 2e-3 * max(1.0, float(plain.float().abs().max()))
    assert all(m.staged_kv is None for m in gd.unet.modules() if hasattr(m, "staged_kv"))     # every staged entry was consumed


def test_graph_replay_of_the_denoise_equals_the_eager_launches(monkeypatch):
    """GIP_GRAPH_DENOISE=1: the two-stream ControlNet + U-Net denoise captured once and replayed with new inputs gives the
    eager result (same kernels in the same order on the same streams)."""
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, ipa_guidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
    g = torch.Generator(device="cuda").manual_seed(8)
    B = 2

    def inputs():
        lat = torch.randn(B, 4, 32, 32, device="cuda", generator=g)
        ctrl = torch.rand(B, 3, 256, 256, device="cuda", generator=g)
        emb = (torch.randn(3 * B, 81, 768, device="cuda", generator=g) * 0.1).half()
        tt = torch.randint(20, 900, (B,), device="cuda", generator=g)
        return torch.cat([lat] * 3), ctrl, torch.cat([tt] * 3), emb
    sets = [inputs() for _ in range(4)]
    with torch.no_grad():
        monkeypatch.setattr(ipa_guidance, "_GRAPH_DENOISE", False)
        gd.forward_unet(*sets[0][:4], True, replicas=3)      # the libraries' first call of a shape may pick another GEMM algorithm
        eager = [gd.forward_unet(x, c, t, e, True, replicas=3) for x, c, t, e in sets]
        monkeypatch.setattr(ipa_guidance, "_GRAPH_DENOISE", True)
        replayed = [gd.forward_unet(x, c, t, e, True, replicas=3) for x, c, t, e in sets]      # eager, capture + replay, replay, replay
    assert gd._graphs and any(isinstance(v, tuple) for v in gd._graphs.values())
    for a, b in zip(eager, replayed):
        # same kernels in the same order; the library GEMMs may pick another algorithm under capture (observed at the
        # test's small shapes: 2e-3, the fp16 rounding level; bit-equal at the training shapes, tools/exp_graph.py)
        assert float((a - b).abs().max()) <= 5e-3 * max(1.0, float(a.abs().max()))
    assert replayed[2].data_ptr() != replayed[3].data_ptr()           # results are copies, not the graph's static buffer


def test_graph_replay_of_the_vae_encoder_equals_the_eager_launches(monkeypatch):
    """GIP_GRAPH_VAE (default on): the differentiable VAE encoder's forward and backward as two HIP-graph launches
    (torch.cuda.make_graphed_callables over `moments`; the latent sampling stays outside) against the eager launches:
    same latents for the same generator state, same gradient with respect to the images."""
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, ipa_guidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
    g = torch.Generator(device="cuda").manual_seed(3)
    imgs = [torch.rand(2, 3, 256, 256, device="cuda", generator=g).half() for _ in range(4)]
    gout = torch.randn(2, 4, 32, 32, device="cuda", generator=g).half()

    def run(img, seed):
        x = img.clone().requires_grad_(True)
        z = gd.encode_images(x, torch.Generator(device="cuda").manual_seed(seed))
        (dx,) = torch.autograd.grad(z, x, gout)
        return z.detach().clone(), dx.clone()
    monkeypatch.setattr(ipa_guidance, "_GRAPH_VAE", False)
    run(imgs[0], 0)
    eager = [run(im, 10 + i) for i, im in enumerate(imgs)]
    monkeypatch.setattr(ipa_guidance, "_GRAPH_VAE", True)
    replayed = [run(im, 10 + i) for i, im in enumerate(imgs)]          # eager (warm), capture, replay, replay
    assert gd._vae_graphs and any(callable(v) for v in gd._vae_graphs.values())
    for (z0, d0), (z1, d1) in zip(eager, replayed):
        assert float((z0.float() - z1.float()).abs().max()) <= 5e-3 * max(1.0, float(z0.float().abs().max()))
        assert float((d0.float() - d1.float()).abs().max()) <= 5e-3 * max(1e-6, float(d0.float().abs().max()))


def test_fused_qkv_projection_equals_three_projections(monkeypatch):
    """Self-attention with frozen weights: one [3C, C] GEMM whose output the attention kernel reads in place through row
    strides (q included: gip_attention_fwd_strided2_f16) against the three separate projections."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import networks as nw
    att = nw.init_for_benchmark(nw.Attention(640, None, 8)).cuda().half().requires_grad_(False)
    x = torch.randn(3, 1024, 640, device="cuda").half()
    res = torch.randn(3, 1024, 640, device="cuda").half()
    with torch.no_grad():
        before = _lib.call_counts.get("gip_attention_fwd_strided2_f16", 0)
        monkeypatch.setenv("GIP_FUSE_QKV", "1")
        fused_out = att(x, None, res)
        monkeypatch.setenv("GIP_FUSE_QKV", "0")
        plain = att(x, None, res)
    assert _lib.call_counts.get("gip_attention_fwd_strided2_f16", 0) - before == 2
    assert float((fused_out.float() - plain.float()).abs().max()) <= 2e-3 * max(1.0, float(plain.float().abs().max()))


def test_denoise_with_the_winograd_layers_equals_the_implicit_gemm_denoise(monkeypatch):
    """The whole ControlNet + U-Net noise prediction at the training shapes' 16 x 16 / 32 x 32 levels (64 x 64 latents, batch 12)
    with the Winograd path on and off: the prediction — what the AHDS gradient is made of — agrees to fp16 rounding level."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, ipa_guidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    monkeypatch.setattr(ipa_guidance, "_GRAPH_DENOISE", False)
    gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
    g = torch.Generator(device="cuda").manual_seed(21)
    B = 4
    lat = torch.randn(B, 4, 64, 64, device="cuda", generator=g)
    ctrl = torch.rand(B, 3, 512, 512, device="cuda", generator=g)
    emb = (torch.randn(3 * B, 81, 768, device="cuda", generator=g) * 0.1).half()
    tt = torch.randint(20, 900, (B,), device="cuda", generator=g)
    args = (torch.cat([lat] * 3), ctrl, torch.cat([tt] * 3), emb, True)
    with torch.no_grad():
        monkeypatch.setenv("GIP_WINOGRAD", "0")
        gd.forward_unet(*args, replicas=3)
        direct = gd.forward_unet(*args, replicas=3).float()
        monkeypatch.setenv("GIP_WINOGRAD", "1")
        ran = lambda: _lib.call_counts.get("gip_winograd_input_f16", 0) + _lib.call_counts.get("gip_winograd_input_gn_f16", 0)  # noqa: E731
        before = ran()
        wino = gd.forward_unet(*args, replicas=3).float()
    assert ran() - before >= 16, "the Winograd layers did not run"      # 18 at these shapes (most with the GroupNorm in their input transform)
    rel = float((wino - direct).norm() / direct.norm())
    cos = float(torch.nn.functional.cosine_similarity(wino.flatten(), direct.flatten(), dim=0))
    assert rel <= 5e-3 and cos >= 0.99995, (rel, cos)
    assert float((wino - direct).abs().max()) <= 1e-2 * max(1.0, float(direct.abs().max()))


def test_wide_head_attention_node_matches_the_op_chain_forward_and_backward():
    """fused._WideHeadAttention (the VAE encoder's 512-channel single-head mid attention: dense GEMMs around csrc/softmax.hip's
    in-place row softmax and its backward) against softmax(q k^T / sqrt(D)) v spelled with torch ops in float32, values and
    all three input gradients; plus the softmax kernels alone on ragged row lengths."""
    import ctypes
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(3)
    B, N, D = 2, 1024, 512
    q, k, v = [(torch.randn(B, N, D, device="cuda", generator=g) * s).half().requires_grad_(True) for s in (1.0, 1.0, 0.5)]
    do = (torch.randn(B, N, D, device="cuda", generator=g) * 0.1).half()
    assert fused.wide_head_attention_supported(q, k)
    out = fused.wide_head_attention(q, k, v)
    out.backward(do)
    got = [out.detach().float()] + [t.grad.float() for t in (q, k, v)]
    q32, k32, v32 = [t.detach().float().requires_grad_(True) for t in (q, k, v)]
    ref = torch.softmax(torch.bmm(q32, k32.transpose(1, 2)) * D ** -0.5, dim=-1) @ v32
    ref.backward(do.float())
    want = [ref.detach()] + [t.grad for t in (q32, k32, v32)]
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, want):
        rel = float((a - b).norm() / b.norm())
        assert rel < 4e-3, (name, rel)
    # the row kernels on their own: every chunk-count instantiation, rows that are not a multiple of the workgroup's four
    lib = _lib.nn_lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (8, 64, 520, 1024, 2056, 4096, 8192):
        s = (torch.randn(7, n, device="cuda", generator=g) * 3).half()
        p = s.clone()
        assert lib.gip_softmax_rows_f16(ctypes.c_void_p(p.data_ptr()), 7, n, 0.37, st) == 0
        ref = torch.softmax(s.float() * 0.37, dim=-1)
        assert float((p.float() - ref).abs().max()) < 1e-3, n
        dp = torch.randn(7, n, device="cuda", generator=g).half()
        gs = dp.clone()
        assert lib.gip_softmax_rows_backward_f16(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(gs.data_ptr()), 7, n, 0.37, st) == 0
        pf = p.float()
        refg = 0.37 * pf * (dp.float() - (dp.float() * pf).sum(-1, keepdim=True))
        assert float((gs.float() - refg).abs().max()) < 2e-3 * max(1.0, float(refg.abs().max())), n
    assert lib.gip_softmax_rows_f16(ctypes.c_void_p(p.data_ptr()), 7, 8200, 1.0, st) == 1           # longer than 8192: rejected
