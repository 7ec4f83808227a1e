"""Whole-network parity of the two networks that are 98 % of the AHDS step.

The per-kernel tests cannot see a wrong skip-connection order, a dropped `replicas` tile, a stale graph input or a stream race
between the ControlNet's side stream and the U-Net.  Here the PRODUCT path — fp16, NHWC, folded LoRA, shared prefix
(`replicas = 3`), ControlNet on the second stream, HIP-graph replay, Winograd / split-K / fused epilogues, every hand-written
kernel — is compared at the TRAINING SHAPE with an independent statement of the same architecture: deep copies of the same
modules converted to float32 and run under `fused.disabled()`, i.e. plain PyTorch ops (F.conv2d, F.group_norm, F.linear,
SDPA in fp32, torch.cat) on NCHW tensors, one stream, no graph, no replica sharing, per-layer time-embedding and key / value
projections.  Reference for what is computed: threestudio/models/guidance/ipa_guidance.py:311-358 (forward_unet: ControlNet
-> 13 residuals -> U-Net) and :522-531 (encode_images).

Bars (fp16 storage of ~60 layers of activations against fp32): relative L2 and cosine of the whole output, and the 13 + 1
ControlNet residuals one by one.  The achieved figures are printed (pytest -s) and written to
gpurun_out/network_parity.json."""
import copy
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

B = 4                     # views per step (configs/exp.yaml:59-61) -> ANPG batch 12
REL_L2 = 5e-3             # VERDICT r3 item 2's bars
COSINE = 0.9999
_report = {}


def _dump():
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "network_parity.json"), "w") as f:
            json.dump(_report, f, indent=1)


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
    cos = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
    mx = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    return rel, cos, mx


@pytest.fixture(scope="module")
def guidance():
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    gd = StableDiffusionGuidance(GuidanceConfig(), image_embeds_provider=lambda _: tokens)
    gd.prepare_for_sds("a", "b", "c")
    return gd


def _fp32_copy(module):
    """The same weights as float32 parameters of an independent module tree (packed inference tables are fp16-only and are
    skipped by the fp32 path: networks._Encoder.stage_* check the dtype)."""
    m = copy.deepcopy(module).float().to(memory_format=torch.contiguous_format)
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def _inputs(dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    lat = torch.randn(B, 4, 64, 64, device=dev, generator=g)
    t = torch.randint(20, 800, (B,), device=dev, generator=g)
    ctrl = torch.rand(B, 3, 512, 512, device=dev, generator=g)
    emb = (torch.randn(3 * B, 81, 768, device=dev, generator=g) * 0.1).half()
    return torch.cat([lat] * 3), ctrl, torch.cat([t] * 3), emb


def test_forward_unet_training_shape_against_fp32_pytorch(guidance):
    """batch 12 (3 prompt branches x 4 views), 64^2 latents, 81 tokens, two streams, graph replay ON (the third call of a
    shape replays the captured graph)."""
    from gaussianip_amd.guidance import fused, ipa_guidance
    assert ipa_guidance._GRAPH_DENOISE and ipa_guidance._TWO_STREAMS, "this test pins the DEFAULT product path"
    dev = guidance.device
    x, ctrl, t, emb = _inputs(dev)
    outs = []
    for call in range(3):                                        # eager (warm) -> capture -> replay
        with torch.no_grad():
            outs.append(guidance.forward_unet(x, ctrl, t, emb, True, replicas=3).float().clone())
    ent = guidance._graphs and [v for v in guidance._graphs.values() if v != "warm"]
    assert ent, "the HIP graph of the denoise was not captured"
    # replay with DIFFERENT inputs of the same shape, then the first inputs again: a stale static input or a pointer baked to a
    # freed temporary would show here
    x2, ctrl2, t2, emb2 = _inputs(dev, seed=5)
    with torch.no_grad():
        other = guidance.forward_unet(x2, ctrl2, t2, emb2, True, replicas=3).float().clone()
        again = guidance.forward_unet(x, ctrl, t, emb, True, replicas=3).float().clone()
    assert torch.equal(again, outs[2]) and torch.equal(outs[1], outs[2]), "graph replay is not reproducible"
    r_eager = _rel(outs[0], outs[2])
    assert r_eager[0] < 1e-6 or torch.equal(outs[0], outs[2]), "eager and graph-replayed denoise differ: %s" % (r_eager,)
    assert not torch.equal(other, outs[2])

    unet32, cn32 = _fp32_copy(guidance.unet), _fp32_copy(guidance.controlnet)
    with fused.disabled(), torch.no_grad():
        ctx32 = emb.float()
        # the reference casts everything to the weights dtype (ipa_guidance.py:324-330): the fp32 statement sees the same
        # fp16-representable inputs
        x32 = x.half().float()
        down32, mid32 = cn32(x32, t, ctx32, ctrl.half().float().repeat(3, 1, 1, 1))
        ref = unet32(x32, t, ctx32, down32, mid32)
        ref_other = unet32(x2.half().float(), t2, emb2.float(), *cn32(x2.half().float(), t2, emb2.float(), ctrl2.half().float().repeat(3, 1, 1, 1)))
    rel, cos, mx = _rel(outs[2], ref)
    rel2, cos2, mx2 = _rel(other, ref_other)
    _report["forward_unet"] = dict(shape=list(ref.shape), rel_l2=rel, cosine=cos, max_over_max=mx, second_input_rel_l2=rel2,
                                   second_input_cosine=cos2, ref_rms=float(ref.double().pow(2).mean().sqrt()))
    print("forward_unet [12,4,64,64] vs fp32 PyTorch: rel L2 %.3e  cosine %.6f  max/max %.3e   (second input: %.3e / %.6f)" % (rel, cos, mx, rel2, cos2))

    # the 13 down residuals + the mid residual of the ControlNet, through the product path (HIP kernels, shared prefix)
    with torch.no_grad():
        xh = x.half().contiguous(memory_format=torch.channels_last)
        down16, mid16 = guidance.controlnet(xh, t, emb, None, 1.0, cond_embedding=guidance.embed_control(ctrl), replicas=3)
    res = []
    for i, (a, b) in enumerate(zip(list(down16) + [mid16], list(down32) + [mid32])):
        assert a.shape == b.shape, (i, a.shape, b.shape)
        r = _rel(a.float(), b)
        res.append(dict(index=i, shape=list(b.shape), rel_l2=r[0], cosine=r[1], max_over_max=r[2]))
        print("  ControlNet residual %2d %-20s rel L2 %.3e  cosine %.6f" % (i, tuple(b.shape), r[0], r[1]))
    _report["controlnet_residuals"] = res
    _dump()
    assert rel < REL_L2 and cos > COSINE, (rel, cos)
    assert rel2 < REL_L2 and cos2 > COSINE, (rel2, cos2)
    assert len(res) == 13 and all(r["rel_l2"] < REL_L2 and r["cosine"] > COSINE for r in res), res


@pytest.mark.parametrize("scale", [1.0, 1024.0])
def test_encode_images_forward_and_image_gradient_against_fp32_autograd(guidance, scale):
    """encode_images (ipa_guidance.py:522-531) at 4 x 512^2: latents AND dL/dimage (the gradient that flows back into the
    rasterizer), product path (graphed VAE forward / backward, one-node ResnetBlock2D, fused GroupNorm backward, parity-class
    stride-2 data gradients) against float32 autograd through plain PyTorch ops.  `scale`: the loss scale in front of the
    backward (a GradScaler multiplies the loss; 1024 is what tests/test_gpu_ahds_step.py trains with)."""
    from gaussianip_amd.guidance import fused
    dev = guidance.device
    g = torch.Generator(device=dev).manual_seed(11)
    img = torch.rand(B, 3, 512, 512, device=dev, generator=g)
    G = torch.randn(B, 4, 64, 64, device=dev, generator=g) * 0.03          # dL/dlatents of the order of the clipped SDS gradient / B
    vae32 = _fp32_copy(guidance.vae)

    def run_hip(call):
        x = img.clone().requires_grad_(True)
        z = guidance.encode_images(x, torch.Generator(device=dev).manual_seed(77))
        (z * G).sum().mul(scale).backward()
        return z.detach().float(), x.grad.detach().float() / scale

    runs = [run_hip(i) for i in range(3)]                         # eager -> capture -> graph replay
    assert guidance._vae_graphs and any(v != "warm" for v in guidance._vae_graphs.values()), "the VAE graph was not captured"
    assert torch.equal(runs[1][0], runs[2][0]) and torch.equal(runs[1][1], runs[2][1]), "graph replay is not reproducible"
    z16, gx16 = runs[2]
    assert torch.isfinite(gx16).all()

    x32 = img.clone().requires_grad_(True)
    with fused.disabled():
        u = x32 * 2.0 - 1.0
        mom = vae32.moments(u + (u.half().float() - u).detach())       # the fp16-representable input the product sees (straight-through), fp32 graph
        mean, logvar = mom.chunk(2, dim=1)
        std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
        # the product draws the latent noise in fp16 from the generator (diffusers' DiagonalGaussianDistribution.sample draws in
        # the parameters' dtype): the same stream, rounded the same way
        noise = torch.randn(mean.shape, device=dev, dtype=torch.float16, generator=torch.Generator(device=dev).manual_seed(77)).float()
        z32 = (mean + std * noise) * vae32.scaling_factor
        (z32 * G).sum().backward()
    rz, rg = _rel(z16, z32.detach()), _rel(gx16, x32.grad)
    r01 = _rel(runs[0][1], runs[2][1])
    _report["encode_images_scale_%g" % scale] = dict(latents=dict(rel_l2=rz[0], cosine=rz[1], max_over_max=rz[2]),
                                                     image_grad=dict(rel_l2=rg[0], cosine=rg[1], max_over_max=rg[2],
                                                                     ref_rms=float(x32.grad.double().pow(2).mean().sqrt())),
                                                     eager_vs_graph_grad_rel_l2=r01[0])
    print("encode_images scale %g: latents rel L2 %.3e cosine %.6f | dL/dimage rel L2 %.3e cosine %.6f max/max %.3e | eager vs graph %.1e" % (
        scale, rz[0], rz[1], rg[0], rg[1], rg[2], r01[0]))
    _dump()
    assert rz[0] < REL_L2 and rz[1] > COSINE, rz
    assert rg[0] < REL_L2 and rg[1] > COSINE, rg       # measured 2.5e-3 .. 2.7e-3 / 0.999996 (profiles/r04_network_parity.json)
    assert r01[0] < 1e-6 or torch.equal(runs[0][1], runs[2][1])


def test_second_vae_forward_before_the_backward_does_not_corrupt_the_first(guidance):
    """make_graphed_callables keeps one set of static activations; a second encode_images of the same shape
    before backward() must not change the first call's gradient (the guard runs the second call eagerly)."""
    dev = guidance.device
    g = torch.Generator(device=dev).manual_seed(3)
    a_img, b_img = torch.rand(B, 3, 512, 512, device=dev, generator=g), torch.rand(B, 3, 512, 512, device=dev, generator=g)
    G = torch.randn(B, 4, 64, 64, device=dev, generator=g) * 0.03

    def grad_alone(img):
        x = img.clone().requires_grad_(True)
        (guidance.encode_images(x, torch.Generator(device=dev).manual_seed(5)) * G).sum().backward()
        return x.grad.clone()
    for _ in range(3):
        ref = grad_alone(a_img)
    xa, xb = a_img.clone().requires_grad_(True), b_img.clone().requires_grad_(True)
    za = guidance.encode_images(xa, torch.Generator(device=dev).manual_seed(5))
    zb = guidance.encode_images(xb, torch.Generator(device=dev).manual_seed(6))       # before za's backward
    (za * G).sum().backward()
    (zb * G).sum().backward()
    assert torch.equal(xa.grad, ref) or _rel(xa.grad.float(), ref.float())[0] < 1e-6
    assert torch.isfinite(xb.grad).all() and float(xb.grad.abs().max()) > 0
    assert torch.equal(grad_alone(a_img), ref)                      # and the graph path is back afterwards


def test_graph_is_dropped_when_a_frozen_input_of_it_changes(guidance):
    """ip_scale is a kernel ARGUMENT inside the captured graph.  Changing it must not replay the stale graph."""
    from gaussianip_amd.guidance import fused
    dev = guidance.device
    x, ctrl, t, emb = _inputs(dev, seed=9)
    with torch.no_grad():
        for _ in range(3):
            base = guidance.forward_unet(x, ctrl, t, emb, True, replicas=3).clone()
        old = [m.ip_scale for m in guidance.unet.modules() if getattr(m, "ip", False)][0]
        guidance.set_ip_scale(old + 0.25)
        changed = guidance.forward_unet(x, ctrl, t, emb, True, replicas=3).clone()
        with fused.disabled():
            pass
        guidance.set_ip_scale(old)
        for _ in range(3):
            back = guidance.forward_unet(x, ctrl, t, emb, True, replicas=3).clone()
    assert not torch.equal(changed, base), "the stale graph (old ip_scale) was replayed"
    assert torch.equal(back, base)


@pytest.mark.parametrize("views", [2, 1])
def test_sharded_shapes_run_on_the_hip_kernels_reproducibly_and_match_fp32(guidance, views):
    """The denoise at the batch of a 2-view / 1-view shard of BASELINE configs[3] (6 / 3 samples).  Round 4 found these shapes
    on MIOpen's atomic split-K kernels for their 20- and 30-tile layers (8 x 8 level; the floor was 32 tiles): the sharded
    step was not reproducible run to run.  Asserted here: no MFMA-shaped 3x3 convolution reaches F.conv2d, no attention reaches
    torch SDPA, repeated calls are bitwise equal (eager and graph replay), and the result matches the fp32 statement."""
    import torch.nn.functional as F
    from gaussianip_amd.guidance import fused
    dev = guidance.device
    g = torch.Generator(device=dev).manual_seed(21 + views)
    lat = torch.randn(views, 4, 64, 64, device=dev, generator=g)
    t = torch.randint(20, 800, (views,), device=dev, generator=g)
    ctrl = torch.rand(views, 3, 512, 512, device=dev, generator=g)
    emb = (torch.randn(3 * views, 81, 768, device=dev, generator=g) * 0.1).half()
    x, tt = torch.cat([lat] * 3), torch.cat([t] * 3)
    lib_convs, sdpa = [], []
    real_conv, real_sdpa = F.conv2d, F.scaled_dot_product_attention

    def spy_conv(x_, w_, *a, **k):
        if w_.dim() == 4 and w_.shape[2:] == (3, 3) and w_.shape[1] % 64 == 0 and w_.shape[0] % 64 == 0:
            lib_convs.append((tuple(x_.shape), tuple(w_.shape)))
        return real_conv(x_, w_, *a, **k)

    def spy_sdpa(q_, *a, **k):
        sdpa.append(tuple(q_.shape))
        return real_sdpa(q_, *a, **k)
    outs = []
    try:
        F.conv2d, F.scaled_dot_product_attention = spy_conv, spy_sdpa
        with torch.no_grad():
            for _ in range(4):           # eager, capture, replay, replay
                outs.append(guidance.forward_unet(x, ctrl, tt, emb, True, replicas=3).float().clone())
    finally:
        F.conv2d, F.scaled_dot_product_attention = real_conv, real_sdpa
    assert not lib_convs, "MFMA-shaped convolutions on the library path at batch %d: %s" % (3 * views, lib_convs[:4])
    assert not sdpa, sdpa[:4]
    assert all(torch.equal(o, outs[0]) for o in outs), [float((o - outs[0]).abs().max()) for o in outs]
    unet32, cn32 = _fp32_copy(guidance.unet), _fp32_copy(guidance.controlnet)
    with fused.disabled(), torch.no_grad():
        x32, ctx32 = x.half().float(), emb.float()
        ref = unet32(x32, tt, ctx32, *cn32(x32, tt, ctx32, ctrl.half().float()))
    rel, cos, mx = _rel(outs[-1], ref)
    _report["forward_unet_batch_%d" % (3 * views)] = dict(rel_l2=rel, cosine=cos, max_over_max=mx)
    print("forward_unet batch %d vs fp32 PyTorch: rel L2 %.3e cosine %.6f" % (3 * views, rel, cos))
    _dump()
    assert rel < REL_L2 and cos > COSINE, (rel, cos)
