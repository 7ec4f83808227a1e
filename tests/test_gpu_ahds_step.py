"""BASELINE.json configs[2]: the full AHDS stage-1 training step on one GPU, asserted (not only benchmarked).

100 000 Gaussians, 1024 x 1024, batch 4: render (one launch set) + GPU pose maps -> StableDiffusionGuidance.__call__
(VAE encode, AHDS timestep, ANPG over ControlNet + U-Net at batch 12, SDS loss) -> loss assembly -> backward through the
VAE encoder and the rasterizer -> densification statistics -> Adam, driven exactly as the reference's system does
(threestudio/systems/GaussianIP.py:355-356, 362-395, 446-475; guidance ipa_guidance.py:602-660).
Networks are random-initialised SD1.5-shaped stacks (no checkpoints offline)."""
from argparse import ArgumentParser

import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu

P, H, W, B = 100000, 1024, 1024, 4


@pytest.fixture(scope="module")
def rig():
    from gaussianip_amd.arguments import OptimizationParams, PipelineParams
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    from gaussianip_amd.guidance.prompts import PromptProcessor
    from gaussianip_amd.poser import Skeleton
    from gaussianip_amd.scene import GaussianModel
    from gaussianip_amd.system import StageOneStep
    from gaussianip_amd.utils import BasicPointCloud
    dev = torch.device("cuda")
    torch.manual_seed(42)
    rng = np.random.default_rng(42)
    gm = GaussianModel(0)
    gm.create_from_pcd(BasicPointCloud(scenes.human_points(P, rng).astype(np.float32), np.full((P, 3), 0.5, np.float32), None), 4.0)
    gm.training_setup(OptimizationParams(ArgumentParser()), fused=True)
    skel = Skeleton(dev)
    skel.scale(-10)                                              # GaussianIP.py:128: the skeleton of the 1.1^10-scaled body
    stage = StageOneStep(gm, PipelineParams(ArgumentParser()), torch.zeros(3, device=dev), skeleton=skel)
    g = torch.Generator(device=dev).manual_seed(1)
    tokens = (torch.randn(1, 4, 768, device=dev, generator=g) * 0.1, torch.zeros(1, 4, 768, device=dev),
              torch.randn(1, 4, 768, device=dev, generator=g) * 0.1)
    guidance = StableDiffusionGuidance(GuidanceConfig(), image_embeds_provider=lambda gd: tokens)

    def encode(texts):                                           # stand-in for the CLIP text encoder (not on this path)
        gg = torch.Generator(device=dev).manual_seed(7)
        return torch.randn(len(texts), 77, 768, device=dev, generator=gg).half() * 0.1
    pp = PromptProcessor("a person wearing a coat", encode, negative_prompt="blurry")
    guidance.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)              # GaussianIP.py:356
    return dict(gm=gm, stage=stage, guidance=guidance, pp=pp, dev=dev)


def test_three_full_ahds_steps(rig):
    from gaussianip_amd import _lib
    gm, stage, guidance, pp, dev = rig["gm"], rig["stage"], rig["guidance"], rig["pp"], rig["dev"]
    cam_rng = np.random.default_rng(3)
    prompt_utils = pp()
    names = [g_["name"] for g_ in gm.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    before = dict(_lib.call_counts)
    losses = []
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)     # precision: 16-mixed; with the fused Adam its step() stays on the device
    import torch.nn.functional as F
    library_convs = []
    real_conv2d = F.conv2d

    def spy_conv2d(x, w, *a, **k):
        library_convs.append((tuple(x.shape), tuple(w.shape)))
        return real_conv2d(x, w, *a, **k)
    sdpa_calls = []
    real_sdpa = F.scaled_dot_product_attention

    def spy_sdpa(q_, *a, **k):           # torch SDPA is an AOTriton kernel on ROCm: no attention of the step may reach it
        sdpa_calls.append(tuple(q_.shape))
        return real_sdpa(q_, *a, **k)
    for step in range(4):
        # steps 0-2: the per-view scalars on the device (a Lightning-transferred batch); step 3: the data module's own CPU
        # tensors (pose visibility rules and prompt lookup then run on the host; bench layout)
        batch = scenes.train_batch(cam_rng, B, H, W, device=dev if step < 3 else None)
        F.conv2d = spy_conv2d if step == 0 else real_conv2d          # step 0 launches every layer eagerly
        F.scaled_dot_product_attention = spy_sdpa if step == 0 else real_sdpa
        if step > 1:
            # steady state: nothing between the render and the optimizer may wait for the GPU (step 0 sizes the
            # rasterizer's capacity synchronously and runs the networks eagerly, step 1 captures their HIP graphs,
            # from step 2 on the VAE encoder and the denoise are graph replays)
            torch.cuda.set_sync_debug_mode("error")
        try:
            loss, out, gout = stage.training_step(step, batch, guidance, prompt_utils, True)
            action = stage.optimizer_step(loss, step, scaler=scaler if step >= 2 else None)
        finally:
            torch.cuda.set_sync_debug_mode("default")
            F.conv2d = real_conv2d
            F.scaled_dot_product_attention = real_sdpa
        assert action is None and set(gout) == {"loss_sds", "grad_norm"}
        assert out["comp_rgb"].shape == (B, H, W, 3) and out["pose"].shape == (B, 512, 512, 3) and out["all_vis_all"].shape == (B,)
        losses.append(loss.detach())
        for g_ in gm.optimizer.param_groups:
            grad = g_["params"][0].grad
            assert grad is not None and bool(torch.isfinite(grad).all()), g_["name"]
            if g_["name"] != "f_rest":                          # sh_degree 0: the higher bands do not exist ([P, 0, 3])
                assert float(grad.abs().max()) > 0, g_["name"]
    assert all(bool(torch.isfinite(x)) for x in losses)
    # the step's statistics reached the model: every visible Gaussian was counted once per step
    assert float(gm.denom.max()) == 4.0 and float(gm.xyz_gradient_accum.max()) > 0 and float(gm.max_radii2D.max()) > 0
    # and the hand-written HIP path is what ran
    ran = {k: _lib.call_counts.get(k, 0) - before.get(k, 0) for k in _lib.call_counts}
    for sym in ("gip_raster_forward", "gip_raster_backward", "gip_openpose_draw", "gip_conv3x3_nhwc_f16",
                "gip_attention_fwd_strided2_f16", "gip_adam_step", "gip_conv3x3_c3_fwd_stats_nhwc_f16", "gip_conv3x3s2_stats_nhwc_f16", "gip_conv3x3_c3_dgrad_nhwc_f16", "gip_gn_silu_forward", "gip_gn_silu_backward", "gip_layernorm_f16",
                "gip_conv3x3_stats_ws_nhwc_f16", "gip_linear_stats_f16", "gip_gn_silu_forward_stats", "gip_cat2_stats_f16",
                    "gip_conv3x3_fewch_nhwc_f16", "gip_upsample2x_conv3x3_nhwc_f16", "gip_conv3x3s2_dgrad_nhwc_f16"):
        assert ran.get(sym, 0) >= (4 if sym.startswith("gip_raster") or sym == "gip_openpose_draw" else 2), (sym, ran.get(sym, 0))
    assert ran["gip_raster_forward"] <= 5                        # one launch set per step (+ one capacity re-run at most)
    from gaussianip_amd.guidance import ipa_guidance
    if ipa_guidance._GRAPH_VAE and ipa_guidance._GRAPH_DENOISE:  # the frozen networks really replayed from their graphs
        assert any(callable(v) for v in guidance._vae_graphs.values()) and any(isinstance(v, tuple) for v in guidance._graphs.values())
    # no 3x3 convolution the MFMA kernel covers (input channels a multiple of 64, >= 64 output channels, 16^2 and larger)
    # may fall back to the library: a tensor that silently lost its NHWC layout (Tensor.repeat, an eager add) once sent
    # a whole ResnetBlock2D there.  What legitimately stays: the 3-channel stems and the 4 / 8-channel output convolutions
    assert not sdpa_calls, sdpa_calls            # every attention (head dims 40 / 80 / 160; the VAE's single wide head runs as GEMMs) is the HIP kernel
    stray = [(xs, ws) for xs, ws in library_convs if ws[2:] == (3, 3) and ws[1] % 64 == 0 and ws[0] >= 64 and xs[2] >= 16]
    assert not stray, stray


def test_amp_gradscaler_reproduces_the_scaled_densification_statistics(rig):
    """`precision: 16-mixed` (configs/exp.yaml:193): parameter gradients are unscaled before the hook, the view-space
    gradients the densification statistics read are not (GaussianIP.py:452-457) — the statistics carry the scale."""
    gm, stage, guidance, pp, dev = rig["gm"], rig["stage"], rig["guidance"], rig["pp"], rig["dev"]
    prompt_utils = pp()
    stats = []
    scale = 1024.0        # GradScaler's default initial scale is 65536; a smaller one keeps this fp16 random-weight VAE backward finite
    for scaler in (None, torch.amp.GradScaler("cuda", init_scale=scale)):
        gm.xyz_gradient_accum.zero_()
        gm.denom.zero_()
        torch.manual_seed(5)                                     # same timesteps / noise for both runs
        batch = scenes.train_batch(np.random.default_rng(11), B, H, W, device=dev)
        state = [g_["params"][0].detach().clone() for g_ in gm.optimizer.param_groups]
        loss, out, gout = stage.training_step(10, batch, guidance, prompt_utils, True)
        opt_state = {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in gm.optimizer.state.items()}
        stage.optimizer_step(loss, 10, scaler=scaler)
        stats.append((gm.xyz_gradient_accum.clone(), gm._xyz.grad.clone()))
        with torch.no_grad():                                    # rewind the parameters and the Adam moments
            for g_, old in zip(gm.optimizer.param_groups, state):
                g_["params"][0].copy_(old)
            for k, v in opt_state.items():
                for kk, vv in v.items():
                    if torch.is_tensor(vv):
                        gm.optimizer.state[k][kk].copy_(vv)
    (acc_plain, g_plain), (acc_amp, g_amp) = stats
    sel = acc_plain[:, 0] > 1e-3 * float(acc_plain.max())
    ratio = acc_amp[sel, 0] / acc_plain[sel, 0]
    assert bool(torch.isfinite(g_amp).all())
    assert abs(float(ratio.median()) / scale - 1) < 1e-2, float(ratio.median())      # statistics carry the scale ...
    # ... parameter gradients do not.  They are not bit-equal either: without the loss scale the fp16 VAE backward
    # flushes its smallest gradients to zero (the reason AMP scales the loss), so compare direction and magnitude
    cos = float(torch.nn.functional.cosine_similarity(g_amp.flatten(), g_plain.flatten(), dim=0))
    mag = float(g_amp.norm() / g_plain.norm())
    assert cos > 0.99 and 0.9 < mag < 1.1, (cos, mag)
