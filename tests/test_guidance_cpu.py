"""Guidance host logic on CPU: AHDS table vs the reference's (golden), SDS / ANPG algebra, shapes of the networks."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ahds_pdf_and_table_match_reference():
    from gaussianip_amd.guidance import ahds
    d = np.load(os.path.join(GOLD, "ahds_schedule.npz"))
    pdf = ahds.optimized_dual_gaussian()
    np.testing.assert_allclose(pdf, d["pdf"], rtol=0, atol=1e-12)
    sums = [pdf[a:b].sum() for a, b in ahds.AHDS_RANGES]
    assert abs(sums[0] - 0.4104) < 2e-3 and abs(sums[1] - 0.2138) < 2e-3 and int(np.argmax(pdf)) == 258
    table = ahds.timestep_table(pdf)
    assert table == list(d["table"])
    assert table[:5] == [799, 799, 798, 797, 796] and table[700] == 496 and table[-5:] == [100, 95, 89, 81, 68]
    sch = ahds.AHDSSchedule(table)
    assert sch.t_min == 68
    assert sch.window(0) == (500, 800) and sch.window(700) == (400, 546) and sch.window(1000)[0] == 150
    assert sch.window(2399) == (20, 118)
    t = sch.sample(1500, 4, "cpu", torch.Generator().manual_seed(0))
    assert t.shape == (4,) and t.dtype == torch.long and int(t.min()) >= 20


def test_rescale_noise_cfg_matches_reference():
    from gaussianip_amd.guidance.sds import rescale_noise_cfg
    d = np.load(os.path.join(GOLD, "rescale_noise_cfg.npz"))
    out = rescale_noise_cfg(torch.from_numpy(d["noise_cfg"]), torch.from_numpy(d["noise_pred_text"]), float(d["guidance_rescale"]))
    np.testing.assert_allclose(out.numpy(), d["out"], atol=1e-6)


def test_sds_algebra():
    from gaussianip_amd.guidance import sds
    acp = sds.alphas_cumprod()
    assert acp.shape == (1000,) and abs(float(acp[0]) - (1 - 0.00085)) < 1e-6 and abs(float(acp[-1]) - 0.004660) < 1e-5
    g = torch.Generator().manual_seed(0)
    B = 4
    lat = torch.randn(B, 4, 8, 8, generator=g, requires_grad=True)
    noise = torch.randn(B, 4, 8, 8, generator=g)
    t = torch.tensor([100, 169, 170, 700])
    x = sds.add_noise(lat.detach(), noise, t, acp)
    np.testing.assert_allclose(x[3].numpy(), (acp[700].sqrt() * lat.detach()[3] + (1 - acp[700]).sqrt() * noise[3]).numpy(), atol=1e-6)
    pred = torch.randn(3 * B, 4, 8, 8, generator=g)
    neg, text, null = pred.chunk(3)
    d = sds.anpg_direction(pred, t, 7.5)
    for b in range(B):
        dd = null[b] if int(t[b]) < 170 else null[b] - neg[b]
        assert torch.allclose(d[b], 7.5 * (text[b] - null[b]) + dd, atol=1e-6)
    w = sds.sds_weight(t, acp, "sds")
    assert torch.allclose(w.view(-1), 1 - acp[t])
    grad = sds.clip_grad_pixel(w * d, 1.0)
    assert float(torch.norm(grad, dim=-1).max()) <= 1.0 + 1e-5
    small = sds.clip_grad_pixel(torch.full((1, 1, 1, 4), 0.1), 1.0)
    assert torch.allclose(small, torch.full((1, 1, 1, 4), 0.1), atol=1e-6)     # below the threshold: unchanged
    loss, gfix = sds.sds_loss(lat, grad)
    loss.backward()
    assert torch.allclose(lat.grad, gfix / B, atol=1e-6)                         # d loss / d latents = grad / B
    with pytest.raises(ValueError):
        sds.sds_weight(t, acp, "nope")


def test_guidance_plugin_token_layout_and_gradient_path():
    """Small-latent run of the full ControlNet -> U-Net stack on CPU (fp32): token layout, ANPG wiring, loss gradient."""
    from gaussianip_amd.guidance import GuidanceConfig, PromptEmbeddings, StableDiffusionGuidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    d = np.load(os.path.join(GOLD, "ahds_schedule.npz"))
    cfg = GuidanceConfig(half_precision_weights=False, channels_last=False)
    gd = StableDiffusionGuidance(cfg, device="cpu", schedule=AHDSSchedule(list(d["table"])))
    assert not any(hasattr(m, "lora_q") for m in gd.unet.modules())          # LoRA folded
    B = 2
    g = torch.Generator().manual_seed(0)
    tabs = [torch.randn(13, 77, 768, generator=g) * 0.1 for _ in range(3)]
    pu = PromptEmbeddings(*tabs, direction_fn=lambda el, az, c, v, dist: (az > 0).long())
    gd.set_image_embeds(torch.randn(1, 4, 768, generator=g) * 0.1, torch.zeros(1, 4, 768), torch.randn(1, 4, 768, generator=g) * 0.1)
    el, az = torch.zeros(B), torch.tensor([-10.0, 20.0])
    emb = gd._prompt_embeds(pu, el, az, None, None, None, 3)
    assert emb.shape == (3 * B, 81, 768)
    assert torch.equal(emb[0, :77], tabs[1][0]) and torch.equal(emb[B + 1, :77], tabs[0][1])   # [neg | pos | null], per-view rows
    assert torch.equal(emb[B, 77:], gd.pos_image_embeds[0]) and float(emb[0, 77:].abs().max()) == 0.0
    latents = torch.randn(B, 4, 8, 8, generator=g, requires_grad=True)
    control = torch.rand(B, 3, 64, 64, generator=g)
    t = torch.tensor([100, 600])
    grad, aux = gd.compute_grad_anpg(latents, control, t, pu, True, None, el, az, None, None, generator=g)
    assert grad.shape == latents.shape and aux["noise_pred"].shape == (3 * B, 4, 8, 8) and torch.isfinite(grad).all()
    assert float(torch.norm(grad, dim=-1).max()) <= cfg.grad_clip_threshold + 1e-5
    grad2, _ = gd.compute_grad_sds(latents, control, t, pu, True, None, el, az, None, None, generator=g)
    assert grad2.shape == latents.shape
    from gaussianip_amd.guidance import sds
    loss, gfix = sds.sds_loss(latents, grad)
    loss.backward()
    assert torch.allclose(latents.grad, gfix / B, atol=1e-6)
