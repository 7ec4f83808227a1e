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
    # the branches of a replicated batch share everything in front of the first cross-attention: same values
    with torch.no_grad():
        x3 = torch.cat([latents.detach()] * 3)
        full = gd.forward_unet(x3, control, torch.cat([t] * 3), emb, True, replicas=1)
        shared = gd.forward_unet(x3, control, torch.cat([t] * 3), emb, True, replicas=3)
    assert float((full - shared).abs().max()) < 1e-4 * max(1.0, float(full.abs().max()))
    from gaussianip_amd.guidance import sds
    loss, gfix = sds.sds_loss(latents, grad)
    loss.backward()
    assert torch.allclose(latents.grad, gfix / B, atol=1e-6)

    # ---- VCR refine pass on the same networks (tiny images, 2 of the 8 steps, one view of each attention branch) ----
    from gaussianip_amd.guidance import refine as rf
    from gaussianip_amd.guidance.networks import VAEDecoder, init_for_benchmark
    dec = init_for_benchmark(VAEDecoder(), 5).eval().requires_grad_(False)
    vcr = rf.ViewConsistentRefiner(gd, dec, num_steps=2)
    rgb = torch.rand(32, 64, 64, 3, generator=g)
    ctrl = torch.rand(32, 64, 64, 3, generator=g)
    cond, uncond = torch.randn(1, 77, 768, generator=g) * 0.1, torch.randn(1, 77, 768, generator=g) * 0.1
    seen = []
    def prompt_fn(name):
        seen.append(name)
        return cond, uncond
    views = ["front", "k0", "v3"]                       # v3 blends k0 and front (0.75 / 0.25)
    out, idx = vcr.refine_rgb(rgb, ctrl, prompt_fn, views=views, generator=torch.Generator().manual_seed(1))
    assert out.shape == (3, 64, 64, 3) and idx == [24, 20, 21] and seen == views
    assert torch.isfinite(out).all() and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert vcr.ctl.state == "normal" and all(len(a.refine.stored_zt) == 0 for a in vcr.targets)
    assert all(a.ip_scale == gd.cfg.ipa_faceid_scale for a in gd.unet.modules() if getattr(a, "ip", False))
    out2, _ = vcr.refine_rgb(rgb, ctrl, prompt_fn, views=views, generator=torch.Generator().manual_seed(1))
    assert torch.equal(out, out2)                       # same seed, same result; the state machine resets cleanly


def test_refine_tables_ddim_and_attention_state_machine():
    import torch.nn.functional as F
    from gaussianip_amd.guidance import refine as rf, sds
    from gaussianip_amd.guidance.networks import Attention, init_for_benchmark
    assert rf.refine_timesteps(8).tolist() == [142, 122, 101, 81, 61, 40, 20, 0]   # tests/golden/refine_timesteps.npz
    assert len(rf.VIEW_IDX_ALL) == 32 and sorted(rf.VIEW_IDX_ALL) == list(range(32)) and rf.VIEW_NAME_ALL[8] == "v0"
    assert rf.KEY_VIEW_NAME_PAIR["v0"] == ("left", "k0") and rf.KEY_VIEW_NAME_PAIR["v5"] == ("k0", "front")
    assert rf.KEY_VIEW_NAME_PAIR["v17"] == ("k2", "back") and rf.KEY_VIEW_NAME_PAIR["v23"] == ("k3", "left")
    assert rf.KEY_VIEW_WEIGHT_PAIR["v9"] == (0.75, 0.25) and rf.KEY_VIEW_WEIGHT_PAIR["v10"] == (0.5, 0.5) and rf.KEY_VIEW_WEIGHT_PAIR["v11"] == (0.25, 0.75)
    # DDIM (eta 0): stepping from t with the true noise recovers the x0-consistent sample at t - 20
    acp = sds.alphas_cumprod()
    g = torch.Generator().manual_seed(0)
    x0, eps = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    xt = sds.add_noise(x0, eps, torch.tensor([142]), acp)
    prev = rf.ddim_step(xt, eps, 142, acp)
    assert torch.allclose(prev, acp[122].sqrt() * x0 + (1 - acp[122]).sqrt() * eps, atol=1e-5)
    last = rf.ddim_step(sds.add_noise(x0, eps, torch.tensor([0]), acp), eps, 0, acp)          # prev < 0 -> alphas[0]
    assert torch.allclose(last, acp[0].sqrt() * x0 + (1 - acp[0]).sqrt() * eps, atol=1e-5)
    # attention state machine against a direct restatement with torch SDPA
    att = init_for_benchmark(Attention(64, None, 8), 3)
    ctl = rf.RefineController(state="refine", total_denoise_step=2, lambda_self=0.55)
    att.refine = rf.RefineAttentionState(ctl)
    xs = {n: [torch.randn(2, 16, 64, generator=g) for _ in range(2)] for n in ("front", "left", "k0", "v1")}

    def plain(q_src, kv_src):
        q, k, v = att.to_q(q_src), att.to_k(kv_src), att.to_v(kv_src)
        h = F.scaled_dot_product_attention(att._split(q), att._split(k), att._split(v)).transpose(1, 2).reshape(q.shape)
        return h
    outs = {}
    for name in ("front", "left", "k0", "v1"):
        ctl.cur_view_name = name
        att.refine.stored_zt[name] = []
        if name == "v1":
            ctl.cur_key_view_name_pair, ctl.cur_key_view_weight_pair = rf.KEY_VIEW_NAME_PAIR[name], rf.KEY_VIEW_WEIGHT_PAIR[name]
        outs[name] = [att(xs[name][s]) for s in range(2)]
        assert att.refine.cur_denoise_step == 0                      # wraps after total_denoise_step calls
    for s in range(2):
        assert torch.allclose(outs["front"][s], att.to_out(plain(xs["front"][s], xs["front"][s])), atol=1e-5)
        mutual = att.to_out(plain(xs["k0"][s], torch.cat([xs["k0"][s], xs["front"][s]], dim=1)))
        assert torch.allclose(outs["k0"][s], mutual, atol=1e-5)
        x = xs["v1"][s]
        blend = 0.55 * plain(x, x) + 0.45 * (0.5 * plain(x, xs["left"][s]) + 0.5 * plain(x, xs["k0"][s]))
        assert torch.allclose(outs["v1"][s], att.to_out(blend), atol=1e-5)
    assert "v1" not in att.refine.stored_zt or att.refine.stored_zt["v1"] == []       # non-key views store nothing
    ctl.state = "normal"
    assert torch.allclose(att(xs["front"][0]), att.to_out(plain(xs["front"][0], xs["front"][0])), atol=1e-5)


def test_checkpoint_key_translation_tables():
    """diffusers <-> this repo's parameter names: the tensor counts are diffusers' own (686 / 340 / 108 + 140 = 248 /
    288 IP-Adapter-FaceID entries), every translated key is unique, well-known diffusers keys are hit, and a
    diffusers-format dict round-trips through the loader."""
    from gaussianip_amd.guidance import checkpoints as ck
    from gaussianip_amd.guidance.networks import ControlNet, UNet, VAEDecoder, VAEEncoder
    with torch.device("meta"):
        nets = {"unet": UNet(0, False, 1.0), "controlnet": ControlNet(), "vae_encoder": VAEEncoder(), "vae_decoder": VAEDecoder()}
        full = UNet(128, True, 0.5)
    counts = {"unet": 686, "controlnet": 340, "vae_encoder": 108, "vae_decoder": 140}
    known = {
        "unet": ["time_embedding.linear_1.weight", "conv_in.bias", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias",
                 "down_blocks.3.resnets.1.time_emb_proj.weight", "down_blocks.2.downsamplers.0.conv.weight",
                 "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight", "mid_block.resnets.1.conv2.bias",
                 "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.3.attentions.2.proj_out.weight",
                 "up_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight", "up_blocks.2.resnets.2.conv_shortcut.weight",
                 "up_blocks.3.attentions.1.transformer_blocks.0.ff.net.2.bias", "conv_norm_out.weight", "conv_out.weight"],
        "controlnet": ["controlnet_cond_embedding.conv_in.weight", "controlnet_cond_embedding.blocks.5.bias",
                       "controlnet_cond_embedding.conv_out.weight", "controlnet_down_blocks.11.weight", "controlnet_mid_block.bias",
                       "down_blocks.1.attentions.1.transformer_blocks.0.norm3.weight", "mid_block.attentions.0.proj_in.weight"],
        "vae_encoder": ["encoder.conv_in.weight", "encoder.down_blocks.1.resnets.0.conv_shortcut.weight",
                        "encoder.down_blocks.2.downsamplers.0.conv.bias", "encoder.mid_block.attentions.0.group_norm.weight",
                        "encoder.mid_block.attentions.0.to_out.0.weight", "encoder.conv_norm_out.bias", "quant_conv.weight"],
        "vae_decoder": ["post_quant_conv.bias", "decoder.conv_in.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight",
                        "decoder.up_blocks.0.upsamplers.0.conv.weight", "decoder.up_blocks.3.resnets.2.norm2.bias",
                        "decoder.mid_block.attentions.0.to_q.bias", "decoder.conv_out.bias"],
    }
    for kind, net in nets.items():
        keys = list(net.state_dict().keys())
        mapped = [ck.diffusers_key(kind, k) for k in keys]
        assert len(keys) == counts[kind] and len(set(mapped)) == len(mapped), kind
        assert not [k for k in known[kind] if k not in set(mapped)], (kind, [k for k in known[kind] if k not in set(mapped)])
        assert not [m for m in mapped if re_search_ours(m)], kind
    # IP-Adapter-FaceID: 32 processors x 4 LoRA pairs + 16 x (to_k_ip, to_v_ip); cross-attention processors are the odd ones
    table = ck.ip_adapter_key_map(full)
    assert len(table) == 288 and len(set(table.values())) == 288
    assert table["1.to_k_ip.weight"] == "down_attn.0.block.attn2.to_k_ip.weight"
    assert table["0.to_q_lora.down.weight"] == "down_attn.0.block.attn1.lora_q.0.weight"
    assert table["12.to_out_lora.up.weight"] == "up_attn.3.block.attn1.lora_out.1.weight"          # first up-block processor
    assert table["31.to_v_ip.weight"] == "mid_attn.block.attn2.to_v_ip.weight"                    # mid block comes last
    assert set(table.values()) == {k for k in full.state_dict() if ".lora_" in k or "_ip." in k}
    # round trip on a small real module: VAE encoder
    src = VAEEncoder()
    fake = {ck.diffusers_key("vae_encoder", k): v.clone() for k, v in src.state_dict().items()}
    old = {k.replace("attentions.0.to_q.", "attentions.0.query.").replace("attentions.0.to_out.0.", "attentions.0.proj_attn."): v
           for k, v in fake.items()}                                                              # pre-0.18 attention names
    for d in (fake, old):
        dst = VAEEncoder()
        assert ck.load_diffusers_state_dict(dst, d, "vae_encoder") == []
        assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    with pytest.raises(KeyError):
        ck.load_diffusers_state_dict(VAEEncoder(), {}, "vae_encoder")


def re_search_ours(mapped_key):
    """True if a translated key still contains one of this repo's own module names (an untranslated segment)."""
    import re
    return re.search(r"(down_res|down_attn|down_sample|up_res|up_attn|up_sample|mid_res|mid_attn|mid_norm|\.block\.|ff_in|ff_out|"
                     r"time_l[12]|cond_stem|zero_convs|mid_zero|^res\.|^down\.|^up\.)", mapped_key) is not None


def test_guidance_config_accepts_the_reference_yaml_section():
    """Every key of `system.guidance` in configs/exp.yaml:78-120 is accepted: per-step keys become fields, the rest is kept."""
    from gaussianip_amd.guidance import GuidanceConfig
    section = {k: None for k in (
        "batch_size enable_memory_efficient_attention grad_clip grad_clip_pixel grad_clip_threshold guidance_rescale guidance_scale "
        "image_encoder_faceid_path image_encoder_path ip_ckpt_faceid_v1_path ip_ckpt_faceid_v2_path ip_ckpt_path ipa_faceid_s_scale "
        "ipa_faceid_scale ipa_scale irr_pil_image_path lw_depth negative_prompt negative_prompt_faceid null_prompt original_size "
        "pil_image_faceid_path pose_controlnet_path pretrained_realistic_model_name_or_path pretrained_sd_model_name_or_path prompt "
        "target_size use_anpg use_ipa_faceid use_pose_controlnet vae_path view_dependent_prompting weighting_strategy").split()}
    section.update(batch_size=4, grad_clip_pixel=True, grad_clip_threshold=1.0, guidance_rescale=0.75, guidance_scale=7.5,
                   ipa_faceid_scale=0.5, ipa_scale=0.6, use_anpg=True, use_ipa_faceid=True, use_pose_controlnet=True,
                   view_dependent_prompting=True, weighting_strategy="sds")
    cfg = GuidanceConfig.from_dict(section)
    assert cfg.use_anpg and cfg.guidance_rescale == 0.75 and cfg.grad_clip_threshold == 1.0 and "vae_path" in cfg.extra
    with pytest.raises(KeyError):
        GuidanceConfig.from_dict({"not_a_reference_key": 1})


def test_lpips_vgg_restatement_properties_and_state_dict_names():
    """guidance/perceptual.py (GaussianIP.py:121,433-436): LPIPS-VGG algebra against a second, loop-written statement
    of the published algorithm on the same weights; identity / symmetry; the lpips package's parameter names."""
    import torch.nn.functional as F
    from gaussianip_amd.guidance.perceptual import LPIPSVGG
    m = LPIPSVGG().init_for_benchmark(3)
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(2, 3, 40, 28, generator=g), torch.rand(2, 3, 40, 28, generator=g)
    d = m(a, b, normalize=True)
    assert d.shape == (2, 1, 1, 1) and float(d.min()) > 0
    assert float(m(a, a, normalize=True).abs().max()) == 0.0
    assert torch.allclose(m(b, a, normalize=True), d)
    # direct restatement: torchvision layer order 0..28 with ReLU after each conv and pools before conv 5, 10, 17, 24
    sd = m.state_dict()
    def feats(x):
        x = ((2 * x - 1) - sd["scaling_layer.shift"]) / sd["scaling_layer.scale"]
        out = []
        for k, ids in enumerate(((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))):
            if k:
                x = F.max_pool2d(x, 2, 2)
            for i in ids:
                x = F.relu(F.conv2d(x, sd["net.slice%d.%d.weight" % (k + 1, i)], sd["net.slice%d.%d.bias" % (k + 1, i)], padding=1))
            out.append(x)
        return out
    want = 0
    for k, (fa, fb) in enumerate(zip(feats(a), feats(b))):
        na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        want = want + F.conv2d((na - nb) ** 2, sd["lin%d.model.1.weight" % k]).mean((2, 3), keepdim=True)
    assert torch.allclose(d, want, rtol=1e-5, atol=1e-7)
    names = set(sd)
    assert len(names) == 2 + 26 + 5 and {"net.slice3.14.bias", "net.slice5.28.weight", "lin4.model.1.weight"} <= names
    # cached target features give the same distance, and the gradient reaches the first argument only
    a.requires_grad_(True)
    tf = m.target_features(b, True, torch.float32)
    d2 = m.distance_to_features(a, tf, True)
    assert torch.allclose(d2, d, rtol=1e-6, atol=1e-8)
    d2.sum().backward()
    assert float(a.grad.abs().max()) > 0 and all(not p.requires_grad for p in m.parameters())


# ---------------------------------------------------------------------------------------------------------------
# drop-in boundary of the guidance plugin: constructed from the cfg mapping and called exactly as
# threestudio/systems/GaussianIP.py:355-356 (on_fit_start) and :362-373 (training_step) do
# ---------------------------------------------------------------------------------------------------------------
class _TinyVAE(torch.nn.Module):
    scaling_factor = 0.18215

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 8, stride=8)

    def encode(self, x, generator=None):
        return self.conv(x) * self.scaling_factor


class _TinyControlNet(torch.nn.Module):
    def embed_condition(self, cond):
        return torch.nn.functional.adaptive_avg_pool2d(cond, 64).mean(1, keepdim=True)

    def forward(self, x, t, ctx, cond, scale=1.0, cond_embedding=None, replicas=1):
        c = cond_embedding
        if c.shape[0] != x.shape[0]:
            c = c.repeat(x.shape[0] // c.shape[0], 1, 1, 1)
        return [c], c


class _TinyUNet(torch.nn.Module):
    def fold_lora(self, scale=1.0):
        return self

    def forward(self, x, t, ctx, down=None, mid=None, replicas=1):
        out = 0.9 * x + ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1) + 1e-3 * t.view(-1, 1, 1, 1).to(x.dtype)
        return out if mid is None else out + 0.1 * mid


_EXP_YAML_GUIDANCE = dict(     # configs/exp.yaml:78-120 (paths shortened), half precision off for the CPU run
    batch_size=4, enable_memory_efficient_attention=True, grad_clip=[0, 1.5, 2.0, 1000], grad_clip_pixel=True,
    grad_clip_threshold=1.0, guidance_rescale=0.75, guidance_scale=7.5, image_encoder_faceid_path="/p/clip",
    image_encoder_path="/p/enc", ip_ckpt_faceid_v1_path="/p/v1.bin", ip_ckpt_faceid_v2_path="/p/v2.bin", ip_ckpt_path="/p/ip.bin",
    ipa_faceid_s_scale=0.4, ipa_faceid_scale=0.5, ipa_scale=0.6, irr_pil_image_path="/p/irr.png", lw_depth=0.5,
    negative_prompt="cloned face", negative_prompt_faceid="cloned face", null_prompt="", original_size=1024,
    pil_image_faceid_path="/p/face.png", pose_controlnet_path="/p/cn", pretrained_realistic_model_name_or_path="/p/rv",
    pretrained_sd_model_name_or_path="/p/sd", prompt="Audrey Hepburn wearing a tailored blazer", target_size=1024, use_anpg=True,
    use_ipa_faceid=True, use_pose_controlnet=True, vae_path="/p/vae", view_dependent_prompting=True, weighting_strategy="sds",
    half_precision_weights=False)


def test_guidance_plugin_is_called_like_the_reference_system_calls_it():
    from gaussianip_amd.guidance import StableDiffusionGuidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    from gaussianip_amd.guidance.prompts import PromptProcessor
    g = torch.Generator().manual_seed(0)
    tokens = (torch.randn(1, 4, 768, generator=g) * 0.1, torch.zeros(1, 4, 768), torch.randn(1, 4, 768, generator=g) * 0.1)
    mk = lambda provider: StableDiffusionGuidance(  # noqa: E731        threestudio.find("ipa-guidance")(cfg)
        _EXP_YAML_GUIDANCE, device="cpu", unet=_TinyUNet(), controlnet=_TinyControlNet(), vae=_TinyVAE(),
        schedule=AHDSSchedule(list(range(799, -1, -1)) * 3), image_embeds_provider=provider)
    guidance = mk(lambda gd: tokens)
    assert guidance.registry_name == "ipa-guidance"
    cfg = guidance.cfg
    assert (cfg.use_anpg, cfg.grad_clip_pixel, cfg.grad_clip_threshold, cfg.ipa_faceid_scale, cfg.guidance_rescale) == (True, True, 1.0, 0.5, 0.75)
    assert cfg.extra["pose_controlnet_path"] == "/p/cn"
    prompt_processor = PromptProcessor(_EXP_YAML_GUIDANCE["prompt"], lambda texts: torch.stack(
        [torch.full((77, 768), 0.01 * len(t)) for t in texts]), negative_prompt="cloned face", null_prompt="")
    # GaussianIP.py:355-356
    guidance.prepare_for_sds(prompt_processor.prompt, prompt_processor.negative_prompt, prompt_processor.null_prompt)
    assert guidance.pos_image_embeds.shape == (4, 4, 768) and guidance.num_samples == 4
    # GaussianIP.py:362-373
    B = 4
    images = torch.rand(B, 64, 64, 3, generator=g, requires_grad=True)
    control_images = torch.rand(B, 32, 32, 3, generator=g)
    all_vis_all = torch.tensor([1.0, 1.0, 0.0, 1.0])
    batch = dict(elevation=torch.tensor([5.0, -10.0, 20.0, 0.0]), azimuth=torch.tensor([30.0, -100.0, 150.0, 90.0]),
                 center=torch.tensor([0.0, 0.65, 0.0, 0.65]), camera_distances=torch.full((B,), 1.5),
                 c2w=torch.eye(4).expand(B, 4, 4), fovy=torch.full((B,), 1.2), height=64, width=64)    # extra keys are absorbed
    prompt_utils = prompt_processor()
    guidance_out = guidance(10, images, control_images, prompt_utils, True, all_vis_all, **batch)
    assert set(guidance_out) == {"loss_sds", "grad_norm"}
    guidance_out["loss_sds"].backward()
    assert torch.isfinite(images.grad).all() and float(images.grad.abs().max()) > 0
    # a cfg mapping that omits keys falls back to the reference Config's defaults (ipa_guidance.py:74-123)
    bare = StableDiffusionGuidance({"half_precision_weights": False}, device="cpu", unet=_TinyUNet(), controlnet=_TinyControlNet(),
                                   vae=_TinyVAE(), schedule=AHDSSchedule(list(range(2400))))
    assert (bare.cfg.use_anpg, bare.cfg.grad_clip_pixel, bare.cfg.grad_clip_threshold, bare.cfg.ipa_faceid_scale) == (False, False, 0.1, 0.6)
    with pytest.raises(RuntimeError, match="image-prompt tokens"):
        bare.prepare_for_sds("p", "n", "")                      # nothing to make the image tokens from: loud failure
    with pytest.raises(KeyError):
        StableDiffusionGuidance({"no_such_key": 1}, device="cpu", unet=_TinyUNet(), controlnet=_TinyControlNet(), vae=_TinyVAE())


def test_prompt_table_gather_equals_the_concatenated_embeddings():
    """`_prompt_embeds_anpg_table` (one gather from a [3 * 13, 81, 768] table, built once) against `_prompt_embeds(..., 3)`
    (ipa_guidance.py:462-470: two table lookups and four concatenations per step) for every view direction, with the
    image-prompt tokens tiled by prepare_for_sds; a replaced token set drops the table."""
    from gaussianip_amd.guidance import StableDiffusionGuidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    from gaussianip_amd.guidance.prompts import PromptProcessor
    g = torch.Generator().manual_seed(3)
    tokens = (torch.randn(1, 4, 768, generator=g) * 0.1, torch.zeros(1, 4, 768), torch.randn(1, 4, 768, generator=g) * 0.1)
    gd = StableDiffusionGuidance(_EXP_YAML_GUIDANCE, device="cpu", unet=_TinyUNet(), controlnet=_TinyControlNet(), vae=_TinyVAE(),
                                 schedule=AHDSSchedule(list(range(2400))), image_embeds_provider=lambda _: tokens)
    pp = PromptProcessor("a person", lambda texts: torch.stack([torch.randn(77, 768, generator=g) for _ in texts]), negative_prompt="blurry")
    gd.prepare_for_sds(pp.prompt, pp.negative_prompt, pp.null_prompt)
    B = 4
    for az, cen, vis in (([30.0, -100.0, 150.0, 90.0], [0.0, 0.65, 0.0, 0.65], [1.0, 1.0, 0.0, 1.0]),
                         ([-170.0, 10.0, -45.0, 179.0], [0.65, 0.65, 0.0, 0.0], [0.0, 1.0, 1.0, 0.0])):
        args = (torch.zeros(B), torch.tensor(az), torch.tensor(cen), torch.tensor(vis), torch.full((B,), 1.5))
        ref = gd._prompt_embeds(pp(), *args, 3).to(gd.weights_dtype)
        tab = gd._prompt_embeds_anpg_table(pp(), *args)
        assert tab is not None and tab.shape == ref.shape == (3 * B, 81, 768) and torch.equal(tab, ref)
    built = gd._embed_table[1]
    gd._prompt_embeds_anpg_table(pp(), *args)
    assert gd._embed_table[1] is built                              # same sources: the table is reused
    gd.set_image_embeds(tokens[0] * 2, tokens[1], tokens[2])         # [1, 4, 768] rows replaced after prepare_for_sds
    tab2 = gd._prompt_embeds_anpg_table(pp(), *args)
    assert tab2 is not None and torch.equal(tab2, gd._prompt_embeds(pp(), *args, 3).to(gd.weights_dtype)) and gd._embed_table[1] is not built


def test_stage_three_densify_schedule_matches_reference_quirks(tmp_path):
    """GaussianIP.on_before_optimizer_step, stage 3 branch (GaussianIP.py:476-506): max_radii2D adopted at step 0,
    statistics every step, ONE densify_and_prune at global step 2500 (stage step 100) with the screen-size limit off
    (the test at :491 uses the stage's own step) and min opacity 0.05, a second accumulation per step in the
    2500 < global < 3000 window, and a prune_only that never fires because of the operator precedence at :504.
    Also the hand-off files before_refine.pth / after_refine.pth (GaussianIP.py:404, refine.py:307-315, GaussianIP.py:359)."""
    from gaussianip_amd import system as S

    class FakeGaussian:
        def __init__(self, n):
            self.max_radii2D = torch.zeros(n)
            self.calls, self.stats, self.lr_steps = [], 0, []
            self.get_xyz = torch.zeros(n, 3)

        def add_densification_stats(self, grad, vis):
            assert grad.shape == (self.max_radii2D.shape[0], 3)
            self.stats += 1

        def densify_and_prune(self, *a):
            self.calls.append(("densify_and_prune",) + a)

        def prune_only(self, **k):
            self.calls.append(("prune_only", k))

        def update_learning_rate(self, it):
            self.lr_steps.append(it)

    n = 7
    g = FakeGaussian(n)
    st = S.StageThreeStep(g, None, None, [], refined_rgbs_small=torch.zeros(4, 3, 8, 8), cfg=S.StageOneConfig())
    st.refine_radii = torch.arange(n, dtype=torch.int32)
    st.refine_visibility_filter = st.refine_radii > 2
    st.viewspace_points = type("VS", (), {"grad": torch.ones(4, n, 3)})()
    actions = {}
    for step in range(0, 700):
        before = g.stats
        a = st.on_before_optimizer_step(step)
        if a:
            actions[step] = a
        per_step = g.stats - before
        assert per_step == (2 if 100 < step < 600 else 1), (step, per_step)
        if step == 0:
            assert torch.equal(g.max_radii2D, st.refine_radii.float())
    assert actions == {100: "densify_and_prune"}
    assert g.calls == [("densify_and_prune", 0.0002, 0.05, 4.0, None, 0.015)]            # no prune_only: `a + b % c == 0`
    # hand-off files
    imgs, ctrl = torch.rand(32, 6, 5, 3), torch.rand(32, 6, 5, 3)
    S.save_before_refine(str(tmp_path / "before_refine.pth"), imgs, ctrl)
    i2, c2 = S.load_before_refine(str(tmp_path / "before_refine.pth"))
    assert torch.equal(i2, imgs) and torch.equal(c2, ctrl)
    from gaussianip_amd.guidance.refine import VIEW_IDX_ALL
    refined = torch.rand(32, 1024, 1024, 3)
    S.save_after_refine(str(tmp_path / "after_refine.pth"), refined, VIEW_IDX_ALL)
    small = S.load_after_refine(str(tmp_path / "after_refine.pth"))
    assert small.shape == (32, 3, 415, 290)                                             # [60:890, 220:800] at half resolution
    idx_mapper = [3, 20, 21, 22, 6, 23, 24, 25, 1, 26, 27, 28, 7, 29, 30, 31, 2, 8, 9, 10, 4, 11, 12, 13, 0, 14, 15, 16, 5, 17, 18, 19]
    want = torch.nn.functional.interpolate(refined[idx_mapper].permute(0, 3, 1, 2)[:, :, 60:890, 220:800], scale_factor=0.5,
                                           mode="bilinear", align_corners=False)     # refine.py:307-311 verbatim
    assert torch.equal(small, want)


# ---- round 3: SD1.5 / ControlNet / VAE pinned key by key against the PUBLIC diffusers layout -------------------------------
# The tables below are written from the published model configurations (runwayml/stable-diffusion-v1-5 unet/config.json:
# block_out_channels 320/640/1280/1280, layers_per_block 2, cross_attention_dim 768, attention_head_dim 8,
# use_linear_projection false; lllyasviel/control_v11p_sd15_openpose: conditioning_embedding_out_channels 16/32/96/256;
# stabilityai/sd-vae-ft-mse: block_out_channels 128/256/512/512, layers_per_block 2, latent_channels 4) and diffusers'
# module naming — independently of gaussianip_amd/guidance/networks.py, whose state dict must translate to exactly them.
def _resnet(t, pre, cin, cout, temb=1280):
    t[pre + "norm1.weight"] = t[pre + "norm1.bias"] = (cin,)
    t[pre + "conv1.weight"], t[pre + "conv1.bias"] = (cout, cin, 3, 3), (cout,)
    if temb:
        t[pre + "time_emb_proj.weight"], t[pre + "time_emb_proj.bias"] = (cout, temb), (cout,)
    t[pre + "norm2.weight"] = t[pre + "norm2.bias"] = (cout,)
    t[pre + "conv2.weight"], t[pre + "conv2.bias"] = (cout, cout, 3, 3), (cout,)
    if cin != cout:
        t[pre + "conv_shortcut.weight"], t[pre + "conv_shortcut.bias"] = (cout, cin, 1, 1), (cout,)


def _transformer(t, pre, c, ctx=768):
    t[pre + "norm.weight"] = t[pre + "norm.bias"] = (c,)
    t[pre + "proj_in.weight"], t[pre + "proj_in.bias"] = (c, c, 1, 1), (c,)
    b = pre + "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        t[b + n + ".weight"] = t[b + n + ".bias"] = (c,)
    for a, kv in (("attn1", c), ("attn2", ctx)):
        t[b + a + ".to_q.weight"] = (c, c)
        t[b + a + ".to_k.weight"] = t[b + a + ".to_v.weight"] = (c, kv)
        t[b + a + ".to_out.0.weight"], t[b + a + ".to_out.0.bias"] = (c, c), (c,)
    t[b + "ff.net.0.proj.weight"], t[b + "ff.net.0.proj.bias"] = (8 * c, c), (8 * c,)
    t[b + "ff.net.2.weight"], t[b + "ff.net.2.bias"] = (c, 4 * c), (c,)
    t[pre + "proj_out.weight"], t[pre + "proj_out.bias"] = (c, c, 1, 1), (c,)


def _sd15_encoder_side(t):
    chans = (320, 640, 1280, 1280)
    t["time_embedding.linear_1.weight"], t["time_embedding.linear_1.bias"] = (1280, 320), (1280,)
    t["time_embedding.linear_2.weight"], t["time_embedding.linear_2.bias"] = (1280, 1280), (1280,)
    t["conv_in.weight"], t["conv_in.bias"] = (320, 4, 3, 3), (320,)
    c = 320
    for i, w in enumerate(chans):
        for j in range(2):
            _resnet(t, "down_blocks.%d.resnets.%d." % (i, j), c, w)
            if i < 3:
                _transformer(t, "down_blocks.%d.attentions.%d." % (i, j), w)
            c = w
        if i < 3:
            t["down_blocks.%d.downsamplers.0.conv.weight" % i], t["down_blocks.%d.downsamplers.0.conv.bias" % i] = (w, w, 3, 3), (w,)
    _resnet(t, "mid_block.resnets.0.", 1280, 1280)
    _transformer(t, "mid_block.attentions.0.", 1280)
    _resnet(t, "mid_block.resnets.1.", 1280, 1280)


def sd15_unet_table():
    t = {}
    _sd15_encoder_side(t)
    rev = (1280, 1280, 640, 320)
    prev = 1280
    for i, out in enumerate(rev):
        inp = rev[min(i + 1, 3)]
        for j in range(3):
            skip = inp if j == 2 else out
            _resnet(t, "up_blocks.%d.resnets.%d." % (i, j), (prev if j == 0 else out) + skip, out)
            if i > 0:
                _transformer(t, "up_blocks.%d.attentions.%d." % (i, j), out)
        if i < 3:
            t["up_blocks.%d.upsamplers.0.conv.weight" % i], t["up_blocks.%d.upsamplers.0.conv.bias" % i] = (out, out, 3, 3), (out,)
        prev = out
    t["conv_norm_out.weight"] = t["conv_norm_out.bias"] = (320,)
    t["conv_out.weight"], t["conv_out.bias"] = (4, 320, 3, 3), (4,)
    return t


def sd15_controlnet_table():
    t = {}
    _sd15_encoder_side(t)
    emb = (16, 32, 96, 256)
    t["controlnet_cond_embedding.conv_in.weight"], t["controlnet_cond_embedding.conv_in.bias"] = (16, 3, 3, 3), (16,)
    for k in range(3):
        a, b = emb[k], emb[k + 1]
        t["controlnet_cond_embedding.blocks.%d.weight" % (2 * k)], t["controlnet_cond_embedding.blocks.%d.bias" % (2 * k)] = (a, a, 3, 3), (a,)
        t["controlnet_cond_embedding.blocks.%d.weight" % (2 * k + 1)], t["controlnet_cond_embedding.blocks.%d.bias" % (2 * k + 1)] = (b, a, 3, 3), (b,)
    t["controlnet_cond_embedding.conv_out.weight"], t["controlnet_cond_embedding.conv_out.bias"] = (320, 256, 3, 3), (320,)
    for k, c in enumerate((320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280)):
        t["controlnet_down_blocks.%d.weight" % k], t["controlnet_down_blocks.%d.bias" % k] = (c, c, 1, 1), (c,)
    t["controlnet_mid_block.weight"], t["controlnet_mid_block.bias"] = (1280, 1280, 1, 1), (1280,)
    return t


def _vae_mid(t, pre):
    _resnet(t, pre + "mid_block.resnets.0.", 512, 512, temb=0)
    a = pre + "mid_block.attentions.0."
    t[a + "group_norm.weight"] = t[a + "group_norm.bias"] = (512,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        t[a + n + ".weight"], t[a + n + ".bias"] = (512, 512), (512,)
    _resnet(t, pre + "mid_block.resnets.1.", 512, 512, temb=0)


def sd_vae_tables():
    enc, dec = {}, {}
    enc["encoder.conv_in.weight"], enc["encoder.conv_in.bias"] = (128, 3, 3, 3), (128,)
    c = 128
    for i, w in enumerate((128, 256, 512, 512)):
        for j in range(2):
            _resnet(enc, "encoder.down_blocks.%d.resnets.%d." % (i, j), c, w, temb=0)
            c = w
        if i < 3:
            enc["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i], enc["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (w, w, 3, 3), (w,)
    _vae_mid(enc, "encoder.")
    enc["encoder.conv_norm_out.weight"] = enc["encoder.conv_norm_out.bias"] = (512,)
    enc["encoder.conv_out.weight"], enc["encoder.conv_out.bias"] = (8, 512, 3, 3), (8,)
    enc["quant_conv.weight"], enc["quant_conv.bias"] = (8, 8, 1, 1), (8,)
    dec["post_quant_conv.weight"], dec["post_quant_conv.bias"] = (4, 4, 1, 1), (4,)
    dec["decoder.conv_in.weight"], dec["decoder.conv_in.bias"] = (512, 4, 3, 3), (512,)
    _vae_mid(dec, "decoder.")
    c = 512
    for i, w in enumerate((512, 512, 256, 128)):
        for j in range(3):
            _resnet(dec, "decoder.up_blocks.%d.resnets.%d." % (i, j), c, w, temb=0)
            c = w
        if i < 3:
            dec["decoder.up_blocks.%d.upsamplers.0.conv.weight" % i], dec["decoder.up_blocks.%d.upsamplers.0.conv.bias" % i] = (w, w, 3, 3), (w,)
    dec["decoder.conv_norm_out.weight"] = dec["decoder.conv_norm_out.bias"] = (128,)
    dec["decoder.conv_out.weight"], dec["decoder.conv_out.bias"] = (3, 128, 3, 3), (3,)
    return enc, dec


def test_sd15_architecture_is_pinned_key_by_key_and_by_parameter_count():
    """Every parameter of the repo's U-Net / ControlNet / VAE translates to a diffusers key of the public SD1.5 layout with
    the public shape — the full key -> shape tables, not a handful of names — and the totals are the published ones:
    UNet2DConditionModel 859 520 964 parameters in 686 tensors, ControlNetModel 361 279 120 in 340, AutoencoderKL
    83 653 863 in 248 (encoder + quant_conv 34 163 664, decoder + post_quant_conv 49 490 199); the IP-Adapter-FaceID
    processors add 25 509 888 (rank-128 LoRA on q / k / v / out of 32 attentions) + 19 169 280 (16 x to_k_ip / to_v_ip)."""
    from math import prod
    from gaussianip_amd.guidance import checkpoints as ck
    from gaussianip_amd.guidance.networks import ControlNet, UNet, VAEDecoder, VAEEncoder
    with torch.device("meta"):
        nets = {"unet": UNet(0, False, 1.0), "controlnet": ControlNet(), "vae_encoder": VAEEncoder(), "vae_decoder": VAEDecoder()}
        full = UNet(128, True, 0.5)
    enc_t, dec_t = sd_vae_tables()
    tables = {"unet": sd15_unet_table(), "controlnet": sd15_controlnet_table(), "vae_encoder": enc_t, "vae_decoder": dec_t}
    totals = {"unet": (859520964, 686), "controlnet": (361279120, 340), "vae_encoder": (34163664, 108), "vae_decoder": (49490199, 140)}
    for kind, net in nets.items():
        ours = {ck.diffusers_key(kind, k): tuple(v.shape) for k, v in net.state_dict().items()}
        want = tables[kind]
        assert set(ours) == set(want), (kind, sorted(set(ours) ^ set(want))[:6])
        wrong = {k: (ours[k], want[k]) for k in want if ours[k] != want[k]}
        assert not wrong, (kind, list(wrong.items())[:4])
        assert (sum(prod(s) for s in want.values()), len(want)) == totals[kind], kind
        assert (sum(p.numel() for p in net.parameters()), len(list(net.parameters()))) == totals[kind], kind
    assert totals["vae_encoder"][0] + totals["vae_decoder"][0] == 83653863
    extra = sum(p.numel() for n, p in full.named_parameters() if ".lora_" in n or "_ip." in n)
    lora = sum(p.numel() for n, p in full.named_parameters() if ".lora_" in n)
    assert (lora, extra - lora) == (25509888, 19169280)
    assert sum(p.numel() for p in full.parameters()) == 859520964 + 44679168
