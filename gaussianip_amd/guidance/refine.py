"""View-consistent refinement (VCR) of the 32 orbit renders and its attention state machine.

Mirrors `refine_rgb` (threestudio/models/guidance/refine.py:115-239), `IPAdapterFaceID.refine_with_small_noise`
(ip_adapter_faceid.py:451-515) and the denoising loop of `__call_refine__` (pipeline_ipa_controlnet.py:1447-1877):
every render is VAE-encoded at full resolution, noised with ONE shared noise tensor to the first of the last 8 of 50
DDIM timesteps (142 ... 0), denoised for those 8 steps with ControlNet + U-Net under classifier-free guidance 7.5, and decoded.
The nine self-attentions of up_blocks.1-3 run in the 'refine' state (`networks.Attention._forward_refine`): the four
canonical views store their tokens, the four diagonal key views attend mutually with front / back, all other views blend
their own attention with attentions over their two neighbouring key views (weights 0.75/0.5/0.25, lambda_self 0.55).

Text embeddings are inputs (`prompt_fn(view_name) -> (cond [1,77,768], uncond [1,77,768])`; the reference appends
", back view" etc. to the prompt for the eight key views, refine.py:121-131 — the tokenizer / text encoder is the
caller's).  No diffusers dependency; DDIM is stated here (eta = 0, clip_sample False, set_alpha_to_one False,
`prev = t - 1000 // 50`, as the reference configures it at refine.py:62-70 and calls it at pipeline :1710,1841).
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import sds
from .networks import IP_TOKENS, Attention, UNet

# refine.py:116-117 (order in which the views are refined; index into the 32-view orbit), :133-145 (neighbour tables)
VIEW_IDX_ALL = [24, 8, 16, 0, 20, 28, 4, 12, 17, 18, 19, 21, 22, 23, 25, 26, 27, 29, 30, 31, 1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15]
VIEW_NAME_ALL = ["front", "back", "left", "right", "k0", "k1", "k2", "k3"] + ["v%d" % i for i in range(24)]
_RING = ["left", "k0", "front", "k1", "right", "k2", "back", "k3", "left"]
KEY_VIEW_NAME_PAIR = {"v%d" % i: (_RING[i // 3], _RING[i // 3 + 1]) for i in range(24)}
KEY_VIEW_WEIGHT_PAIR = {"v%d" % i: ((0.75, 0.25), (0.5, 0.5), (0.25, 0.75))[i % 3] for i in range(24)}
PROMPT_SUFFIX = {"front": "", "back": ", back view", "left": ", left view", "right": ", right view", "k0": ", left front view",
                 "k1": ", right front view", "k2": ", right back view", "k3": ", left back view"}
NEGATIVE_PROMPT = "blurry face, bad face, poorly drawn face, duplicate face, extra fingers, blurry, fused fingers"


def refine_timesteps(num_steps=8, num_inference_steps=50, device=None):
    """torch.linspace(0, 999, 50, dtype=torch.int64).round().flip()[-num_steps:]  (refine.py:176-178).  The int64
    linspace TRUNCATES each sample (the .round() after it is a no-op), so the last eight are
    142, 122, 101, 81, 61, 40, 20, 0 — not the rounded 143, 122, 102, 82, 61, 41, 20, 0.  Pinned by
    tests/golden/refine_timesteps.npz (the reference's expression evaluated as written)."""
    ts = torch.linspace(0, 999, num_inference_steps, dtype=torch.int64).flip(dims=[0])
    return ts[-num_steps:].to(device) if device is not None else ts[-num_steps:]


def ddim_step(sample, eps, t, alphas, num_inference_steps=50, num_train_timesteps=1000):
    """DDIMScheduler.step with eta = 0, epsilon prediction, clip_sample False, set_alpha_to_one False."""
    prev = int(t) - num_train_timesteps // num_inference_steps
    a_t = alphas[int(t)].to(torch.float32)
    a_prev = (alphas[prev] if prev >= 0 else alphas[0]).to(torch.float32)
    x = sample.to(torch.float32)
    e = eps.to(torch.float32)
    x0 = (x - (1 - a_t).sqrt() * e) / a_t.sqrt()
    return (a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * e).to(sample.dtype)


@dataclass
class RefineController:
    """The fields the reference sets on every target processor (refine.py:161-171, 205-211), shared by reference."""
    state: str = "normal"
    total_denoise_step: int = 8
    lambda_self: float = 0.55
    cur_view_name: str = "front"
    cur_key_view_name_pair: Tuple[str, str] = ("left", "k0")
    cur_key_view_weight_pair: Tuple[float, float] = (0.5, 0.5)


@dataclass
class RefineAttentionState:
    """Per-layer part: the stored key-view tokens and the denoising-step counter (attention_processor_faceid.py:217-233)."""
    ctl: RefineController
    stored_zt: Dict[str, List[torch.Tensor]] = field(default_factory=dict)
    cur_denoise_step: int = 0


def target_attentions(unet: UNet) -> List[Attention]:
    """attn1 of up_blocks.{1,2,3}.attentions.{0,1,2} (refine.py:147-157) = up_attn[3..11] here."""
    return [unet.up_attn[k].block.attn1 for k in range(3, 12)]


class ViewConsistentRefiner:
    def __init__(self, guidance, vae_decoder, num_steps=8, lambda_self=0.55, guidance_scale=7.5, ip_scale=0.6):
        """`guidance`: a StableDiffusionGuidance (its unet / controlnet / vae encoder / alphas are used);
        `vae_decoder`: networks.VAEDecoder in the same dtype / layout."""
        self.g = guidance
        self.decoder = vae_decoder
        self.guidance_scale = guidance_scale
        self.ip_scale = ip_scale                      # refine_with_small_noise(scale=0.6): set_scale on the IP branches
        self.ctl = RefineController(total_denoise_step=num_steps, lambda_self=lambda_self)
        self.targets = target_attentions(guidance.unet)
        for a in self.targets:
            a.refine = RefineAttentionState(self.ctl)

    # ------------------------------------------------------------------ one view
    @torch.no_grad()
    def refine_latents(self, latents_noisy, embeds, control_img, timesteps):
        """8-step DDIM with classifier-free guidance; embeds = cat[uncond, cond] [2, 81, 768]; control_img [1,3,H,W]."""
        lat = latents_noisy
        hint = self.g.embed_control(control_img)      # timestep-independent: once per view, not once per DDIM step
        for t in timesteps:
            tt = t.reshape(1).expand(2)
            noise_pred = self.g.forward_unet(torch.cat([lat] * 2), None, tt, embeds, True, control_embedding=hint, replicas=2)
            uncond, text = noise_pred.float().chunk(2)
            noise_pred = uncond + self.guidance_scale * (text - uncond)
            lat = ddim_step(lat, noise_pred, t, self.g.alphas)
        return lat

    @torch.no_grad()
    def decode(self, latents):
        z = (latents / self.g.vae.scaling_factor).to(self.g.weights_dtype)
        if self.g.cfg.channels_last:
            z = z.contiguous(memory_format=torch.channels_last)
        img = self.decoder(z).float()
        return (img / 2 + 0.5).clamp(0, 1)            # image_processor.postprocess(output_type="pt", do_denormalize)

    # ------------------------------------------------------------------ all views
    @torch.no_grad()
    def refine_rgb(self, rgb, control_img, prompt_fn: Callable[[str], Tuple[torch.Tensor, torch.Tensor]],
                   image_embeds: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, views: Optional[Sequence[str]] = None,
                   generator=None):
        """rgb, control_img [n_views, H, W, 3] in orbit order -> (refined [len(views), H, W, 3], view_idx_all).
        `views` restricts the pass to a prefix-closed subset of VIEW_NAME_ALL (tests); default = all 32."""
        g = self.g
        names = list(VIEW_NAME_ALL if views is None else views)
        dev, dt = g.device, g.weights_dtype
        H, W = rgb.shape[1], rgb.shape[2]
        timesteps = refine_timesteps(self.ctl.total_denoise_step, 50, dev)
        noise = torch.randn((1, 4, H // 8, W // 8), device=dev, dtype=torch.float16, generator=generator)      # refine.py:182
        pos_img, neg_img = image_embeds if image_embeds is not None else (g.pos_image_embeds, g.neg_image_embeds)
        old_scales = [(a, a.ip_scale) for a in g.unet.modules() if isinstance(a, Attention) and a.ip]
        for a, _ in old_scales:
            a.ip_scale = self.ip_scale
        self.ctl.state = "refine"
        for a in self.targets:
            a.refine.stored_zt.clear()
            a.refine.cur_denoise_step = 0
        out = []
        try:
            for name in names:
                idx = VIEW_IDX_ALL[VIEW_NAME_ALL.index(name)]
                self.ctl.cur_view_name = name
                for a in self.targets:
                    a.refine.stored_zt[name] = []
                if "v" in name:
                    self.ctl.cur_key_view_name_pair = KEY_VIEW_NAME_PAIR[name]
                    self.ctl.cur_key_view_weight_pair = KEY_VIEW_WEIGHT_PAIR[name]
                cur = rgb[idx].permute(2, 0, 1)[None].to(dev)
                ctrl = control_img[idx].permute(2, 0, 1)[None].to(dev)
                lat = g.encode_images(cur.to(dt), generator)
                lat_noisy = sds.add_noise(lat, noise.to(lat.dtype), timesteps[:1], g.alphas)
                cond, uncond = prompt_fn(name)
                embeds = torch.cat([torch.cat([uncond.to(dev, dt), neg_img[:1]], dim=1),
                                    torch.cat([cond.to(dev, dt), pos_img[:1]], dim=1)], dim=0)
                assert embeds.shape[1] == 77 + IP_TOKENS
                lat = self.refine_latents(lat_noisy, embeds, ctrl, timesteps)
                out.append(self.decode(lat))
        finally:
            self.ctl.state = "normal"
            for a, sc in old_scales:
                a.ip_scale = sc
            for a in self.targets:
                a.refine.stored_zt.clear()
        refined = torch.cat(out, dim=0).permute(0, 2, 3, 1)
        return refined, [VIEW_IDX_ALL[VIEW_NAME_ALL.index(n)] for n in names]
