"""SDS / ANPG gradient algebra around the denoiser (everything except the networks).

Reference: threestudio/models/guidance/ipa_guidance.py — scheduler betas :139-147 (DDIM, scaled_linear
0.00085 -> 0.012, 1000 steps), add_noise (diffusers DDIMScheduler.add_noise), ANPG combine :411-416, weighting
:418-425, per-"pixel" clip :427-431 (L2 norm over the LAST axis = latent width), loss :645-653,
rescale_noise_cfg :41-53 (pinned by tests/golden/rescale_noise_cfg.npz).
"""
import torch
import torch.nn.functional as F


def per_sample(draw, n, generator):
    """`draw(k, g)` -> a tensor of k samples from generator g.  `generator` may be one torch.Generator (or None: the global
    one) — the reference's behaviour, one stream for the whole batch — or a LIST of n generators, one per batch row: every
    view then draws from its own stream, so a view's noise / timestep does not depend on which other views share its batch
    (the view-sharded step of BASELINE configs[3] draws, per view, what the unsharded 4-view step draws)."""
    if isinstance(generator, (list, tuple)):
        if len(generator) != n:
            raise ValueError("need one generator per batch row: %d for %d rows" % (len(generator), n))
        return torch.cat([draw(1, g) for g in generator], dim=0)
    return draw(n, generator)


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, device=None):
    """cumprod(1 - beta) for the "scaled_linear" schedule (linear in sqrt(beta)), float32 like diffusers."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).to(device)


def add_noise(latents, noise, t, acp):
    """x_t = sqrt(acp_t) x_0 + sqrt(1 - acp_t) eps."""
    a = acp.to(device=latents.device, dtype=latents.dtype)[t]
    sa = a.sqrt().view(-1, 1, 1, 1)
    sb = (1 - a).sqrt().view(-1, 1, 1, 1)
    return sa * latents + sb * noise


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def anpg_direction(noise_pred, t, guidance_scale=7.5, t_switch=170):
    """noise_pred [3B,4,h,w] ordered (neg | text | null) -> delta_c + delta_d  (ipa_guidance.py:411-416)."""
    eps_neg, eps_text, eps_null = noise_pred.chunk(3)
    B = eps_text.shape[0]
    delta_c = guidance_scale * (eps_text - eps_null)
    mask = (t < t_switch).int().view(B, 1, 1, 1)
    delta_d = mask * eps_null + (1 - mask) * (eps_null - eps_neg)
    return delta_c + delta_d


def cfg_direction(noise_pred, noise, guidance_scale=7.5, guidance_rescale=0.0):
    """Plain SDS (use_anpg = False, :443-519): noise_pred [2B,...] ordered (neg | pos) like the reference's batch
    (`final_prompt_embeds = cat([neg, pos])` :470, `noise_pred_neg, noise_pred_pos = chunk(2)` :495)."""
    eps_uncond, eps_text = noise_pred.chunk(2)
    eps = eps_uncond + guidance_scale * (eps_text - eps_uncond)
    if guidance_rescale > 0:
        eps = rescale_noise_cfg(eps, eps_text, guidance_rescale)
    return eps - noise


def sds_weight(t, acp, strategy="sds"):
    a = acp[t]
    if strategy == "sds":
        return (1 - a).view(-1, 1, 1, 1)
    if strategy == "uniform":
        return 1
    if strategy == "fantasia3d":
        return (a ** 0.5 * (1 - a)).view(-1, 1, 1, 1)
    raise ValueError("Unknown weighting strategy: %s" % strategy)


def clip_grad_pixel(grad, threshold):
    """The reference's "pixel" clip: L2 norm over the last axis, clamped to `threshold` (:427-431)."""
    n = torch.norm(grad, dim=-1, keepdim=True) + 1e-8
    return n.clamp(max=threshold) * grad / n


def sds_loss(latents, grad):
    """0.5 * || latents - (latents - grad).detach() ||^2 / B: d loss / d latents == grad."""
    grad = torch.nan_to_num(grad)
    lat32 = latents.float()   # the reference evaluates this under autocast, where mse_loss runs in float32
    target = (lat32 - grad.float()).detach()
    return 0.5 * F.mse_loss(lat32, target, reduction="sum") / latents.shape[0], grad
