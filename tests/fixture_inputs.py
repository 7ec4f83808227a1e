"""Seeded INPUTS of the guidance golden fixtures, shared by tools/make_golden.py (which feeds them to the reference's own
functions in the build container) and by the tests (which feed them to this repo's code).  The fixtures under
tests/golden/ then only need to hold the reference's OUTPUTS.  numpy's PCG64 streams are stable across platforms and
numpy versions, so the inputs are identical wherever they are regenerated.  Nothing here comes from the reference."""
import numpy as np
import torch


def _n(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def attn_weights(seed, dim, ctx_dim=None, rank=128, ip=False):
    """Weights of one attention layer with LoRA branches (and the IP-Adapter key / value projections when `ip`)."""
    rng = np.random.default_rng(seed)
    ctx = ctx_dim or dim
    w = dict(to_q=_n(rng, dim, dim, scale=dim ** -0.5), to_k=_n(rng, dim, ctx, scale=ctx ** -0.5),
             to_v=_n(rng, dim, ctx, scale=ctx ** -0.5), to_out_w=_n(rng, dim, dim, scale=dim ** -0.5),
             to_out_b=_n(rng, dim, scale=0.1))
    for name, cin in (("q", dim), ("k", ctx), ("v", ctx), ("out", dim)):
        w["lora_%s_down" % name] = _n(rng, rank, cin, scale=cin ** -0.5)
        w["lora_%s_up" % name] = _n(rng, dim, rank, scale=0.5 * rank ** -0.5)
    if ip:
        w["to_k_ip"] = _n(rng, dim, ctx, scale=ctx ** -0.5)
        w["to_v_ip"] = _n(rng, dim, ctx, scale=ctx ** -0.5)
    return w


def attn_tokens(seed, batch, tokens, dim, scale=1.0):
    return _n(np.random.default_rng(seed), batch, tokens, dim, scale=scale)


# the refine-state scenario of the self-attention fixture: (view name, neighbour pair, weight pair); two denoising steps
REFINE_VIEWS = [("front", None, None), ("back", None, None), ("left", None, None), ("k0", None, None), ("k2", None, None),
                ("v3", ("k0", "front"), (0.75, 0.25)), ("v16", ("k2", "back"), (0.5, 0.5)), ("v1", ("left", "k0"), (0.25, 0.75))]
REFINE_STEPS = 2
REFINE_TOKENS = 16
REFINE_LAMBDA_SELF = 0.55


def refine_tokens(view_index, step, batch, tokens, dim):
    return attn_tokens(1000 + 10 * view_index + step, batch, tokens, dim)


# ---- SDS / ANPG fixture ----
def sds_case(seed, B=4, h=16, w=16):
    rng = np.random.default_rng(seed)
    return dict(
        latents=_n(rng, B, 4, h, w), control=rng.uniform(0, 1, (B, 3, 16, 16)).astype(np.float32),
        t=np.array([100, 169, 170, 640][:B], dtype=np.int64),
        text_vd=_n(rng, 13, 77, 768, scale=0.1), uncond_vd=_n(rng, 13, 77, 768, scale=0.1), null=_n(rng, 1, 77, 768, scale=0.1),
        text=_n(rng, 1, 77, 768, scale=0.1), uncond=_n(rng, 1, 77, 768, scale=0.1),
        pos_image=_n(rng, B, 4, 768, scale=0.1), neg_image=np.zeros((B, 4, 768), np.float32), null_image=_n(rng, B, 4, 768, scale=0.1),
        elevation=np.array([5.0, -10.0, 20.0, 0.0][:B], np.float32), azimuth=np.array([30.0, -100.0, 150.0, 90.0][:B], np.float32),
        center=np.array([0.0, 0.65, 0.0, 0.65][:B], np.float32), all_vis_all=np.array([1.0, 1.0, 0.0, 1.0][:B], np.float32),
        camera_distances=np.array([1.5, 1.4, 1.6, 1.3][:B], np.float32))


def fake_forward_unet(noisy_latents, control_img, t, encoder_hidden_states, *args, **kwargs):
    """A deterministic stand-in for ControlNet -> U-Net with the same call signature as forward_unet
    (ipa_guidance.py:311-358): every input reaches the output, so the [neg | pos | null] batch layout, the
    [77 text | 4 image] token layout, the timestep tiling and the pose-map tiling are all visible in the result."""
    n = noisy_latents.shape[0]
    ctx = encoder_hidden_states.float()
    text = torch.tanh(ctx[:, :77, :4].mean(1)).view(n, 4, 1, 1)
    image = ctx[:, 77:, 4:8].mean(1).view(n, 4, 1, 1)
    ctrl = control_img.float()
    if ctrl.shape[0] != n:            # this repo hands over the B distinct pose maps; the reference tiles them to 3B
        ctrl = ctrl.repeat(n // ctrl.shape[0], 1, 1, 1)
    c = ctrl.mean(dim=(1, 2, 3)).view(n, 1, 1, 1)
    tt = (t.float() / 1000.0).view(n, 1, 1, 1)
    return (0.8 * noisy_latents.float() + 0.5 * text + 2.0 * image + 0.3 * c + 0.2 * tt).to(noisy_latents.dtype)


def direction_grid():
    """(elevation, azimuth, center, all_vis_all, camera_distances) probes for the view-dependent prompt lookup: a sweep
    of azimuths including every window boundary, both visibility states, head-zoom centre or not."""
    az = np.concatenate([np.arange(-180.0, 180.1, 7.5), np.array([0.0, 45.0, 135.0, -45.0, -135.0, 0.5, 44.9, 45.1, 179.9, -179.9])])
    rows = []
    for vis in (0.0, 1.0):
        for cent in (0.0, 0.65):
            for a in az:
                rows.append((10.0, a, cent, vis, 1.5))
    g = np.array(rows, dtype=np.float32)
    return g[:, 0].copy(), g[:, 1].copy(), g[:, 2].copy(), g[:, 3].copy(), g[:, 4].copy()
