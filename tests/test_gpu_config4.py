"""BASELINE.json configs[4] as written: 1 M Gaussians (post-densify look), 1024 x 1024, the 36-view orbit (elevation 5,
distance 1.8, fovy 70; configs/exp.yaml:37-40) rendered in launch sets of 12, and the VCR refine pass at 1024 x 1024.

Orbit: two views of it are compared with the CPU oracle at full size (integer buffers bit-exact, images 1e-4 outside
proven knife-edge pixels); all 36 through size-independent properties (batched == single view bitwise, alpha in range,
every tile list sorted, determinism).  Refine: the three attention branches of the reference's state machine
(canonical / mutual / blended view, attention_processor_faceid.py:291-364) at 128 x 128 latents = 16 384 tokens."""
import numpy as np
import pytest
import torch

import scenes
from test_gpu_raster_parity import _check_forward_scene, _settings

pytestmark = pytest.mark.gpu
P, H, W, NV, SET = 1000000, 1024, 1024, 36, 12


def _scene():
    sc = scenes.make_scene("human", P, seed=42)
    sc["scales"] = (sc["scales"] / 1.6).astype(np.float32)     # gaussian_model.py:371: scales / 1.6 per split
    sc["opacities"][:] = 0.6
    return sc


def test_orbit_36_views_of_1m_gaussians(oracle):
    from gaussianip_amd import GaussianRasterizer, rasterize_views
    from gaussianip_amd import rasterizer as R
    sc = _scene()
    cams = [scenes.camera(5.0, -180.0 + 10.0 * i, 1.8, 70.0, H, W) for i in range(NV)]
    oracle.set_threads(oracle.max_threads())
    try:
        for i in (0, 13):                                      # full-size oracle comparison of two orbit views
            _check_forward_scene(oracle, sc, cams[i], H, W, 0, (0.0, 0.0, 0.0))
    finally:
        oracle.set_threads(1)
    t = {k: torch.from_numpy(v).cuda() for k, v in sc.items()}
    sts = [_settings(c, H, W, (0.0, 0.0, 0.0), 0) for c in cams]
    with torch.no_grad():
        colors, alphas = [], []
        for s in range(0, NV, SET):
            color, radii, depth, alpha = rasterize_views(t["means3D"], None, t["opacities"], sts[s:s + SET], shs=t["shs"],
                                                         scales=t["scales"], rotations=t["rotations"])
            assert torch.isfinite(color).all() and float(alpha.min()) >= 0 and float(alpha.max()) <= 1 + 1e-5
            assert float((color - 0.5 * alpha).abs().max()) < 2e-4          # grey Gaussians on black: colour = 0.5 alpha
            colors.append(color)
            alphas.append(alpha)
        colors = torch.cat(colors)
        for i in (5, 23, 35):                                  # a view of a launch set == the same view rendered alone
            c1 = GaussianRasterizer(sts[i])(means3D=t["means3D"], means2D=None, opacities=t["opacities"], shs=t["shs"],
                                            scales=t["scales"], rotations=t["rotations"])[0]
            assert torch.equal(c1, colors[i])
        # opposite views of the orbit see the same silhouette area to a few per cent (sanity of the camera path)
        a = torch.cat(alphas).mean(dim=(1, 2, 3))
        assert float(((a[:18] - a[18:]).abs() / a[:18]).max()) < 0.15
        (outs, plan) = R.forward_with_state(t["means3D"], t["opacities"], sts[:SET], shs=t["shs"], scales=t["scales"],
                                            rotations=t["rotations"])
        sv = R.state_views(plan)
        hdr = sv["header"].cpu().numpy()
        keys = sv["keys"][:int(hdr[1])]
        ts = sv["tile_start"].cpu().numpy().astype(np.int64)
        assert int(hdr[2]) == 0 and ts[-1] == int(hdr[1]) and int(hdr[3]) > 16384      # long lists: the beyond-LDS sort path
        seg = torch.zeros(int(hdr[1]), dtype=torch.bool, device="cuda")
        seg[torch.from_numpy(ts[:-1][ts[:-1] < ts[1:]]).cuda()] = True
        assert bool(((keys[1:] > keys[:-1]) | seg[1:]).all()), "a tile list is not strictly increasing in (depth, index)"


def test_vcr_refine_at_1024(monkeypatch):
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, fused, networks, refine as rf
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    gd = StableDiffusionGuidance(GuidanceConfig(), schedule=AHDSSchedule(list(range(2400))))
    dec = networks.init_for_benchmark(networks.VAEDecoder(), 5).to("cuda", torch.float16).eval().requires_grad_(False)
    dec = dec.to(memory_format=torch.channels_last)
    vcr = rf.ViewConsistentRefiner(gd, dec, num_steps=2)       # 2 of the 8 DDIM steps: the state machine wraps at total_denoise_step
    g = torch.Generator(device="cuda").manual_seed(0)
    rgb = torch.rand(32, H, W, 3, device="cuda", generator=g)
    ctrl = torch.rand(32, H, W, 3, device="cuda", generator=g)
    cond = torch.randn(1, 77, 768, device="cuda", generator=g) * 0.1
    uncond = torch.randn(1, 77, 768, device="cuda", generator=g) * 0.1
    gd.set_image_embeds(torch.randn(1, 4, 768, device="cuda", generator=g) * 0.1, torch.zeros(1, 4, 768), torch.zeros(1, 4, 768))
    keys = []
    orig = fused.attention
    monkeypatch.setattr(fused, "attention", lambda *a: (keys.append((a[0].shape[1], a[1].shape[1])), orig(*a))[1])
    views = ["front", "k0", "v3"]                              # canonical, mutual with front, blend of k0 / front
    out, idx = vcr.refine_rgb(rgb, ctrl, lambda n: (cond, uncond), views=views, generator=torch.Generator(device="cuda").manual_seed(3))
    assert out.shape == (3, H, W, 3) and idx == [24, 20, 21]
    assert torch.isfinite(out).all() and float(out.min()) >= 0 and float(out.max()) <= 1
    assert (16384, 16384) in keys and (16384, 32768) in keys   # 128^2 latent tokens; mutual self-attention over 2N keys
    assert rf.refine_timesteps(8).tolist() == [142, 122, 101, 81, 61, 40, 20, 0]
