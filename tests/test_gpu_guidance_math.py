"""compute_grad_anpg / compute_grad_sds on CUDA tensors against the reference-generated fixture (tests/golden/sds_grad.npz,
produced by calling the reference's own functions, ipa_guidance.py:361-519): the same cases tests/test_golden_guidance.py
runs on the CPU, here with every tensor on the GPU.  The reference draws its noise from the CPU
generator under torch.manual_seed; the draw is reproduced on the host and moved to the device, everything else runs on
the device."""
import os

import numpy as np
import pytest
import torch

import fixture_inputs as fx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _NoNet(torch.nn.Module):
    def fold_lora(self, scale=1.0):
        return self


CASES = [
    ("anpg", dict(use_anpg=True, view_dependent_prompting=True, grad_clip_pixel=True, grad_clip_threshold=1.0, weighting_strategy="sds")),
    ("anpg_flat_noclip", dict(use_anpg=True, view_dependent_prompting=False, grad_clip_pixel=False, weighting_strategy="fantasia3d")),
    ("sds", dict(use_anpg=False, view_dependent_prompting=True, grad_clip_pixel=True, grad_clip_threshold=1.0, weighting_strategy="sds", guidance_rescale=0.0)),
    ("sds_rescale", dict(use_anpg=False, view_dependent_prompting=True, grad_clip_pixel=False, weighting_strategy="uniform", guidance_rescale=0.75)),
]


@pytest.mark.parametrize("tag,over", CASES)
@pytest.mark.parametrize("host_scalars", [False, True])
def test_compute_grad_on_the_gpu_matches_the_reference_function(tag, over, host_scalars, monkeypatch):
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    from gaussianip_amd.guidance.prompts import PromptProcessorOutput
    dev = torch.device("cuda")
    d = np.load(os.path.join(GOLD, "sds_grad.npz"))
    case = fx.sds_case(int(d["case_seed"]))
    T = lambda a: torch.from_numpy(np.asarray(a)).to(dev)                 # noqa: E731
    S = (lambda a: torch.from_numpy(np.asarray(a))) if host_scalars else T  # the per-view scalars: host batch or device batch
    gd = StableDiffusionGuidance(GuidanceConfig(half_precision_weights=False, channels_last=False, **over), device=dev,
                                 unet=_NoNet(), controlnet=_NoNet(), vae=_NoNet(), schedule=AHDSSchedule(list(range(2400))))
    gd.forward_unet = fx.fake_forward_unet
    gd.set_image_embeds(T(case["pos_image"]), T(case["neg_image"]), T(case["null_image"]))
    pu = PromptProcessorOutput(T(case["text"]), T(case["uncond"]), T(case["null"]), T(case["text_vd"]), T(case["uncond_vd"]))
    real_randn = torch.randn

    def host_randn(*size, **kw):            # the reference's draw: CPU generator, then onto the device
        kw = dict(kw)
        target = kw.pop("device", None)
        kw.pop("generator", None)
        out = real_randn(*size, **kw)
        return out.to(target) if target is not None else out
    monkeypatch.setattr(torch, "randn", host_randn)
    torch.manual_seed(int(d["seed"]))
    fn = gd.compute_grad_anpg if tag.startswith("anpg") else gd.compute_grad_sds
    grad, util = fn(T(case["latents"]), T(case["control"]), T(case["t"]), pu, True, S(case["all_vis_all"]),
                    S(case["elevation"]), S(case["azimuth"]), S(case["center"]), S(case["camera_distances"]))
    assert grad.is_cuda and util["latents_noisy"].is_cuda
    np.testing.assert_allclose(util["latents_noisy"].cpu().numpy(), d[tag + "_latents_noisy"], atol=2e-6)
    np.testing.assert_allclose(grad.cpu().numpy(), d[tag + "_grad"], atol=4e-6, rtol=2e-5)


def test_tuned_gemm_table_is_active_and_changes_no_values_beyond_rounding():
    """guidance/tunableop_gfx950.csv (hipBLASLt solutions picked by PyTorch's TunableOp on an MI355X box, tuning OFF at run time):
    constructing the guidance switches it on; a listed shape then runs its tuned solution — same product to half rounding, and
    bitwise reproducible call to call (the table holds hipBLASLt solutions only)."""
    import os
    import torch
    import torch.nn.functional as F
    from gaussianip_amd.guidance import fused
    if os.environ.get("GIP_TUNABLEOP", "1") == "0":
        pytest.skip("GIP_TUNABLEOP=0")
    assert fused.enable_tuned_gemms(), "the shipped TunableOp results were rejected (library versions differ from the file's validators?)"
    import torch.cuda.tunable as tun
    assert tun.is_enabled() and not tun.tuning_is_enabled()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn(12288, 640, device="cuda", generator=g) * 0.5).half()            # ff_in at 32^2: tn_5120_12288_640 is in the table
    w = (torch.randn(5120, 640, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(5120, device="cuda", generator=g).half()
    y1, y2 = F.linear(x, w, b), F.linear(x, w, b)
    assert torch.equal(y1, y2)
    ref = F.linear(x.float(), w.float(), b.float())
    assert float((y1.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
