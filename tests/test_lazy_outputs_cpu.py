"""The lazily materialised entries of the step's output dicts (renderer._LazyDict, system._StepOutputs): what a training step never
reads is never computed, and everything the reference's dicts hold is still there when asked for (GaussianIP.forward :218-230)."""
import torch

from gaussianip_amd.renderer import _LazyDict
from gaussianip_amd.system import _StepOutputs


def test_lazy_dict_computes_on_first_access_only():
    calls = []
    radii = torch.tensor([[0, 3], [2, 0]], dtype=torch.int32)
    d = _LazyDict({"radii": radii}, {"visibility_filter": lambda: calls.append(1) or radii > 0})
    assert "visibility_filter" in d and "nothing" not in d and not calls
    assert set({**d}) == {"radii"}                                  # unpacking carries the materialised entries only
    assert d.get("missing", 7) == 7
    v = d["visibility_filter"]
    assert torch.equal(v, radii > 0) and len(calls) == 1
    assert d["visibility_filter"] is v and d.get("visibility_filter") is v and len(calls) == 1
    assert set({**d}) == {"radii", "visibility_filter"}


def test_step_outputs_make_opacity_scale_and_visibility_when_read():
    depth = torch.tensor([[[[0.0], [2.0]], [[4.0], [1.0]]]])      # [1, 2, 2, 1]
    radii = torch.tensor([[0, 5, 1]], dtype=torch.int32)
    scale = torch.rand(3, 3)
    out = _StepOutputs({"depth": depth, "radii": radii}, lazy={"dmax": lambda: depth.max(), "scale": lambda: scale})
    assert "opacity" in out and "scale" in out and "visibility_filter" in out
    assert not dict.__contains__(out, "opacity")
    # the thunks are not on the mapping: copies / iteration / logging see reference keys with tensor values only
    assert set(out) == {"depth", "radii"} and all(torch.is_tensor(v) for v in {**out}.values())
    assert out.get("no such key", 7) == 7
    assert torch.allclose(out.get("opacity"), depth / (4.0 + 1e-5)) and dict.__contains__(out, "opacity")     # get() materialises too
    assert torch.allclose(out["opacity"], depth / (4.0 + 1e-5))
    assert out["scale"] is scale
    assert torch.equal(out["visibility_filter"], radii > 0)
    try:
        out["no such key"]
        raise AssertionError("KeyError expected")
    except KeyError:
        pass
    plain = _StepOutputs({"depth": depth, "radii": radii}, lazy={"dmax": lambda: depth.max(), "scale": lambda: scale}).materialize()
    assert set(dict(plain)) == {"depth", "radii", "opacity", "scale", "visibility_filter"}
    bare = _StepOutputs({"depth": depth})
    assert "opacity" not in bare and "visibility_filter" not in bare and bare.get("opacity") is None


def test_fused_glue_is_not_offered_for_cpu_tensors_or_other_dtypes():
    """guidance/glue.py: the single-launch stages exist for fp16 CUDA tensors only; everything else keeps the reference's op chains
    (there is no CPU implementation behind the kernels)."""
    from gaussianip_amd.guidance import glue, sds
    rgb = torch.rand(2, 3, 1024, 1024)
    assert not glue.image_prep_supported(rgb, (512, 512))
    moments = torch.randn(2, 8, 64, 64).half()
    eps = torch.randn(2, 4, 64, 64).half()
    t = torch.tensor([10, 500])
    acp = sds.alphas_cumprod()
    assert not glue.latent_sample_supported(moments, eps, eps, t, acp)
    assert not glue.anpg_loss_supported(eps, torch.cat([eps] * 3), t, acp, "sds")
    assert not glue.timestep_embedding_supported(t, torch.float16)
