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
    out = _StepOutputs({"depth": depth, "radii": radii, "_dmax": lambda: depth.max(), "_scale": lambda: scale})
    assert "opacity" in out and "scale" in out and "visibility_filter" in out
    assert not dict.__contains__(out, "opacity")
    assert torch.allclose(out["opacity"], depth / (4.0 + 1e-5)) and dict.__contains__(out, "opacity")
    assert out["scale"] is scale
    assert torch.equal(out["visibility_filter"], radii > 0)
    try:
        out["no such key"]
        raise AssertionError("KeyError expected")
    except KeyError:
        pass
