"""Guidance math pinned to the REFERENCE ITSELF (CPU).  The fixtures under tests/golden/ hold the outputs of the
reference's own classes / functions (tools/make_golden.py `guidance` / `backward`, run in the build container with
/root/reference imported by path); their inputs are regenerated here from tests/fixture_inputs.py.

  attention_processors.npz  LoRAAttnProcessor2_0 'normal' / 'refine', LoRAIPAttnProcessor2_0
                            (attention_processor_faceid.py:211-395, 398-523)             -> networks.Attention
  sds_grad.npz              compute_grad_anpg / compute_grad_sds (ipa_guidance.py:361-519) -> StableDiffusionGuidance
  prompt_directions.npz     13 directions + get_text_embeddings (prompt_processors/base.py:52-81, 252-335) -> prompts.py
  refine_timesteps.npz      refine.py:176-178                                             -> refine.refine_timesteps
  preprocess_backward.npz   autograd through sh_utils.eval_sh / general_utils.build_scaling_rotation -> the ORACLE's
                            per-Gaussian backward stages (oracle_sh_backward, oracle_cov3D_backward)
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import fixture_inputs as fx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = torch.from_numpy


# ------------------------------------------------------------------------------------------ attention processors
def make_attention(w, dim, ctx_dim, heads, rank, ip, ip_scale=1.0, fold=False, device="cpu", dtype=torch.float32):
    from gaussianip_amd.guidance.networks import Attention
    a = Attention(dim, ctx_dim, heads, lora_rank=rank, ip=ip, ip_scale=ip_scale)
    with torch.no_grad():
        a.to_q.weight.copy_(T(w["to_q"])); a.to_k.weight.copy_(T(w["to_k"])); a.to_v.weight.copy_(T(w["to_v"]))   # noqa: E702
        a.to_out.weight.copy_(T(w["to_out_w"])); a.to_out.bias.copy_(T(w["to_out_b"]))                               # noqa: E702
        for n in ("q", "k", "v", "out"):
            lora = getattr(a, "lora_" + n)
            lora[0].weight.copy_(T(w["lora_%s_down" % n]))
            lora[1].weight.copy_(T(w["lora_%s_up" % n]))
        if ip:
            a.to_k_ip.weight.copy_(T(w["to_k_ip"])); a.to_v_ip.weight.copy_(T(w["to_v_ip"]))                         # noqa: E702
    if fold:
        a.fold_lora(1.0)                    # the reference's lora_scale = 1.0 (attention_processor_faceid.py:222,284)
    return a.to(device, dtype).eval().requires_grad_(False)


@pytest.fixture(scope="module")
def gold_attn():
    return np.load(os.path.join(GOLD, "attention_processors.npz"))


@pytest.mark.parametrize("fold", [False, True])
def test_self_attention_normal_state_matches_reference_processor(gold_attn, fold):
    d = gold_attn
    dim, heads, rank, B = int(d["dim"]), int(d["heads"]), int(d["rank"]), int(d["batch"])
    a = make_attention(fx.attn_weights(11, dim, None, rank), dim, None, heads, rank, False, fold=fold)
    for n_tok, nb in ((64, B), (256, 1)):
        x = T(fx.attn_tokens(21 + n_tok, nb, n_tok, dim))
        np.testing.assert_allclose(a(x).numpy(), d["self_normal_%d" % n_tok], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("fold", [False, True])
def test_decoupled_cross_attention_matches_reference_ip_processor(gold_attn, fold):
    """Text keys over tokens [:77], image-prompt keys over the last 4 (attention_processor_faceid.py:462-466), own
    softmax each, h = h_text + scale * h_ip (:506)."""
    d = gold_attn
    dim, heads, rank, B = int(d["dim"]), int(d["heads"]), int(d["rank"]), int(d["batch"])
    a = make_attention(fx.attn_weights(12, dim, 768, rank, ip=True), dim, 768, heads, rank, True, float(d["ip_scale"]), fold=fold)
    for n_tok, nb in ((64, B), (256, 1)):
        x = T(fx.attn_tokens(31 + n_tok, nb, n_tok, dim))
        ctx = T(fx.attn_tokens(41, nb, 81, 768))
        np.testing.assert_allclose(a(x, ctx).numpy(), d["cross_ip_%d" % n_tok], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("fold", [False, True])
def test_refine_state_machine_matches_reference_processor(gold_attn, fold):
    """The 'refine' state of LoRAAttnProcessor2_0 (:291-364): key views store their tokens per denoising step, k-views
    attend over [own | front-or-back], v-views blend self / left / right attentions with the reference's weights; the
    step counter wraps at total_denoise_step.  A non-target layer in the same state takes the plain path."""
    from gaussianip_amd.guidance.refine import RefineAttentionState, RefineController
    d = gold_attn
    dim, heads, rank = int(d["dim"]), int(d["heads"]), int(d["rank"])
    w = fx.attn_weights(11, dim, None, rank)
    a = make_attention(w, dim, None, heads, rank, False, fold=fold)
    other = make_attention(w, dim, None, heads, rank, False, fold=fold)          # not a target: no refine state attached
    ctl = RefineController(state="refine", total_denoise_step=fx.REFINE_STEPS, lambda_self=fx.REFINE_LAMBDA_SELF)
    a.refine = RefineAttentionState(ctl)
    for vi, (view, pair, weights) in enumerate(fx.REFINE_VIEWS):
        ctl.cur_view_name = view
        a.refine.stored_zt[view] = []
        if "v" in view:
            ctl.cur_key_view_name_pair, ctl.cur_key_view_weight_pair = pair, weights
        for step in range(fx.REFINE_STEPS):
            x = T(fx.refine_tokens(vi, step, 1, fx.REFINE_TOKENS, dim))
            np.testing.assert_allclose(a(x).numpy(), d["refine_%s_%d" % (view, step)], atol=2e-5, rtol=1e-5,
                                       err_msg="%s step %d" % (view, step))
            if view == "front":
                np.testing.assert_allclose(other(x).numpy(), d["refine_other_front_%d" % step], atol=2e-5, rtol=1e-5)
        assert a.refine.cur_denoise_step == 0                 # wrapped after total_denoise_step calls
    assert len(a.refine.stored_zt["front"]) == fx.REFINE_STEPS and len(a.refine.stored_zt["v3"]) == 0


# ------------------------------------------------------------------------------------------ prompt directions
def test_prompt_directions_match_reference():
    from gaussianip_amd.guidance import prompts as P
    d = np.load(os.path.join(GOLD, "prompt_directions.npz"))
    assert [x.name for x in P.DIRECTIONS] == list(d["names"])
    assert list(P.DIRECTION2IDX.keys()) == list(d["direction2idx_keys"])
    assert list(P.DIRECTION2IDX.values()) == list(d["direction2idx_values"])          # the name-keyed quirk: 6..12
    assert P.view_dependent_prompts(str(d["prompt"])) == list(d["prompts_vd"])
    el, az, cent, vis, dist = [T(a) for a in fx.direction_grid()]
    idx = P.direction_index(el, az, cent, vis, dist, head_offset=0.65)
    assert np.array_equal(idx.numpy(), d["direction_idx"])
    assert set(np.unique(d["direction_idx"])) == {0, 6, 7, 8, 9, 10, 11, 12}          # 1..5 are unreachable
    marker = torch.arange(13, dtype=torch.float32).view(13, 1, 1).expand(13, 2, 3).contiguous()
    out = P.PromptProcessorOutput(torch.full((1, 2, 3), 100.0), torch.full((1, 2, 3), 200.0), torch.full((1, 2, 3), 300.0),
                                  marker, marker + 1000.0)
    n = el.shape[0]
    emb = out.get_text_embeddings(el, az, cent, vis, dist, True)
    assert emb.shape == (3 * n, 2, 3)
    assert np.array_equal(emb[:n, 0, 0].numpy().astype(np.int64), d["direction_idx"])
    assert np.array_equal((emb[n:2 * n, 0, 0] - 1000).numpy().astype(np.int64), d["uncond_idx"])
    assert np.array_equal(emb[2 * n:, 0, 0].numpy(), d["null_value"])
    flat = out.get_text_embeddings(el, az, cent, vis, dist, False)
    assert np.array_equal(flat[:, 0, 0].numpy(), d["not_view_dependent"])


def test_prompt_processor_builds_the_reference_tables():
    from gaussianip_amd.guidance.prompts import PromptProcessor
    d = np.load(os.path.join(GOLD, "prompt_directions.npz"))
    seen = []

    def encode(texts):
        seen.extend(texts)
        return torch.stack([torch.full((77, 768), float(len(t))) for t in texts])
    pp = PromptProcessor(str(d["prompt"]), encode, negative_prompt="ugly", null_prompt="")
    assert pp.prompts_vd == list(d["prompts_vd"]) and pp.negative_prompts_vd == list(d["negative_prompts_vd"])
    assert len(seen) == len(set(seen)) and "" in seen                    # each distinct prompt encoded once
    out = pp()
    assert out.text_embeddings_vd.shape == (13, 77, 768) and out.uncond_text_embeddings_vd.shape == (13, 77, 768)
    assert float(out.text_embeddings_vd[6, 0, 0]) == len(d["prompts_vd"][6]) and float(out.null_embeddings[0, 0, 0]) == 0.0


# ------------------------------------------------------------------------------------------ SDS / ANPG gradients
class _NoNet(torch.nn.Module):
    def fold_lora(self, scale=1.0):
        return self


def _guidance(**cfg_over):
    from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
    from gaussianip_amd.guidance.ahds import AHDSSchedule
    cfg = GuidanceConfig(half_precision_weights=False, channels_last=False, **cfg_over)
    gd = StableDiffusionGuidance(cfg, device="cpu", unet=_NoNet(), controlnet=_NoNet(), vae=_NoNet(),
                                 schedule=AHDSSchedule(list(range(2400))))
    gd.forward_unet = fx.fake_forward_unet
    return gd


@pytest.mark.parametrize("tag,over", [
    ("anpg", dict(use_anpg=True, view_dependent_prompting=True, grad_clip_pixel=True, grad_clip_threshold=1.0, weighting_strategy="sds")),
    ("anpg_flat_noclip", dict(use_anpg=True, view_dependent_prompting=False, grad_clip_pixel=False, weighting_strategy="fantasia3d")),
    ("sds", dict(use_anpg=False, view_dependent_prompting=True, grad_clip_pixel=True, grad_clip_threshold=1.0, weighting_strategy="sds", guidance_rescale=0.0)),
    ("sds_rescale", dict(use_anpg=False, view_dependent_prompting=True, grad_clip_pixel=False, weighting_strategy="uniform", guidance_rescale=0.75)),
])
def test_compute_grad_matches_reference_function(tag, over):
    """compute_grad_anpg / compute_grad_sds called exactly as __call__ does, against the reference's own functions run
    on the same inputs with the same stand-in forward_unet and the same torch seed (the noise draw)."""
    from gaussianip_amd.guidance.prompts import PromptProcessorOutput
    d = np.load(os.path.join(GOLD, "sds_grad.npz"))
    case = fx.sds_case(int(d["case_seed"]))
    gd = _guidance(**over)
    gd.set_image_embeds(T(case["pos_image"]), T(case["neg_image"]), T(case["null_image"]))
    pu = PromptProcessorOutput(T(case["text"]), T(case["uncond"]), T(case["null"]), T(case["text_vd"]), T(case["uncond_vd"]))
    fn = gd.compute_grad_anpg if tag.startswith("anpg") else gd.compute_grad_sds
    torch.manual_seed(int(d["seed"]))
    grad, util = fn(T(case["latents"]), T(case["control"]), T(case["t"]), pu, True, T(case["all_vis_all"]),
                    T(case["elevation"]), T(case["azimuth"]), T(case["center"]), T(case["camera_distances"]))
    np.testing.assert_allclose(util["latents_noisy"].numpy(), d[tag + "_latents_noisy"], atol=1e-6)
    if tag == "sds":
        # the reference's plain-SDS path (compute_grad_sds :443-519) has NO pixel clip; grad_clip_pixel only acts in ANPG
        pass
    np.testing.assert_allclose(grad.numpy(), d[tag + "_grad"], atol=2e-6, rtol=1e-5)


def test_refine_timesteps_match_reference_expression():
    from gaussianip_amd.guidance.refine import refine_timesteps
    d = np.load(os.path.join(GOLD, "refine_timesteps.npz"))
    assert refine_timesteps(8).tolist() == d["timesteps_sub"].tolist() == [142, 122, 101, 81, 61, 40, 20, 0]
    assert refine_timesteps(50).tolist() == d["timesteps"].tolist()


# ------------------------------------------------------------------------------------------ oracle backward stages
def _oracle_lib(oracle):
    oracle.build()
    return ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), "libraster_oracle.so"))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_oracle_sh_backward_matches_reference_autograd(oracle, deg):
    d = np.load(os.path.join(GOLD, "preprocess_backward.npz"))
    lib = _oracle_lib(oracle)
    K = (deg + 1) ** 2
    Pn = d["xyz"].shape[0]
    feats = np.ascontiguousarray(d["features"][:, :K])
    clamped = np.ascontiguousarray((d["rgb%d" % deg] <= 0.0).astype(np.uint8))         # the forward's clamp flags
    dsh = np.zeros((Pn, K, 3), np.float32)
    dxyz = np.zeros((Pn, 3), np.float32)
    xyz, campos, gcol = [np.ascontiguousarray(d[k]) for k in ("xyz", "campos", "gcol")]
    lib.oracle_sh_backward(Pn, deg, K, _ptr(xyz), _ptr(campos), _ptr(feats), _ptr(clamped), _ptr(gcol), _ptr(dsh), _ptr(dxyz))
    np.testing.assert_allclose(dsh, d["dfeatures%d" % deg], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(dxyz, d["dxyz%d" % deg], atol=5e-6, rtol=2e-4)
    assert clamped.any() and not clamped.all()                                          # both branches of the clamp exercised


@pytest.mark.parametrize("tag,mod", [("m1", 1.0), ("m17", 1.7)])
def test_oracle_cov3D_backward_matches_reference_autograd(oracle, tag, mod):
    """Two documented relations to the reference's Python statement: the fork differentiates w.r.t. mod * scale
    (reference autograd = mod x oracle) and differentiates the quaternion as given, while build_rotation normalises
    inside (reference autograd = oracle projected orthogonally to the unit quaternion, general_utils.py:78-81)."""
    d = np.load(os.path.join(GOLD, "preprocess_backward.npz"))
    lib = _oracle_lib(oracle)
    s, q, dcov = [np.ascontiguousarray(d[k]) for k in ("scales", "rotations", "dcov")]
    Pn = s.shape[0]
    ds, dq = np.zeros((Pn, 3), np.float32), np.zeros((Pn, 4), np.float32)
    lib.oracle_cov3D_backward(Pn, _ptr(s), ctypes.c_float(mod), _ptr(q), _ptr(dcov), _ptr(ds), _ptr(dq))
    ref_ds, ref_dq = d["dscales_" + tag], d["drotations_" + tag]
    np.testing.assert_allclose(mod * ds, ref_ds, atol=1e-6 * np.abs(ref_ds).max(), rtol=2e-4)
    proj = dq - (dq * q).sum(1, keepdims=True) * q
    np.testing.assert_allclose(proj, ref_dq, atol=2e-5 * np.abs(ref_dq).max(), rtol=1e-3)


def test_mvp_matrices_and_bce_match_reference_ops():
    from gaussianip_amd.system import binary_cross_entropy
    from gaussianip_amd.utils.graphics import get_mvp_matrix, get_projection_matrix
    d = np.load(os.path.join(GOLD, "mvp.npz"))
    fovy, c2w = T(d["fovy"]), T(d["c2w"])
    proj = get_projection_matrix(fovy, 1.0, 0.1, 1000.0)
    assert np.array_equal(proj.numpy(), d["proj"])
    assert np.array_equal(get_projection_matrix(fovy, 1.5, 0.1, 1000.0).numpy(), d["proj_aspect_1p5"])
    np.testing.assert_allclose(get_mvp_matrix(c2w, proj).numpy(), d["mvp"], atol=1e-6)
    x = T(d["bce_x"])
    np.testing.assert_allclose(binary_cross_entropy(x, x).numpy(), d["bce"], rtol=1e-6)
