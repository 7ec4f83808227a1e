"""Host-side mirror of the reference's "ipa-guidance" plugin.

Reference: threestudio/models/guidance/ipa_guidance.py — registry name :71, Config :74-123, forward_unet :311-358,
compute_grad_anpg :361-440, compute_grad_sds :443-519, encode_images :522-531, __call__ :602-660.  Same call signature
and returned dict ({"loss_sds", "grad_norm"}), same token layout into the U-Net ([neg x B | pos x B | null x B], each
[77 text | 4 image tokens]; :382-386), same ANPG combine, weighting, clip and loss.

Differences: networks come from gaussianip_amd.guidance.networks (no diffusers dependency; random weights unless state
dicts are supplied); text / face-ID embeddings are inputs (`PromptEmbeddings`, `set_image_embeds`) because CLIP,
insightface and the checkpoints are outside this path; the frozen LoRA is folded into the base weights once.
"""
from collections.abc import Mapping
from dataclasses import dataclass, field, fields
from typing import Any, Callable, Optional

import torch
import torch.nn.functional as F

from . import fused, glue, sds
from .ahds import AHDSSchedule
from .networks import IP_TOKENS, TEXT_TOKENS, ControlNet, UNet, VAEEncoder, init_for_benchmark


_NO_SHARED_PREFIX = False      # True: the layers in front of the first cross-attention run on the full ANPG / CFG batch (measured: +1.25 ms, §4b)
_TWO_STREAMS = __import__("os").environ.get("GIP_GUIDANCE_STREAMS", "2") != "1"    # A/B switch: ControlNet beside the U-Net encoder
# The frozen, fixed-shape networks replay from HIP graphs (round 3 default): the Python host needs ~28 us per launch, which
# made a ONE-view shard of configs[3] (1305 launches, 20 ms of GPU work) launch-bound at 30 ms.  GIP_GRAPH_DENOISE=0 /
# GIP_GRAPH_VAE=0 restore the eager launches (same kernels, same values).
_GRAPH_DENOISE = __import__("os").environ.get("GIP_GRAPH_DENOISE", "1") == "1"
_GRAPH_VAE = __import__("os").environ.get("GIP_GRAPH_VAE", "1") == "1"
# round 6: the latents-independent head of the denoise as its own graph on the side stream, beside the VAE encoder (launch_denoise_prologue).
# False = one graph as in round 5 (same-box A/B: 33.56 -> 33.45 ms with it; tests flip the attribute)
_PROLOGUE_GRAPH = True


class _no_gc:
    """No cyclic garbage collection while a HIP graph is being captured.  A collection that happens to run inside the capture can
    finalise ANOTHER object that owns captured graphs (an earlier guidance instance of the same process): destroying a graph / its
    memory pool is not permitted while a stream is capturing, the error surfaces in a destructor and aborts the process (seen once in
    the GPU suite, round 6: "Fatal Python error: Aborted ... Garbage-collecting" under _forward_unet_graph).  torch.cuda.graph()
    collects BEFORE the capture for the same reason; this closes the window during it."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        if self._was:
            import gc
            gc.enable()


@dataclass
class GuidanceConfig:
    # the subset of ipa_guidance.py:74-123 that the per-step path reads (values of configs/exp.yaml:78-120 as defaults)
    use_ipa_faceid: bool = True
    use_pose_controlnet: bool = True
    batch_size: int = 4
    guidance_scale: float = 7.5
    ipa_scale: float = 0.6
    ipa_faceid_scale: float = 0.5
    half_precision_weights: bool = True
    use_anpg: bool = True
    weighting_strategy: str = "sds"
    view_dependent_prompting: bool = True
    guidance_rescale: float = 0.0
    grad_clip_pixel: bool = True
    grad_clip_threshold: float = 1.0
    fold_lora: bool = True
    channels_last: bool = True    # NHWC end to end: MIOpen's MFMA igemm layout + the fused GroupNorm(+SiLU) kernels
    seed: int = 0
    extra: dict = field(default_factory=dict)     # reference keys that do not reach the per-step path (paths, prompts, ...)

    # every key of the reference's Config (ipa_guidance.py:74-123) that this class does not model itself
    _PASSIVE = ("pretrained_sd_model_name_or_path", "pretrained_realistic_model_name_or_path", "vae_path", "image_encoder_path",
                "image_encoder_faceid_path", "ip_ckpt_path", "ip_ckpt_faceid_v1_path", "ip_ckpt_faceid_v2_path",
                "pose_controlnet_path", "prompt", "negative_prompt", "negative_prompt_faceid", "null_prompt", "pil_image_path",
                "pil_image_faceid_path", "irr_pil_image_path", "enable_memory_efficient_attention",
                "enable_sequential_cpu_offload", "enable_attention_slicing", "enable_channels_last_format",
                "ipa_faceid_s_scale", "grad_clip", "max_items_eval", "lw_depth", "original_size", "target_size")

    # the reference's own Config defaults (ipa_guidance.py:74-123) for the keys modelled above: a cfg mapping that
    # omits a key gets THESE (what `threestudio.find("ipa-guidance")(cfg)` would give), not the shipped-YAML values
    # that the dataclass defaults carry for the benchmark and the tests
    _REFERENCE_DEFAULTS = dict(use_ipa_faceid=True, use_pose_controlnet=True, batch_size=4, guidance_scale=7.5,
                               ipa_scale=0.6, ipa_faceid_scale=0.6, half_precision_weights=True, use_anpg=False,
                               weighting_strategy="sds", view_dependent_prompting=True, guidance_rescale=0.0,
                               grad_clip_pixel=False, grad_clip_threshold=0.1)

    @classmethod
    def from_dict(cls, d: dict) -> "GuidanceConfig":
        """Build from the `system.guidance` section of the reference's YAML (configs/exp.yaml:78-120): the keys the
        per-step path reads become fields (missing ones take the reference Config's defaults), the remaining reference
        keys are kept in `extra`, unknown keys raise."""
        names = {f.name for f in fields(cls)} - {"extra"}
        own = dict(cls._REFERENCE_DEFAULTS)
        own.update({k: v for k, v in d.items() if k in names})
        rest = {k: v for k, v in d.items() if k not in names}
        unknown = [k for k in rest if k not in cls._PASSIVE]
        if unknown:
            raise KeyError("unknown guidance config keys: %s" % unknown)
        if "enable_channels_last_format" in rest:
            own.setdefault("channels_last", bool(rest["enable_channels_last_format"]) or True)
        return cls(**own, extra=rest)


class PromptEmbeddings:
    """Stand-in for PromptProcessorOutput (prompt_processors/base.py:52-81): view-dependent text embeddings are looked
    up from [D,77,768] tables by a direction index; returns cat[pos, neg, null] = [3B,77,768]."""

    def __init__(self, pos, neg, null, direction_fn=None):
        self.pos, self.neg, self.null, self.direction_fn = pos, neg, null, direction_fn

    def get_text_embeddings(self, elevation, azimuth, center, all_vis_all, camera_distances, view_dependent_prompting=True):
        B = elevation.shape[0]
        if view_dependent_prompting and self.direction_fn is not None:
            idx = self.direction_fn(elevation, azimuth, center, all_vis_all, camera_distances)
        else:
            idx = torch.zeros(B, dtype=torch.long, device=self.pos.device)
        idx = idx.to(self.pos.device, non_blocking=True)
        pick = lambda tab: tab[idx % tab.shape[0]]  # noqa: E731
        return torch.cat([pick(self.pos), pick(self.neg), pick(self.null)], dim=0)


class StableDiffusionGuidance:
    registry_name = "ipa-guidance"

    def __init__(self, cfg=None, device="cuda", unet=None, controlnet=None, vae=None,
                 schedule: Optional[AHDSSchedule] = None, checkpoints: Optional[dict] = None,
                 image_embeds_provider: Optional[Callable] = None):
        """`cfg`: a GuidanceConfig, or the mapping `threestudio.find("ipa-guidance")(cfg)` hands over (the
        `system.guidance` YAML section, dict or DictConfig; GaussianIP.py:355) -> GuidanceConfig.from_dict.
        `image_embeds_provider(guidance) -> (pos, neg, null)` image-prompt tokens, each [1, 4, 768]: what the reference's
        prepare_for_sds computes with insightface + the IP-Adapter projection (ipa_guidance.py:236-275); called by
        prepare_for_sds.  `checkpoints` (optional): {"unet": path, "controlnet": path, "vae": path, "ip_adapter": path} — diffusers-format
        .safetensors / .bin files (runwayml/stable-diffusion-v1-5 unet, lllyasviel/control_v11p_sd15_openpose,
        stabilityai/sd-vae-ft-mse, h94/IP-Adapter-FaceID ip-adapter-faceid-plusv2_sd15.bin; the names the reference loads at
        ipa_guidance.py:127-185).  Without it the networks keep their deterministic random initialisation."""
        if isinstance(cfg, Mapping):
            cfg = GuidanceConfig.from_dict(dict(cfg))
        self.cfg = cfg or GuidanceConfig()
        self.image_embeds_provider = image_embeds_provider
        self._image_embeds_set = False
        self.device = torch.device(device)
        self.weights_dtype = torch.float16 if self.cfg.half_precision_weights else torch.float32
        scale = self.cfg.ipa_faceid_scale if self.cfg.use_ipa_faceid else self.cfg.ipa_scale
        self.unet = unet if unet is not None else init_for_benchmark(UNet(128, True, scale), self.cfg.seed)
        self.controlnet = controlnet if controlnet is not None else init_for_benchmark(ControlNet(), self.cfg.seed + 1)
        self.vae = vae if vae is not None else init_for_benchmark(VAEEncoder(), self.cfg.seed + 2)
        if checkpoints:
            self.load_checkpoints(**checkpoints)          # real weights: before the LoRA branches are folded away
        if self.cfg.fold_lora:
            self.unet.fold_lora(1.0)
        for m in (self.unet, self.controlnet, self.vae):
            m.to(self.device, self.weights_dtype).eval().requires_grad_(False)
            if self.cfg.channels_last:
                m.to(memory_format=torch.channels_last)
        if self.device.type == "cuda" and self.weights_dtype == torch.float16:
            fused.enable_tuned_gemms()             # the library GEMMs of the step on their tuned hipBLASLt solutions (once per process)
            self.unet.prepare_inference()          # one batched time-embedding projection per forward
            self.controlnet.prepare_inference()
        self.num_train_timesteps = 1000
        self.alphas = sds.alphas_cumprod(device=self.device)
        self.schedule = schedule or AHDSSchedule()
        self.ahds_chosen_t_all = self.schedule.table
        self.ahds_chosen_t_min = self.schedule.t_min
        z = torch.zeros(1, IP_TOKENS, 768, device=self.device, dtype=self.weights_dtype)
        self.pos_image_embeds, self.neg_image_embeds, self.null_image_embeds = z, z, z

    def invalidate_graphs(self):
        """Drop every captured HIP graph (denoise and VAE) and release the derived-weight cache entries they pinned.  Called
        by whatever changes something a graph froze at capture: load_checkpoints, fold_lora, prepare_inference, an ip_scale
        change (those also bump fused._weights_epoch, which is part of the graph keys — a stale graph can never be replayed
        even when this method is not reached, e.g. when a caller edits the modules directly)."""
        self._drop_graphs()
        fused.bump_weights_epoch()

    def _drop_graphs(self):
        self._graphs = None
        self._vae_graphs = None
        self._vae_live = None
        fused._wt_cache.unpin(id(self))        # only what THIS instance's captures pinned: other instances' graphs stay valid

    def __del__(self):
        # an instance that goes away with live graphs must not leave its derived weights pinned for the rest of the process
        try:
            fused._wt_cache.unpin(id(self))
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    _graph_sig = None

    def _check_graph_signature(self):
        """Graphs captured under another weights epoch / switch setting can never be replayed again (the signature is part of
        their keys): release them and their private pools instead of keeping them for good."""
        sig = fused.graph_signature()
        if self._graph_sig != sig:
            if self._graph_sig is not None:
                self._drop_graphs()
            self._graph_sig = sig

    def set_ip_scale(self, scale):
        """IPAdapter.set_scale (ip_adapter_faceid.py:330-333): the image-prompt weight of every cross-attention."""
        from .networks import Attention
        for m in self.unet.modules():
            if isinstance(m, Attention) and m.ip:
                m.ip_scale = float(scale)
        self.invalidate_graphs()

    def load_checkpoints(self, unet=None, controlnet=None, vae=None, ip_adapter=None):
        from . import checkpoints as ck
        self.invalidate_graphs()

        def read(path):
            return ck.load_safetensors(path) if str(path).endswith(".safetensors") else torch.load(path, map_location="cpu")
        if unet:
            ck.load_diffusers_state_dict(self.unet, read(unet), "unet")
        if controlnet:
            ck.load_diffusers_state_dict(self.controlnet, read(controlnet), "controlnet")
        if vae:
            ck.load_diffusers_state_dict(self.vae, read(vae), "vae_encoder")
        if ip_adapter:
            state = read(ip_adapter)
            ck.load_ip_adapter_faceid(self.unet, state.get("ip_adapter", state))

    def prepare_for_sds(self, prompt=None, negative_prompt=None, null_prompt=None, image_embeds=None):
        """Same entry point and positional signature as ipa_guidance.py:236 — `guidance.prepare_for_sds(prompt,
        negative_prompt, null_prompt)` at GaussianIP.py:356.  The reference runs insightface FaceAnalysis + the
        IP-Adapter image projection here to turn the identity photo into (pos, null, neg) image tokens; those models are
        the caller's (not shippable, not on the per-step path), so the tokens come from, in this order: the
        `image_embeds` keyword, the `image_embeds_provider` given at construction, or an earlier `set_image_embeds`.
        With none of the three the call fails loudly.  The tokens are tiled to `cfg.batch_size` rows like :279-288.
        The text prompts are consumed by the prompt processor (`prompt_utils`), exactly as in the reference's
        view-dependent path; they are kept as attributes."""
        self.prompt, self.negative_prompt, self.null_prompt = prompt, negative_prompt, null_prompt
        if image_embeds is None and self.image_embeds_provider is not None:
            image_embeds = self.image_embeds_provider(self)
        if image_embeds is not None:
            self.set_image_embeds(*image_embeds)
        if not self._image_embeds_set:
            raise RuntimeError("prepare_for_sds: no image-prompt tokens — construct the guidance with "
                               "image_embeds_provider=..., or call set_image_embeds(pos, neg, null) first (face-ID "
                               "analysis and the IP-Adapter image projection run outside this package)")
        self.num_samples = int(self.cfg.batch_size)
        tile = lambda e: e.expand(self.num_samples, -1, -1).contiguous() if e.shape[0] == 1 else e  # noqa: E731
        single = [e if e.shape[0] == 1 else None for e in (self.pos_image_embeds, self.neg_image_embeds, self.null_image_embeds)]
        self.pos_image_embeds, self.neg_image_embeds, self.null_image_embeds = (
            tile(self.pos_image_embeds), tile(self.neg_image_embeds), tile(self.null_image_embeds))
        # the one-row originals of the tiled tokens (every view shares them): what the prompt table of the fused call is built from
        self._image_embeds_single = None if any(e is None for e in single) else (
            tuple(single), tuple(id(e) for e in (self.pos_image_embeds, self.neg_image_embeds, self.null_image_embeds)))
        self.bs_embed, self.seq_len = 1, self.pos_image_embeds.shape[1]

    def set_image_embeds(self, pos, neg, null):
        """[1 or B, 4, 768] face-ID image tokens: pos = identity, null = irrelevant face, neg = zeros
        (ip_adapter_faceid.py:362-382, ipa_guidance.py:250-257)."""
        cast = lambda t: t.to(self.device, self.weights_dtype)  # noqa: E731
        self.pos_image_embeds, self.neg_image_embeds, self.null_image_embeds = cast(pos), cast(neg), cast(null)
        self._image_embeds_set = True

    # ------------------------------------------------------------------ networks
    def embed_control(self, control_img):
        """The ControlNet's hint embedding of `control_img` [n, 3, H, W] (timestep-independent): pass it to forward_unet as
        `control_embedding` when the same pose map is denoised more than once."""
        cond = control_img.to(self.weights_dtype)
        if self.cfg.channels_last:
            cond = cond.contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", enabled=False):
            return self.controlnet.embed_condition(cond)

    def forward_unet(self, noisy_latents, control_img, t, encoder_hidden_states, use_pose_controlnet=True,
                     control_embedding=None, replicas=1, borrow_output=False, prologue=None, **_unused):
        """`replicas` = r: the batch is r copies of the same (latents, t, pose map) with different prompt embeddings
        (ANPG: 3, classifier-free guidance: 2); the layers in front of the first cross-attention then run on one copy
        (networks._Encoder.encode) — identical algebra."""
        if _NO_SHARED_PREFIX:
            replicas = 1
        if (_GRAPH_DENOISE and noisy_latents.is_cuda and control_embedding is None and control_img is not None and
                not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
            return self._forward_unet_graph(noisy_latents, control_img, t, encoder_hidden_states, use_pose_controlnet, replicas,
                                            borrow_output, prologue)
        return self._forward_unet_eager(noisy_latents, control_img, t, encoder_hidden_states, use_pose_controlnet,
                                        control_embedding, replicas)

    _graphs = None

    def _graph_key(self, noisy_shape, noisy_dtype, control_img, t, ctx, use_pose, replicas, dev_index):
        return (tuple(noisy_shape), noisy_dtype, tuple(control_img.shape), control_img.dtype, tuple(t.shape), t.dtype,
                tuple(ctx.shape), ctx.dtype, bool(use_pose), int(replicas), dev_index, fused.graph_signature(),
                bool(glue.timestep_embedding_supported(t, self.weights_dtype)))

    def launch_denoise_prologue(self, noisy_shape, noisy_dtype, control_img, t, ctx, use_pose, replicas):
        """Round 6.  What the denoise needs that does NOT depend on the latents — the ControlNet's hint stem, both networks' time
        embeddings / ResnetBlock2D addends and prompt-token keys / values (~0.2 ms of small launches at the head of the ControlNet's
        chain, i.e. of the critical path) — is its own captured graph, launched HERE on the side stream.  The fused plugin call
        invokes this before it enqueues the VAE encoder, whose large kernels the small ones then run beside; forward_unet(...,
        prologue=token) joins the side stream and replays the rest.  Returns None (nothing launched: forward_unet does everything
        itself) until the graphs of this shape exist."""
        if not (_GRAPH_DENOISE and use_pose and _TWO_STREAMS and control_img.is_cuda and self._graphs):
            return None
        self._check_graph_signature()
        if self._graphs is None:
            return None
        key = self._graph_key(noisy_shape, noisy_dtype, control_img, t, ctx, use_pose, replicas, control_img.device.index)
        ent = self._graphs.get(key)
        if not isinstance(ent, tuple) or ent[3] is None:
            return None
        graph, static, out, g0 = ent[:4]
        main, side = torch.cuda.current_stream(control_img.device), self._side_stream(control_img.device)
        side.wait_stream(main)          # the inputs come from the main stream; the previous replay of the main graph (which reads the
        with torch.cuda.stream(side):   # prologue's outputs) was enqueued there too
            for d_, s_ in zip(static[1:], (control_img, t, ctx)):
                d_.copy_(s_)
            g0.replay()
        return key

    def _forward_unet_graph(self, noisy_latents, control_img, t, ctx, use_pose, replicas, borrow_output=False, prologue=None):
        """GIP_GRAPH_DENOISE=1: the frozen, fixed-shape denoise (~700 launches on two streams) as ONE HIP-graph launch per call.
        The GPU time is the same (measured: 24.7 vs 24.6 ms); it saves host time — 16.5 -> 9 ms of the 40 ms the host needs
        to enqueue a training step.  On the pool's hosts the step is GPU-bound either way (43.3-44.0 eager vs 43.6-43.7 ms
        with the graph), so it is opt-in, for hosts that cannot keep the queue filled.
        First call of a shape runs eagerly (lazy one-time initialisations must not be captured), the second captures."""
        self._check_graph_signature()
        if self._graphs is None:
            self._graphs = {}
        key = self._graph_key(noisy_latents.shape, noisy_latents.dtype, control_img, t, ctx, use_pose, replicas, noisy_latents.device.index)
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs[key] = "warm"
            return self._forward_unet_eager(noisy_latents, control_img, t, ctx, use_pose, None, replicas)
        if ent == "warm":
            static = [torch.empty_like(a) for a in (noisy_latents, control_img, t, ctx)]
            for d_, s_ in zip(static, (noisy_latents, control_img, t, ctx)):
                d_.copy_(s_)
            g0, pro = None, None
            if use_pose and _TWO_STREAMS and _PROLOGUE_GRAPH:
                # graph 0: hint stem + both networks' latents-independent staging (launch_denoise_prologue); its outputs live in
                # its private pool and are read in place by graph 1
                g0 = torch.cuda.CUDAGraph()
                with _no_gc(), fused.capture_owner(id(self)), torch.cuda.graph(g0):
                    pro = self._denoise_prologue_eager(static[1], static[2], static[3])
            graph = torch.cuda.CUDAGraph()
            with _no_gc(), fused.capture_owner(id(self)), torch.cuda.graph(graph):
                out = self._forward_unet_eager(static[0], static[1], static[2], static[3], use_pose, None, replicas, pro=pro)
            ent = self._graphs[key] = (graph, static, out, g0, pro)
        graph, static, out, g0 = ent[:4]
        if prologue is not None and prologue == key and g0 is not None:
            # graph 0 of THIS call is already running on the side stream (launch_denoise_prologue): join it, hand over the latents
            torch.cuda.current_stream(noisy_latents.device).wait_stream(self._side_stream(noisy_latents.device))
            static[0].copy_(noisy_latents)
        else:
            for d_, s_ in zip(static, (noisy_latents, control_img, t, ctx)):
                d_.copy_(s_)
            if g0 is not None:
                g0.replay()
        graph.replay()
        # `borrow_output`: the caller consumes the prediction before the next replay (same stream) and does not keep it:
        # the graph's own output buffer is handed out instead of a copy
        return out if borrow_output else out.clone()

    def _denoise_prologue_eager(self, control_img, t, ctx):
        """(hint embedding, U-Net handle, ControlNet handle): see launch_denoise_prologue / networks._Encoder.prologue."""
        dt = self.weights_dtype
        c = ctx.to(dt)
        with torch.no_grad(), torch.autocast("cuda", enabled=False):
            return (self.embed_control(control_img), self.unet.prologue(t, c, dt), self.controlnet.prologue(t, c, dt))

    def _forward_unet_eager(self, noisy_latents, control_img, t, encoder_hidden_states, use_pose_controlnet, control_embedding,
                            replicas, pro=None):
        dt = self.weights_dtype
        x = noisy_latents.to(dt)
        ctx = encoder_hidden_states.to(dt)
        if self.cfg.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", enabled=False):
            if not use_pose_controlnet:
                return self.unet(x, t, ctx, replicas=replicas).to(noisy_latents.dtype)
            if not (x.is_cuda and _TWO_STREAMS):
                if control_embedding is None:
                    control_embedding = self.embed_control(control_img)
                down, mid = self.controlnet(x, t, ctx, None, 1.0, cond_embedding=control_embedding, replicas=replicas)
                return self.unet(x, t, ctx, down, mid, replicas=replicas).to(noisy_latents.dtype)
            # The ControlNet and the U-Net's encoder + mid block are independent (the 13 residuals enter the U-Net after its
            # mid block, ipa_guidance.py:331-352): the ControlNet is enqueued on a second HIP stream and the U-Net joins it
            # where it needs the residuals.  Many layers of both have too few tiles to fill 256 CUs on their own (0.47-0.94
            # of a round at 16^2 / 32^2), so the two kernel sequences fill each other's idle CUs.  Same kernels, same values.
            main = torch.cuda.current_stream(x.device)
            side = self._side_stream(x.device)
            side.wait_stream(main)                     # x, ctx, the pose maps were produced on the main stream
            with torch.cuda.stream(side):
                if pro is not None:             # hint stem and staging came out of the prologue graph
                    down, mid = self.controlnet(x, t, ctx, None, 1.0, cond_embedding=pro[0], replicas=replicas, pro=pro[2])
                else:
                    emb = self.embed_control(control_img) if control_embedding is None else control_embedding
                    down, mid = self.controlnet(x, t, ctx, None, 1.0, cond_embedding=emb, replicas=replicas)

            def join():
                main.wait_stream(side)
                for r in list(down) + [mid]:
                    r.record_stream(main)              # allocated on the side stream, consumed (and freed) on the main one
                return down, mid
            return self.unet(x, t, ctx, join, None, replicas=replicas, pro=None if pro is None else pro[1]).to(noisy_latents.dtype)

    _side = None

    def _side_stream(self, dev):
        if self._side is None:
            self._side = {}
        st = self._side.get(dev)
        if st is None:
            st = self._side[dev] = torch.cuda.Stream(device=dev)
        return st

    def encode_images(self, imgs, generator=None):
        x = (imgs * 2.0 - 1.0).to(self.weights_dtype)
        if self.cfg.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", enabled=False):
            if (_GRAPH_VAE and x.is_cuda and x.requires_grad and torch.is_grad_enabled() and self.cfg.channels_last and
                    not torch.cuda.is_current_stream_capturing()):
                return self._encode_graphed(x, generator).to(imgs.dtype)
            return self._vae_encode(x, generator).to(imgs.dtype)

    def _moments(self, x):
        """The VAE encoder's (mean | logvar) of an already prepared image (half, channels-last, in [-1, 1])."""
        with torch.autocast("cuda", enabled=False):
            if _GRAPH_VAE and x.requires_grad and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing():
                return self._encode_graphed(x, None, moments_only=True)
            return self._vae_moments(x)

    def _vae_encode(self, x, generator):
        return self.vae.encode(x, generator)

    def _vae_moments(self, x):
        """(The batch as two halves on two streams measured neutral in round 4 — 33.77 vs 33.70 ms per step — and was removed.)"""
        return self.vae.moments(x)

    _vae_graphs = None
    _vae_live = None

    def _encode_graphed(self, x, generator, moments_only=False):
        """The differentiable VAE encoder (fixed shape, frozen weights: ~350 launches forward, ~400 backward) as two HIP-graph
        launches: torch.cuda.make_graphed_callables captures `moments` and its backward; the stochastic part of
        latent_dist.sample() (ipa_guidance.py:522-531) stays outside the graph.  First call of a shape runs eagerly (lazy
        one-time initialisations must not be captured), the second captures."""
        self._check_graph_signature()
        if self._vae_graphs is None:
            self._vae_graphs = {}
        key = (tuple(x.shape), x.dtype, x.device.index, fused.graph_signature())
        ent = self._vae_graphs.get(key)
        if ent is None:
            self._vae_graphs[key] = "warm"
            return self._vae_moments(x) if moments_only else self._vae_encode(x, generator)
        if ent == "warm":
            sample = torch.zeros_like(x, memory_format=torch.channels_last).requires_grad_(True)
            with _no_gc(), fused.capture_owner(id(self)):
                ent = self._vae_graphs[key] = torch.cuda.make_graphed_callables(lambda t_: self._vae_moments(t_), (sample,), num_warmup_iters=2)
        # make_graphed_callables keeps ONE set of static activations: a second forward before the first one's backward would
        # overwrite what that backward reads.  While an earlier output of this graph is still alive and has not been
        # back-propagated, the call runs eagerly instead (same kernels, same values).
        if self._vae_live is None:
            self._vae_live = {}
        live = self._vae_live.get(key)
        if live is not None and live[0]() is not None and not live[1][0]:
            return self._vae_moments(x) if moments_only else self._vae_encode(x, generator)
        import weakref
        moments = ent(x)
        done = [False]
        moments.register_hook(lambda g_, d_=done: d_.__setitem__(0, True))
        self._vae_live[key] = (weakref.ref(moments), done)
        return moments if moments_only else self.vae.sample(moments, generator)

    # ------------------------------------------------------------------ gradients
    def _prompt_embeds(self, prompt_utils, elevation, azimuth, center, all_vis_all, camera_distances, n_sets):
        B = elevation.shape[0]
        text = prompt_utils.get_text_embeddings(elevation, azimuth, center, all_vis_all, camera_distances,
                                                self.cfg.view_dependent_prompting).to(self.weights_dtype)
        pos_t, neg_t, null_t = text[:B], text[B:2 * B], text[2 * B:3 * B]
        ex = lambda e: e.expand(B, -1, -1) if e.shape[0] == 1 else e[:B]  # noqa: E731
        pos = torch.cat([pos_t, ex(self.pos_image_embeds)], dim=1)
        neg = torch.cat([neg_t, ex(self.neg_image_embeds)], dim=1)
        if n_sets == 2:
            return torch.cat([neg, pos], dim=0)                    # ipa_guidance.py:470
        null = torch.cat([null_t, ex(self.null_image_embeds)], dim=1)
        return torch.cat([neg, pos, null], dim=0)

    _embed_table = None

    def _prompt_embeds_anpg_table(self, prompt_utils, elevation, azimuth, center, all_vis_all, camera_distances):
        """`_prompt_embeds(..., 3)` as ONE gather: the 13 view directions x (neg | pos | null) rows, image-prompt tokens
        appended, are laid out once as a [3 * 13, 81, 768] table (rebuilt when the prompt tables or the image embeddings
        change); a step selects row set * 13 + direction(view).  Same values (a gather instead of gather + four concatenations)."""
        from .prompts import direction_index
        if not hasattr(prompt_utils, "text_embeddings_vd"):
            return None
        img = (self.pos_image_embeds, self.neg_image_embeds, self.null_image_embeds)
        one = getattr(self, "_image_embeds_single", None)
        if one is not None and one[1] == tuple(id(e) for e in img):
            img = one[0]                           # prepare_for_sds tiled these single rows to batch_size identical rows
        srcs = (prompt_utils.text_embeddings_vd, prompt_utils.uncond_text_embeddings_vd, prompt_utils.null_embeddings) + tuple(img)
        if not self.cfg.view_dependent_prompting or any(e.shape[0] != 1 for e in srcs[2:]):
            return None
        key = tuple((id(e), e._version, e.data_ptr()) for e in srcs) + (self.weights_dtype,)
        if self._embed_table is None or self._embed_table[0] != key:
            dt, dev = self.weights_dtype, self.device
            vd, un, nl, pos, neg, nul = (e.to(device=dev, dtype=dt) for e in srcs)
            D = vd.shape[0]
            rows = torch.cat([torch.cat([un, neg.expand(D, -1, -1)], dim=1),            # neg  (ipa_guidance.py:470 order)
                              torch.cat([vd, pos.expand(D, -1, -1)], dim=1),            # pos
                              torch.cat([nl.expand(D, -1, -1), nul.expand(D, -1, -1)], dim=1)], dim=0).contiguous()
            self._embed_table = (key, rows, D, srcs)
        _, rows, D, _ = self._embed_table
        idx = direction_index(elevation, azimuth, center, all_vis_all, camera_distances, prompt_utils.head_offset)
        flat = torch.cat([idx, idx + D, idx + 2 * D]).to(rows.device, non_blocking=True)       # host-side batch: host arithmetic
        return rows.index_select(0, flat)

    def compute_grad_anpg(self, latents, control_img, t, prompt_utils, use_pose_controlnet, all_vis_all, elevation,
                          azimuth, center, camera_distances, generator=None, control_embedding=None):
        B = elevation.shape[0]
        embeds = self._prompt_embeds(prompt_utils, elevation, azimuth, center, all_vis_all, camera_distances, 3)
        assert embeds.shape[1] == TEXT_TOKENS + IP_TOKENS
        with torch.no_grad():
            noise = sds.per_sample(lambda k, g: torch.randn((k,) + tuple(latents.shape[1:]), device=latents.device, dtype=latents.dtype,
                                                          generator=g), latents.shape[0], generator)
            latents_noisy = sds.add_noise(latents, noise, t, self.alphas)
            # the three branches share their pose maps: the ControlNet hint stem runs on the B distinct maps and tiles
            noise_pred = self.forward_unet(torch.cat([latents_noisy] * 3, dim=0), control_img,
                                           torch.cat([t] * 3), embeds, use_pose_controlnet, replicas=3,
                                           control_embedding=control_embedding)
            direction = sds.anpg_direction(noise_pred, t, self.cfg.guidance_scale)
        grad = sds.sds_weight(t, self.alphas, self.cfg.weighting_strategy) * direction
        if self.cfg.grad_clip_pixel:
            grad = sds.clip_grad_pixel(grad, self.cfg.grad_clip_threshold)
        return grad, {"t_orig": t, "latents_noisy": latents_noisy, "noise_pred": noise_pred, "neg_guidance_weights": None}

    def compute_grad_sds(self, latents, control_img, t, prompt_utils, use_pose_controlnet, all_vis_all, elevation,
                         azimuth, center, camera_distances, generator=None, control_embedding=None):
        embeds = self._prompt_embeds(prompt_utils, elevation, azimuth, center, all_vis_all, camera_distances, 2)
        with torch.no_grad():
            noise = sds.per_sample(lambda k, g: torch.randn((k,) + tuple(latents.shape[1:]), device=latents.device, dtype=latents.dtype,
                                                          generator=g), latents.shape[0], generator)
            latents_noisy = sds.add_noise(latents, noise, t, self.alphas)
            noise_pred = self.forward_unet(torch.cat([latents_noisy] * 2, dim=0), control_img,
                                           torch.cat([t] * 2), embeds, use_pose_controlnet, replicas=2,
                                           control_embedding=control_embedding)
            direction = sds.cfg_direction(noise_pred, noise, self.cfg.guidance_scale, self.cfg.guidance_rescale)
        # no per-pixel clip here: the reference applies grad_clip_pixel only on the ANPG path (:427-431 vs :513)
        grad = sds.sds_weight(t, self.alphas, self.cfg.weighting_strategy) * direction
        return grad, {"t_orig": t, "latents_noisy": latents_noisy, "noise_pred": noise_pred}

    # ------------------------------------------------------------------ the plugin call, glue fused (guidance/glue.py)
    def _fused_call_ok(self, rgb, use_pose_controlnet):
        """The fp16 CUDA ANPG training path: every element-wise chain between the rasterizer and the networks is one launch."""
        return (glue.ENABLED and self.cfg.use_anpg and rgb.is_cuda and self.weights_dtype == torch.float16 and self.cfg.channels_last and
                self.cfg.weighting_strategy in ("sds", "fantasia3d") and glue.image_prep_supported(rgb.permute(0, 3, 1, 2), (512, 512)) and
                not (use_pose_controlnet and _TWO_STREAMS and not _GRAPH_DENOISE))

    def _call_fused(self, step, rgb, control, prompt_utils, use_pose_controlnet, all_vis_all, elevation, azimuth, center,
                    camera_distances, generator):
        """Same values as the op-chain path below (same random draws in the same order: the VAE's sample noise, the timesteps,
        the diffusion noise), with the chains of ipa_guidance.py:612-614 / :524-531 / :395-431 / :645-653 as four launches."""
        B = rgb.shape[0]
        x = glue.image_prep(rgb.permute(0, 3, 1, 2), (512, 512))
        # The three random draws (same generator order as the op chain: VAE sample noise, timesteps, diffusion noise) and the prompt
        # table do not depend on the VAE encoder: they are enqueued in front of it, so that the latents-independent head of the
        # denoise (hint stem, time embeddings, prompt-token keys / values) can start on the side stream beside it (round 6)
        lat_shape = (self.vae.conv_out.out_channels // 2, x.shape[2] // 8, x.shape[3] // 8)
        draw = lambda k, g: torch.randn((k,) + lat_shape, device=x.device, dtype=self.weights_dtype, generator=g)  # noqa: E731
        eps = sds.per_sample(draw, B, generator)                     # VAEEncoder.sample's draw
        t = self.schedule.sample(step, B, self.device, generator)
        noise = sds.per_sample(draw, B, generator)                   # compute_grad_anpg's draw
        embeds = self._prompt_embeds_anpg_table(prompt_utils, elevation, azimuth, center, all_vis_all, camera_distances)
        if embeds is None:
            embeds = self._prompt_embeds(prompt_utils, elevation, azimuth, center, all_vis_all, camera_distances, 3)
        assert embeds.shape[1] == TEXT_TOKENS + IP_TOKENS
        t3 = t.repeat(3)
        token = None
        if use_pose_controlnet:
            token = self.launch_denoise_prologue((3 * B,) + lat_shape, self.weights_dtype, control, t3, embeds, True, 3)
        moments = self._moments(x)
        assert tuple(moments.shape[1:]) == (2 * lat_shape[0],) + lat_shape[1:] and moments.dtype == self.weights_dtype
        if not glue.latent_sample_supported(moments, eps, noise, t, self.alphas):
            raise RuntimeError("fused guidance glue: unsupported tensors (set GIP_FUSED_GLUE=0 for the op-chain path)")
        latents, latents_noisy = glue.latent_sample(moments, eps, noise, t, self.alphas, self.vae.scaling_factor, 3)
        with torch.no_grad():
            noise_pred = self.forward_unet(latents_noisy, control, t3, embeds, use_pose_controlnet, replicas=3, borrow_output=True, prologue=token)
        clip = self.cfg.grad_clip_threshold if self.cfg.grad_clip_pixel else None
        loss_sds, grad, grad_norm = glue.anpg_loss(latents, noise_pred, t, self.alphas, self.cfg.guidance_scale,
                                                   self.cfg.weighting_strategy, clip)
        return {"loss_sds": loss_sds, "grad_norm": grad_norm}

    # ------------------------------------------------------------------ the plugin call
    def __call__(self, step, rgb, control_img, prompt_utils, use_pose_controlnet, all_vis_all, elevation, azimuth,
                 center, camera_distances, generator=None, **kwargs: Any):
        """rgb [B,H,W,3], control_img [B,h,w,3] -> {"loss_sds", "grad_norm"}."""
        B = rgb.shape[0]
        control = control_img.permute(0, 3, 1, 2)
        if self._fused_call_ok(rgb, use_pose_controlnet):
            return self._call_fused(step, rgb, control, prompt_utils, use_pose_controlnet, all_vis_all, elevation, azimuth, center,
                                    camera_distances, generator)
        rgb_512 = F.interpolate(rgb.permute(0, 3, 1, 2), (512, 512), mode="bilinear", align_corners=False)
        hint = None
        if use_pose_controlnet and control.is_cuda and _TWO_STREAMS and not _GRAPH_DENOISE:
            # the ControlNet's hint stem depends on the pose maps only: it runs on the side stream beside the VAE encoder
            # (with the graph-captured denoise the stem is part of the graph instead)
            main, side = torch.cuda.current_stream(control.device), self._side_stream(control.device)
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                hint = self.embed_control(control)       # consumed on the side stream (forward_unet)
        latents = self.encode_images(rgb_512.to(self.weights_dtype), generator)
        t = self.schedule.sample(step, B, self.device, generator)
        fn = self.compute_grad_anpg if self.cfg.use_anpg else self.compute_grad_sds
        grad, _ = fn(latents, control, t, prompt_utils, use_pose_controlnet, all_vis_all, elevation, azimuth, center,
                     camera_distances, generator, control_embedding=hint)
        loss_sds, grad = sds.sds_loss(latents, grad)
        return {"loss_sds": loss_sds, "grad_norm": grad.norm()}


def _register_with_threestudio():
    """`@threestudio.register("ipa-guidance")` (ipa_guidance.py:71) when the host framework is importable: the system's
    `threestudio.find(self.cfg.guidance_type)(self.cfg.guidance)` (GaussianIP.py:355) then returns this class."""
    try:
        import threestudio
        register = threestudio.register
    except Exception:       # threestudio itself needs pytorch_lightning / omegaconf / tinycudann at import time
        return False
    register(StableDiffusionGuidance.registry_name)(StableDiffusionGuidance)
    return True


REGISTERED = _register_with_threestudio()
