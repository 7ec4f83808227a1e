"""SD1.5-shaped denoiser stack in plain PyTorch-ROCm: U-Net, pose ControlNet and the VAE encoder.

The reference runs diffusers 0.27 modules (threestudio/models/guidance/ipa_guidance.py:127-233, forward_unet :311-358,
encode_images :522-531) with IP-Adapter-FaceID attention processors installed on every U-Net attention
(ip_adapter/attention_processor_faceid.py:211-523, ip_adapter_faceid.py:286-329).  diffusers is not a dependency here:
this file states the same architectures directly (channel widths, block order, token layout), so that

  * the per-step compute — ControlNet(12) -> U-Net(12) at 64x64 latents, VAE encode of 4x512x512 — has exactly the
    reference's shapes and FLOPs (weights are random-initialised when no checkpoint is given: there is no network in
    the build environment, and `bench.py` says so in `data`);
  * the frozen LoRA branches (rank 128, scale 1.0; attention_processor_faceid.py:284,330-331,378) can be FOLDED into the
    base projections once (`fold_lora`), and the text / image-token cross-attentions of LoRAIPAttnProcessor2_0
    (:462-500) run as two SDPA calls on one projected query: `h = SDPA(q, k_text, v_text) + scale * SDPA(q, k_ip, v_ip)`.

Real checkpoints load through `checkpoints.load_diffusers_state_dict` / `load_ip_adapter_faceid` (conv / linear /
norm names are identical inside each block; only container names differ; tensor counts equal diffusers': 686 / 340 / 248).

GEMMs / convolutions go to hipBLASLt / MIOpen through PyTorch (the library path the task statement allows for plain
GEMM-shaped work); attention is torch SDPA.
"""
import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .fused import GroupNormAct, add_bias_residual, conv1x1, conv3x3, fusable, geglu

TEXT_TOKENS = 77
# measured: the fused GEGLU projection wins at the 64x64 level, hipBLASLt + geglu below (12288 rows = the 32x32 level: neutral)
_GEGLU_FUSE_MIN_ROWS = 32768
IP_TOKENS = 4


def timestep_embedding(t, dim=320, max_period=10000.0):
    """diffusers Timesteps(320, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _tile_batch(x, n):
    """x.repeat(n, 1, 1, 1) that keeps an NHWC tensor NHWC (Tensor.repeat returns NCHW-contiguous memory, which sends
    every consumer of the result — here the last up-block of the U-Net through its skip connection — down the
    unfused library path)."""
    if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return x.permute(0, 2, 3, 1).repeat(n, 1, 1, 1).permute(0, 3, 1, 2)
    return x.repeat(n, 1, 1, 1)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, temb_dim=1280, eps=1e-5):
        super().__init__()
        self.norm1 = GroupNormAct(32, cin, eps=eps, act=True)       # GroupNorm + SiLU fused (csrc/groupnorm.hip)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout) if temb_dim else None
        self.norm2 = GroupNormAct(32, cout, eps=eps, act=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        if fusable(x) and not self.conv1.weight.requires_grad:
            return self._forward_fused(x, temb)
        h = self.conv1(self.norm1(x))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(self.norm2(h))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h

    def _forward_fused(self, x, temb):
        """Same block with the pointwise work folded into the HIP kernels: conv1's bias and the time-embedding
        projection enter norm2 as its per-(sample, channel) addend; conv2's bias, the shortcut's bias and the residual
        add are one pass.  `self.staged_addend` (set by _Encoder.stage_time_embedding) already holds
        conv1.bias + time_emb_proj(silu(temb)) for this block when the batched projection is in use."""
        if fused.resblock_grad_supported(x, self):
            return fused.resblock_with_grad(x, self)         # differentiable path (VAE encoder): one autograd node per block
        addend, self.staged_addend = self.staged_addend, None
        if addend is not None and addend.shape[0] != x.shape[0]:
            addend = addend[:x.shape[0]]           # shared prefix of a replicated batch (see _Encoder.encode)
        if addend is None:
            addend = self.conv1.bias
            if self.time_emb_proj is not None:
                fused.fallback("time-embedding projection", temb, library=True)
                addend = F.linear(F.silu(temb), self.time_emb_proj.weight, self.time_emb_proj.bias + addend)
        # both convolutions feed a GroupNorm (conv1 -> norm2; conv2 -> the next block's norm1 / a transformer's norm /
        # norm_out): their epilogues take its statistics, so that GroupNorm reads its input once (apply) instead of twice
        # (conv3x3_gn: where the convolution runs as Winograd, the GroupNorm + SiLU is applied inside its input transform)
        h = fused.conv3x3_gn(x, self.norm1, None, self.conv1.weight, gn_next=True)
        if self.conv_shortcut is None:
            return fused.conv3x3_gn(h, self.norm2, addend, self.conv2.weight, self.conv2.bias, x, gn_next=True)
        return fused.conv3x3_gn(h, self.norm2, addend, self.conv2.weight, self.conv2.bias,
                                conv1x1(x, self.conv_shortcut.weight, self.conv_shortcut.bias), gn_next=True)

    staged_addend = None


class Attention(nn.Module):
    """Multi-head attention with optional LoRA branches (foldable) and optional IP-Adapter image-token branch."""

    def __init__(self, dim, ctx_dim=None, heads=8, lora_rank=0, ip=False, ip_scale=1.0):
        super().__init__()
        self.heads = heads
        ctx_dim = ctx_dim or dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.Linear(dim, dim)
        self.ip, self._ip_scale = ip, ip_scale
        if ip:
            self.to_k_ip = nn.Linear(ctx_dim, dim, bias=False)
            self.to_v_ip = nn.Linear(ctx_dim, dim, bias=False)
        self.lora_rank = lora_rank
        if lora_rank:
            mk = lambda i, o: nn.Sequential(nn.Linear(i, lora_rank, bias=False), nn.Linear(lora_rank, o, bias=False))  # noqa: E731
            self.lora_q, self.lora_k, self.lora_v, self.lora_out = mk(dim, dim), mk(ctx_dim, dim), mk(ctx_dim, dim), mk(dim, dim)

    @property
    def ip_scale(self):
        return self._ip_scale

    @ip_scale.setter
    def ip_scale(self, value):
        if value != self._ip_scale:
            fused.bump_weights_epoch()       # a captured graph holds the old value as a kernel argument
        self._ip_scale = value

    @torch.no_grad()
    def fold_lora(self, scale=1.0):
        """W' = W + scale * up @ down for q/k/v/out; removes the LoRA modules (inference-time identity)."""
        if not self.lora_rank:
            return
        fused.bump_weights_epoch()
        for base, lora in ((self.to_q, self.lora_q), (self.to_k, self.lora_k), (self.to_v, self.lora_v), (self.to_out, self.lora_out)):
            base.weight.add_(scale * (lora[1].weight.float() @ lora[0].weight.float()).to(base.weight.dtype))
        del self.lora_q, self.lora_k, self.lora_v, self.lora_out
        self.lora_rank = 0

    def _split(self, x):
        B, N, C = x.shape
        return x.view(B, N, self.heads, C // self.heads).transpose(1, 2)

    def _qkv(self, x, ctx):
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        if self.lora_rank:
            q, k, v = q + self.lora_q(x), k + self.lora_k(ctx), v + self.lora_v(ctx)
        return q, k, v

    def _kv(self, ctx):
        k, v = self.to_k(ctx), self.to_v(ctx)
        if self.lora_rank:
            k, v = k + self.lora_k(ctx), v + self.lora_v(ctx)
        return k, v

    def _sdpa(self, q, k, v):
        """[B, N, C] projections -> [B, Nq, C]; the HIP kernel where it applies, torch SDPA otherwise."""
        if fused.attention_supported(q, k, self.heads):
            return fused.attention(q, k, v, self.heads)
        fused.fallback("attention (_sdpa)", q)
        h = F.scaled_dot_product_attention(self._split(q), self._split(k), self._split(v))
        return h.transpose(1, 2).reshape(q.shape)

    refine = None        # RefineAttentionState when this layer is one of the VCR target self-attentions
    staged_kv = None     # (k, v, k_ip, v_ip) of the prompt tokens for the next call (_Encoder.stage_context)

    def _forward_refine(self, x):
        """Self-attention in the 'refine' state of LoRAAttnProcessor2_0 (attention_processor_faceid.py:291-364):
        key views store their tokens per denoising step; k0..k3 attend over [own | front-or-back] tokens (mutual
        self-attention); every other view blends its own attention with the attentions over its two neighbouring key
        views' stored tokens: lambda_self * self + (1 - lambda_self) * (w_l * left + w_r * right)."""
        st = self.refine
        name, step = st.ctl.cur_view_name, st.cur_denoise_step
        if "v" not in name:
            st.stored_zt.setdefault(name, []).append(x)
        if name in ("front", "back", "left", "right"):
            q, k, v = self._qkv(x, x)
            h = self._sdpa(q, k, v)
        elif name in ("k0", "k1", "k2", "k3"):
            other = st.stored_zt["front" if name in ("k0", "k1") else "back"][step]
            q, k, v = self._qkv(x, torch.cat([x, other], dim=1))
            h = self._sdpa(q, k, v)
        else:
            (ln, rn), (lw, rw) = st.ctl.cur_key_view_name_pair, st.ctl.cur_key_view_weight_pair
            q, k, v = self._qkv(x, x)
            kl, vl = self._kv(st.stored_zt[ln][step])
            kr, vr = self._kv(st.stored_zt[rn][step])
            h = st.ctl.lambda_self * self._sdpa(q, k, v) + (1.0 - st.ctl.lambda_self) * (
                lw * self._sdpa(q, kl, vl) + rw * self._sdpa(q, kr, vr))
        st.cur_denoise_step += 1
        if st.cur_denoise_step == st.ctl.total_denoise_step:
            st.cur_denoise_step = 0
        out = self.to_out(h)
        return out + self.lora_out(h) if self.lora_rank else out

    def _out(self, h, residual=None):
        """to_out (+ LoRA) (+ the block's residual, added in the GEMM epilogue when the MFMA linear applies)."""
        if self.lora_rank:
            out = self.to_out(h) + self.lora_out(h)
            return out if residual is None else residual + out
        if fused.linear_supported(h, self.to_out.weight) and (residual is None or residual.is_contiguous()):
            # with the block's residual the result is the next LayerNorm's input: leave its row sums (LayerNorm fold)
            return fused.linear_auto(h, self.to_out.weight, self.to_out.bias, residual, want_rows=residual is not None)
        out = self.to_out(h)
        return out if residual is None else residual + out

    def forward(self, x, ctx=None, residual=None):
        """`x` may be a fused.LNInput (the block's LayerNorm not yet applied): the q | k | v projection of the self-attention and
        the to_q projection of the cross-attention fold it into their GEMM; every other path materialises it first."""
        lnin = x if isinstance(x, fused.LNInput) else None
        if lnin is not None:
            hot_self = (ctx is None and not (self.refine is not None and self.refine.ctl.state == "refine") and not self.lora_rank and
                        self.to_q.bias is None and fused.qkv_fusable(lnin.x, self.to_q.weight))
            hot_cross = (ctx is not None and self.staged_kv is not None and self.staged_kv[0].shape[0] == lnin.x.shape[0] and
                         not self.lora_rank)
            if not (hot_self or hot_cross):
                x, lnin = lnin.materialize(), None
            else:
                x = lnin.x            # (shape / identity tests below; the projections take `lnin`)
        ip_ctx = None
        if ctx is None:
            if self.refine is not None and self.refine.ctl.state == "refine":
                out = self._forward_refine(x)
                return out if residual is None else residual + out
            ctx = x
        elif self.ip:
            ctx, ip_ctx = ctx[:, :-IP_TOKENS], ctx[:, -IP_TOKENS:]
        staged, self.staged_kv = self.staged_kv, None
        if staged is not None and staged[0].shape[0] == x.shape[0] and not self.lora_rank:
            # key / value projections of the prompt tokens, made for all layers at once by _Encoder.stage_context
            q = lnin.linear(self.to_q.weight, self.to_q.bias) if lnin is not None else fused.linear_auto(x, self.to_q.weight, self.to_q.bias)
            k, v, k_ip, v_ip = staged
            if fused.attention_supported(q, k, self.heads):
                if k_ip is None:
                    return self._out(fused.attention(q, k, v, self.heads), residual)
                return self._out(fused.attention(q, k, v, self.heads, k_ip, v_ip, self.ip_scale), residual)
            if lnin is not None:
                x, lnin = lnin.materialize(), None
        if ctx is x and not self.lora_rank and self.to_q.bias is None and fused.qkv_fusable(x, self.to_q.weight):
            # self-attention with frozen, folded weights: ONE [3C, C] projection (the tokens are read once, not three times);
            # q, k, v are column ranges of its output and the attention kernel reads them in place through row strides
            C = x.shape[-1]
            wqkv = fused.qkv_weight(self.to_q.weight, self.to_k.weight, self.to_v.weight)
            qkv = lnin.linear(wqkv) if lnin is not None else fused.linear_auto(x, wqkv)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            if fused.attention_supported(q, k, self.heads):
                return self._out(fused.attention(q, k, v, self.heads), residual)
        if lnin is not None:             # no folding path applied after all: the LayerNorm kernel, and self-attention attends to ITS output
            self_attn = ctx is x
            x, lnin = lnin.materialize(), None
            if self_attn:
                ctx = x
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        if self.lora_rank:
            q, k, v = q + self.lora_q(x), k + self.lora_k(ctx), v + self.lora_v(ctx)
        if fused.attention_supported(q, k, self.heads):       # [B, N, C] in and out: no head transposes
            if ip_ctx is None:
                return self._out(fused.attention(q, k, v, self.heads), residual)
            # decoupled cross-attention: text and image-prompt keys in one pass over the queries
            return self._out(fused.attention(q, k, v, self.heads, self.to_k_ip(ip_ctx), self.to_v_ip(ip_ctx), self.ip_scale), residual)
        if self.heads == 1 and ip_ctx is None and q.shape[-1] >= 256 and fused.wide_head_attention_supported(q, k):
            # single wide head (the VAE's 512-channel mid attention): dense GEMMs at the FLOP minimum around this repo's in-place
            # softmax kernels (fused._WideHeadAttention); the flash kernels lose at head dim 512 (forward and backward; measured in
            # tools/exp_ab_vae.py)
            return self._out(fused.wide_head_attention(q, k, v), residual)
        fused.fallback("attention", q)
        q = self._split(q)
        h = F.scaled_dot_product_attention(q, self._split(k), self._split(v))
        if ip_ctx is not None:
            h = h + self.ip_scale * F.scaled_dot_product_attention(q, self._split(self.to_k_ip(ip_ctx)), self._split(self.to_v_ip(ip_ctx)))
        B, H, N, D = h.shape
        return self._out(h.transpose(1, 2).reshape(B, N, H * D), residual)


class TransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads, lora_rank, ip, ip_scale):
        super().__init__()
        self.norm1 = fused.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, lora_rank)
        self.norm2 = fused.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads, lora_rank, ip=ip, ip_scale=ip_scale)
        self.norm3 = fused.LayerNorm(dim)
        self.ff_in = nn.Linear(dim, dim * 8)     # GEGLU: value | gate
        self.ff_out = nn.Linear(dim * 4, dim)

    def forward(self, x, ctx, replicas=1):
        # the three LayerNorms are handed to their consumers un-applied (fused.LNInput): where the consuming projection runs on
        # this repo's MFMA linear and x carries its producer's row sums, the normalisation is folded into that GEMM's epilogue
        x = self.attn1(fused.LNInput(x, self.norm1), None, x)  # residual adds ride in the out-projection's epilogue
        if replicas > 1:                                       # x held one copy of `replicas` identical samples so far
            x = x.repeat(replicas, 1, 1)
        x = self.attn2(fused.LNInput(x, self.norm2), ctx, x)
        h = None
        if x.shape[0] * x.shape[1] >= _GEGLU_FUSE_MIN_ROWS:
            h = fused.linear_ln(x, self.norm3, self.ff_in.weight, self.ff_in.bias, True)      # LayerNorm AND GEGLU in the GEMM's epilogue
        if h is not None:
            pass
        elif x.shape[0] * x.shape[1] >= _GEGLU_FUSE_MIN_ROWS and fused.linear_supported(x, self.ff_in.weight):
            h = fused.linear(self.norm3(x), self.ff_in.weight, self.ff_in.bias, None, True)        # GEGLU in the GEMM epilogue
        else:
            h = geglu(self.ff_in(self.norm3(x)))
        if fused.linear_supported(h, self.ff_out.weight) and x.is_contiguous():
            return fused.linear_auto(h, self.ff_out.weight, self.ff_out.bias, x)
        return x + self.ff_out(h)


class SpatialTransformer(nn.Module):
    def __init__(self, dim, ctx_dim=768, heads=8, lora_rank=0, ip=False, ip_scale=1.0):
        super().__init__()
        self.norm = GroupNormAct(32, dim, eps=1e-6, act=False)
        self.proj_in = nn.Conv2d(dim, dim, 1)
        self.block = TransformerBlock(dim, ctx_dim, heads, lora_rank, ip, ip_scale)
        self.proj_out = nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx, replicas=1):
        """`replicas` > 1: x holds ONE copy of `replicas` samples that are identical up to the first cross-attention
        (same latents / timestep / pose map, different prompts); norm, proj_in and the self-attention run on that
        copy and the result is tiled before attn2 — the same values, 1/replicas of the work."""
        B, C, H, W = x.shape
        if fusable(x):      # NHWC: the 1x1 projections are GEMMs on the token view, no layout change anywhere
            t = fused.linear_auto(self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C), self.proj_in.weight.reshape(C, C), self.proj_in.bias,
                                  want_rows=True)      # norm1 of the block reads this tensor: its row sums come out of the epilogue
            res = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
            if replicas > 1:
                res = res.repeat(replicas, 1, 1)
            holder = []
            t = fused.linear(self.block(t, ctx, replicas), self.proj_out.weight.reshape(C, C), self.proj_out.bias, res, stats=holder)   # the block's residual rides in the epilogue
            # the next ResnetBlock2D's norm1 reads this tensor: its statistics came out of the projection's epilogue
            return fused.attach_stats(t.reshape(B * replicas, H, W, C).permute(0, 3, 1, 2), holder[0] if holder else None)
        h = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.block(h, ctx, replicas).reshape(B * replicas, H, W, C).permute(0, 3, 1, 2)
        return (_tile_batch(x, replicas) if replicas > 1 else x) + self.proj_out(h)


class Downsample(nn.Module):
    def __init__(self, c, asymmetric=False):
        super().__init__()
        self.asymmetric = asymmetric
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0 if asymmetric else 1)

    def forward(self, x):
        if self.asymmetric:
            return fused.downsample_asym(x, self.conv.weight, self.conv.bias)
        return fused.downsample_sym(x, self.conv.weight, self.conv.bias)


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return fused.upsample2x_conv3x3(x, self.conv.weight, self.conv.bias)


class _Encoder(nn.Module):
    """conv_in + time embedding + the four down blocks + mid block shared by the U-Net and the ControlNet."""
    widths = (320, 640, 1280, 1280)

    def __init__(self, lora_rank, ip, ip_scale):
        super().__init__()
        self.time_l1, self.time_l2 = nn.Linear(320, 1280), nn.Linear(1280, 1280)
        self.conv_in = nn.Conv2d(4, 320, 3, padding=1)
        self.down_res, self.down_attn, self.down_sample = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        c = 320
        for i, w in enumerate(self.widths):
            for _ in range(2):
                self.down_res.append(ResBlock(c, w))
                self.down_attn.append(SpatialTransformer(w, 768, 8, lora_rank, ip, ip_scale) if i < 3 else nn.Identity())
                c = w
            self.down_sample.append(Downsample(w) if i < 3 else nn.Identity())
        self.mid_res1, self.mid_attn, self.mid_res2 = ResBlock(1280, 1280), SpatialTransformer(1280, 768, 8, lora_rank, ip, ip_scale), ResBlock(1280, 1280)

    def temb(self, t, dtype):
        from . import glue
        e = glue.timestep_embedding(t) if glue.timestep_embedding_supported(t, dtype) else timestep_embedding(t).to(dtype)
        temb = self.time_l2(F.silu(self.time_l1(e)))
        self.stage_time_embedding(temb)
        return temb

    _temb_pack = None

    def prologue(self, t, ctx, dtype):
        """Everything of a forward that does not depend on the latents: the time embedding, every ResnetBlock2D's addend (one packed
        projection) and every cross-attention's prompt-token keys / values (one or two packed projections).  Returns a handle for
        `forward(..., pro=handle)`; the modules' staging slots are left empty.  The guidance runs this on its side stream while the
        VAE encoder is still busy (ipa_guidance.launch_denoise_prologue); needs prepare_inference()."""
        temb = self.temb(t, dtype)
        self.stage_context(ctx)
        blocks = self._temb_pack[0] if self._temb_pack is not None else []
        cross = self._ctx_pack[0] if self._ctx_pack is not None else []
        pro = (temb, [b.staged_addend for b in blocks], [a.staged_kv for a in cross])
        for b in blocks:
            b.staged_addend = None
        for a in cross:
            a.staged_kv = None
        return pro

    def install(self, pro):
        """Puts a prologue()'s staged tensors where the blocks look for them; returns the time embedding."""
        temb, adds, kvs = pro
        for b, v in zip(self._temb_pack[0] if self._temb_pack is not None else [], adds):
            b.staged_addend = v
        for a, v in zip(self._ctx_pack[0] if self._ctx_pack is not None else [], kvs):
            a.staged_kv = v
        return temb

    @torch.no_grad()
    def prepare_inference(self):
        """Call once the (frozen) weights are final: packs every ResnetBlock2D's time_emb_proj into ONE [sum(C), 1280]
        projection whose bias also carries conv1's bias, so a forward runs a single GEMM for all the blocks' addends."""
        fused.bump_weights_epoch()
        blocks = [m for m in self.modules() if isinstance(m, ResBlock) and m.time_emb_proj is not None]
        W = torch.cat([b.time_emb_proj.weight for b in blocks]).contiguous()
        bias = torch.cat([b.time_emb_proj.bias + b.conv1.bias for b in blocks]).contiguous()
        offs, o = [], 0
        for b in blocks:
            offs.append((o, o + b.conv1.out_channels))
            o += b.conv1.out_channels
        self._temb_pack = (blocks, W, bias, offs)
        # every cross-attention layer projects the SAME prompt tokens: one [sum(2 C), 768] matrix for all their to_k / to_v
        # (and one for the IP-Adapter's to_k_ip / to_v_ip over the 4 image tokens) turns 4 tiny GEMMs per layer into 2 per
        # forward.  Only for folded LoRA (a live LoRA branch keeps the per-layer path).
        cross = [m.attn2 for m in self.modules() if isinstance(m, TransformerBlock)]
        self._ctx_pack = None
        if cross and not any(a.lora_rank for a in cross):
            Wt = torch.cat([torch.cat([a.to_k.weight, a.to_v.weight]) for a in cross]).contiguous()
            Wi = torch.cat([torch.cat([a.to_k_ip.weight, a.to_v_ip.weight]) for a in cross]).contiguous() if all(a.ip for a in cross) else None
            offs, o = [], 0
            for a in cross:
                c = a.to_k.weight.shape[0]
                offs.append((o, c))
                o += 2 * c
            self._ctx_pack = (cross, Wt, Wi, offs)
        return self

    _ctx_pack = None

    def stage_context(self, ctx):
        """Projects the prompt tokens `ctx` [B, 77 (+ 4 image tokens), 768] for every cross-attention layer of this
        network at once; each layer then reads its keys / values as a column range of the result (strided in place)."""
        if self._ctx_pack is None or not (ctx.is_cuda and ctx.dtype == torch.float16) or fused._DISABLED or \
                (torch.is_grad_enabled() and ctx.requires_grad):
            return
        cross, Wt, Wi, offs = self._ctx_pack
        B = ctx.shape[0]
        fused.fallback("prompt-token key / value projections of all layers (one wide GEMM)", ctx, library=True)
        if Wi is not None:
            text = F.linear(ctx[:, :-IP_TOKENS].reshape(-1, ctx.shape[-1]), Wt).view(B, -1, Wt.shape[0])
            ip = F.linear(ctx[:, -IP_TOKENS:].reshape(-1, ctx.shape[-1]), Wi).view(B, IP_TOKENS, Wi.shape[0])
        else:
            text, ip = F.linear(ctx.reshape(-1, ctx.shape[-1]), Wt).view(B, -1, Wt.shape[0]), None
        for a, (o, c) in zip(cross, offs):
            a.staged_kv = (text[:, :, o:o + c], text[:, :, o + c:o + 2 * c],
                           None if ip is None else ip[:, :, o:o + c], None if ip is None else ip[:, :, o + c:o + 2 * c])

    def stage_time_embedding(self, temb):
        if self._temb_pack is None or not (temb.is_cuda and temb.dtype == torch.float16):
            return
        blocks, W, bias, offs = self._temb_pack
        fused.fallback("time-embedding projections of all blocks (one wide GEMM)", temb, library=True)
        allp = F.linear(F.silu(temb), W, bias)          # [N, sum(C)]; each block reads its column range in place
        for b, (lo, hi) in zip(blocks, offs):
            b.staged_addend = allp[:, lo:hi]

    def encode(self, h, temb, ctx, replicas=1):
        """`replicas` > 1: `h` (and the rows of `temb` that matter) hold one copy of `replicas` identical samples — the
        [neg | pos | null] branches of compute_grad_anpg share latents, timestep and pose map (ipa_guidance.py:397-399)
        and first differ at the first cross-attention — so conv_in, the first ResnetBlock2D and the first transformer's
        self-attention half run once per distinct sample; everything from attn2 on sees the full batch."""
        skips = [_tile_batch(h, replicas) if replicas > 1 else h]
        for i in range(4):
            for j in range(2):
                first = replicas > 1 and i == 0 and j == 0
                h = self.down_res[2 * i + j](h, temb[:h.shape[0]] if first and temb is not None else temb)
                if i < 3:
                    h = self.down_attn[2 * i + j](h, ctx, replicas) if first else self.down_attn[2 * i + j](h, ctx)
                skips.append(h)
            if i < 3:
                h = self.down_sample[i](h)
                skips.append(h)
        return h, skips

    def mid(self, h, temb, ctx):
        return self.mid_res2(self.mid_attn(self.mid_res1(h, temb), ctx), temb)


class UNet(_Encoder):
    """UNet2DConditionModel, SD1.5 configuration (block_out_channels 320/640/1280/1280, 2 layers per block,
    cross_attention_dim 768, 8 heads).  16 attention layers of each kind = the 32 processors the reference replaces."""

    def __init__(self, lora_rank=128, ip_adapter=True, ip_scale=0.5):
        super().__init__(lora_rank, ip_adapter, ip_scale)
        skip_c = [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
        self.up_res, self.up_attn, self.up_sample = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        c = 1280
        for i, w in enumerate((1280, 1280, 640, 320)):
            for _ in range(3):
                self.up_res.append(ResBlock(c + skip_c.pop(), w))
                self.up_attn.append(SpatialTransformer(w, 768, 8, lora_rank, ip_adapter, ip_scale) if i > 0 else nn.Identity())
                c = w
            self.up_sample.append(Upsample(w) if i < 3 else nn.Identity())
        self.norm_out = GroupNormAct(32, 320, act=True)
        self.conv_out = nn.Conv2d(320, 4, 3, padding=1)

    def forward(self, x, t, ctx, down_residuals: Optional[List[torch.Tensor]] = None, mid_residual=None, replicas=1, pro=None):
        if pro is not None:
            temb = self.install(pro)
        else:
            temb = self.temb(t, x.dtype)
            self.stage_context(ctx)
        b = x.shape[0] // replicas
        h, skips = self.encode(fused.conv3x3_latent_in(x[:b] if replicas > 1 else x, self.conv_in.weight, self.conv_in.bias), temb, ctx, replicas)
        h = self.mid(h, temb, ctx)
        if callable(down_residuals):          # the ControlNet ran on another stream beside the encoder: join here
            down_residuals, mid_residual = down_residuals()
        residuals = [None] * len(skips) if down_residuals is None else list(down_residuals)
        if down_residuals is not None:
            h = h + mid_residual
        k = 0
        for i in range(4):
            for _ in range(3):
                # skip + ControlNet residual, the concatenation and norm1's statistics: one pass (fused.cat_skip)
                h = self.up_res[k](fused.cat_skip(h, skips.pop(), residuals.pop()), temb)
                if i > 0:
                    h = self.up_attn[k](h, ctx)
                k += 1
            if i < 3:
                h = self.up_sample[i](h)
        return fused.conv3x3_narrow_out(self.norm_out(h), self.conv_out.weight, self.conv_out.bias)      # 320 -> 4

    @torch.no_grad()
    def fold_lora(self, scale=1.0):
        for m in self.modules():
            if isinstance(m, Attention):
                m.fold_lora(scale)
        return self


class ControlNet(_Encoder):
    """ControlNetModel (control_v11p_sd15_openpose shape): U-Net encoder copy + conditioning stem + 13 zero convs.
    The reference installs NO special processor here, so it cross-attends over all 81 tokens."""

    def __init__(self):
        super().__init__(0, False, 1.0)
        chans = (16, 32, 96, 256)
        stem = [nn.Conv2d(3, 16, 3, padding=1)]
        for a, b in zip(chans[:-1], chans[1:]):
            stem += [nn.Conv2d(a, a, 3, padding=1), nn.Conv2d(a, b, 3, padding=1, stride=2)]
        stem.append(nn.Conv2d(256, 320, 3, padding=1))
        self.cond_stem = nn.ModuleList(stem)
        self.zero_convs = nn.ModuleList([nn.Conv2d(c, c, 1) for c in (320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280)])
        self.mid_zero = nn.Conv2d(1280, 1280, 1)

    def embed_condition(self, cond):
        """controlnet_cond_embedding: the 8-convolution hint stem (image -> 320 channels at 1/8 resolution).  It depends on
        the pose map only — not on the latents or the timestep — so a caller that denoises the same view repeatedly (the 8
        DDIM steps of the refine pass) computes it once and hands it to forward() as `cond_embedding`."""
        c = cond
        for i, conv in enumerate(self.cond_stem):
            if i < len(self.cond_stem) - 1:
                c = fused.conv3x3_fewch(c, conv.weight, conv.bias, conv.stride[0], act=True)     # bias + SiLU in the epilogue
            else:
                c = conv3x3(c, conv.weight, conv.bias)       # 256 -> 320 at 1/8 resolution: the MFMA kernel's shape
        return c

    def forward(self, x, t, ctx, cond, conditioning_scale=1.0, cond_embedding=None, replicas=1, pro=None) -> Tuple[List[torch.Tensor], torch.Tensor]:
        """`cond` may hold fewer samples than x (B / k): the hint stem then runs once per distinct hint and its output
        is tiled k times — the three guidance branches of compute_grad_anpg share their pose maps (ipa_guidance.py:397-399)."""
        if pro is not None:
            temb = self.install(pro)
        else:
            temb = self.temb(t, x.dtype)
            self.stage_context(ctx)
        c = self.embed_condition(cond) if cond_embedding is None else cond_embedding
        b = x.shape[0] // replicas
        if replicas > 1:
            x, c = x[:b], c[:b]
        elif c.shape[0] != x.shape[0]:
            c = c.repeat(x.shape[0] // c.shape[0], 1, 1, 1)
        h, skips = self.encode(fused.conv3x3_latent_in(x, self.conv_in.weight, self.conv_in.bias) + c, temb, ctx, replicas)
        h = self.mid(h, temb, ctx)
        down = [conv1x1(s, z.weight, z.bias) for z, s in zip(self.zero_convs, skips)]
        mid = conv1x1(h, self.mid_zero.weight, self.mid_zero.bias)
        if conditioning_scale != 1.0:
            down, mid = [d * conditioning_scale for d in down], mid * conditioning_scale
        return down, mid


class VAEEncoder(nn.Module):
    """AutoencoderKL.encode (sd-vae-ft-mse shape): 3 -> 128/256/512/512, mid attention, 8-channel moments, quant_conv."""
    scaling_factor = 0.18215

    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv2d(3, 128, 3, padding=1)
        self.res, self.down = nn.ModuleList(), nn.ModuleList()
        c = 128
        for i, w in enumerate((128, 256, 512, 512)):
            for _ in range(2):
                self.res.append(ResBlock(c, w, temb_dim=0, eps=1e-6))
                c = w
            self.down.append(Downsample(w, asymmetric=True) if i < 3 else nn.Identity())
        self.mid_res1, self.mid_res2 = ResBlock(512, 512, 0, 1e-6), ResBlock(512, 512, 0, 1e-6)
        self.mid_norm = GroupNormAct(32, 512, eps=1e-6, act=False)
        self.mid_attn = Attention(512, None, heads=1)
        self.mid_attn.to_q, self.mid_attn.to_k, self.mid_attn.to_v = nn.Linear(512, 512), nn.Linear(512, 512), nn.Linear(512, 512)
        self.norm_out = GroupNormAct(32, 512, eps=1e-6, act=True)
        self.conv_out = nn.Conv2d(512, 8, 3, padding=1)
        self.quant_conv = nn.Conv2d(8, 8, 1)

    def moments(self, x):
        h = fused.conv3x3_few_inputs(x, self.conv_in.weight, self.conv_in.bias)
        for i in range(4):
            h = self.res[2 * i + 1](self.res[2 * i](h))
            h = self.down[i](h)
        h = self.mid_res1(h)
        B, C, H, W = h.shape
        a = self.mid_attn(self.mid_norm(h).permute(0, 2, 3, 1).reshape(B, H * W, C))
        h = h + a.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self.mid_res2(h)
        if fusable(h) and not self.conv_out.weight.requires_grad and self.conv_out.weight.dtype == torch.float16:
            # quant_conv (1x1, 8 -> 8) composed into conv_out: one kernel forward, one backward (no library convolution in the VAE)
            w2, b2 = fused.folded_quant_conv(self.conv_out, self.quant_conv)
            return fused.conv3x3_narrow_out(self.norm_out(h), w2, b2)
        return self.quant_conv(fused.conv3x3_narrow_out(self.norm_out(h), self.conv_out.weight, self.conv_out.bias))      # 512 -> 8

    def encode(self, x, generator=None):
        """latent_dist.sample() * scaling_factor — stochastic and differentiable, like ipa_guidance.py:522-531."""
        return self.sample(self.moments(x), generator)

    def sample(self, moments, generator=None):
        mean, logvar = moments.chunk(2, dim=1)
        std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
        from .sds import per_sample
        noise = per_sample(lambda k, g: torch.randn((k,) + tuple(mean.shape[1:]), device=mean.device, dtype=mean.dtype, generator=g),
                           mean.shape[0], generator)
        return (mean + std * noise) * self.scaling_factor


class VAEDecoder(nn.Module):
    """AutoencoderKL.decode (sd-vae-ft-mse shape): post_quant_conv, 4 -> 512, mid attention, up blocks 512/512/256/128
    with three ResnetBlock2D each and nearest-2x upsampling after the first three, 128 -> 3.  Used by the VCR refine pass
    (refine.py:220-239 -> pipeline_ipa_controlnet.py:1856-1859)."""

    def __init__(self):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        self.conv_in = nn.Conv2d(4, 512, 3, padding=1)
        self.mid_res1, self.mid_res2 = ResBlock(512, 512, 0, 1e-6), ResBlock(512, 512, 0, 1e-6)
        self.mid_norm = GroupNormAct(32, 512, eps=1e-6, act=False)
        self.mid_attn = Attention(512, None, heads=1)
        self.mid_attn.to_q, self.mid_attn.to_k, self.mid_attn.to_v = nn.Linear(512, 512), nn.Linear(512, 512), nn.Linear(512, 512)
        self.res, self.up = nn.ModuleList(), nn.ModuleList()
        c = 512
        for i, w in enumerate((512, 512, 256, 128)):
            for _ in range(3):
                self.res.append(ResBlock(c, w, temb_dim=0, eps=1e-6))
                c = w
            self.up.append(Upsample(w) if i < 3 else nn.Identity())
        self.norm_out = GroupNormAct(32, 128, eps=1e-6, act=True)
        self.conv_out = nn.Conv2d(128, 3, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(self.post_quant_conv(z))
        h = self.mid_res1(h)
        B, C, H, W = h.shape
        a = self.mid_attn(self.mid_norm(h).permute(0, 2, 3, 1).reshape(B, H * W, C))
        h = h + a.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self.mid_res2(h)
        for i in range(4):
            for j in range(3):
                h = self.res[3 * i + j](h)
            h = self.up[i](h)
        return self.conv_out(self.norm_out(h))


def init_for_benchmark(module, seed=0):
    """Deterministic random initialisation with activations of order one in fp16 (the real checkpoints are not
    shippable): normal(0, 0.02)-style weights scaled per fan-in, zero biases, LoRA up-projections small."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(fan_in)))
            else:
                p.zero_()
        for m in module.modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):   # GroupNormAct is an nn.GroupNorm
                m.weight.fill_(1.0)
    return module
