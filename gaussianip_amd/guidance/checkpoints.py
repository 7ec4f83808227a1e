"""Loading real SD1.5 / ControlNet / VAE / IP-Adapter-FaceID weights into this repo's networks.

The reference builds its modules with diffusers 0.27 (`ipa_guidance.py:127-233`, `refine.py:38-100`) and loads
`runwayml/stable-diffusion-v1-5`-shaped checkpoints; `networks.py` states the same architectures with shorter module
names.  This file is the name translation (the tensors are identical in shape and meaning):

    diffusers_key(kind, ours) -> the key a diffusers state_dict uses for our parameter `ours`
    load_diffusers_state_dict(module, state_dict, kind) -> loads a diffusers-format dict (e.g. from safetensors)
    load_ip_adapter_faceid(unet, ip_state)              -> the LoRA + image-prompt projections of IP-Adapter-FaceID

kinds: "unet" (UNet2DConditionModel, 686 tensors), "controlnet" (ControlNetModel, 340), "vae_encoder" (AutoencoderKL
encoder + quant_conv, 108), "vae_decoder" (decoder + post_quant_conv, 140) — the counts are diffusers' own.
IP-Adapter-FaceID stores its processor weights as a ModuleList state dict indexed by the position of the processor in
`unet.attn_processors` (ip_adapter_faceid.py:286-329, 331-336); diffusers registers `down_blocks`, then `up_blocks`, then
`mid_block`, so that order is down (12 processors), up (18), mid (2), attn1 before attn2 inside every block.
No checkpoint can be fetched in the build environment: the tests check the translation tables (counts, uniqueness, known
diffusers keys, round trips), not a real file.
"""
import re
from typing import Dict

import torch

from .networks import Attention, UNet

_INNER = (
    (r"\.block\.attn([12])\.to_out\.", r".transformer_blocks.0.attn\1.to_out.0."),
    (r"\.block\.ff_in\.", r".transformer_blocks.0.ff.net.0.proj."),
    (r"\.block\.ff_out\.", r".transformer_blocks.0.ff.net.2."),
    (r"\.block\.", r".transformer_blocks.0."),
)


def _encoder_side(k: str) -> str:
    """Keys shared by the U-Net and the ControlNet (time embedding, conv_in, down blocks, mid block)."""
    k = re.sub(r"^time_l([12])\.", lambda m: "time_embedding.linear_%s." % m.group(1), k)
    k = re.sub(r"^down_res\.(\d+)\.", lambda m: "down_blocks.%d.resnets.%d." % divmod(int(m.group(1)), 2), k)
    k = re.sub(r"^down_attn\.(\d+)\.", lambda m: "down_blocks.%d.attentions.%d." % divmod(int(m.group(1)), 2), k)
    k = re.sub(r"^down_sample\.(\d+)\.conv\.", r"down_blocks.\1.downsamplers.0.conv.", k)
    k = re.sub(r"^mid_res([12])\.", lambda m: "mid_block.resnets.%d." % (int(m.group(1)) - 1), k)
    k = re.sub(r"^mid_attn\.", "mid_block.attentions.0.", k)
    for pat, rep in _INNER:
        k = re.sub(pat, rep, k)
    return k


def diffusers_key(kind: str, ours: str) -> str:
    k = ours
    if kind == "unet":
        k = re.sub(r"^up_res\.(\d+)\.", lambda m: "up_blocks.%d.resnets.%d." % divmod(int(m.group(1)), 3), k)
        k = re.sub(r"^up_attn\.(\d+)\.", lambda m: "up_blocks.%d.attentions.%d." % divmod(int(m.group(1)), 3), k)
        k = re.sub(r"^up_sample\.(\d+)\.conv\.", r"up_blocks.\1.upsamplers.0.conv.", k)
        k = re.sub(r"^norm_out\.", "conv_norm_out.", k)
        return _encoder_side(k)
    if kind == "controlnet":
        def stem(m):
            i = int(m.group(1))
            return "controlnet_cond_embedding." + ("conv_in." if i == 0 else "conv_out." if i == 7 else "blocks.%d." % (i - 1))
        k = re.sub(r"^cond_stem\.(\d+)\.", stem, k)
        k = re.sub(r"^zero_convs\.(\d+)\.", r"controlnet_down_blocks.\1.", k)
        k = re.sub(r"^mid_zero\.", "controlnet_mid_block.", k)
        return _encoder_side(k)
    if kind in ("vae_encoder", "vae_decoder"):
        side = "encoder" if kind == "vae_encoder" else "decoder"
        if k.startswith(("quant_conv.", "post_quant_conv.")):
            return k
        k = re.sub(r"^mid_res([12])\.", lambda m: "mid_block.resnets.%d." % (int(m.group(1)) - 1), k)
        k = re.sub(r"^mid_norm\.", "mid_block.attentions.0.group_norm.", k)
        k = re.sub(r"^mid_attn\.to_out\.", "mid_block.attentions.0.to_out.0.", k)
        k = re.sub(r"^mid_attn\.", "mid_block.attentions.0.", k)
        k = re.sub(r"^norm_out\.", "conv_norm_out.", k)
        if kind == "vae_encoder":
            k = re.sub(r"^res\.(\d+)\.", lambda m: "down_blocks.%d.resnets.%d." % divmod(int(m.group(1)), 2), k)
            k = re.sub(r"^down\.(\d+)\.conv\.", r"down_blocks.\1.downsamplers.0.conv.", k)
        else:
            k = re.sub(r"^res\.(\d+)\.", lambda m: "up_blocks.%d.resnets.%d." % divmod(int(m.group(1)), 3), k)
            k = re.sub(r"^up\.(\d+)\.conv\.", r"up_blocks.\1.upsamplers.0.conv.", k)
        return side + "." + k
    raise ValueError("unknown kind %r" % kind)


# AutoencoderKL checkpoints saved before diffusers 0.18 name the mid attention differently
_VAE_OLD = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}


def load_diffusers_state_dict(module: torch.nn.Module, state_dict: Dict[str, torch.Tensor], kind: str, strict: bool = True):
    """Copy a diffusers-format `state_dict` into `module` (one of this repo's networks, WITHOUT LoRA / IP branches in
    its own key set: they come from load_ip_adapter_faceid).  Returns the list of our keys that were not found."""
    from . import fused
    fused.bump_weights_epoch()          # captured HIP graphs hold derived copies of the old weights
    own = module.state_dict()
    missing, loaded = [], {}
    for ours, dst in own.items():
        if ".lora_" in ours or "_ip." in ours:
            continue
        key = diffusers_key(kind, ours)
        src = state_dict.get(key)
        if src is None and kind.startswith("vae"):
            for new, old in _VAE_OLD.items():
                alt = key.replace("attentions.0." + new + ".", "attentions.0." + old + ".")
                if alt in state_dict:
                    src = state_dict[alt]
                    break
        if src is None:
            missing.append(ours)
            continue
        if src.shape != dst.shape:
            if src.numel() == dst.numel():            # linear <-> 1x1 convolution (use_linear_projection, old VAE attention)
                src = src.reshape(dst.shape)
            else:
                raise ValueError("%s: checkpoint shape %s, module shape %s" % (key, tuple(src.shape), tuple(dst.shape)))
        loaded[ours] = src.to(dst.dtype)
    if strict and missing:
        raise KeyError("%d parameters not in the checkpoint, e.g. %s" % (len(missing), missing[:4]))
    module.load_state_dict(loaded, strict=False)
    return missing


def attention_processor_order(unet: UNet):
    """Our Attention modules in the order of diffusers' `unet.attn_processors`: down_blocks, up_blocks, mid_block."""
    out = []
    for seq in (unet.down_attn, unet.up_attn):
        for st in seq:
            if hasattr(st, "block"):
                out += [st.block.attn1, st.block.attn2]
    out += [unet.mid_attn.block.attn1, unet.mid_attn.block.attn2]
    return out


def ip_adapter_key_map(unet: UNet) -> Dict[str, str]:
    """{key inside the checkpoint's "ip_adapter" dict: our state-dict key} for a U-Net built with lora_rank and ip_adapter."""
    names = {m: n for n, m in unet.named_modules() if isinstance(m, Attention)}
    table = {}
    for idx, att in enumerate(attention_processor_order(unet)):
        base = names[att]
        for ours, theirs in (("lora_q", "to_q_lora"), ("lora_k", "to_k_lora"), ("lora_v", "to_v_lora"), ("lora_out", "to_out_lora")):
            if att.lora_rank:
                table["%d.%s.down.weight" % (idx, theirs)] = "%s.%s.0.weight" % (base, ours)
                table["%d.%s.up.weight" % (idx, theirs)] = "%s.%s.1.weight" % (base, ours)
        if att.ip:
            table["%d.to_k_ip.weight" % idx] = base + ".to_k_ip.weight"
            table["%d.to_v_ip.weight" % idx] = base + ".to_v_ip.weight"
    return table


def load_ip_adapter_faceid(unet: UNet, ip_state: Dict[str, torch.Tensor], strict: bool = True):
    """`ip_state` = torch.load(ip-adapter-faceid-*.bin)["ip_adapter"].  Call before UNet.fold_lora()."""
    from . import fused
    fused.bump_weights_epoch()
    own = unet.state_dict()
    table = ip_adapter_key_map(unet)
    loaded = {}
    for theirs, ours in table.items():
        if theirs in ip_state:
            loaded[ours] = ip_state[theirs].to(own[ours].dtype)
        elif strict:
            raise KeyError(theirs)
    unet.load_state_dict(loaded, strict=False)
    return [k for k in table if k not in ip_state]


def load_safetensors(path: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return load_file(path)
