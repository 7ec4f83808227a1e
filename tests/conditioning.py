"""Random-initialised denoisers ignore their conditioning: eps_pos ~ eps_null, so ANPG's 7.5 (eps_pos - eps_null) amplifies the
fp16 rounding of `noise_pred` 229x (profiles/r05_anpg_sensitivity.json) and "sharded == unsharded" can only be asserted at the
10 % level there — a bar that would also pass a real 5 % sharding bug.  `strengthen_conditioning` makes the
conditioning MATTER on the same random weights: the key / value projections of every cross-attention (text tokens and image-prompt
tokens; attention_processor_faceid.py:433-523) of the U-Net and the ControlNet are scaled, which sharpens the attention over the
prompt tokens and raises the weight of what it reads.  Measured in float32 (16^2 latents): |eps_pos - eps_null| / |eps_pos| =
0.037 as initialised, 0.15 at x4, 0.60 at x16 (the default), 1.37 at x64.  The fp16 floor of the ANPG gradient then sits at a few
1e-3 and a 1 % bar on batch-6-vs-batch-12 gradients means something."""
import torch


@torch.no_grad()
def strengthen_conditioning(guidance, factor=16.0):
    from gaussianip_amd.guidance.networks import Attention
    n = 0
    for net in (guidance.unet, guidance.controlnet):
        for m in net.modules():
            if isinstance(m, Attention) and m.to_k.in_features == 768:       # cross-attention: keys / values come from the 768-wide prompt tokens
                for lin in (m.to_k, m.to_v) + ((m.to_k_ip, m.to_v_ip) if m.ip else ()):
                    lin.weight.mul_(factor)
                n += 1
    assert n == 16 + 7, n       # SD1.5: 16 transformer blocks in the U-Net, 7 in the ControlNet's encoder + mid block
    # everything derived from the weights: the packed prompt-token projections, pinned graph operands, captured graphs
    if guidance.device.type == "cuda" and guidance.weights_dtype == torch.float16:
        guidance.unet.prepare_inference()
        guidance.controlnet.prepare_inference()
    guidance.invalidate_graphs()
    return guidance
