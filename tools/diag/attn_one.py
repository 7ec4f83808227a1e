"""One shape of the self-attention kernel, a few launches (for PMC passes: tools/pmc_nn.sh tools/diag/attn_one.py attn_fwd)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused  # noqa: E402

B, H, N, D = 12, 8, 4096, 40
q, k, v = [torch.randn(B, N, H * D, device="cuda").half() for _ in range(3)]
with torch.no_grad():
    for _ in range(4):
        fused.attention(q, k, v, H)
torch.cuda.synchronize()
