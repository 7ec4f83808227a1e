"""distCUDA2 replacement: mean squared distance to the 3 nearest neighbours, on the GPU through the C-ABI.

Reference: simple_knn._C.distCUDA2 (gaussiansplatting/submodules/simple-knn/ext.cpp:16, spatial.cu:16-25,
simple_knn.cu:185-221); called once at initialisation (scene/gaussian_model.py:123)."""
import ctypes

import torch

from . import _lib


MODES = {"auto": 0, "all_pairs": 1, "box_pruned": 2}


def distCUDA2(points: torch.Tensor, mode: str = "auto") -> torch.Tensor:
    """`mode`: "auto" (all pairs up to 32 768 points, Morton-sorted box pruning above — simple_knn.cu:45-185), "all_pairs" or
    "box_pruned"; every mode returns the identical float per point (exact 3-NN, the same arithmetic)."""
    if not points.is_cuda:
        raise ValueError("distCUDA2: points must be a GPU tensor (no CPU path in the product)")
    pts = points.detach().float().contiguous()
    P = int(pts.shape[0])
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    lib = _lib.knn_lib()
    m = MODES[mode]
    with torch.cuda.device(pts.device):
        ws_bytes = lib.gip_knn_workspace_bytes_mode(P, m)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pts.device)
        rc = lib.gip_knn_mean_dist2_mode(P, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(ws.data_ptr()), ws_bytes, m,
                                         ctypes.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("gip_knn_mean_dist2_mode failed with status %d" % rc)
    return out
