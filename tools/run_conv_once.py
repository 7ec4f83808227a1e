import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused
N, ci, co, H = 12, 640, 640, 32
if len(sys.argv) > 1: N, ci, co, H = [int(a) for a in sys.argv[1:5]]
x = torch.randn(N, ci, H, H, device="cuda").half().contiguous(memory_format=torch.channels_last)
w = (torch.randn(co, ci, 3, 3, device="cuda") * 0.01).half().contiguous(memory_format=torch.channels_last)
if len(sys.argv) > 5:        # timing ablation of the K loop (WRONG results): bit 0 no activation DMA, 1 no weight DMA, 2 no MFMA
    import ctypes
    from gaussianip_amd import _lib
    ctypes.c_int.in_dll(_lib.nn_lib()._lib, "gip_dbg_conv_ablate").value = int(sys.argv[5])
for _ in range(3):
    fused.conv3x3(x, w)
torch.cuda.synchronize()
