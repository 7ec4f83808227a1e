"""The sixteen GEMMs of a Winograd F(2x2, 3x3) convolution: one batched hipBLASLt call (torch.bmm, what fused._winograd_conv
issues) against sixteen launches of the own MFMA linear (gip_linear_f16) and against ONE launch of the batched own GEMM
(gip_linear_batched_f16, blockIdx.y = product; round 5), all replayed from a HIP graph so that launch overhead is what it is
inside the captured denoise.  Shapes: every (grid, Cin -> Cout) the Winograd path takes at 12 / 6 / 3
samples (fused._WINOGRAD_DEFAULT).  VERDICT r3 item 5: keep the library only where it is measured faster, shape by shape."""
import os
import sys

import ctypes

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import fused  # noqa: E402


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1000.0        # us


SHAPES = [(16, 1280, 1280), (16, 2560, 1280), (16, 1920, 1280), (16, 640, 1280), (32, 1280, 640), (32, 960, 640), (32, 1920, 640)]
gen = torch.Generator(device="cuda").manual_seed(0)
from gaussianip_amd import _lib  # noqa: E402
print("%-34s | %12s | %12s | %12s | own16/lib  batched/lib" % ("tiles x Cin -> Cout (16 GEMMs)", "bmm us", "16 own us", "batched own us"))
with torch.no_grad():
    for batch in (12, 6, 3):
        for H, cin, cout in SHAPES:
            T = batch * (H // 2) * (H // 2)
            V = (torch.randn(16, T, cin, device="cuda", generator=gen) * 0.5).half()
            U = (torch.randn(16, cout, cin, device="cuda", generator=gen) * 0.05).half()
            Ut = U.transpose(1, 2)
            outs = [None]

            def lib():
                outs[0] = torch.bmm(V, Ut)

            def own():
                outs[0] = [fused.linear(V[i], U[i]) for i in range(16)]
            if not fused.linear_supported(V[0], U[0]):
                continue
            Mb = torch.empty((16, T, cout), dtype=torch.float16, device="cuda")

            def batched():
                rc = _lib.nn_lib().gip_linear_batched_f16(ctypes.c_void_p(V.data_ptr()), ctypes.c_void_p(U.data_ptr()), ctypes.c_void_p(Mb.data_ptr()),
                                                          16, T, cin, cout, T * cin, cout * cin, T * cout,
                                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            t_lib = graph_time(lib)
            ref = outs[0]
            t_own = graph_time(own)
            err = max(float((outs[0][i].float() - ref[i].float()).abs().max()) for i in range(16))
            t_b = graph_time(batched)
            errb = float((Mb.float() - ref.float()).abs().max())
            fl = 2.0 * 16 * T * cin * cout
            print("b%2d %2d^2 %5d x %4d -> %4d       | %7.1f %4.0fTF | %7.1f %4.0fTF | %7.1f %4.0fTF | %.2f  %.2f  maxdiff %.3g %.3g" %
                  (batch, H, T, cin, cout, t_lib, fl / t_lib / 1e6, t_own, fl / t_own / 1e6, t_b, fl / t_b / 1e6, t_own / t_lib, t_b / t_lib, err, errb),
                  flush=True)
