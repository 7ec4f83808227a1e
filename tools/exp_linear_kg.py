"""Two K groups inside a workgroup (conv3x3_kernel<..., KG = 2>, round 6) against the one-group kernel on the GEMMs whose grids leave
at most one workgroup per CU: HIP-graph replay of 12 launches cycling through 12 weight copies (cold weights, as in the denoise);
outputs compared (the sum is first half + second half of K instead of one chain: equal to fp32 rounding of the partial sums)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianip_amd import _lib  # noqa: E402
from exp_linear_narrow import graph_time, COPIES, p  # noqa: E402

lib = _lib.nn_lib()
KG = ctypes.c_int.in_dll(lib._lib, "gip_dbg_linear_kg")


def run(M, K, N):
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(M, K, device="cuda", generator=gen).half()
    wts = [(torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5).half() for _ in range(COPIES)]
    bias = torch.randn(N, device="cuda", generator=gen).half()
    res = torch.randn(M, N, device="cuda", generator=gen).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)

    def call(w):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.gip_linear_f16(p(x), p(w), p(bias), p(res), p(out), M, K, N, 0, st) == 0
    ts, outs = [], []
    for mode in (0, 1):
        KG.value = mode
        call(wts[0])
        torch.cuda.synchronize()
        outs.append(out.float().clone())
        ts.append(min(graph_time(lambda: [call(w) for w in wts]) for _ in range(3)))
    KG.value = -1
    ref = torch.addmm(bias.float(), x.float(), wts[0].float().t()) + res.float()
    lib_t = min(graph_time(lambda: [torch.addmm(bias, x, w.t()) for w in wts]) for _ in range(3))
    e0, e1 = float((outs[0] - ref).abs().max()), float((outs[1] - ref).abs().max())
    return ts, lib_t, float((outs[0] - outs[1]).abs().max()), e0, e1


if __name__ == "__main__":
    print("M, K, N | us: one K group, two K groups, hipBLASLt addmm (no residual) | max |KG1 - KG2|, max error vs fp32: KG1, KG2")
    for M in (192, 768, 3072):
        for K, N in ((1280, 1280), (1280, 3840), (5120, 1280), (640, 640), (640, 1920), (2560, 640), (320, 320), (1280, 320), (768, 1280)):
            ts, lib_t, d, e0, e1 = run(M, K, N)
            print("%-20s %6.1f  %6.1f  %6.1f   | %.1e  %.1e  %.1e" % ((M, K, N), ts[0], ts[1], lib_t, d, e0, e1), flush=True)
