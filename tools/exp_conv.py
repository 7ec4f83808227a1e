"""Conv shapes of the SD1.5 denoiser: MIOpen (NHWC fp16) vs the equivalent dense GEMM through hipBLASLt."""
import torch, time
dev = "cuda"
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
shapes = [(12, 320, 320, 64), (12, 640, 640, 32), (12, 1280, 1280, 16), (12, 1280, 1280, 8), (12, 2560, 1280, 8), (12, 2560, 1280, 16),
          (12, 1920, 1280, 16), (12, 1920, 640, 32), (12, 1280, 640, 32), (12, 960, 640, 32), (12, 960, 320, 64), (12, 640, 320, 64),
          (4, 128, 128, 512), (4, 128, 256, 256), (4, 256, 256, 256), (4, 256, 512, 128), (4, 512, 512, 128), (4, 512, 512, 64)]
for N, ci, co, hw in shapes:
    x = torch.randn(N, ci, hw, hw, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=dev, dtype=torch.half).contiguous(memory_format=torch.channels_last) * 0.01
    b = torch.zeros(co, device=dev, dtype=torch.half)
    fl = 2.0 * N * hw * hw * ci * co * 9
    t_c = timed(lambda: torch.nn.functional.conv2d(x, w, None, padding=1))
    t_cb = timed(lambda: torch.nn.functional.conv2d(x, w, b, padding=1))
    A = torch.randn(N * hw * hw, 9 * ci, device=dev, dtype=torch.half)
    B = torch.randn(co, 9 * ci, device=dev, dtype=torch.half)
    t_g = timed(lambda: torch.nn.functional.linear(A, B))
    # 9 shifted GEMMs accumulating in place over the padded flat index
    Mp = N * (hw + 2) * (hw + 2)
    xin = torch.randn(Mp + 2 * (hw + 3), ci, device=dev, dtype=torch.half)
    out = torch.zeros(Mp, co, device=dev, dtype=torch.half)
    Wk = [torch.randn(ci, co, device=dev, dtype=torch.half) * 0.01 for _ in range(9)]
    offs = [(dy * (hw + 2) + dx) for dy in range(3) for dx in range(3)]
    def nine():
        torch.mm(xin[offs[0]:offs[0] + Mp], Wk[0], out=out)
        for k in range(1, 9):
            out.addmm_(xin[offs[k]:offs[k] + Mp], Wk[k])
    t_9 = timed(nine)
    print("N%2d %4d->%4d @%3d  %6.1f GF | miopen %.3f ms %5.0f TF/s (bias +%.3f) | gemm K=9C %.3f ms %5.0f TF/s | 9 shifted %.3f ms %5.0f TF/s" %
          (N, ci, co, hw, fl / 1e9, t_c, fl / t_c / 1e9, t_cb - t_c, t_g, fl / t_g / 1e9, t_9, fl / t_9 / 1e9), flush=True)
