import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance
from gaussianip_amd.guidance.ahds import AHDSSchedule
dev = torch.device("cuda")
def timed(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
sched = AHDSSchedule(list(range(2400)))
for cl in (True, False):
  for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    g = StableDiffusionGuidance(GuidanceConfig(channels_last=cl), schedule=sched)
    B = 4
    lat = torch.randn(B, 4, 64, 64, device=dev); ctrl = torch.rand(B, 3, 512, 512, device=dev)
    emb = torch.randn(3 * B, 81, 768, device=dev, dtype=torch.float16) * 0.1
    tt = torch.randint(20, 800, (B,), device=dev)
    x3, c3, t3 = torch.cat([lat] * 3), torch.cat([ctrl] * 3), torch.cat([tt] * 3)
    den = timed(lambda: g.forward_unet(x3, c3, t3, emb, True))
    img = torch.rand(B, 3, 512, 512, device=dev, requires_grad=True)
    def vae_fb():
        z = g.encode_images(img); z.sum().backward()
    vae = timed(vae_fb)
    vf = timed(lambda: g.encode_images(img.detach()))
    # CUDA graph of the denoise
    gr_ms = None
    try:
        sx, sc, st, se = x3.clone(), c3.clone(), t3.clone(), emb.clone()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): g.forward_unet(sx, sc, st, se, True)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = g.forward_unet(sx, sc, st, se, True)
        gr_ms = timed(lambda: graph.replay())
    except Exception as e:
        gr_ms = "graph failed: %r" % (e,)
    print("channels_last=%s benchmark=%s denoise=%.2f ms  graph=%s  vae f+b=%.2f ms  vae fwd=%.2f ms" % (cl, bench, den, gr_ms, vae, vf), flush=True)
    del g; torch.cuda.empty_cache()
