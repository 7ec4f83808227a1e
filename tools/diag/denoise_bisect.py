"""Which layer of the denoise is not reproducible at the 2-view-shard shape (batch 6)?  Forward hooks record a checksum of every
block's output over repeated eager calls on identical inputs; the first module (in execution order) whose checksum moves is
printed.  GIP_GUIDANCE_STREAMS=1 runs it on one stream (a cross-stream race then disappears)."""
import os
import sys

os.environ["GIP_GRAPH_DENOISE"] = "0"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance import GuidanceConfig, StableDiffusionGuidance, fused, networks  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
RUNS = 5
dev = torch.device("cuda")
gd = StableDiffusionGuidance(GuidanceConfig())
gg = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(B, 4, 64, 64, device=dev, generator=gg)
t = torch.randint(20, 800, (B,), device=dev, generator=gg)
ctrl = torch.rand(B, 3, 512, 512, device=dev, generator=gg)
emb = (torch.randn(3 * B, 81, 768, device=dev, generator=gg) * 0.1).half()
x, tt = torch.cat([lat] * 3), torch.cat([t] * 3)

log = []
kinds = (networks.ResBlock, networks.SpatialTransformer, networks.TransformerBlock, networks.Attention, networks.Downsample,
         networks.Upsample, fused.LayerNorm, fused.GroupNormAct, torch.nn.Conv2d, torch.nn.Linear)


def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(o):
            log.append((name, tuple(o.shape), o.double().sum(), o.double().abs().sum()))
    return f


for net, tag in ((gd.unet, "unet"), (gd.controlnet, "controlnet")):
    for n, m in net.named_modules():
        if isinstance(m, kinds):
            m.register_forward_hook(hook(tag + "." + n))

runs = []
with torch.no_grad():
    for r in range(RUNS):
        log.clear()
        out = gd.forward_unet(x, ctrl, tt, emb, True, replicas=3)
        torch.cuda.synchronize()
        runs.append(([(n, s, float(a), float(b)) for n, s, a, b in log], out.clone()))
print("B=%d streams=%s: outputs equal to run 0: %s" % (B, os.environ.get("GIP_GUIDANCE_STREAMS", "2"), [bool(torch.equal(o, runs[0][1])) for _, o in runs]))
base = runs[0][0]
# per network (the two run concurrently: execution order is only meaningful inside one network)
for tag in ("controlnet", "unet"):
    ref = [e for e in base if e[0].startswith(tag)]
    for r in range(1, RUNS):
        cur = [e for e in runs[r][0] if e[0].startswith(tag)]
        bad = [(i, a, b) for i, (a, b) in enumerate(zip(ref, cur)) if a != b]
        if bad:
            i, a, b = bad[0]
            print("  run %d %s: first differing module #%d of %d: %s %s   sum %.6f vs %.6f   |sum| %.6f vs %.6f   (%d modules differ)" % (
                r, tag, i, len(ref), a[0], a[1], a[2], b[2], a[3], b[3], len(bad)))
            if i > 0:
                print("      previous module (equal): %s %s" % (ref[i - 1][0], ref[i - 1][1]))
        else:
            print("  run %d %s: all %d module outputs equal" % (r, tag, len(ref)))
