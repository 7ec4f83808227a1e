"""Stage-3 training step (GaussianIP.py:424-436): 4 of the 32 refine views rendered at 1024^2, cropped, halved, L1 + LPIPS
against the cached refined-image features, backward, Adam.  Prints ms per step with the MFMA path and with plain ops."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from argparse import ArgumentParser
from gaussianip_amd.arguments import OptimizationParams, PipelineParams
from gaussianip_amd.guidance import fused
from gaussianip_amd.guidance.perceptual import LPIPSVGG
from gaussianip_amd.scene import GaussianModel
from gaussianip_amd.scene.cameras import Camera
from gaussianip_amd.system import StageThreeStep, create_refine_batch
from gaussianip_amd.utils import BasicPointCloud

dev = torch.device("cuda")
P = int(os.environ.get("P", 100000))
pts = scenes.human_points(P, np.random.default_rng(0)).astype(np.float32)
gm = GaussianModel(0)
gm.create_from_pcd(BasicPointCloud(pts, np.full((P, 3), 0.5, np.float32), None), 4.0)
gm.training_setup(OptimizationParams(ArgumentParser()))
pipe = PipelineParams(ArgumentParser())
bg = torch.ones(3, device=dev)
b = create_refine_batch()
cams = [Camera(c2w=b["c2w"][i], FoVy=float(b["fovy"][i]), height=1024, width=1024, data_device=dev) for i in range(32)]
refined = torch.rand(32, 1024, 1024, 3, device=dev)
from gaussianip_amd.guidance.refine import VIEW_IDX_ALL

def run(lp, n=10):
    st = StageThreeStep(gm, pipe, bg, cams, refined, VIEW_IDX_ALL, lambda_l1=10.0, lambda_lpips=15.0 if lp is not None else 0.0,
                        perceptual=lp, train_bs=4)
    def step():
        out = st.training_step()
        gm.optimizer.zero_grad(set_to_none=True)
        out["loss"].backward()
        gm.optimizer.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("L1 only                      : %.2f ms / step" % run(None))
lp16 = LPIPSVGG().init_for_benchmark(0).prepare_inference(dev)
print("L1 + LPIPS fp16, MFMA convs  : %.2f ms / step" % run(lp16))
with fused.disabled():
    print("L1 + LPIPS fp16, MIOpen convs: %.2f ms / step" % run(lp16))
lp32 = LPIPSVGG().init_for_benchmark(0).to(dev)
print("L1 + LPIPS fp32 (reference's precision), MIOpen: %.2f ms / step" % run(lp32))
