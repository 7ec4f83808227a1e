import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scenes
from gaussianip_amd.knn import distCUDA2
for P in (100000, 1000000):
    pts = torch.from_numpy(scenes.human_points(P, np.random.default_rng(0)).astype(np.float32)).cuda()
    distCUDA2(pts[:1000]); torch.cuda.synchronize()
    t0 = time.perf_counter(); d = distCUDA2(pts); torch.cuda.synchronize()
    print("P=%d  %.1f ms  mean dist2 %.3e" % (P, (time.perf_counter() - t0) * 1e3, float(d.mean())), flush=True)
