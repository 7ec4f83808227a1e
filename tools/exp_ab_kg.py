#!/usr/bin/env python3
"""Same-process A/B of the two-K-group GEMM (gip_dbg_linear_kg 0 / 1): the 1-view shard of configs[3] (proxy_group = 4, batch-3 networks)
and the full 4-view step, alternating, graphs re-captured after every flip."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_ahds  # noqa: E402
from gaussianip_amd import _lib  # noqa: E402

KG = ctypes.c_int.in_dll(_lib.nn_lib()._lib, "gip_dbg_linear_kg")
for rep in range(2):
    for mode in (0, 1):
        KG.value = mode
        g = bench_ahds.cached_guidance()
        if g is not None:
            g.invalidate_graphs()
        shard = bench_ahds.measure(steps=10, warmup=4, proxy_group=4, pieces=False)
        full = bench_ahds.measure(steps=10, warmup=4, pieces=False)
        print("two K groups %s | 1-view shard %.2f ms | full step %.2f ms" % ("on " if mode else "off", shard["ms_per_step"], full["ms_per_step"]), flush=True)
