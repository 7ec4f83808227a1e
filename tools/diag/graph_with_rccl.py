"""HIP-graph capture of the denoise / VAE with a LIVE RCCL process group (its watchdog thread polls events from another
thread; a capture in the default 'global' error mode would be invalidated by such a call).  One rank is enough to start
the watchdog: init the group, run a collective, then let tools/bench_ahds.py capture its graphs during the warm-up steps
and keep issuing a collective per step, as the sharded layout does.
usage: graph_with_rccl.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
import bench_ahds  # noqa: E402

# proxy_group=4: one rank's share of a 4-rank seed group, exchange packed / unpacked; here the collectives are REAL (world 1)
out = bench_ahds.measure(steps=4, warmup=4, pieces=False, proxy_group=0)
for _ in range(3):
    dist.all_reduce(t)
torch.cuda.synchronize()
print("graphs captured and replayed with a live RCCL group:", out["ms_per_step"], "ms per step")
dist.destroy_process_group()
