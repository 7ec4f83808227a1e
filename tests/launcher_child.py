"""Rank process for tests/test_bench_launcher_cpu.py: what bench.py's launcher starts, minus the GPU — joins a gloo group
from the RANK / WORLD_SIZE / MASTER_* environment bench.spawn_ranks hands out, all-reduces, prints noise on stdout and (rank
0) one result line."""
import json
import os
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
if mode == "hang":
    # every rank blocks for good and ignores SIGTERM (a rank stuck inside the driver): only the launcher's deadline + SIGKILL end it
    import signal
    import time
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    print("[noise] rank %s is stuck" % os.environ["RANK"], flush=True)
    while True:
        time.sleep(1)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print("[noise] rank %d of %d says hello on stdout" % (rank, world), flush=True)
if mode == "fail" and rank == world - 1:
    sys.exit(7)
if mode == "silent":
    dist.barrier()
    sys.exit(0)
if rank == 0:
    print(json.dumps({"not_the_line": True}), flush=True)
    print(json.dumps({"metric": "launcher_selftest", "value": float(t.item()), "n_gpus": world, "dist_world_size": dist.get_world_size()}), flush=True)
if mode != "fail":
    dist.barrier()
dist.destroy_process_group()
