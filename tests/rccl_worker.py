"""Child process of tests/test_gpu_rccl.py: ONE rank on the RCCL backend ("nccl" on ROCm).  A 1-GPU box cannot form a
multi-rank RCCL group, but a single-rank group still drives every call of the per-step exchange through the backend the
multi-GPU bench uses (process-group creation with device_id, sub-groups, int32 MAX and float SUM all-reduces, async
handles, the differentiable group maximum) — an unsupported dtype / op / argument combination fails here, not on the
8-GPU node."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, out_path = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from gaussianip_amd import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    parallel._on = lambda group: True                     # a 1-rank group is "off" by default: force the collective path
    sub = dist.new_group([0])
    g = torch.Generator(device=dev).manual_seed(0)
    P, V = 5000, 2
    params = [torch.nn.Parameter(torch.randn(P, k, device=dev, generator=g)) for k in (3, 1, 4)]
    for p in params:
        p.grad = torch.randn(p.shape, device=dev, generator=g)
    want = [p.grad.clone() for p in params]
    vsg = torch.randn(V, P, 3, device=dev, generator=g)
    res = {}
    for name, group in (("world", None), ("sub", sub)):
        norm = parallel.exchange_sum(params, group=group, viewspace_grads=vsg)
        res[name + "_sum"] = all(torch.equal(p.grad, w) for p, w in zip(params, want)) and \
            bool(torch.allclose(norm, torch.linalg.vector_norm(vsg[..., :2], dim=-1).sum(0), rtol=1e-6, atol=1e-6))
        radii = torch.randint(0, 50, (V, P), device=dev, dtype=torch.int32, generator=g)
        depth = torch.rand(V, 1, 64, 64, device=dev, generator=g) * 3
        st = parallel.exchange_forward_stats(radii, depth, group=group)       # async int32 MAX bucket
        rmax, dmax = st.wait()
        res[name + "_max"] = bool(torch.equal(rmax, radii.amax(0))) and float(dmax) == float(depth.amax())
        r1, d1 = radii.amax(0).contiguous(), depth.amax()
        parallel.exchange_max(r1, d1, group=group)
        res[name + "_max_sync"] = bool(torch.equal(r1, radii.amax(0))) and float(d1) == float(depth.amax())
        x = torch.tensor(2.5, device=dev, requires_grad=True)
        y = parallel._GroupMax.apply(x * 2, group)
        (y * 3).backward()
        res[name + "_groupmax"] = float(y) == 5.0 and float(x.grad) == 6.0
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main()
