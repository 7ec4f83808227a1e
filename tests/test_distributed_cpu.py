"""world_size-2 gloo tests of the per-step exchange (gaussianip_amd/parallel.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussianip_amd import parallel
    g = torch.Generator().manual_seed(100 + rank)
    P = 257
    params = [torch.nn.Parameter(torch.zeros(P, 3)), torch.nn.Parameter(torch.zeros(P, 1, 3)), torch.nn.Parameter(torch.zeros(P, 4))]
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    local = [p.grad.clone() for p in params]
    vs = torch.rand(P, generator=g)
    radii = torch.randint(0, 50, (P,), generator=g, dtype=torch.int32)
    dmax = torch.tensor(1.0 + rank)
    res = parallel.exchange_step(params, vs.clone(), radii.clone(), dmax.clone())
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(local=local, vs=vs, radii=radii))
    for i, p in enumerate(params):
        ref = sum(gathered[r]["local"][i] for r in range(world))
        assert torch.allclose(p.grad, ref, atol=1e-6)
    assert torch.allclose(res["viewspace_grad_norm"], sum(gathered[r]["vs"] for r in range(world)))
    assert torch.equal(res["radii"], torch.stack([gathered[r]["radii"] for r in range(world)]).max(0).values)
    assert float(res["depth_max"]) == float(world)
    # the MAX bucket started early (forward outputs) and waited on later, like bench.py does around the backward
    r2, d2 = radii.clone(), torch.tensor(0.25 * (1 + rank))
    h = parallel.exchange_max(r2, d2, async_op=True)
    h.wait()
    assert torch.equal(r2, res["radii"]) and float(d2) == 0.25 * world
    per_view = torch.stack([radii, radii // 2])
    rm, dm = parallel.exchange_forward_stats(per_view, torch.full((2, 1, 4, 4), 0.5 + rank)).wait()
    assert torch.equal(rm, res["radii"]) and float(dm) == 0.5 + (world - 1)
    v2 = vs.clone()
    parallel.exchange_sum(params, v2, average=True)          # gradients (already summed) averaged, norms summed
    assert torch.allclose(v2, res["viewspace_grad_norm"])
    for i, p in enumerate(params):
        ref = sum(gathered[r]["local"][i] for r in range(world))
        assert torch.allclose(p.grad, ref, atol=1e-5)        # sum over ranks of the same reduced value / world
    g2 = torch.rand(3, P, 3, generator=g)                    # raw means2D gradients of three local views
    red = parallel.exchange_sum([], viewspace_grads=g2.clone())
    all_g2 = [None] * world
    dist.all_gather_object(all_g2, g2)
    assert torch.allclose(red, sum(torch.linalg.vector_norm(x[..., :2], dim=-1).sum(0) for x in all_g2), atol=1e-5)
    # broadcast of a model-like object
    class M:
        pass
    m = M()
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "max_radii2D", "xyz_gradient_accum", "denom"):
        setattr(m, n, torch.full((5, 2), float(rank)))
    parallel.broadcast_gaussians(m, src=1)
    assert float(m._xyz.mean()) == 1.0 and float(m.denom.mean()) == 1.0
    assert parallel.shard_views(4, rank, world) == ([0, 2] if rank == 0 else [1, 3])
    assert parallel.shard_views(4, 5, 8) == [1]
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_exchange_step_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


def test_single_process_is_a_no_op():
    from gaussianip_amd import parallel
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    parallel.exchange_step([p], torch.ones(3), torch.ones(3, dtype=torch.int32), torch.tensor(2.0))
    assert torch.equal(p.grad, torch.ones(3))
