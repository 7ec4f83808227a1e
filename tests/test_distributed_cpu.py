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


# ---- round 3: scaler / exchange order, tie-breaking of the group maximum, the hand mask under sharding ----
class _FakeGaussian:
    def __init__(self, p, lr=0.1):
        self.optimizer = torch.optim.Adam([{"params": [p], "name": "xyz"}], lr=lr)
        self.get_xyz = p.detach()


def _scaler_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussianip_amd import parallel
    from gaussianip_amd.system import StageOneConfig, StageOneStep
    p = torch.nn.Parameter(torch.ones(4, 3))
    stage = StageOneStep(_FakeGaussian(p), None, torch.zeros(3), StageOneConfig(refine_start_step=0))   # hook returns at once
    scaler = torch.amp.GradScaler("cpu", init_scale=65536.0)

    def exchange(st):
        parallel.allreduce_gradients([p])

    # step 0: rank 1's LOCAL gradient overflows (3e34 * 65536 = inf in fp32); rank 0's is finite
    coef = 3e34 if rank == 1 else 1.0
    stage.optimizer_step((p * coef).sum(), 0, scaler=scaler, exchange=exchange)
    skipped = bool(torch.equal(p.detach(), torch.ones(4, 3)))
    scale0 = float(scaler.get_scale())
    # step 1: finite everywhere -> both ranks take the same Adam step from the same summed gradient
    stage.optimizer_step((p * (1.0 + rank)).sum(), 1, scaler=scaler, exchange=exchange)
    res = [None] * world
    dist.all_gather_object(res, dict(skipped=skipped, scale0=scale0, p=p.detach().tolist(), scale1=float(scaler.get_scale())))
    # group maximum with a TIE: both ranks hold 2.0; torch.max over the concatenated data gives the gradient to ONE element
    x = torch.tensor(2.0, requires_grad=True)
    y = parallel._GroupMax.apply(x * 1.0, None)
    (y * (3.0 + rank)).backward()                 # total gradient 3 + 4 = 7 goes to the lowest rank only
    tie = [None] * world
    dist.all_gather_object(tie, (float(y.detach()), float(x.grad)))
    # and without a tie the holder gets it, whichever rank it is
    x2 = torch.tensor(1.0 + rank, requires_grad=True)
    y2 = parallel._GroupMax.apply(x2 * 1.0, None)
    y2.backward()
    notie = [None] * world
    dist.all_gather_object(notie, (float(y2.detach()), float(x2.grad)))
    # a NaN local maximum (rank 1's depth map is poisoned) makes the group maximum NaN on EVERY rank, like depths.max() of the
    # single-process step — it must not be hidden behind the other rank's finite maximum
    x3 = torch.tensor(float("nan") if rank == 1 else 5.0)
    y3 = parallel._GroupMax.apply(x3, None)
    nan = [None] * world
    dist.all_gather_object(nan, bool(torch.isnan(y3)))
    if rank == 0:
        out.put(dict(res=res, tie=tie, notie=notie, nan=nan))
    dist.destroy_process_group()


def test_scaler_sees_the_exchanged_gradients_and_group_max_breaks_ties():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scaler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r = q.get(timeout=5)
    a, b = r["res"]
    # an overflow on ONE rank's local gradients: BOTH ranks skip the step and halve their scale (system.py:168)
    assert a["skipped"] and b["skipped"]
    assert a["scale0"] == b["scale0"] == 32768.0
    assert a["p"] == b["p"] and a["p"] != torch.ones(4, 3).tolist()
    assert a["scale1"] == b["scale1"]
    assert r["tie"] == [(2.0, 7.0), (2.0, 0.0)]
    assert r["notie"] == [(2.0, 0.0), (2.0, 2.0)]
    assert r["nan"] == [True, True]


def test_view_sharding_rejects_uneven_replication_and_keeps_the_hand_mask():
    import pytest
    from gaussianip_amd import parallel
    from gaussianip_amd.system import StageOneConfig, StageOneStep
    with pytest.raises(ValueError):
        parallel.ViewSharding(4, rank=5, world=6, make_groups=False)
    assert parallel.ViewSharding(4, rank=5, world=8, make_groups=False).views == [1]
    assert parallel.ViewSharding(4, rank=1, world=3, make_groups=False).views == [1]
    # the exchange re-derives the visibility filter from the group-wide radii WITH the hand exclusion (GaussianIP.py:212-216)
    p = torch.nn.Parameter(torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [2.0, 0.0, 0.0]]))
    stage = StageOneStep(_FakeGaussian(p), None, torch.zeros(3), StageOneConfig(disable_hand_densification=True, hand_radius=0.05),
                         hand_centers=torch.tensor([[1.0, 0.0, 0.01]]))
    stage.radii = torch.tensor([3, 5, 0], dtype=torch.int32)
    stage.viewspace_points = torch.zeros(1, 3, 3, requires_grad=True)
    stage.viewspace_points.grad = torch.ones(1, 3, 3)
    parallel.ViewSharding(4, rank=0, world=1, make_groups=False).exchange(stage)
    assert stage.visibility_filter.tolist() == [True, False, False]
    assert stage.visibility(stage.radii).tolist() == [True, False, False]


def test_view_sharding_seed_group_sizes():
    """ViewSharding(group_size=g): configs[3]'s 4 views x 2 seeds (default), 2 views x 4 seeds (g = 2), seeds only (g = 1)."""
    import pytest
    from gaussianip_amd.parallel import ViewSharding
    lay = lambda g: [(v.seed_id, v.views, v.n_seed_groups, v.share) for v in (ViewSharding(4, r, 8, make_groups=False, group_size=g) for r in range(8))]  # noqa: E731
    assert lay(None) == [(r // 4, [r % 4], 2, 0.25) for r in range(8)]
    assert lay(2) == [(r // 2, [r % 2, r % 2 + 2], 4, 0.5) for r in range(8)]
    assert lay(1) == [(r, [0, 1, 2, 3], 8, 1.0) for r in range(8)]
    assert not ViewSharding(4, 3, 8, make_groups=False, group_size=1).active
    with pytest.raises(ValueError):
        ViewSharding(4, 0, 8, make_groups=False, group_size=3)
    with pytest.raises(ValueError):
        ViewSharding(4, 0, 8, make_groups=False, group_size=8)


def test_gemm_dispatch_table_matches_the_measurements():
    """fused.linear_prefers_own against the rows of profiles/r04_gemm_own_vs_hipblaslt.txt: the own kernel is chosen wherever it was
    >= 5 % faster, the library wherever IT was >= 5 % faster — except rows within the launch-floor noise of the eager timing
    loop (both under 21 us), which the table decides by shape class."""
    import json
    import os
    from gaussianip_amd.guidance import fused
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_gemm_own_vs_hipblaslt.txt")
    rows = [json.loads(ln) for ln in open(path) if ln.startswith("{")]
    assert len(rows) > 60
    wrong = []
    for r in rows:
        own = fused.linear_prefers_own(r["M"], r["K"], r["N"])
        ratio = r["own_over_lib"]
        floor = max(r["own_us"], r["lib_us"]) < 27.0          # both at the host-launch floor of the timing loop
        if floor:
            continue
        if (ratio < 0.95 and not own) or (ratio > 1.08 and own):
            wrong.append((r["name"], r["M"], r["K"], r["N"], ratio, own))
    assert len(wrong) <= 2, wrong          # (M 6144: qkv 1.12 and ff_out 1.08 sit on the class boundary)
