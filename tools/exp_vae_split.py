"""Does the VAE encoder's forward + backward get faster when the batch of 4 images is cut into sub-batches that run on
different HIP streams (one chain's bandwidth-bound GroupNorm / elementwise passes under the other's MFMA-bound
convolutions)?  Everything is replayed from HIP graphs, so the host's launch rate plays no part.
usage: exp_vae_split.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianip_amd.guidance.networks import VAEEncoder, init_for_benchmark  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
vae = init_for_benchmark(VAEEncoder()).to(dev).half().to(memory_format=torch.channels_last).requires_grad_(False)
B = 4
x_all = (torch.rand(B, 3, 512, 512, device=dev) * 2 - 1).half().contiguous(memory_format=torch.channels_last)
g_all = torch.randn(B, 8, 64, 64, device=dev).half()


def make(parts, nstreams):
    """graph of: for each of `parts` equal sub-batches, moments(x_i).backward(g_i), sub-batch i on stream i % nstreams"""
    n = B // parts
    xs = [x_all[i * n:(i + 1) * n].clone(memory_format=torch.channels_last).requires_grad_(True) for i in range(parts)]
    gs = [g_all[i * n:(i + 1) * n].clone() for i in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]

    def body():
        cur = torch.cuda.current_stream(dev)
        if nstreams == 1:
            for x, g in zip(xs, gs):
                vae.moments(x).backward(g)
            return
        for s in streams:
            s.wait_stream(cur)
        ys = []
        for i, x in enumerate(xs):
            with torch.cuda.stream(streams[i % nstreams]):
                ys.append(vae.moments(x))
        for i, (y, g) in enumerate(zip(ys, gs)):
            with torch.cuda.stream(streams[i % nstreams]):
                y.backward(g)
        for s in streams:
            cur.wait_stream(s)

    # warm-up on a side stream (lazy initialisations must not be captured)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            for x in xs:
                x.grad = None
            body()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    for x in xs:
        x.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    return graph, xs, body


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


variants = {"1 x 4 images, 1 stream": (1, 1), "2 x 2 images, 1 stream": (2, 1), "2 x 2 images, 2 streams": (2, 2),
            "4 x 1 image, 2 streams": (4, 2), "4 x 1 image, 4 streams": (4, 4)}
built = {}
for name, (parts, ns) in variants.items():
    t0 = time.time()
    try:
        built[name] = make(parts, ns)
    except Exception as e:          # noqa: BLE001
        print("%-28s capture failed: %r" % (name, e), flush=True)
        continue
    print("%-28s captured in %.1f s" % (name, time.time() - t0), flush=True)
ref_grad = None
res = {k: [] for k in built}
for rnd in range(5):
    for name, (graph, xs, body) in built.items():
        res[name].append(timed(graph.replay))
for name, (graph, xs, body) in built.items():
    graph.replay()
    torch.cuda.synchronize()
    grad = torch.cat([x.grad for x in xs]).float()
    if ref_grad is None:
        ref_grad = grad
    err = float((grad - ref_grad).abs().max() / ref_grad.abs().max())
    eager = timed(lambda: ([setattr(x, "grad", None) for x in xs], body()), 5)
    print("%-28s graph replay %.3f ms (median of 5 x 10)   eager %.3f ms   grad diff vs first variant %.2e" %
          (name, sorted(res[name])[2], eager, err), flush=True)
