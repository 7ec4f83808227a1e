"""LayerNorm folded into the projection that consumes it (include/gip_nn.h: gip_linear_rows_f16, gip_linear_ln_f16):
    LN(x) W^T + b = rstd (x (W gamma)^T) - rstd mu s + t
with the row statistics taken from the partial sums the PRODUCER of x left in its epilogue.  Against plain PyTorch in float32
(F.layer_norm -> F.linear -> GEGLU), at the channel widths / tile shapes of the denoiser's three levels; and the transformer
block with and without the fold."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _producer(M, C, seed):
    """x = linear(h) + residual through the own kernel with row sums, as attn.to_out produces the block's residual stream."""
    from gaussianip_amd.guidance import fused
    g = torch.Generator(device="cuda").manual_seed(seed)
    h = torch.randn(M, C, device="cuda", generator=g).half()
    w = (torch.randn(C, C, device="cuda", generator=g) / C ** 0.5).half()
    b = torch.randn(C, device="cuda", generator=g).half()
    r = (torch.randn(M, C, device="cuda", generator=g) * 2 + 0.7).half()          # a mean well away from zero: the cancellation in rstd (acc - mu s)
    rows = []
    with torch.no_grad():
        x = fused.linear(h, w, b, r, rows=rows)
        plain = fused.linear(h, w, b, r)
    assert rows and torch.equal(x, plain), "the row-sum epilogue changed the output"
    return x, rows[0]


@pytest.mark.parametrize("M,C", [(8192, 320), (4096 + 128, 640), (3072, 1280), (768, 1280), (200, 320)])
def test_row_sums_of_the_producer(M, C):
    from gaussianip_amd import _lib
    x, rows = _producer(M, C, 1)
    parts = _lib.nn_lib().gip_linear_row_parts(M, C)
    assert rows.shape == (M, parts, 2)
    xf = x.double()
    assert float((rows[..., 0].double().sum(1) - xf.sum(1)).abs().max()) <= 1e-4 * float(xf.abs().sum(1).max())
    assert float((rows[..., 1].double().sum(1) - (xf * xf).sum(1)).abs().max()) <= 1e-4 * float((xf * xf).sum(1).max())
    bn = C // parts if C % parts == 0 else None
    if bn:                                               # each part = one channel tile
        tiles = xf.reshape(M, parts, bn)
        assert float((rows[..., 0].double() - tiles.sum(2)).abs().max()) <= 1e-4 * float(tiles.abs().sum(2).max())


@pytest.mark.parametrize("M,C,N,geglu", [(8192, 320, 960, False), (8192, 320, 320, False), (4224, 640, 640, False), (3072, 1280, 1280, False),
                                          (768, 1280, 1280, False), (49152, 320, 1280, True), (200, 320, 320, False)])
def test_linear_with_folded_layernorm_against_fp32(M, C, N, geglu, monkeypatch):
    from gaussianip_amd.guidance import fused
    monkeypatch.setenv("GIP_OWN_GEMM", "2")              # test the kernel at every shape, whatever the dispatch table prefers
    x, rows = _producer(M, C, 2)
    setattr(x, fused._ROWS_ATTR, rows)
    g = torch.Generator(device="cuda").manual_seed(3)
    norm = fused.LayerNorm(C).cuda().half().requires_grad_(False)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, device="cuda", generator=g) * 0.3 + 1.0)
        norm.bias.copy_(torch.randn(C, device="cuda", generator=g) * 0.2)
    rows_w = 2 * N if geglu else N
    w = (torch.randn(rows_w, C, device="cuda", generator=g) / C ** 0.5).half()
    b = torch.randn(rows_w, device="cuda", generator=g).half()
    with torch.no_grad():
        out = fused.linear_ln(x, norm, w, b, geglu)
    assert out is not None, "the fold did not apply"
    y = F.linear(F.layer_norm(x.float(), (C,), norm.weight.float(), norm.bias.float(), norm.eps), w.float(), b.float())
    if geglu:
        v, gate = y.chunk(2, dim=-1)
        y = v * F.gelu(gate)
    err = float((out.float() - y).abs().max()) / float(y.abs().max())
    rel = float((out.float() - y).norm() / y.norm())
    # the unfused product path for comparison: LayerNorm kernel (half output) + projection
    with torch.no_grad():
        two = fused.linear(norm(x), w, b, None, geglu)
    rel_two = float((two.float() - y).norm() / y.norm())
    print("M %d C %d N %d geglu %d: folded rel L2 %.2e max %.2e | LayerNorm kernel + GEMM rel L2 %.2e" % (M, C, N, geglu, rel, err, rel_two))
    assert err < 4e-3 and rel < 1.5e-3
    assert rel < 2.5 * rel_two + 1e-4                    # no worse than the two-kernel path beyond rounding noise


def test_transformer_block_with_and_without_the_fold(monkeypatch):
    """BasicTransformerBlock at the 64^2 level's shape: the folded path (3 LayerNorm kernels gone) against the LayerNorm-kernel
    path and against float32; the LayerNorm kernel must not run where the fold applies."""
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused, networks
    torch.manual_seed(0)
    C, B, T = 320, 8, 4096               # 32 768 rows: the GEGLU projection folds its LayerNorm from here on (networks._GEGLU_FUSE_MIN_ROWS)
    blk = networks.TransformerBlock(C, 768, 8, 0, True, 0.5)
    networks.init_for_benchmark(blk, 5)
    blk = blk.cuda().half().eval().requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(1)
    h = torch.randn(B, T, C, device="cuda", generator=g).half()
    wp = (torch.randn(C, C, device="cuda", generator=g) / C ** 0.5).half()
    bp = torch.zeros(C, device="cuda").half()
    ctx = (torch.randn(B, 81, 768, device="cuda", generator=g) * 0.1).half()
    lib = _lib.nn_lib()

    def run():
        with torch.no_grad():
            t = fused.linear_auto(h, wp, bp, want_rows=True)          # proj_in's role
            # the prompt tokens' key / value projections staged as _Encoder.stage_context does for every cross-attention layer
            a = blk.attn2
            text, ip = ctx[:, :-networks.IP_TOKENS], ctx[:, -networks.IP_TOKENS:]
            a.staged_kv = (a.to_k(text), a.to_v(text), a.to_k_ip(ip), a.to_v_ip(ip))
            before = dict(_lib.call_counts)
            y = blk(t, ctx)
            ln_calls = _lib.call_counts.get("gip_layernorm_f16", 0) - before.get("gip_layernorm_f16", 0)
            fold_calls = _lib.call_counts.get("gip_linear_ln_f16", 0) - before.get("gip_linear_ln_f16", 0)
        return y, ln_calls, fold_calls
    y_fold, ln1, f1 = run()
    monkeypatch.setenv("GIP_LN_FOLD", "0")
    y_plain, ln0, f0 = run()
    monkeypatch.delenv("GIP_LN_FOLD")
    assert (ln0, f0) == (3, 0) and f1 == 3 and ln1 == 0, (ln0, f0, ln1, f1)
    blk32 = __import__("copy").deepcopy(blk).float()
    with fused.disabled(), torch.no_grad():
        t32 = F.linear(h.float(), wp.float(), bp.float()).half().float()
        ref = blk32(t32, ctx.float())
    r_fold = float((y_fold.float() - ref).norm() / ref.norm())
    r_plain = float((y_plain.float() - ref).norm() / ref.norm())
    print("transformer block vs fp32: folded %.2e, LayerNorm kernels %.2e" % (r_fold, r_plain))
    assert r_fold < 3e-3 and r_fold < 2.0 * r_plain + 1e-4


@pytest.mark.parametrize("M,K,N", [(768, 1280, 1280), (192, 640, 640), (3072, 320, 960), (1000, 1280, 320), (3072, 640, 1920)])
def test_narrow_channel_tiles_of_small_grids_are_bit_identical(M, K, N, monkeypatch):
    """GEMMs whose 128-wide (160) tiles leave most CUs with at most one workgroup run on 128 x 64 tiles (csrc/conv3x3.hip:
    narrow_tiles): same K order per element, so every entry point — plain, GroupNorm statistics, LayerNorm row sums, folded
    LayerNorm — gives the bits of the wide tile; the row-sum parts follow the tile width (gip_linear_row_parts(M, Nout))."""
    import ctypes
    from gaussianip_amd import _lib
    from gaussianip_amd.guidance import fused
    lib = _lib.nn_lib()
    monkeypatch.setenv("GIP_OWN_GEMM", "2")                      # every supported shape on the own kernel (the dispatch is not under test)
    knob = ctypes.c_int.in_dll(lib._lib, "gip_dbg_linear_narrow")
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
    b = torch.randn(N, device="cuda", generator=g).half()
    r = torch.randn(M, N, device="cuda", generator=g).half()
    norm = fused.LayerNorm(K).cuda().half()
    with torch.no_grad():
        norm.weight.add_(torch.randn(K, device="cuda", generator=g).half() * 0.2)
        norm.bias.add_(torch.randn(K, device="cuda", generator=g).half() * 0.2)
    w_in = (torch.randn(K, K, device="cuda", generator=g) / K ** 0.5).half()

    def run(lim):
        knob.value = lim
        try:
            with torch.no_grad():
                plain = fused.linear(x, w, b, r)
                st, rows = [], []
                with_stats = fused.linear(x, w, b, r, stats=st) if M % 128 == 0 else plain
                h = fused.linear(x, w_in, None, None, rows=rows)              # producer of a LayerNorm input, with its row sums
                setattr(h, fused._ROWS_ATTR, rows[0])
                folded = fused.linear_ln(h, norm, w, b)
            parts = lib.gip_linear_row_parts(M, K)
            return plain, with_stats, (st[0] if st else None), h, rows[0], folded, parts
        finally:
            knob.value = -1

    wide = run(0)
    narrow = run(1 << 20)
    assert narrow[6] == (K + 63) // 64 and wide[6] in ((K + 127) // 128, (K + 159) // 160)
    assert narrow[4].shape[1] == narrow[6] and wide[4].shape[1] == wide[6]
    for i in (0, 1, 3):
        assert torch.equal(narrow[i], wide[i]), i
    if wide[2] is not None:
        # per-128-row channel sums of the same values: a thread's rows are grouped by the tile width, so the float32 order differs
        assert float((narrow[2].double() - wide[2].double()).abs().max()) <= 1e-5 * float(wide[2].double().abs().max())
    # row sums: more, narrower parts of the same rows
    assert float((narrow[4].double().sum(1) - wide[4].double().sum(1)).abs().max()) <= 1e-5 * float(wide[4].double().sum(1).abs().max())
    assert narrow[5] is not None and wide[5] is not None
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(wide[3].float(), (K,), norm.weight.float(), norm.bias.float(), norm.eps), w.float(), b.float())
    for got in (narrow[5], wide[5]):
        assert float((got.float() - ref).abs().max()) <= 8e-3 * max(1.0, float(ref.abs().max()))
    ref0 = torch.nn.functional.linear(x.float(), w.float(), b.float()) + r.float()
    assert float((narrow[0].float() - ref0).abs().max()) <= 4e-3 * max(1.0, float(ref0.abs().max()))


@pytest.mark.parametrize("M,K,N", [(768, 1280, 1280), (192, 5120, 1280), (3072, 1280, 320), (768, 2560, 640), (200, 1344, 320)])
def test_two_k_groups_per_workgroup_equal_the_one_group_gemm(M, K, N):
    """Round 6: on grids of at most one workgroup per CU the own GEMM runs TWO K groups of four waves per workgroup (conv3x3_kernel<...,
    KG = 2>): each takes half of the K steps in its own stage buffers, group 1 hands its accumulators over through LDS.  Against the
    one-group kernel (gip_dbg_linear_kg = 0): the sum is (first half) + (second half) instead of one chain — equal to the rounding of
    two fp32 partial sums (<= one half ulp of the output after the final rounding) — same error against float32, bitwise reproducible;
    an odd number of K steps (1344 = 21 x 64) gives the second group the extra step."""
    import ctypes
    from gaussianip_amd import _lib
    lib = _lib.nn_lib()
    knob = ctypes.c_int.in_dll(lib._lib, "gip_dbg_linear_kg")
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
    b = torch.randn(N, device="cuda", generator=g).half()
    r = torch.randn(M, N, device="cuda", generator=g).half()
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    try:
        for mode in (0, 1, 1):
            knob.value = mode
            o = torch.empty(M, N, device="cuda", dtype=torch.float16)
            assert lib.gip_linear_f16(p_(x), p_(w), p_(b), p_(r), p_(o), M, K, N, 0, st) == 0
            outs.append(o)
    finally:
        knob.value = -1
    ref = (torch.addmm(b.float(), x.float(), w.float().t()).half().float() + r.float())
    scale = float(ref.abs().max())
    assert torch.equal(outs[1], outs[2])
    assert float((outs[0].float() - outs[1].float()).abs().max()) <= 2.0 ** -9 * scale
    e0, e1 = float((outs[0].float() - ref).abs().max()), float((outs[1].float() - ref).abs().max())
    assert e1 <= 2.0 ** -8 * scale and e1 <= 1.5 * e0 + 2.0 ** -10 * scale, (e0, e1)
