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
    parts = _lib.nn_lib().gip_linear_row_parts(C)
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
