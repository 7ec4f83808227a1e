"""Host logic of the two parity-class convolutions (no GPU): the weight blocks that `fused._upsample_conv_weight` and
`fused._s2_dgrad_weight` hand to gip_upsample2x_conv3x3_nhwc_f16 / gip_conv3x3s2_dgrad_nhwc_f16, applied the way the
kernel applies them — a pad-1 3x3 correlation over the SOURCE grid with the class's tap subset, scattered to the output
pixels of that parity — must equal F.interpolate + conv2d, and autograd's data gradient of the pad(0,1,0,1) stride-2
convolution.  fp64, so the only tolerance is summation order."""
import torch
import torch.nn.functional as F

from gaussianip_amd.guidance import fused

# tap subsets of the kernel (csrc/conv3x3.hip, tapsel tables): bit 3 * dy + dx, tap (dy, dx) = input offset (dy - 1, dx - 1)
UPSAMPLE_MASKS = {(0, 0): 0x01b, (0, 1): 0x036, (1, 0): 0x0d8, (1, 1): 0x1b0}
S2_DGRAD_MASKS = {(0, 0): 0x01b, (0, 1): 0x012, (1, 0): 0x018, (1, 1): 0x010}


def _apply_classes(x, wt4, masks, bias=None):
    """x [N, Cin, H, W]; wt4 [4][Cout][3][3][Cin] -> out [N, Cout, 2H, 2W] as the kernel computes it."""
    N, _, H, W = x.shape
    out = torch.zeros((N, wt4.shape[1], 2 * H, 2 * W), dtype=x.dtype)
    for (pi, pj), mask in masks.items():
        w = wt4[2 * pi + pj].permute(0, 3, 1, 2).clone()             # [Cout][Cin][3][3]
        for t in range(9):
            if not (mask >> t) & 1:
                assert float(w[:, :, t // 3, t % 3].abs().max()) == 0.0 or True
                w[:, :, t // 3, t % 3] = 0                           # the kernel never reads the taps outside the subset
        out[:, :, pi::2, pj::2] = F.conv2d(x, w, bias, padding=1)
    return out


def test_upsample_then_convolution_weight_blocks():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(4, generator=g, dtype=torch.float64)
    wt4 = fused._upsample_conv_weight(w)
    assert wt4.shape == (4, 4, 3, 3, 5) and wt4.dtype == w.dtype
    got = _apply_classes(x, wt4, UPSAMPLE_MASKS, b)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    assert float((got - ref).abs().max()) < 1e-12
    # the taps outside a class's subset hold zeros (nothing is hidden behind the mask)
    for (pi, pj), mask in UPSAMPLE_MASKS.items():
        for t in range(9):
            if not (mask >> t) & 1:
                assert float(wt4[2 * pi + pj][:, t // 3, t % 3].abs().max()) == 0.0


def test_stride2_data_gradient_weight_blocks():
    g = torch.Generator().manual_seed(1)
    N, C, Co, H, W = 2, 3, 5, 8, 6
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (ref,) = torch.autograd.grad(y, x, dy)
    wt4 = fused._s2_dgrad_weight(w)                                   # [4][C][3][3][Co]: the data gradient maps Co -> C channels
    assert wt4.shape == (4, C, 3, 3, Co)
    got = _apply_classes(dy, wt4, S2_DGRAD_MASKS)
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-12
    for (pi, pj), mask in S2_DGRAD_MASKS.items():
        for t in range(9):
            if not (mask >> t) & 1:
                assert float(wt4[2 * pi + pj][:, t // 3, t % 3].abs().max()) == 0.0


def test_gpu_only_entry_points_fall_back_to_the_library_ops_on_cpu():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 64, 4, 4, generator=g)
    w = torch.randn(8, 64, 3, 3, generator=g) * 0.05
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1)
    assert torch.allclose(fused.upsample2x_conv3x3(x, w, None), ref)
    h, s = torch.randn(1, 64, 4, 4, generator=g), torch.randn(1, 64, 4, 4, generator=g)
    assert torch.equal(fused.cat_skip(h, s, s), torch.cat([h, s + s], dim=1))
    assert torch.equal(fused.cat_skip(h, s), torch.cat([h, s], dim=1))
    c = torch.randn(1, 16, 16, 16, generator=g)
    wc, bc = torch.randn(32, 16, 3, 3, generator=g) * 0.1, torch.randn(32, generator=g)
    assert torch.allclose(fused.conv3x3_fewch(c, wc, bc, 2, act=True), F.silu(F.conv2d(c, wc, bc, stride=2, padding=1)))


def test_winograd_weight_block_and_the_two_transforms():
    """fused._winograd_weight (U = G g G^T, the host's share of Winograd F(2x2, 3x3)) with the input / output transforms the
    kernels of csrc/winograd.hip apply, written with the matrices: V = B^T d B per 4 x 4 patch at (2 ty - 1, 2 tx - 1), sixteen
    products M[p] = V[p] U[p]^T, Y = A^T M A — equals conv2d(padding=1) in fp64."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 6, 8, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    U = fused._winograd_weight(w.float()).double()                      # the builder works in fp32: compare at fp32 accuracy
    assert U.shape == (16, 4, 5)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros(2, 4, 6, 8, dtype=torch.float64)
    for n in range(2):
        for ty in range(3):
            for tx in range(4):
                d = xp[n, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                    # [ci, 4, 4]
                V = torch.einsum("ik,ckl,jl->ijc", Bt, d, Bt).reshape(16, 5)            # position p = 4 i + j
                M = torch.einsum("pc,poc->po", V, U).reshape(4, 4, 4)                   # [i, j, co]
                out[n, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = torch.einsum("ai,ijo,bj->oab", At, M, At)
    ref = F.conv2d(x, w, padding=1)
    assert float((out - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    # the default shape table parses and only names shapes with even grids
    shapes = fused._winograd_shapes()
    assert shapes and all(h % 2 == 0 and c % 8 == 0 and rows > 0 for (h, c), rows in shapes.items())
