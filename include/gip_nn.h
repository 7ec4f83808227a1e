/*
 * gip_nn.h — C-ABI of the fused normalisation kernels used by the SD1.5 + ControlNet denoise step and the VAE
 * encoder (the "U-Net denoise step that drives AHDS/SDS guidance" half of the hot path).
 *
 * The reference runs diffusers' ResnetBlock2D / Transformer2DModel, whose GroupNorm -> SiLU pairs execute as separate
 * PyTorch kernels (threestudio/models/guidance/ipa_guidance.py:311-358 -> diffusers 0.27).  On MI355X the tensors are
 * kept channels-last (NHWC, the layout MIOpen's MFMA implicit-GEMM convolutions use natively) and each
 * GroupNorm(+SiLU) is ONE statistics pass + ONE apply pass in fp16 with fp32 accumulation:
 *
 *   gip_gn_silu_forward   y = silu?( (x - mean_g) * rstd_g * gamma_c + beta_c )
 *   gip_gn_silu_backward  dL/dx of the same (weights are frozen in this path: no dgamma / dbeta)
 *
 * Both take an optional per-(sample, channel) `addend` (half, row stride `addend_stride` elements; stride 0 = one row
 * shared by all samples; NULL = none) that is added to x on load: ResnetBlock2D's `conv1(x) + bias + time_emb_proj(...)`
 * feeds norm2, so the bias / time-embedding adds never run as separate passes over the activation.
 *
 *   gip_add_bias_residual  out = a + b + bias[c]            (ResnetBlock2D output: shortcut + conv2 + biases; out may alias a)
 *   gip_geglu              out[m, :D] = in[m, :D] * gelu(in[m, D:2D])   (diffusers GEGLU, exact erf GELU)
 *
 * Plain C, raw device pointers, caller-owned buffers (workspace sized by gip_gn_workspace_bytes), work enqueued on
 * `stream` (hipStream_t as void*), integer status (0 ok, 1 bad argument, 2 workspace too small, 3 HIP error).
 * Layout: x, y, dy, dx are [N, HW, C] half-precision with C fastest (torch channels_last memory of an NCHW tensor);
 * gamma, beta [C] half; mean, rstd [N, G] float (outputs of forward, inputs of backward).  C % 8 == 0, C % G == 0.
 */
#ifndef GIP_NN_H
#define GIP_NN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t gip_gn_workspace_bytes(int32_t N, int32_t G);
int gip_gn_silu_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                        int32_t N, int64_t HW, int32_t C, int32_t G, float eps, int32_t apply_silu,
                        const void* addend, int32_t addend_stride,
                        void* workspace, size_t workspace_bytes, void* stream);
int gip_gn_silu_backward(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                         const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t apply_silu,
                         const void* addend, int32_t addend_stride,
                         void* workspace, size_t workspace_bytes, void* stream);
/* gip_gn_silu_backward with `accum` [N, HW, C] half added to the result in the same pass: dx = dL/dx of the GroupNorm +
 * accum (fp32 sum, one rounding).  In ResnetBlock2D's backward x receives two gradients — through norm1 and through the
 * shortcut — which autograd would add in a separate pass over the tensor.  dx may alias accum. */
int gip_gn_silu_backward_accum(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                               const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t apply_silu,
                               const void* addend, int32_t addend_stride, const void* accum,
                               void* workspace, size_t workspace_bytes, void* stream);
/* Backward with the reduction pass taken out: `chan_sums` [N * blocks_per_sample][C][2] float holds, per 128-row block and
 * channel, sum(dxh) and sum(dxh * xh) — written by the epilogue of the data-gradient convolution that PRODUCED dy
 * (gip_conv3x3_gnbwd_nhwc_f16).  One small launch folds them into the two per-group means, then the apply pass runs
 * (+ `accum` when not NULL, as gip_gn_silu_backward_accum).  workspace: gip_gn_workspace_bytes(N, G). */
int gip_gn_silu_backward_sums(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                              const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t apply_silu,
                              const void* addend, int32_t addend_stride, const void* accum, const float* chan_sums,
                              int32_t blocks_per_sample, void* workspace, size_t workspace_bytes, void* stream);
/* Forward with the statistics pass taken out: `chan_stats` [N * blocks_per_sample][C][2] float holds, per 128-row block of
 * x and channel, the sum and the sum of squares of x's elements — written by the epilogue of the kernel that PRODUCED x
 * (gip_conv3x3_stats_nhwc_f16 / gip_linear_stats_f16; HW % 128 == 0 so that a block never straddles two samples).  One
 * small launch folds them into mean / rstd (the addend enters algebraically), then the same apply pass runs: x is read
 * once instead of twice.  No workspace. */
int gip_gn_silu_forward_stats(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                              int32_t N, int64_t HW, int32_t C, int32_t G, float eps, int32_t apply_silu,
                              const void* addend, int32_t addend_stride, const float* chan_stats,
                              int32_t blocks_per_sample, void* stream);
int gip_add_bias_residual(const void* a, const void* b, const void* bias, void* out, int64_t M, int32_t C, void* stream);
int gip_geglu(const void* in, void* out, int64_t M, int32_t D, void* stream);
/* Skip-connection concatenation of the U-Net decoder with the ControlNet residual and the consuming GroupNorm's statistics
 * in one pass: out[m] = [ a[m] | half(b[m] + b_add[m]) ], a [M, Ca], b / b_add [M, Cb], out [M, Ca + Cb] half rows (NHWC
 * tensors seen as rows); b_add may be NULL.  chan_stats (may be NULL) [M / 128][Ca + Cb][2] float receives, per 128-row
 * block and channel, the sum and the sum of squares of the values written — the partials gip_gn_silu_forward_stats takes
 * (M % 128 == 0 then).  Ca % 64 == 0, Cb % 64 == 0.  Replaces, in the reference's U-Net (diffusers
 * UNet2DConditionModel.forward as driven by ipa_guidance.py:338-356): `down_block_res_sample + controlnet residual`,
 * `torch.cat([hidden_states, res_hidden_states], dim=1)` of every up-block layer, and the statistics read of the
 * ResnetBlock2D.norm1 that follows. */
int gip_cat2_stats_f16(const void* a, const void* b, const void* b_add, void* out, float* chan_stats, int64_t M, int32_t Ca,
                       int32_t Cb, void* stream);

/* LayerNorm over the last dimension: y[m, :] = (x[m, :] - mean_m) * rstd_m * weight + bias, x / y [M, C] half, weight /
 * bias [C] half, fp32 statistics (biased variance, like torch.nn.LayerNorm).  C % 8 == 0, C <= 2048.  Replaces the
 * BasicTransformerBlock norm1 / norm2 / norm3 calls the reference's U-Net makes through diffusers (forward only: the
 * denoiser is frozen in ipa_guidance.py:143-172). */
int gip_layernorm_f16(const void* x, const void* weight, const void* bias, void* y, int64_t M, int32_t C, float eps,
                      void* stream);

/* Row softmax in place, and its backward in place over the upstream gradient (csrc/softmax.hip): the softmax between the two
 * GEMMs of the VAE encoder's single-head mid attention (diffusers AttnBlock inside AutoencoderKL.encode, reference call
 * ipa_guidance.py:522-531; scale = 512^-1/2).  s / p / dp: [rows, n] half, n % 8 == 0, n <= 8192.
 *   forward   s[r, :] <- softmax(scale * s[r, :])
 *   backward  dp[r, :] <- scale * p[r, :] * (dp[r, :] - sum_j dp[r, j] p[r, j])     (= dL/d(raw scores)) */
int gip_softmax_rows_f16(void* s, int64_t rows, int32_t n, float scale, void* stream);
int gip_softmax_rows_backward_f16(const void* p, void* dp, int64_t rows, int32_t n, float scale, void* stream);

/* One LPIPS layer term against cached target features (csrc/lpips.hip); replaces, per VGG tap, the
 * normalize_tensor -> (a - b)^2 -> lin -> spatial_average chain of `lpips.LPIPS.forward` that the reference calls at
 * threestudio/systems/GaussianIP.py:435 (third-party `lpips` package; published LPIPS v0.1 algorithm).
 *   feat         [N, HW, C] half  raw VGG features of the rendered images (NHWC)
 *   target_unit  [N, HW, C] half  channel-normalised features of the fixed refined images
 *   lin          [C] float        the layer's 1x1 weights
 * forward : partial[n * blocks + b] = this block's share of sum_hw sum_c lin[c] (feat/(|feat|+1e-10) - target)^2 ; the
 *           caller sums over b (fixed order: deterministic) and divides by HW.  blocks = gip_lpips_layer_blocks(N, HW).
 * backward: grad_feat [N, HW, C] half = d(sum_hw ...)/dfeat * coef[n], saturated to the fp16 range (coef carries
 *           gout / HW and the caller's power-of-two loss scale).
 * C % 8 == 0, C <= 512. */
int32_t gip_lpips_layer_blocks(int32_t N, int64_t HW);
int gip_lpips_layer_forward(const void* feat, const void* target_unit, const float* lin, float* partial, int32_t N,
                            int64_t HW, int32_t C, int32_t blocks, void* stream);
int gip_lpips_layer_backward(const void* feat, const void* target_unit, const float* lin, const float* coef,
                             void* grad_feat, int32_t N, int64_t HW, int32_t C, void* stream);

/* 3x3 / stride 1 / pad 1 convolution as an MFMA implicit GEMM (csrc/conv3x3.hip): x [N,H,W,Cin] half (NHWC),
 * w [Cout,3,3,Cin] half (the channels_last memory of a torch [Cout,Cin,3,3] weight), out [N,H,W,Cout] half, fp32
 * accumulation.  Epilogue (fp32, before the single rounding to half): + bias[Cout] (NULL = none) + residual
 * [N,H,W,Cout] (NULL = none; ResnetBlock2D's shortcut).  Cin % 64 == 0, Cout % 4 == 0.
 * `workspace` (optional, caller-owned device memory): with it, problems whose 128 x BN output tiles cannot fill the
 * chip (the 8x8 latent level) run split-K — fp32 slabs [split][N*H*W][Cout] + a fixed-order reduce with the epilogue.
 * Replaces the MIOpen call behind diffusers' ResnetBlock2D / Upsample2D convolutions in the denoiser and the VAE. */
int gip_conv3x3_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int32_t N,
                         int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* workspace, size_t workspace_bytes,
                         void* stream);

/* gip_conv3x3_nhwc_f16 that also writes `chan_stats` [ceil(N*H*W / 128)][Cout][2] float: per 128-pixel block and output
 * channel the sum and the sum of squares of the final (half-rounded, bias and residual included) outputs — the
 * statistics pass of the GroupNorm that reads `out` next (gip_gn_silu_forward_stats).  Never split-K.  Cout % 8 == 0. */
int gip_conv3x3_stats_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int32_t N,
                               int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* chan_stats, void* stream);
/* mean / rstd [N, G] of GroupNorm(x + addend) from the per-(128-row block, channel) sums `chan_stats` the kernel that produced x
 * left in its epilogue — the first half of gip_gn_silu_forward_stats on its own (no apply pass: gip_conv3x3_gnin_nhwc_f16 below
 * normalises while it loads). */
int gip_gn_stats_from_partials(float* mean, float* rstd, int32_t N, int64_t HW, int32_t C, int32_t G, float eps,
                               const void* addend, int32_t addend_stride, const float* chan_stats, int32_t blocks_per_sample,
                               void* stream);
/* out = conv3x3(silu?(GroupNorm(x + addend)), w) + bias (+ residual) WITHOUT materialising the normalised tensor: the
 * halo-resident kernel (Cin = 128, H % 8 == 0, W % 16 == 0, >= 256 output tiles) normalises its 18 x 10 pixel halo in LDS, once per
 * tile, with the arithmetic of the separate apply pass.  ResnetBlock2D's `conv1(silu(norm1(x)))` / `conv2(silu(norm2(h)))` of the
 * VAE encoder's 128-channel level (diffusers resnet.py, driven from ipa_guidance.py:522-531).  chan_stats as in
 * gip_conv3x3_stats_nhwc_f16 (may be NULL).  Returns 1 for shapes the halo kernel does not take: run gip_gn_silu_forward + the
 * plain convolution then. */
int gip_conv3x3_gnin_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int32_t N,
                              int32_t H, int32_t W, int32_t Cin, int32_t Cout, const void* gamma, const void* beta,
                              const float* mean, const float* rstd, int32_t G, int32_t apply_silu, const void* addend,
                              int32_t addend_stride, float* chan_stats, void* stream);
/* Data gradient of a 3x3 convolution whose INPUT was y = silu?(GroupNorm(gn_x + addend)): out = dL/dy = conv3x3(dy_in, w) with
 * w the flipped-transposed weight (no bias, no residual), and `chan_sums` [N*H*W / 128][Cout][2] = per 128-pixel block and
 * channel sum(dxh), sum(dxh xh) of that GroupNorm's backward (xh = (gn_x + addend - mean) rstd, dxh = dL/dy dsilu?(gamma xh
 * + beta) gamma) — the reductions gip_gn_silu_backward would take in a separate pass over gn_x and dL/dy.  (H*W) % 128 == 0
 * (% 256 where the 256-row tile applies); never split-K. */
int gip_conv3x3_gnbwd_nhwc_f16(const void* dy_in, const void* w, void* out, int32_t N, int32_t H, int32_t W, int32_t Cin,
                               int32_t Cout, const void* gn_x, const void* gamma, const void* beta, const float* mean,
                               const float* rstd, int32_t G, int32_t apply_silu, const void* addend, int32_t addend_stride,
                               float* chan_sums, void* stream);
/* The same with a split-K workspace (may be NULL): layers with too few output tiles for the chip (the 16 x 16 and 8 x 8 levels)
 * run split-K as in gip_conv3x3_nhwc_f16, and the kernel that sums the fp32 slabs also writes the statistics — per
 * `stats_rows`-row block: 128, or 64 where a sample has only 64 pixels (H * W % stats_rows == 0; 64 requires the split-K
 * route, i.e. a workspace and < 256 output tiles: returns 1 otherwise).  chan_stats [N * H * W / stats_rows][Cout][2]. */
int gip_conv3x3_stats_ws_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int32_t N,
                                  int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* chan_stats, int32_t stats_rows,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* gip_linear_f16 (no GEGLU) with the same per-(128-row block, column) statistics of its output. */
int gip_linear_stats_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                         int32_t K, int32_t Nout, float* chan_stats, void* stream);

/* The VAE encoder's conv_in and its data gradient (csrc/conv_small.hip) — 3x3 / stride 1 / pad 1 between 3 and 128 channels,
 * both bound by one pass over the 128-channel tensor:
 *   forward  x [N,H,W,3] half, w [128,3,3,3] half (channels_last memory of the torch weight), bias [128] or NULL -> out [N,H,W,128];
 *            H % 16 == 0, W % 16 == 0.  Replaces MIOpen's kernel + bias kernel + NCHW -> NHWC copy (AutoencoderKL.encoder.conv_in).
 *   dgrad    dy [N,H,W,128] half, wt [3][9][128] half with wt[c][3 ty + tx][co] = w[co][2 - ty][2 - tx][c] -> dx [N,H,W,3]
 *            (the gradient that leaves the VAE towards the bilinear resize and the rasterizer); H % 8 == 0, W % 16 == 0. */
int gip_conv3x3_c3_fwd_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t H, int32_t W,
                                int32_t Cout, void* stream);
/* The same, leaving the next GroupNorm's statistics: chan_stats [N * H * W / 128][128][2] float32 = (sum, sum of squares) per
 * 16 x 8 half tile and channel of the half-rounded outputs (AutoencoderKL.encode: conv_in -> down_blocks[0].resnets[0].norm1). */
int gip_conv3x3_c3_fwd_stats_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t H, int32_t W,
                                      int32_t Cout, float* chan_stats, void* stream);
int gip_conv3x3_c3_dgrad_nhwc_f16(const void* dy, const void* wt, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                  void* stream);

/* Few-channel 3x3 / pad 1 convolutions with bias and optional SiLU in the epilogue (csrc/conv_small.hip): the ControlNet's
 * conditioning stem, diffusers ControlNetConditioningEmbedding (conv_in 3 -> 16, then 16 -> 16, 16 -> 32 /2, 32 -> 32,
 * 32 -> 96 /2, 96 -> 96, 96 -> 256 /2, each followed by F.silu), which the reference runs inside self.controlnet(...)
 * (ipa_guidance.py:338-346).  x [N,Hin,Win,Cin] half, w [Cout,3,3,Cin] half (channels_last memory of the torch weight),
 * bias [Cout] or NULL -> out [N,Hin/stride,Win/stride,Cout]; act != 0: out = silu(half(conv + bias)) (torch's two
 * roundings).  Supported (Cin, Cout, stride): (3, 16 | 128, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2),
 * (96, 96, 1), (96, 256, 2); output width % 16 == 0, output height % 8 == 0 (% 16 for Cin = 3, % 4 for 96 -> 256); anything else returns 1.
 * Also the two narrow OUTPUT convolutions, (320, 4, 1) = conv_out of the U-Net and (512, 8, 1) = conv_out of the VAE encoder
 * (output height % 8 / % 4): there w holds 16 rows [16,3,3,Cin] of which rows >= Cout are zero; bias / out have Cout channels. */
int gip_conv3x3_fewch_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t Hin, int32_t Win,
                               int32_t Cin, int32_t Cout, int32_t stride, int32_t act, void* stream);

/* The same kernel at stride 2 (diffusers Downsample2D): out [N, Hin/2, Win/2, Cout]; pad_top / pad_left = 1 with the
 * symmetric padding of the U-Net / ControlNet (padding=1), 0 for the VAE's F.pad(x, (0, 1, 0, 1)) + padding=0 form (the
 * missing bottom / right rows are the usual out-of-range zeros).  Hin, Win even. */
int gip_conv3x3s2_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t Hin, int32_t Win,
                           int32_t Cin, int32_t Cout, int32_t pad_top, int32_t pad_left, void* workspace,
                           size_t workspace_bytes, void* stream);
/* gip_conv3x3s2_nhwc_f16 whose epilogue also takes the next GroupNorm's statistics: chan_stats [N * (Hin/2) * (Win/2) / 128][Cout][2]
 * float32 (sum, sum of squares per 128 output pixels and channel; (Hin/2) * (Win/2) % 128 == 0, Cout % 8 == 0, no split-K). */
int gip_conv3x3s2_stats_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t Hin, int32_t Win,
                                 int32_t Cin, int32_t Cout, int32_t pad_top, int32_t pad_left, float* chan_stats, void* stream);

/* DATA GRADIENT of that stride-2 convolution in its pad_top = pad_left = 0 form (the VAE's Downsample2D, whose gradient the
 * reference gets from autograd through AutoencoderKL.encode, ipa_guidance.py:309-314): dy [N,Ho,Wo,Cin] half (Cin = the forward
 * convolution's output channels), dx [N,2 Ho,2 Wo,Cout] half (every element written).  wt4 [4][Cout][3][3][Cin] half, parity
 * class c = 2 (i & 1) + (j & 1) of the dx pixel: wt4[c][ci][dy][dx][co] = w[co][ci][ky][kx] with ky = 2 for dy = 0, ky = i & 1
 * for dy = 1, unused otherwise (same in x; taps a class does not use are never read).  Four launches of the implicit GEMM
 * over dy's own grid with 4 / 2 / 2 / 1 taps: the minimal FLOPs, no zero-dilated copy.  Cin % 64 == 0, Cout % 8 == 0. */
int gip_conv3x3s2_dgrad_nhwc_f16(const void* dy, const void* wt4, void* dx, int32_t N, int32_t Ho, int32_t Wo, int32_t Cin,
                                 int32_t Cout, void* stream);

/* Nearest-neighbour 2x upsampling + 3x3 / pad 1 convolution (diffusers Upsample2D in the U-Net decoder the reference runs,
 * ipa_guidance.py:349-356) without materialising the upsampled tensor: x [N,Hin,Win,Cin] half -> out [N,2 Hin,2 Win,Cout] half
 * (+ bias [Cout] or NULL).  Each output parity class (i & 1, j & 1) is a 2 x 2-tap convolution over x's own grid, the
 * weights of the taps that fall on the same source pixel summed by the caller: wt4 [4][Cout][3][3][Cin] half, class
 * c = 2 (i & 1) + (j & 1); rows: parity 0 -> tap 0 holds w[ky = 0], tap 1 holds w[1] + w[2]; parity 1 -> tap 1 holds
 * w[0] + w[1], tap 2 holds w[2] (tap t = input offset t - 1; same in x; the sums taken in fp32, rounded to half once).
 * 4 tap-GEMMs per output pixel instead of 9.  Cin % 64 == 0, Cout % 8 == 0. */
int gip_upsample2x_conv3x3_nhwc_f16(const void* x, const void* wt4, const void* bias, void* out, int32_t N, int32_t Hin,
                                    int32_t Win, int32_t Cin, int32_t Cout, void* stream);

/* Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions of the 16 x 16 level (csrc/winograd.hip): the two transforms
 * around ONE batched library GEMM.  T = N * (H / 2) * (W / 2) output tiles of 2 x 2 pixels; H, W even; C % 8 == 0.
 *   gip_winograd_input_f16   x [N,H,W,C] half -> V [16][T][C] half, V[4 i + j] = (B^T d B)[i][j] of the 4 x 4 patch at
 *                            (2 ty - 1, 2 tx - 1) (zeros outside the image);
 *   (caller)                 M[p] = V[p] U[p]^T for p = 0..15, U [16][Cout][Cin] half = (G g G^T)[i][j] of the weight, made once in
 *                            fp32 — a batched GEMM [16, T, Cin] x [16, Cin, Cout] -> M [16][T][Cout] half;
 *   gip_winograd_output_f16  M -> out [N,H,W,Cout] half = A^T M A + bias (+ residual, added to the half-rounded result as the
 *                            implicit GEMM does).
 * 4 multiplications per output element and channel pair instead of 9; replaces the same F.conv2d calls as
 * gip_conv3x3_nhwc_f16 where the GEMMs dominate the transforms (measured: only the 1280 / 1920 / 2560-channel layers at
 * 16 x 16).  fp16 Winograd adds rounding of V and M: ~2x the implicit GEMM's error against fp32 (tests/test_gpu_conv.py). */
int gip_winograd_input_f16(const void* x, void* V, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* gip_winograd_input_f16 on y = silu?(GroupNorm(x + addend)) WITHOUT materialising y: x is the raw tensor, mean / rstd [N, G] come from
 * gip_gn_stats_from_partials (or any GroupNorm statistics pass), the normalisation — the apply pass' arithmetic, one rounding of y to
 * half — happens while the 4 x 4 patches are loaded.  ResnetBlock2D's conv(silu(norm(x))) at the levels that run as Winograd. */
int gip_winograd_input_gn_f16(const void* x, void* V, int32_t N, int32_t H, int32_t W, int32_t C, const void* gamma, const void* beta,
                              const float* mean, const float* rstd, int32_t G, int32_t apply_silu, const void* addend,
                              int32_t addend_stride, void* stream);
int gip_winograd_output_f16(const void* M, const void* bias, const void* residual, void* out, int32_t N, int32_t H, int32_t W,
                            int32_t C, void* stream);
/* gip_winograd_output_f16 that also writes chan_stats [N * H * W / 128][C][2] (may be NULL): the statistics of the GroupNorm
 * that consumes `out`, per 128 consecutive pixels and channel, as gip_conv3x3_stats_nhwc_f16 does.  W = 16 or 32, H * W % 128 == 0. */
int gip_winograd_output_stats_f16(const void* M, const void* bias, const void* residual, void* out, float* chan_stats, int32_t N,
                                  int32_t H, int32_t W, int32_t C, void* stream);

/* nn.Linear on the same MFMA machinery (TAPS = 1): out[m][n] = sum_k x[m][k] w[n][k] (+ bias[n]) (+ residual[m][n]),
 * x [M,K], w [Nout,K] (torch Linear weight), out [M,Nout] half, fp32 accumulation, K % 64 == 0.  geglu != 0: w has
 * 2*Nout rows [value | gate], bias 2*Nout, out = (xWv + bv) * gelu(xWg + bg) — diffusers' GEGLU feed-forward input
 * projection with the activation in the GEMM epilogue (the 2*Nout-wide intermediate never reaches HBM); Nout % 64 == 0.
 * Replaces hipBLASLt + separate bias / residual / GEGLU kernels for the transformer blocks' memory-bound projections. */
int gip_linear_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int32_t K,
                   int32_t Nout, int32_t geglu, void* stream);
/* B independent GEMMs in ONE launch: out[b] [M, Nout] = x[b] [M, K] . w[b]^T ([Nout, K]), entry b at x + b * bs_x, w + b * bs_w,
 * out + b * bs_o (element strides; bs_x, bs_w multiples of 8, bs_o of 4).  The sixteen products M[i] = V[i] U[i]^T of a Winograd
 * F(2x2, 3x3) convolution (csrc/winograd.hip) — round 4 ran them as one batched hipBLASLt call; sixteen separate launches of the
 * own kernel left the chip under-filled (60 tiles each).  K % 64 == 0, Nout % 4 == 0, no bias / residual / split-K.  Round 6: where
 * Nout % 256 == 0 and the batch of products fills the chip with 256 x 256 tiles (one 8-wave workgroup per CU) the launch takes those
 * (the 16 x 16 level: 3 x 5 x 16 = 240 tiles): at or below the batched library call's time at K <= 1280.
 * gip_linear_f16 and its variants (round 6): where the grid leaves at most one workgroup per CU (<= 256 tiles) and K >= 512, a
 * workgroup runs TWO K groups of four waves (each half of the K steps in its own stage buffers, accumulators handed over through
 * LDS): the result is (first half) + (second half) of the K sum — deterministic, same error against float32. */
int gip_linear_batched_f16(const void* x, const void* w, void* out, int32_t B, int64_t M, int32_t K, int32_t Nout,
                           int64_t bs_x, int64_t bs_w, int64_t bs_o, void* stream);

/* LayerNorm folded into the projection that consumes it (BasicTransformerBlock: norm1 -> q|k|v, norm2 -> to_q, norm3 -> GEGLU ff_in;
 * reference: diffusers BasicTransformerBlock as run by threestudio/models/guidance/ipa_guidance.py:311-358).
 *   out = LN(x) W^T + b  =  rstd_m (x_m . (W gamma)_n) - rstd_m mu_m s_n + t_n,   s_n = sum_k (W gamma)[n][k],  t_n = sum_k W[n][k] beta_k + b_n
 * gip_linear_rows_f16: gip_linear_f16 (no GEGLU) whose epilogue also leaves, per output row, the (sum, sum of squares) of the final
 *   half-rounded output over each channel tile: rows_out [M][gip_linear_row_parts(M, Nout)][2] float32.
 * gip_linear_ln_f16: x raw [M, K]; wg = W * gamma (half; GEGLU: [2 Nout, K] = [value | gate]); s, t float32 [Nout] ([2 Nout]);
 *   ln_rows [M][ln_parts][2] = the partial sums the producer of x left; eps = the LayerNorm's.  No LayerNorm kernel, no normalised copy. */
int32_t gip_linear_row_parts(int64_t M, int32_t Nout);
int gip_linear_rows_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M, int32_t K,
                        int32_t Nout, float* rows_out, void* stream);
int gip_linear_ln_f16(const void* x, const void* wg, const float* s, const float* t, void* out, int64_t M, int32_t K, int32_t Nout,
                      int32_t geglu, const float* ln_rows, int32_t ln_parts, float eps, void* stream);

/* Attention forward o = softmax(q k^T * scale) v  [+ weight2 * softmax(q k2^T * scale) v2]  (csrc/attention.hip):
 * q, o [B, Nq, H*D], k, v [B, Nkv, H*D], k2, v2 [B, Nkv2, H*D] half — the projection outputs / to_out input, heads
 * interleaved along the last axis (no head transposes).  fp32 softmax and accumulation.  No mask, no gradient (the
 * denoiser is frozen).  D in {40, 64, 80, 160}; Nq % 32 == 0 (a workgroup takes 128 queries, a wave 32); any Nkv, Nkv2 >= 1 (ragged tails are masked).  k2 = v2 = NULL:
 * one key set — the self-attention of LoRAAttnProcessor2_0 (attention_processor_faceid.py:300-318).  With k2 / v2: the
 * decoupled cross-attention of LoRAIPAttnProcessor2_0 (:462-500), text keys (77) and image-prompt keys (4) with their
 * own softmax each, hidden = text + scale * ip, in ONE pass over the queries. */
int gip_attention_fwd_f16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H, int32_t Nq,
                          int32_t Nkv, int32_t D, float scale, const void* k2, const void* v2, int32_t Nkv2,
                          float weight2, void* stream);
/* Same, with the key / value rows of a sample `ld_kv` (`ld_kv2` for k2 / v2) halves apart instead of H * D: k and v may
 * then be column ranges of ONE wide projection matrix that holds the to_k / to_v outputs of every cross-attention layer
 * of a network (the prompt embeddings are the same input for all 16 layers of attention_processor_faceid.py:433-523, so
 * the host projects them with one GEMM per forward).  ld % 8 == 0, ld >= H * D; a sample's rows are consecutive. */
int gip_attention_fwd_strided_f16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H, int32_t Nq,
                                  int32_t Nkv, int32_t D, float scale, const void* k2, const void* v2, int32_t Nkv2,
                                  float weight2, int32_t ld_kv, int32_t ld_kv2, void* stream);
/* Same, with the query rows `ld_q` halves apart as well: q, k and v may all be column ranges of ONE [B, N, 3 H D] matrix — the
 * fused to_q | to_k | to_v projection of a self-attention layer (one GEMM that reads the tokens once instead of three). */
int gip_attention_fwd_strided2_f16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H, int32_t Nq,
                                   int32_t Nkv, int32_t D, float scale, const void* k2, const void* v2, int32_t Nkv2,
                                   float weight2, int32_t ld_q, int32_t ld_kv, int32_t ld_kv2, void* stream);

/* Guidance glue (csrc/guidance_glue.hip): the element-wise algebra around the denoiser, one launch per stage instead of the
 * reference's op chains; same expressions, same order, same intermediate half roundings.
 * gip_image_prep_f16: rgb [B,C,2 Hout,2 Wout] float32 contiguous -> out [B,Hout,Wout,C] half (the channels-last image the VAE reads)
 *   = (F.interpolate(rgb, (Hout, Wout), "bilinear", align_corners=False).half() * 2 - 1)   (ipa_guidance.py:612-614, :524);
 *   _backward: g_out [B,Hout,Wout,C] half -> g_rgb [B,C,2 Hout,2 Wout] float32.
 * gip_latent_sample_f16: moments [B,2C,H,W] half with element strides m_strides[4] (host array: b, c, h, w), eps / noise [B,C,H,W]
 *   half contiguous, t [B] int64, acp [1000] float32 (alphas_cumprod) -> latents [B,C,H,W] half =
 *   (mean + exp(0.5 clamp(logvar, -30, 20)) eps) * scaling   (latent_dist.sample() * scaling_factor, :529) and
 *   noisy [replicas B,C,H,W] half = `replicas` copies of sqrt(acp_t) latents + sqrt(1 - acp_t) noise   (add_noise, :395-399);
 *   _backward: g_latents [B,C,H,W] half -> g_moments in the moments' layout.
 * gip_anpg_loss_f16: noise_pred [3B,C,H,W] half (neg | text | null; strides np_strides[4]), latents [B,C,H,W] half (lat_strides[4]) ->
 *   grad_out [B,C,H,W] float32 = nan_to_num(clip(w(t) * (guidance_scale (text - null) + (t < t_switch ? null : null - neg))))
 *   (:411-431; weighting 0 "sds" 1 - acp_t, 1 "uniform", 2 "fantasia3d"; clip_threshold <= 0: no clip; the clip's norm runs over
 *   the LAST axis), diff_out = lat32 - (lat32 - grad)  (what the MSE's backward multiplies, :645-653),
 *   scalars[0] = 0.5 sum diff^2 / B  (loss_sds), scalars[1] = ||grad||_2  (grad_norm); one wave per (b, h) row (W <= 64 with
 *   the clip), fixed summation order.
 * gip_scale_cast_f16: out[i] = half(x[i] * scale[0] * mult)  (scale: a device scalar, e.g. the upstream gradient of the loss). */
int gip_image_prep_f16(const float* rgb, int32_t B, int32_t C, int32_t Hout, int32_t Wout, void* out, void* stream);
int gip_image_prep_backward_f16(const void* g_out, int32_t B, int32_t C, int32_t Hout, int32_t Wout, float* g_rgb, void* stream);
int gip_latent_sample_f16(const void* moments, const int64_t* m_strides, const void* eps, const void* noise, const int64_t* t,
                          const float* acp, float scaling, int32_t B, int32_t C, int32_t H, int32_t W, int32_t replicas,
                          void* latents, void* noisy, void* stream);
int gip_latent_sample_backward_f16(const void* moments, const int64_t* m_strides, const void* eps, const void* g_latents,
                                   float scaling, int32_t B, int32_t C, int32_t H, int32_t W, void* g_moments, void* stream);
int gip_anpg_loss_f16(const void* noise_pred, const int64_t* np_strides, const void* latents, const int64_t* lat_strides,
                      const int64_t* t, const float* acp, int32_t B, int32_t C, int32_t H, int32_t W, float guidance_scale,
                      int32_t t_switch, int32_t weighting, float clip_threshold, float* grad_out, float* diff_out,
                      float* scalars, float* partials /* [B * H][2] scratch */, void* stream);
int gip_scale_cast_f16(const float* x, const float* scale, float mult, void* out, int64_t n, void* stream);
/* diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) as one launch: out [B, dim] half =
 * [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(max_period) k / (dim / 2)), float32 arithmetic, t [B] int64
 * (the U-Net's / ControlNet's time_proj, driven by ipa_guidance.py:311-358). */
int gip_timestep_embedding_f16(const int64_t* t, int32_t B, int32_t dim, float max_period, void* out, void* stream);
#ifdef __cplusplus
}
#endif
#endif
