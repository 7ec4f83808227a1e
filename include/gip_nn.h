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
                        void* workspace, size_t workspace_bytes, void* stream);
int gip_gn_silu_backward(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                         const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t apply_silu,
                         void* workspace, size_t workspace_bytes, void* stream);
#ifdef __cplusplus
}
#endif
#endif
