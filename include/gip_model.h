/*
 * gip_model.h — C-ABI of the device-side Gaussian-set surgery used by densify / prune.
 *
 *   gip_gather_rows  <->  the chain of boolean-mask indexing and torch.cat calls that rebuilds the six parameter
 *       tensors and their twelve Adam moment tensors whenever the reference densifies or prunes
 *       (gaussiansplatting/scene/gaussian_model.py:292-355 prune_points / cat_tensors_to_optimizer /
 *       densification_postfix, driven by densify_and_prune :395-411 and prune_only :413-418).
 *
 * The final row order of clone -> split -> prune is fully described by one index list over the virtual concatenation
 * [old rows | new rows]; every tensor is then rebuilt by ONE gather, all tensors in ONE launch:
 *       dst_t[j] = index[j] < n_old ? old_t[index[j]] : (new_t ? new_t[index[j] - n_old] : 0)
 * (new_t == NULL writes zero rows: the Adam moments of freshly created Gaussians).  Byte-exact data movement: rows
 * are copied as 4-byte words, row_bytes % 4 == 0.  Plain C, raw device pointers, caller-owned buffers, work enqueued
 * on `stream`, integer status (0 ok, 1 bad argument, 3 HIP error).  `tensors` is a HOST array (at most 24 entries).
 */
#ifndef GIP_MODEL_H
#define GIP_MODEL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define GIP_GATHER_MAX_TENSORS 24
typedef struct {
  const void* old_rows; /* [n_old, row_bytes] device */
  const void* new_rows; /* [n_new, row_bytes] device, or NULL for zero rows */
  void* dst;            /* [n_out, row_bytes] device */
  int32_t row_bytes;
  int32_t reserved;
} GipGatherTensor;
int gip_gather_rows(const GipGatherTensor* tensors, int32_t n_tensors, const int64_t* index /* [n_out] device */,
                    int64_t n_out, int64_t n_old, void* stream);

/* Bucket packing for the per-step multi-GPU exchange (gaussianip_amd/parallel.py; SURVEY.md §8e: all_reduce(sum) of the
 * six parameter gradients and of the per-Gaussian view-space gradient norms that GaussianIP.py:452-457 accumulates):
 *   gip_pack_bucket    flat = [seg_0 | seg_1 | ... | tail], tail[p] = sum_v sqrt(g2d[v,p,0]^2 + g2d[v,p,1]^2) when g2d is
 *                      given (g2d [V, P, 3] float, the means2D gradients of the local views; tail has P floats), in ONE
 *                      launch instead of a norm, a sum and a concatenation;
 *   gip_unpack_bucket  seg_i <- flat[offset_i : offset_i + n_i] * scale (scale = 1 / world for averaged gradients), and
 *                      tail_dst <- flat tail (unscaled) when given, in one launch.
 * Segments are float32 device arrays (at most GIP_PACK_MAX_SEGS), `segs` / `counts` are HOST arrays. */
#define GIP_PACK_MAX_SEGS 12
int gip_pack_bucket(const void* const* segs, const int64_t* counts, int32_t n_segs, const void* g2d, int32_t V, int64_t P,
                    void* flat, void* stream);
int gip_unpack_bucket(void* const* segs, const int64_t* counts, int32_t n_segs, void* tail_dst, int64_t tail_count,
                      const void* flat, float scale, void* stream);

/* The MAX bucket of the same exchange (GaussianIP.py:225 batch-global depth maximum, :452-457 max_radii2D), straight
 * from the rasterizer's forward outputs in one launch:
 *   out[p] = max_v radii[v, p]  (p < P, int32),   out[P] = bit pattern of max(depth[0 .. n_depth))  (float32 >= 0:
 *   non-negative IEEE floats order like their bit patterns, so an int32 all_reduce(max) of the whole bucket is exact). */
int gip_max_bucket(const int32_t* radii, int32_t V, int64_t P, const float* depth, int64_t n_depth, int32_t* out, void* stream);

/* One Adam step for ALL parameter groups of the Gaussian model in ONE launch (torch.optim.Adam, no weight decay, no amsgrad;
 * gaussiansplatting/scene/gaussian_model.py:145-155 builds six single-tensor groups with their own learning rates, eps 1e-15).
 * torch's fused Adam launches three multi-tensor kernels per group (18 launches, 0.29 ms of a 39 ms step for 5.6 MB of state).
 *   step_t += 1;  m = m + (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^t) * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps))        (torch.optim.Adam's operation order; the bias
 *   corrections and 1 - beta in double arithmetic like torch's Python-side constants)
 * `step` of every group is a device float scalar (torch's own state["step"] of a fused / capturable Adam) and is incremented by
 * the kernel.  found_inf (device float, may be NULL): a GradScaler's verdict — when non-zero nothing is written (the skipped
 * step of torch.amp; the gradients were unscaled before).  `groups` is a HOST array (at most GIP_ADAM_MAX_GROUPS). */
#define GIP_ADAM_MAX_GROUPS 8
typedef struct {
  void* param;        /* [n] float32 device */
  const void* grad;   /* [n] float32 device */
  void* exp_avg;      /* [n] float32 device */
  void* exp_avg_sq;   /* [n] float32 device */
  float* step;        /* device scalar */
  int64_t n;
  float lr;
  int32_t reserved;
} GipAdamGroup;
int gip_adam_step(const GipAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps, const float* found_inf,
                  void* stream);   /* betas / eps as doubles: the caller's Python floats, no re-rounding (0 <= beta < 1, else status 1) */

/* The sparsity term of the stage-1 loss (threestudio/systems/GaussianIP.py:225, :377-380):
 *   mean(sqrt((depth / (max(depth) + 1e-5))^2 + 0.01))   over the n = B * H * W depths of a step,
 * three launches forward, two backward (the reference's op chain: ~20 launches on 4 M elements), fixed summation orders.
 * workspace: gip_sparsity_workspace_bytes() bytes, kept between forward and backward;
 * after forward: ((float*)workspace)[0] = max(depth), [1] = the term.  backward: g_depth[i] = d term / d depth[i] * g_loss[0] * mult
 * (the maximum's share through the denominator goes evenly to the elements equal to it, like torch.max()'s backward). */
size_t gip_sparsity_workspace_bytes(void);
int gip_sparsity_loss_forward(const float* depth, int64_t n, void* workspace, void* stream);
int gip_sparsity_loss_backward(const float* depth, int64_t n, const float* g_loss, float mult, void* workspace, float* g_depth,
                               void* stream);

/* The three parameter activations of GaussianModel (gaussiansplatting/scene/gaussian_model.py:36-41, getters :72-89) in one launch:
 *   opacity [P] = sigmoid(opacity_raw), scaling [P,3] = exp(scaling_raw), rotation [P,4] = rotation_raw / max(||rotation_raw||, 1e-12)
 * (float32, contiguous), and their backward in one launch (g_* may be NULL = no gradient arrived for that output; d_* may be NULL). */
int gip_activate_gaussians(const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, int64_t P, float* opacity,
                           float* scaling, float* rotation, void* stream);
int gip_activate_gaussians_backward(const float* opacity, const float* scaling, const float* rotation_raw, const float* g_opacity,
                                    const float* g_scaling, const float* g_rotation, int64_t P, float* d_opacity_raw,
                                    float* d_scaling_raw, float* d_rotation_raw, void* stream);

/* Densification statistics of one step in one launch (threestudio/systems/GaussianIP.py:451-457, gaussian_model.py:420-422):
 *   grad = sum over the V views of viewspace_grad [V,P,3];  where visible: max_radii2D = max(max_radii2D, radii);
 *   xyz_gradient_accum += ||grad[:, :2]|| * visible;  denom += visible.   (visible: bytes 0 / 1; radii int32; the rest float32 [P].) */
int gip_densify_stats(const float* viewspace_grad, int32_t V, int64_t P, const uint8_t* visible, const int32_t* radii,
                      float* max_radii2D, float* xyz_gradient_accum, float* denom, void* stream);
#ifdef __cplusplus
}
#endif
#endif
