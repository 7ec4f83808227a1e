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
#ifdef __cplusplus
}
#endif
#endif
