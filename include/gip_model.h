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
#ifdef __cplusplus
}
#endif
#endif
