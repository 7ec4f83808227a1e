/*
 * gip_knn.h — C-ABI of the MI355X-native replacement of simple_knn's distCUDA2.
 *
 *   gip_knn_mean_dist2  <->  simple_knn._C.distCUDA2(points) -> Tensor[P]
 *       reference: gaussiansplatting/submodules/simple-knn/ext.cpp:16 (binding), spatial.cu:16-25 (wrapper),
 *       simple_knn.cu:185-221 (SimpleKNN::knn), called once at gaussiansplatting/scene/gaussian_model.py:123.
 *
 * out[i] = (d1 + d2 + d3) / 3 with d1 <= d2 <= d3 the three smallest squared distances from point i to the
 * other points (self excluded) — exactly what simple_knn.cu:147-183 computes (its Morton sort / box rejection
 * only prunes the search).  Plain C, raw device pointers, caller-owned workspace, work enqueued on `stream`,
 * integer status (0 = ok, 1 = bad argument, 2 = workspace too small, 3 = HIP error).
 */
#ifndef GIP_KNN_H
#define GIP_KNN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t gip_knn_workspace_bytes(int32_t P);
int gip_knn_mean_dist2(int32_t P, const float* points /* [P,3] device */, float* out /* [P] device */,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same result by a chosen algorithm: mode 0 = by size (what gip_knn_mean_dist2 does: all pairs up to gip_knn_prune_from
 * points, box-pruned above), 1 = exact tiled all-pairs, 2 = Morton sort + 1024-point boxes that prune the exact search
 * (simple_knn.cu:45-185: coord2Morton, boxMinMax, distBoxPoint, boxMeanDist).  Both give the identical float per point.
 * gip_knn_workspace_bytes_mode: the workspace that mode needs for P points. */
size_t gip_knn_workspace_bytes_mode(int32_t P, int32_t mode);
int gip_knn_mean_dist2_mode(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes, int32_t mode,
                            void* stream);
#ifdef __cplusplus
}
#endif
#endif
