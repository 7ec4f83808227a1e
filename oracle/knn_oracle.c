/*
 * knn_oracle.c — CPU restatement of simple_knn's distCUDA2 (TEST INFRASTRUCTURE ONLY).
 *
 * Follows gaussiansplatting/submodules/simple-knn/simple_knn.cu:
 *   updateKBest<3>   :131-145  (insertion into the 3 smallest squared distances, self excluded)
 *   boxMeanDist      :147-183  (result = (best0 + best1 + best2) / 3.0f, written at the point's own index)
 * The Morton sort + box pruning of :185-221 only accelerates the search: the rejection test
 * (:170-172) never discards a box that could hold one of the 3 nearest neighbours, so the result
 * is the exact 3-NN mean and a brute-force scan reproduces it.  Parity is pinned by the golden
 * vector tests/golden/knn_dist2.npz (brute-force torch.cdist/topk stub used when importing the
 * reference's GaussianModel).
 * The CUDA file cannot be compiled here (no nvcc; needs cub/thrust) — "unbuildable", see DESIGN.md.
 */
#include <float.h>
#include <stdint.h>

void oracle_knn_mean_dist2(int P, const float* pts, float* out) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const float rx = pts[3 * i], ry = pts[3 * i + 1], rz = pts[3 * i + 2];
    for (int j = 0; j < P; j++) {
      if (j == i) continue;
      float dx = pts[3 * j] - rx, dy = pts[3 * j + 1] - ry, dz = pts[3 * j + 2] - rz;
      float dist = dx * dx + dy * dy + dz * dz;
      for (int k = 0; k < 3; k++) {
        if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
      }
    }
    out[i] = (best[0] + best[1] + best[2]) / 3.0f;
  }
}
