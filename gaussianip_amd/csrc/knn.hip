// knn.hip — mean squared distance to the 3 nearest neighbours (distCUDA2 replacement) for gfx950.
//
// Reference: gaussiansplatting/submodules/simple-knn/simple_knn.cu — updateKBest<3> :131-145,
// boxMeanDist :147-183, SimpleKNN::knn :185-221 (Morton sort + 1024-point boxes used only to prune).
//
// MI355X formulation: exact tiled all-pairs.  A workgroup owns 256 query points (one per lane, best-3 kept
// in registers) and streams the whole cloud through LDS in tiles of 1024 points stored as SoA float arrays;
// every LDS read is a wave-wide broadcast, the global loads are perfectly coalesced, and there is no sort, no
// host read-back and no data-dependent control flow.  P = 100k (the shipped init) is 1e10 pair evaluations
// ~ 5 ms on one MI355X, run once per training job.  (For P >> 1M a cell-pruned variant is the next step.)
// Compiled with -ffp-contract=off so dx*dx + dy*dy + dz*dz rounds exactly like the CPU oracle.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/gip_knn.h"

#define KNN_BLOCK 256
#define KNN_TILE 1024

__global__ void __launch_bounds__(KNN_BLOCK)
gip_knn_kernel(int P, const float* __restrict__ pts, float* __restrict__ out) {
  __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
  const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  float rx = 0.f, ry = 0.f, rz = 0.f;
  if (i < P) { rx = pts[3 * i]; ry = pts[3 * i + 1]; rz = pts[3 * i + 2]; }
  float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
  for (int base = 0; base < P; base += KNN_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < KNN_TILE; t += KNN_BLOCK) {
      const int j = base + t;
      if (j < P) { sx[t] = pts[3 * j]; sy[t] = pts[3 * j + 1]; sz[t] = pts[3 * j + 2]; }
    }
    __syncthreads();
    const int cnt = min(KNN_TILE, P - base);
    for (int t = 0; t < cnt; t++) {
      const float dx = sx[t] - rx, dy = sy[t] - ry, dz = sz[t] - rz;
      float d = dx * dx + dy * dy + dz * dz;
      if (base + t == i) d = FLT_MAX;                 // self excluded
      // keep the three smallest in ascending order (branch-free insertion)
      const float n0 = fminf(b0, d), m0 = fmaxf(b0, d);
      const float n1 = fminf(b1, m0), m1 = fmaxf(b1, m0);
      b0 = n0; b1 = n1; b2 = fminf(b2, m1);
    }
  }
  if (i < P) out[i] = (b0 + b1 + b2) / 3.0f;
}

extern "C" size_t gip_knn_workspace_bytes(int32_t P) { (void)P; return 256; }

extern "C" int gip_knn_mean_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  if (P < 0 || (P > 0 && (!points || !out))) return 1;
  (void)workspace; (void)workspace_bytes;
  if (P == 0) return 0;
  hipLaunchKernelGGL(gip_knn_kernel, dim3((P + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), 0, (hipStream_t)stream, P,
                     points, out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
