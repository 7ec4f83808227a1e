// pose.hip — OpenPose control maps for all views of a step in one launch (include/gip_pose.h).  Byte arithmetic, one
// thread per pixel; the per-view primitives (18 discs, 17 ellipses) are prepared by the first lanes of each workgroup
// in LDS.  Output-bound (12 bytes per pixel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gip_pose.h"

__constant__ unsigned char c_colors[GIP_POSE_POINTS][3] = {
    {255, 0, 0},   {255, 85, 0},  {255, 170, 0}, {255, 255, 0}, {170, 255, 0}, {85, 255, 0},  {0, 255, 0},  {0, 255, 85}, {0, 255, 170},
    {0, 255, 255}, {0, 170, 255}, {0, 85, 255},  {0, 0, 255},   {85, 0, 255},  {170, 0, 255}, {255, 0, 255}, {255, 0, 170}, {255, 0, 85}};
__constant__ int c_lines[GIP_POSE_LIMBS][2] = {{0, 1}, {1, 2},  {2, 3},   {3, 4},   {1, 5},   {5, 6},  {6, 7},   {1, 8},  {8, 9},
                                               {9, 10}, {1, 11}, {11, 12}, {12, 13}, {0, 14}, {14, 16}, {0, 15}, {15, 17}};
// half-width of the radius-4 midpoint-circle footprint on row |dy| (cv2.circle, filled): rows 0..4
__constant__ int c_disc_half[5] = {4, 3, 3, 2, 0};

struct Limb {
  int cx, cy, a, on;
  float cs, sn;
};

__device__ __forceinline__ unsigned char blend(unsigned char canvas, unsigned char src) {
  // cv2.addWeighted on uint8: float arithmetic, round half to even, saturate
  const float t = (float)canvas * 0.4f + (float)src * 0.6f;
  const int r = __float2int_rn(t);
  return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

__global__ void __launch_bounds__(256)
gip_openpose_draw_kernel(const int32_t* __restrict__ pts, const uint8_t* __restrict__ visible, const float* __restrict__ limbs,
                         float* __restrict__ out, int H, int W) {
  __shared__ int s_px[GIP_POSE_POINTS], s_py[GIP_POSE_POINTS], s_vis[GIP_POSE_POINTS];
  __shared__ Limb s_limb[GIP_POSE_LIMBS];
  const int v = blockIdx.y;
  if (threadIdx.x < GIP_POSE_POINTS) {
    s_px[threadIdx.x] = pts[((size_t)v * GIP_POSE_POINTS + threadIdx.x) * 2];
    s_py[threadIdx.x] = pts[((size_t)v * GIP_POSE_POINTS + threadIdx.x) * 2 + 1];
    s_vis[threadIdx.x] = visible[(size_t)v * GIP_POSE_POINTS + threadIdx.x];
  }
  if (threadIdx.x >= 32 && threadIdx.x < 32 + GIP_POSE_LIMBS) {
    const int l = threadIdx.x - 32;
    const float* q = limbs + ((size_t)v * GIP_POSE_LIMBS + l) * 6;
    Limb L;
    L.cx = (int)q[0]; L.cy = (int)q[1]; L.a = (int)q[2]; L.on = q[3] != 0.f; L.cs = q[4]; L.sn = q[5];
    s_limb[l] = L;
  }
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int y = pix / W, x = pix - y * W;
  unsigned char c0 = 0, c1 = 0, c2 = 0;
  for (int i = 0; i < GIP_POSE_POINTS; i++) {
    if (!s_vis[i]) continue;
    const int dy = abs(y - s_py[i]), dx = abs(x - s_px[i]);
    if (dy <= 4 && dx <= c_disc_half[dy]) { c0 = c_colors[i][0]; c1 = c_colors[i][1]; c2 = c_colors[i][2]; }
  }
  for (int l = 0; l < GIP_POSE_LIMBS; l++) {
    const Limb L = s_limb[l];
    if (!L.on) continue;
    const float dx = (float)(x - L.cx), dy = (float)(y - L.cy);
    const float u = dx * L.cs + dy * L.sn, w = -dx * L.sn + dy * L.cs;
    const float ua = u / ((float)L.a + 0.5f), wb = w / 4.5f;
    const bool inside = ua * ua + wb * wb <= 1.0f;
    c0 = blend(c0, inside ? c_colors[l][0] : c0);
    c1 = blend(c1, inside ? c_colors[l][1] : c1);
    c2 = blend(c2, inside ? c_colors[l][2] : c2);
  }
  float* o = out + ((size_t)v * H * W + pix) * 3;
  o[0] = (float)c0 / 255.f; o[1] = (float)c1 / 255.f; o[2] = (float)c2 / 255.f;
}

extern "C" int gip_openpose_draw(const int32_t* points_px, const uint8_t* visible, const float* limbs, float* out, int32_t V,
                                 int32_t H, int32_t W, void* stream) {
  if (!points_px || !visible || !limbs || !out || V < 1 || H < 1 || W < 1) return 1;
  hipLaunchKernelGGL(gip_openpose_draw_kernel, dim3((H * W + 255) / 256, V), dim3(256), 0, (hipStream_t)stream, points_px, visible,
                     limbs, out, H, W);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
