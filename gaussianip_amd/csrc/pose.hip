// pose.hip — OpenPose control maps for all views of a step (include/gip_pose.h).  Integer / byte arithmetic throughout.
//
// Two launches: gip_pose_limb_spans_kernel (one wave per (view, limb)) restates what OpenCV does for
//   polygon = cv2.ellipse2Poly((int(mX), int(mY)), (int(length / 2), 4), int(angle), 0, 360, 1); cv2.fillConvexPoly(canvas, polygon, colour)
// (threestudio/utils/poser.py:895-897) — the 361-point 1-degree polygon from the float SinTable in double arithmetic,
// cvRound, duplicate removal, the outline by 8-connected Bresenham lines (clipLine + LineIterator, left to right) and the
// XY_SHIFT = 16 fixed-point scanline fill with its shared `edges` counter — and leaves the painted footprint as one
// [lo, hi] span per image row (a convex polygon's rows are single runs; tests/test_pose_oracle.py checks it on the
// oracle, tests/test_gpu_pose.py bit-exactly on the kernel).  gip_openpose_draw_kernel then replays the draw order per
// pixel: 18 discs (cv::Circle's filled midpoint circle of radius 4), 17 limbs blended 0.4 / 0.6 with uint8 rounding.
// oracle/pose_oracle.py is the same statement in numpy, function by function.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gip_pose.h"

__constant__ unsigned char c_colors[GIP_POSE_POINTS][3] = {
    {255, 0, 0},   {255, 85, 0},  {255, 170, 0}, {255, 255, 0}, {170, 255, 0}, {85, 255, 0},  {0, 255, 0},  {0, 255, 85}, {0, 255, 170},
    {0, 255, 255}, {0, 170, 255}, {0, 85, 255},  {0, 0, 255},   {85, 0, 255},  {170, 0, 255}, {255, 0, 255}, {255, 0, 170}, {255, 0, 85}};
// half-width of the radius-4 midpoint-circle footprint on row |dy| (cv::Circle, fill): rows 0..4
__constant__ int c_disc_half[5] = {4, 3, 3, 2, 0};
// OpenCV's `static const float SinTable[]` (drawing.cpp): sin(degrees), 0..450, seven-decimal literals
__constant__ float c_sin_table[451] = {
    0.0000000f, 0.0174524f, 0.0348995f, 0.0523360f, 0.0697565f, 0.0871557f, 0.1045285f, 0.1218693f,
    0.1391731f, 0.1564345f, 0.1736482f, 0.1908090f, 0.2079117f, 0.2249511f, 0.2419219f, 0.2588190f,
    0.2756374f, 0.2923717f, 0.3090170f, 0.3255682f, 0.3420201f, 0.3583679f, 0.3746066f, 0.3907311f,
    0.4067366f, 0.4226183f, 0.4383711f, 0.4539905f, 0.4694716f, 0.4848096f, 0.5000000f, 0.5150381f,
    0.5299193f, 0.5446390f, 0.5591929f, 0.5735764f, 0.5877853f, 0.6018150f, 0.6156615f, 0.6293204f,
    0.6427876f, 0.6560590f, 0.6691306f, 0.6819984f, 0.6946584f, 0.7071068f, 0.7193398f, 0.7313537f,
    0.7431448f, 0.7547096f, 0.7660444f, 0.7771460f, 0.7880108f, 0.7986355f, 0.8090170f, 0.8191520f,
    0.8290376f, 0.8386706f, 0.8480481f, 0.8571673f, 0.8660254f, 0.8746197f, 0.8829476f, 0.8910065f,
    0.8987940f, 0.9063078f, 0.9135455f, 0.9205049f, 0.9271839f, 0.9335804f, 0.9396926f, 0.9455186f,
    0.9510565f, 0.9563048f, 0.9612617f, 0.9659258f, 0.9702957f, 0.9743701f, 0.9781476f, 0.9816272f,
    0.9848078f, 0.9876883f, 0.9902681f, 0.9925462f, 0.9945219f, 0.9961947f, 0.9975641f, 0.9986295f,
    0.9993908f, 0.9998477f, 1.0000000f, 0.9998477f, 0.9993908f, 0.9986295f, 0.9975641f, 0.9961947f,
    0.9945219f, 0.9925462f, 0.9902681f, 0.9876883f, 0.9848078f, 0.9816272f, 0.9781476f, 0.9743701f,
    0.9702957f, 0.9659258f, 0.9612617f, 0.9563048f, 0.9510565f, 0.9455186f, 0.9396926f, 0.9335804f,
    0.9271839f, 0.9205049f, 0.9135455f, 0.9063078f, 0.8987940f, 0.8910065f, 0.8829476f, 0.8746197f,
    0.8660254f, 0.8571673f, 0.8480481f, 0.8386706f, 0.8290376f, 0.8191520f, 0.8090170f, 0.7986355f,
    0.7880108f, 0.7771460f, 0.7660444f, 0.7547096f, 0.7431448f, 0.7313537f, 0.7193398f, 0.7071068f,
    0.6946584f, 0.6819984f, 0.6691306f, 0.6560590f, 0.6427876f, 0.6293204f, 0.6156615f, 0.6018150f,
    0.5877853f, 0.5735764f, 0.5591929f, 0.5446390f, 0.5299193f, 0.5150381f, 0.5000000f, 0.4848096f,
    0.4694716f, 0.4539905f, 0.4383711f, 0.4226183f, 0.4067366f, 0.3907311f, 0.3746066f, 0.3583679f,
    0.3420201f, 0.3255682f, 0.3090170f, 0.2923717f, 0.2756374f, 0.2588190f, 0.2419219f, 0.2249511f,
    0.2079117f, 0.1908090f, 0.1736482f, 0.1564345f, 0.1391731f, 0.1218693f, 0.1045285f, 0.0871557f,
    0.0697565f, 0.0523360f, 0.0348995f, 0.0174524f, 0.0000000f, -0.0174524f, -0.0348995f, -0.0523360f,
    -0.0697565f, -0.0871557f, -0.1045285f, -0.1218693f, -0.1391731f, -0.1564345f, -0.1736482f, -0.1908090f,
    -0.2079117f, -0.2249511f, -0.2419219f, -0.2588190f, -0.2756374f, -0.2923717f, -0.3090170f, -0.3255682f,
    -0.3420201f, -0.3583679f, -0.3746066f, -0.3907311f, -0.4067366f, -0.4226183f, -0.4383711f, -0.4539905f,
    -0.4694716f, -0.4848096f, -0.5000000f, -0.5150381f, -0.5299193f, -0.5446390f, -0.5591929f, -0.5735764f,
    -0.5877853f, -0.6018150f, -0.6156615f, -0.6293204f, -0.6427876f, -0.6560590f, -0.6691306f, -0.6819984f,
    -0.6946584f, -0.7071068f, -0.7193398f, -0.7313537f, -0.7431448f, -0.7547096f, -0.7660444f, -0.7771460f,
    -0.7880108f, -0.7986355f, -0.8090170f, -0.8191520f, -0.8290376f, -0.8386706f, -0.8480481f, -0.8571673f,
    -0.8660254f, -0.8746197f, -0.8829476f, -0.8910065f, -0.8987940f, -0.9063078f, -0.9135455f, -0.9205049f,
    -0.9271839f, -0.9335804f, -0.9396926f, -0.9455186f, -0.9510565f, -0.9563048f, -0.9612617f, -0.9659258f,
    -0.9702957f, -0.9743701f, -0.9781476f, -0.9816272f, -0.9848078f, -0.9876883f, -0.9902681f, -0.9925462f,
    -0.9945219f, -0.9961947f, -0.9975641f, -0.9986295f, -0.9993908f, -0.9998477f, -1.0000000f, -0.9998477f,
    -0.9993908f, -0.9986295f, -0.9975641f, -0.9961947f, -0.9945219f, -0.9925462f, -0.9902681f, -0.9876883f,
    -0.9848078f, -0.9816272f, -0.9781476f, -0.9743701f, -0.9702957f, -0.9659258f, -0.9612617f, -0.9563048f,
    -0.9510565f, -0.9455186f, -0.9396926f, -0.9335804f, -0.9271839f, -0.9205049f, -0.9135455f, -0.9063078f,
    -0.8987940f, -0.8910065f, -0.8829476f, -0.8746197f, -0.8660254f, -0.8571673f, -0.8480481f, -0.8386706f,
    -0.8290376f, -0.8191520f, -0.8090170f, -0.7986355f, -0.7880108f, -0.7771460f, -0.7660444f, -0.7547096f,
    -0.7431448f, -0.7313537f, -0.7193398f, -0.7071068f, -0.6946584f, -0.6819984f, -0.6691306f, -0.6560590f,
    -0.6427876f, -0.6293204f, -0.6156615f, -0.6018150f, -0.5877853f, -0.5735764f, -0.5591929f, -0.5446390f,
    -0.5299193f, -0.5150381f, -0.5000000f, -0.4848096f, -0.4694716f, -0.4539905f, -0.4383711f, -0.4226183f,
    -0.4067366f, -0.3907311f, -0.3746066f, -0.3583679f, -0.3420201f, -0.3255682f, -0.3090170f, -0.2923717f,
    -0.2756374f, -0.2588190f, -0.2419219f, -0.2249511f, -0.2079117f, -0.1908090f, -0.1736482f, -0.1564345f,
    -0.1391731f, -0.1218693f, -0.1045285f, -0.0871557f, -0.0697565f, -0.0523360f, -0.0348995f, -0.0174524f,
    -0.0000000f, 0.0174524f, 0.0348995f, 0.0523360f, 0.0697565f, 0.0871557f, 0.1045285f, 0.1218693f,
    0.1391731f, 0.1564345f, 0.1736482f, 0.1908090f, 0.2079117f, 0.2249511f, 0.2419219f, 0.2588190f,
    0.2756374f, 0.2923717f, 0.3090170f, 0.3255682f, 0.3420201f, 0.3583679f, 0.3746066f, 0.3907311f,
    0.4067366f, 0.4226183f, 0.4383711f, 0.4539905f, 0.4694716f, 0.4848096f, 0.5000000f, 0.5150381f,
    0.5299193f, 0.5446390f, 0.5591929f, 0.5735764f, 0.5877853f, 0.6018150f, 0.6156615f, 0.6293204f,
    0.6427876f, 0.6560590f, 0.6691306f, 0.6819984f, 0.6946584f, 0.7071068f, 0.7193398f, 0.7313537f,
    0.7431448f, 0.7547096f, 0.7660444f, 0.7771460f, 0.7880108f, 0.7986355f, 0.8090170f, 0.8191520f,
    0.8290376f, 0.8386706f, 0.8480481f, 0.8571673f, 0.8660254f, 0.8746197f, 0.8829476f, 0.8910065f,
    0.8987940f, 0.9063078f, 0.9135455f, 0.9205049f, 0.9271839f, 0.9335804f, 0.9396926f, 0.9455186f,
    0.9510565f, 0.9563048f, 0.9612617f, 0.9659258f, 0.9702957f, 0.9743701f, 0.9781476f, 0.9816272f,
    0.9848078f, 0.9876883f, 0.9902681f, 0.9925462f, 0.9945219f, 0.9961947f, 0.9975641f, 0.9986295f,
    0.9993908f, 0.9998477f, 1.0000000f,
};

#define POSE_MAX_ROWS 2048
#define POSE_XY_SHIFT 16
#define POSE_XY_ONE (1 << POSE_XY_SHIFT)
#define POSE_EMPTY_LO 32767

__device__ __forceinline__ long long trunc_div(long long a, long long b) { return a / b; }      // C division truncates toward zero

// cv::clipLine(Size2l, Point2l&, Point2l&)
__device__ bool clip_line(int W, int H, long long& x1, long long& y1, long long& x2, long long& y2) {
  const long long right = W - 1, bottom = H - 1;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

__global__ void __launch_bounds__(64)
gip_pose_limb_spans_kernel(const float* __restrict__ limbs, int* __restrict__ spans, int H, int W) {
  __shared__ int s_raw[362][2];
  __shared__ int s_pts[362][2];
  __shared__ int s_lo[POSE_MAX_ROWS], s_hi[POSE_MAX_ROWS];
  const int l = blockIdx.x, v = blockIdx.y, lane = threadIdx.x;
  const float* q = limbs + ((size_t)v * GIP_POSE_LIMBS + l) * 6;
  int* out = spans + ((size_t)v * GIP_POSE_LIMBS + l) * H;
  for (int y = lane; y < H; y += 64) { s_lo[y] = POSE_EMPTY_LO; s_hi[y] = -1; }
  const bool on = q[3] != 0.f;
  if (!on) {                                           // uniform over the workgroup
    for (int y = lane; y < H; y += 64) out[y] = POSE_EMPTY_LO | (0 << 16);
    return;
  }
  const int cx = (int)q[0], cy = (int)q[1], ax = (int)q[2];
  int angle = (int)q[4];
  while (angle < 0) angle += 360;
  while (angle > 360) angle -= 360;
  // ---- ellipse2Poly: 361 points, double arithmetic on the float table, cvRound (round half to even)
  const double alpha = (double)c_sin_table[450 - angle], beta = (double)c_sin_table[angle];
  for (int i = lane; i < 361; i += 64) {
    const double x = (double)ax * (double)c_sin_table[450 - i], y = 4.0 * (double)c_sin_table[i];
    s_raw[i][0] = __double2int_rn((double)cx + x * alpha - y * beta);
    s_raw[i][1] = __double2int_rn((double)cy + x * beta + y * alpha);
  }
  __syncthreads();
  // consecutive duplicates dropped (order-preserving compaction by ballots)
  int count = 0;
  for (int base = 0; base < 361; base += 64) {
    const int i = base + lane;
    const bool keep = i < 361 && (i == 0 || s_raw[i][0] != s_raw[i - 1][0] || s_raw[i][1] != s_raw[i - 1][1]);
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
      s_pts[pos][0] = s_raw[i][0];
      s_pts[pos][1] = s_raw[i][1];
    }
    count += __popcll(m);
  }
  if (count == 1) {                                    // "a zero-size polygon": pts.assign(2, center)
    if (lane == 0) { s_pts[0][0] = s_pts[1][0] = cx; s_pts[0][1] = s_pts[1][1] = cy; }
    count = 2;
  }
  __syncthreads();
  const int n = count;
  // ---- outline: Line(p[i-1], p[i]) for every vertex = clipLine + LineIterator(8-connected, left to right)
  for (int s = lane; s < n; s += 64) {
    const int s0 = s == 0 ? n - 1 : s - 1;
    long long x1 = s_pts[s0][0], y1 = s_pts[s0][1], x2 = s_pts[s][0], y2 = s_pts[s][1];
    if (!clip_line(W, H, x1, y1, x2, y2)) continue;
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1), sy = 1;
    int x = (int)x1, y = (int)y1;
    if (dx < 0) { dx = -dx; dy = -dy; x = (int)x2; y = (int)y2; }
    if (dy < 0) { dy = -dy; sy = -1; }
    const bool vert = dy > dx;
    if (vert) { const int t = dx; dx = dy; dy = t; }
    int err = dx - (dy + dy);
    const int plus_delta = dx + dx, minus_delta = -(dy + dy);
    for (int k = 0; k <= dx; k++) {
      atomicMin(&s_lo[y], x);
      atomicMax(&s_hi[y], x);
      const bool m = err < 0;
      err += minus_delta + (m ? plus_delta : 0);
      if (vert) { y += sy; if (m) x += 1; }
      else { x += 1; if (m) y += sy; }
    }
  }
  __syncthreads();
  // ---- FillConvexPoly's scanline part, as written (one lane: the two edge chains share the `edges` counter)
  if (lane == 0) {
    long long xmin = s_pts[0][0], xmax = xmin, ymin = s_pts[0][1], ymax = ymin;
    int imin = 0;
    for (int i = 0; i < n; i++) {
      const long long px = s_pts[i][0], py = s_pts[i][1];
      if (py < ymin) { ymin = py; imin = i; }
      ymax = py > ymax ? py : ymax;
      xmax = px > xmax ? px : xmax;
      xmin = px < xmin ? px : xmin;
    }
    if (!(n < 3 || xmax < 0 || ymax < 0 || xmin >= W || ymin >= H)) {
      if (ymax > H - 1) ymax = H - 1;
      int e_idx[2] = {imin, imin}, e_di[2] = {1, n - 1}, e_ye[2] = {(int)ymin, (int)ymin};
      long long e_x[2] = {-POSE_XY_ONE, -POSE_XY_ONE}, e_dx[2] = {0, 0};
      int edges = n, y = (int)ymin;
      const int delta = POSE_XY_ONE >> 1;
      do {
        for (int i = 0; i < 2; i++) {
          if (y >= e_ye[i]) {
            int idx0 = e_idx[i];
            const int di = e_di[i];
            int idx = idx0 + di;
            if (idx >= n) idx -= n;
            for (; edges-- > 0;) {
              const int ty = s_pts[idx][1];
              if (ty > y) {
                const long long xs = (long long)s_pts[idx0][0] << POSE_XY_SHIFT, xe = (long long)s_pts[idx][0] << POSE_XY_SHIFT;
                e_ye[i] = ty;
                e_dx[i] = trunc_div((xe - xs) * 2 + (ty - y), 2 * (long long)(ty - y));
                e_x[i] = xs;
                e_idx[i] = idx;
                break;
              }
              idx0 = idx;
              idx += di;
              if (idx >= n) idx -= n;
            }
          }
        }
        if (edges < 0) break;
        if (y >= 0) {
          const int left = e_x[0] > e_x[1] ? 1 : 0, right = 1 - left;
          int xx1 = (int)((e_x[left] + delta) >> POSE_XY_SHIFT), xx2 = (int)((e_x[right] + delta) >> POSE_XY_SHIFT);
          if (xx2 >= 0 && xx1 < W) {
            if (xx1 < 0) xx1 = 0;
            if (xx2 >= W) xx2 = W - 1;
            if (xx1 <= xx2) {
              if (xx1 < s_lo[y]) s_lo[y] = xx1;
              if (xx2 > s_hi[y]) s_hi[y] = xx2;
            }
          }
        }
        e_x[0] += e_dx[0];
        e_x[1] += e_dx[1];
      } while (++y <= (int)ymax);
    }
  }
  __syncthreads();
  for (int y = lane; y < H; y += 64) out[y] = s_hi[y] < s_lo[y] ? (POSE_EMPTY_LO | (0 << 16)) : (s_lo[y] | (s_hi[y] << 16));
}

__device__ __forceinline__ unsigned char blend(unsigned char canvas, unsigned char src) {
  // cv2.addWeighted on uint8: float arithmetic, round half to even, saturate
  const float t = (float)canvas * 0.4f + (float)src * 0.6f;
  const int r = __float2int_rn(t);
  return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

__global__ void __launch_bounds__(256)
gip_openpose_draw_kernel(const int32_t* __restrict__ pts, const uint8_t* __restrict__ visible, const int* __restrict__ spans,
                         float* __restrict__ out, int H, int W) {
  __shared__ int s_px[GIP_POSE_POINTS], s_py[GIP_POSE_POINTS], s_vis[GIP_POSE_POINTS];
  const int v = blockIdx.y;
  if (threadIdx.x < GIP_POSE_POINTS) {
    s_px[threadIdx.x] = pts[((size_t)v * GIP_POSE_POINTS + threadIdx.x) * 2];
    s_py[threadIdx.x] = pts[((size_t)v * GIP_POSE_POINTS + threadIdx.x) * 2 + 1];
    s_vis[threadIdx.x] = visible[(size_t)v * GIP_POSE_POINTS + threadIdx.x];
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
  const int* sp = spans + (size_t)v * GIP_POSE_LIMBS * H + y;
  for (int l = 0; l < GIP_POSE_LIMBS; l++) {
    const int s = sp[(size_t)l * H];
    const int lo = s & 0xffff, hi = s >> 16;
    if (lo == POSE_EMPTY_LO && hi == 0) continue;      // limb not drawn (or no pixel on this row): canvas = blend(canvas, canvas)
    const bool inside = x >= lo && x <= hi;
    c0 = blend(c0, inside ? c_colors[l][0] : c0);
    c1 = blend(c1, inside ? c_colors[l][1] : c1);
    c2 = blend(c2, inside ? c_colors[l][2] : c2);
  }
  float* o = out + ((size_t)v * H * W + pix) * 3;
  o[0] = (float)c0 / 255.f; o[1] = (float)c1 / 255.f; o[2] = (float)c2 / 255.f;
}

extern "C" size_t gip_openpose_workspace_bytes(int32_t V, int32_t H) {
  return (size_t)(V > 0 ? V : 0) * GIP_POSE_LIMBS * (size_t)(H > 0 ? H : 0) * sizeof(int);
}

extern "C" int gip_openpose_draw(const int32_t* points_px, const uint8_t* visible, const float* limbs, float* out, int32_t V,
                                 int32_t H, int32_t W, void* workspace, size_t workspace_bytes, void* stream) {
  if (!points_px || !visible || !limbs || !out || !workspace || V < 1 || H < 1 || W < 1 || H > POSE_MAX_ROWS || W > 32766) return 1;
  if (workspace_bytes < gip_openpose_workspace_bytes(V, H)) return 2;
  hipLaunchKernelGGL(gip_pose_limb_spans_kernel, dim3(GIP_POSE_LIMBS, V), dim3(64), 0, (hipStream_t)stream, limbs, (int*)workspace, H, W);
  hipLaunchKernelGGL(gip_openpose_draw_kernel, dim3((H * W + 255) / 256, V), dim3(256), 0, (hipStream_t)stream, points_px, visible,
                     (const int*)workspace, out, H, W);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
