// render_backward.hip — per-pixel reverse-order replay + wavefront segmented reduction (gfx950).
//
// Replaces the fork's backward renderCUDA<3> (SURVEY.md §2.1 "bwd 1").  The reference lets every
// pixel thread atomicAdd nine floats per (pixel, Gaussian) pair into per-Gaussian accumulators — the
// contention hot spot named in SURVEY.md §8a3.  Here no float atomic is issued at all:
//
//   * a workgroup (4 x wave64, one 8x8 pixel quadrant per wave) walks its tile's depth-sorted list
//     back to front in batches staged through LDS;
//   * for each list entry every wave that has at least one contributing pixel reduces the ten
//     per-pixel gradient terms across its 64 lanes with DPP row-shift / row-broadcast adds (no LDS
//     traffic), and one lane deposits the wave's sums in LDS;
//   * after the batch the four wave slots are added in fixed order and written as ONE 64-byte row per
//     (tile, Gaussian) instance at row index inst_offset[g] + (tile's position inside g's rectangle).
//
// A Gaussian's rows are therefore contiguous and the follow-up kernel (gather_backward.hip) sums them
// in a fixed order: gradients are bitwise reproducible run to run, unlike the atomic formulation.
//
// Row layout (floats): 0,1 dL/dmean2D.xy (NDC units)  2,3,4 dL/dconic (x, y, w slots)  5 dL/dopacity
//                      6,7,8 dL/dcolour  9 dL/ddepth  10..15 zero
#include "gip_internal.h"

#define BWD_BATCH 128
#define BWD_NV 12   // floats kept per (wave, entry) slot; 10 used

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
// Sum of the 64 lanes, valid in lane 63.
__device__ __forceinline__ float wave_reduce_to_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the total
  return v;
}

__global__ void __launch_bounds__(GIP_BLOCK)
gip_render_backward_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_start,
                           const unsigned long long* __restrict__ keys, const GipRecord* __restrict__ records,
                           const uint32_t* __restrict__ inst_offset, const float* __restrict__ bg,
                           const uint32_t* __restrict__ n_contrib, const float* __restrict__ alpha_out,
                           const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                           const float* __restrict__ dL_dalpha_in, float* __restrict__ partial) {
  const uint32_t vt = blockIdx.x;
  const uint32_t v = vt / kp.T, tile = vt - v * kp.T;
  const uint32_t tx = tile % kp.tiles_x, ty = tile / kp.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = ((wave & 1) << 3) | (lane & 7), ly = ((wave >> 1) << 3) | (lane >> 3);
  const int px = tx * GIP_TILE + lx, py = ty * GIP_TILE + ly;
  const bool inside = px < kp.W && py < kp.H;
  const float pxf = (float)px, pyf = (float)py;

  const uint32_t start = tile_start[vt];
  uint32_t end = tile_start[vt + 1];
  if (end > kp.capacity) end = kp.capacity;
  if (end <= start) return;
  const int n = (int)(end - start);
  const GipRecord* recs = records + (size_t)v * kp.P;
  const uint32_t* ioff = inst_offset + (size_t)v * kp.P;

  __shared__ float2 s_xy[BWD_BATCH];
  __shared__ float4 s_con[BWD_BATCH];
  __shared__ float4 s_col[BWD_BATCH];
  __shared__ uint32_t s_row[BWD_BATCH];
  __shared__ float s_part[4][BWD_BATCH][BWD_NV];
  __shared__ uint32_t s_max[4];

  const size_t HW = (size_t)kp.H * kp.W;
  const size_t pix = (size_t)py * kp.W + px;
  float T_final = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f, ga = 0.f;
  uint32_t last_contributor = 0;
  if (inside) {
    T_final = 1.f - alpha_out[(size_t)v * HW + pix];
    last_contributor = n_contrib[(size_t)v * HW + pix];
    if (dL_dcolor) {
      const float* p = dL_dcolor + (size_t)v * 3 * HW;
      g0 = p[pix]; g1 = p[HW + pix]; g2 = p[2 * HW + pix];
    }
    if (dL_ddepth) gd = dL_ddepth[(size_t)v * HW + pix];
    if (dL_dalpha_in) ga = dL_dalpha_in[(size_t)v * HW + pix];
  }
  const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
  const float ddelx_dx = 0.5f * kp.W, ddely_dy = 0.5f * kp.H;

  // longest replay needed by any pixel of the tile
  {
    uint32_t m = gip_wave_max_u32(last_contributor);
    if (lane == 0) s_max[wave] = m;
  }
  __syncthreads();
  const int maxc = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

  float T = T_final;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f, acca = 0.f;
  float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;

  for (int hi = n; hi > 0; hi -= BWD_BATCH) {
    const int lo = max(0, hi - BWD_BATCH);
    const int cnt = hi - lo;
    __syncthreads();   // previous batch fully flushed
    if ((int)threadIdx.x < cnt) {
      const int i = hi - 1 - (int)threadIdx.x;   // slot j holds list entry hi-1-j: back to front
      const uint32_t g = (uint32_t)keys[start + i];
      const float4* rp = reinterpret_cast<const float4*>(recs + g);
      const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
      const uint4 q3 = reinterpret_cast<const uint4*>(recs + g)[3];
      s_xy[threadIdx.x] = make_float2(q0.x, q0.y);
      s_con[threadIdx.x] = make_float4(q1.x, q1.y, q1.z, q0.w);
      s_col[threadIdx.x] = make_float4(q2.x, q2.y, q2.z, q0.z);
      const uint32_t rminx = q3.x & 0xffff, rminy = q3.x >> 16, rmaxx = q3.y & 0xffff;
      s_row[threadIdx.x] = ioff[g] + (ty - rminy) * (rmaxx - rminx) + (tx - rminx);
    }
    for (int e = threadIdx.x; e < 4 * BWD_BATCH * BWD_NV; e += GIP_BLOCK) (&s_part[0][0][0])[e] = 0.f;
    __syncthreads();

    if (lo < maxc) {
      for (int j = 0; j < cnt; j++) {
        const int i = hi - 1 - j;
        if (i >= maxc) continue;                       // uniform over the workgroup
        const float2 xy = s_xy[j];
        const float4 co = s_con[j];
        const float dx = xy.x - pxf, dy = xy.y - pyf;
        const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
        const float G = __expf(power);
        const float alpha = fminf(GIP_ALPHA_MAX, co.w * G);
        const bool c = (uint32_t)i < last_contributor && power <= 0.0f && alpha >= GIP_ALPHA_MIN;
        if (__any(c)) {                                 // uniform over the wave
          float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f, v9 = 0.f;
          if (c) {
            const float4 cl = s_col[j];
            const float rcp1ma = __builtin_amdgcn_rcpf(1.f - alpha);
            T = T * rcp1ma;
            const float dchannel_dcolor = alpha * T;
            float dL_dalpha;
            acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = cl.x;
            acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = cl.y;
            acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = cl.z;
            dL_dalpha = (cl.x - acc0) * g0 + (cl.y - acc1) * g1 + (cl.z - acc2) * g2;
            v6 = dchannel_dcolor * g0; v7 = dchannel_dcolor * g1; v8 = dchannel_dcolor * g2;
            accd = last_alpha * last_depth + (1.f - last_alpha) * accd; last_depth = cl.w;
            dL_dalpha += (cl.w - accd) * gd;
            v9 = dchannel_dcolor * gd;
            acca = last_alpha + (1.f - last_alpha) * acca;
            dL_dalpha += (1.f - acca) * ga;
            dL_dalpha *= T;
            last_alpha = alpha;
            dL_dalpha += (-T_final * rcp1ma) * bg_dot;
            const float dL_dG = co.w * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
            v0 = dL_dG * (-gdx * co.x - gdy * co.y) * ddelx_dx;
            v1 = dL_dG * (-gdy * co.z - gdx * co.y) * ddely_dy;
            v2 = -0.5f * gdx * dx * dL_dG;
            v3 = -0.5f * gdx * dy * dL_dG;
            v4 = -0.5f * gdy * dy * dL_dG;
            v5 = G * dL_dalpha;
          }
          v0 = wave_reduce_to_lane63(v0); v1 = wave_reduce_to_lane63(v1);
          v2 = wave_reduce_to_lane63(v2); v3 = wave_reduce_to_lane63(v3);
          v4 = wave_reduce_to_lane63(v4); v5 = wave_reduce_to_lane63(v5);
          v6 = wave_reduce_to_lane63(v6); v7 = wave_reduce_to_lane63(v7);
          v8 = wave_reduce_to_lane63(v8); v9 = wave_reduce_to_lane63(v9);
          if (lane == 63) {
            float4* dst = reinterpret_cast<float4*>(&s_part[wave][j][0]);
            dst[0] = make_float4(v0, v1, v2, v3);
            dst[1] = make_float4(v4, v5, v6, v7);
            dst[2] = make_float4(v8, v9, 0.f, 0.f);
          }
        }
      }
    }
    __syncthreads();
    // flush: 4 lanes per entry, one 16-byte quad each -> one 64-byte row per (tile, Gaussian)
    for (int e = threadIdx.x; e < cnt * 4; e += GIP_BLOCK) {
      const int j = e >> 2, q = e & 3;
      const uint32_t row = s_row[j];
      if (row < kp.capacity) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < 3) {
          const float4 a = reinterpret_cast<const float4*>(&s_part[0][j][0])[q];
          const float4 b = reinterpret_cast<const float4*>(&s_part[1][j][0])[q];
          const float4 c4 = reinterpret_cast<const float4*>(&s_part[2][j][0])[q];
          const float4 d = reinterpret_cast<const float4*>(&s_part[3][j][0])[q];
          s.x = (a.x + b.x) + (c4.x + d.x); s.y = (a.y + b.y) + (c4.y + d.y);
          s.z = (a.z + b.z) + (c4.z + d.z); s.w = (a.w + b.w) + (c4.w + d.w);
        }
        reinterpret_cast<float4*>(partial + (size_t)row * GIP_PARTIAL_FLOATS)[q] = s;
      }
    }
  }
}

void gip_launch_render_backward(const GipKernelParams& kp, const float* bg, GipStatePtrs st, const GipRasterGradsIn& gin,
                                float* partial, hipStream_t s) {
  hipLaunchKernelGGL(gip_render_backward_kernel, dim3(kp.V * kp.T), dim3(GIP_BLOCK), 0, s, kp, st.tile_start, st.keys,
                     st.records, st.inst_offset, bg, st.n_contrib, gin.alpha, gin.dL_dcolor, gin.dL_ddepth,
                     gin.dL_dalpha, partial);
}
