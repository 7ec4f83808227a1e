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
// lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l]   (v_permlane32_swap + add)
__device__ __forceinline__ float fold32(float a, float b) {
  // inline asm: the clang builtin's second result is mis-selected for float operands on ROCm 7.2 (both
  // extracts return the first register); the swap updates both registers in place.
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// rows (16 lanes) 0,2 <- a.row(r) + a.row(r+1), rows 1,3 <- b.row(r-1) + b.row(r)   (v_permlane16_swap + add)
__device__ __forceinline__ float fold16(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
// sum of each 16-lane row, valid in the row's lane 15
__device__ __forceinline__ float row_reduce(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  return v;
}

__global__ void __launch_bounds__(GIP_BLOCK)
gip_render_backward_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_order,
                           const uint32_t* __restrict__ tile_start,
                           const unsigned long long* __restrict__ keys, const GipRecord* __restrict__ records,
                           const uint32_t* __restrict__ inst_offset, const float* __restrict__ bg,
                           const uint32_t* __restrict__ n_contrib, const float* __restrict__ alpha_out,
                           const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                           const float* __restrict__ dL_dalpha_in, float* __restrict__ partial) {
  const uint32_t vt = tile_order[blockIdx.x];
  const uint32_t start = tile_start[vt];
  uint32_t end = tile_start[vt + 1];
  if (end > kp.capacity) end = kp.capacity;
  if (end <= start) return;
  const uint32_t v = vt / kp.T, tile = vt - v * kp.T;
  const uint32_t tx = tile % kp.tiles_x, ty = tile / kp.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = ((wave & 1) << 3) | (lane & 7), ly = ((wave >> 1) << 3) | (lane >> 3);
  const int px = tx * GIP_TILE + lx, py = ty * GIP_TILE + ly;
  const bool inside = px < kp.W && py < kp.H;
  const float pxf = (float)px, pyf = (float)py;
  const float tile_x0 = (float)(tx * GIP_TILE), tile_y0 = (float)(ty * GIP_TILE);
  const int n = (int)(end - start);
  const GipRecord* recs = records + (size_t)v * kp.P;
  const uint32_t* ioff = inst_offset + (size_t)v * kp.P;

  __shared__ float2 s_xy[BWD_BATCH];
  __shared__ float4 s_con[BWD_BATCH];
  __shared__ float4 s_col[BWD_BATCH];
  __shared__ uint32_t s_row[BWD_BATCH];
  __shared__ uint32_t s_mask[BWD_BATCH];
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
  const int wave_maxc = (int)gip_wave_max_u32(last_contributor);
  if (lane == 0) s_max[wave] = (uint32_t)wave_maxc;
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
      // quadrant mask from the extent of { alpha >= 1/255 } (same conservative test as the forward kernel)
      uint32_t mask = 0xf;
      const float t2 = 2.0f * __logf(255.0f * q0.w) + 0.02f;
      const float det = q1.x * q1.z - q1.y * q1.y;
      if (t2 <= 0.f || i >= maxc) {
        mask = 0;
      } else if (det > 0.f) {
        const float inv = t2 / det;
        const float hx = sqrtf(inv * q1.z) * 1.01f + 0.05f, hy = sqrtf(inv * q1.x) * 1.01f + 0.05f;
        const float rx = q0.x - tile_x0, ry = q0.y - tile_y0;
        const bool xl = rx - hx <= 7.f, xr = rx + hx >= 8.f, yt = ry - hy <= 7.f, yb = ry + hy >= 8.f;
        mask = (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
      }
      s_mask[threadIdx.x] = mask;
    }
    for (int e = threadIdx.x; e < 4 * BWD_BATCH * BWD_NV; e += GIP_BLOCK) (&s_part[0][0][0])[e] = 0.f;
    __syncthreads();

    if (lo < wave_maxc) {
      for (int c0 = 0; c0 < cnt; c0 += 64) {
        const int e = c0 + lane;
        const bool want = e < cnt && ((s_mask[e] >> wave) & 1u) && (hi - 1 - e) < wave_maxc;
        unsigned long long m = __ballot(want);
        while (m) {
          const int j = c0 + __builtin_ctzll(m);
          m &= m - 1;
          const int i = hi - 1 - j;
          const float2 xy = s_xy[j];
          const float4 co = s_con[j];
          const float dx = xy.x - pxf, dy = xy.y - pyf;
          const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
          const float G = __expf(power);
          const float alpha = fminf(GIP_ALPHA_MAX, co.w * G);
          const bool c = (uint32_t)i < last_contributor && power <= 0.0f && alpha >= GIP_ALPHA_MIN;
          if (!__any(c)) continue;                      // uniform over the wave
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
          // 64-lane sums of the ten terms in 27 cross-lane ops: fold halves (permlane32_swap), fold row
          // pairs (permlane16_swap), then one DPP row reduction per packed register.
          const float p02 = fold32(v0, v2), p13 = fold32(v1, v3);   // [v0|v2], [v1|v3]
          const float p46 = fold32(v4, v6), p57 = fold32(v5, v7);
          const float p89 = fold32(v8, v9);
          float qa = fold16(p02, p13);   // rows: v0, v1, v2, v3
          float qb = fold16(p46, p57);   // rows: v4, v5, v6, v7
          qa = row_reduce(qa);
          qb = row_reduce(qb);
          float qc = row_reduce(p89);    // rows 0,1: v8 ; rows 2,3: v9
          qc = dpp_add<0x142, 0xa>(qc);  // row_bcast:15 -> lane 31 = v8, lane 63 = v9
          if ((lane & 15) == 15) {
            const int r = lane >> 4;
            float* dst = &s_part[wave][j][0];
            dst[r] = qa;
            dst[4 + r] = qb;
            if (r & 1) dst[8 + (r >> 1)] = qc;
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
  hipLaunchKernelGGL(gip_render_backward_kernel, dim3(kp.V * kp.T), dim3(GIP_BLOCK), 0, s, kp, st.tile_order,
                     st.tile_start, st.keys,
                     st.records, st.inst_offset, bg, st.n_contrib, gin.alpha, gin.dL_dcolor, gin.dL_ddepth,
                     gin.dL_dalpha, partial);
}
