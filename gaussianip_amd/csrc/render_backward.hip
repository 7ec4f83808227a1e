// render_backward.hip — segment-parallel backward of the blend + wavefront segmented reduction (gfx950).
//
// Replaces the fork's backward renderCUDA<3> (SURVEY.md §2.1 "bwd 1").  The reference walks every
// tile's list back to front, one thread per pixel, and atomicAdds nine floats per (pixel, Gaussian)
// pair into per-Gaussian accumulators (the contention hot spot of SURVEY.md §8a3); the walk is serial
// per tile, so the longest list sets the kernel time.
//
// MI355X-first formulation:
//   * NO serial walk.  The forward kernel checkpoints the per-pixel blend state (T, C, D) every
//     GIP_SEGMENT = 64 list entries.  With kappa_j = c_j.gC + d_j gD, S_tot = C_tot.gC + D_tot gD and the
//     running prefix Sp_{j+1} = sum_{k<=j} a_k T_k kappa_k, the gradient of entry j is
//         dL/dalpha_j = T_j kappa_j + [Sp_{j+1} - S_tot + T_final (gA - bg.gC)] / (1 - a_j)
//     which is algebraically the reference's back-to-front recurrence (its accum_rec is
//     (C_tot - Cp_{j+1}) / T_{j+1} and its 1 - accum_alpha is T_final / T_{j+1}).  Every (tile, segment)
//     is therefore an independent, equally sized work item; a persistent grid walks the flat segment list.
//   * ONE WAVE per work item, four pixels per lane (one per 8x8 quadrant, as in the forward kernel): no
//     workgroup barrier, no cross-wave LDS traffic, four-way ILP in the per-pixel math.
//   * NO float atomics.  The ten per-pixel terms of an entry are first added over a lane's four pixels
//     in registers, then reduced across the 64 lanes in 27 cross-lane ops (v_permlane32_swap,
//     v_permlane16_swap, DPP row shifts), and ONE row per (tile, Gaussian) instance is stored at row
//     inst_offset[g] + (tile's position in g's rectangle).  gather_backward.hip sums a Gaussian's
//     contiguous rows in fixed order: gradients are bitwise reproducible.
//   * quadrants that the entry's alpha >= 1/255 ellipse misses are skipped with wave-uniform branches
//     (same conservative quadrant mask as the forward kernel).
//
// Row layout (floats): 0,1 M1,M2  2,3,4 M3,M4,M5  5 M0   (moments of Q = dL/dG * G over the pixel offsets dx, dy)
//                      6,7,8 dL/dcolour  9 dL/ddepth  10..15 unused
#include "gip_internal.h"

#ifndef BWD_GRID
#define BWD_GRID 16384 // persistent single-wave workgroups (grid-stride over the segment list)
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
// Cross-lane folds.  v_permlane32_swap(a, b): lanes 0-31 keep a, take b's low half into a's high half ...
// after the swap a = [a_lo | b_lo], b = [a_hi | b_hi], so a + b = [sum of a's halves | sum of b's halves].
// v_permlane16_swap does the same on 16-lane rows (odd rows of a <-> even rows of b).
// Inline asm because the clang builtin mis-selects its second result for float operands on ROCm 7.2 (both extracts
// return the first register); one s_nop pair per GROUP of independent swaps covers the VALU->permlane hazards.
__device__ __forceinline__ void fold32x5(float& a0, float& b0, float& a1, float& b1, float& a2, float& b2, float& a3,
                                         float& b3, float& a4, float& b4) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
               "v_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9\n\ts_nop 1"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3), "+v"(a4), "+v"(b4));
}
__device__ __forceinline__ void fold16x2(float& a0, float& b0, float& a1, float& b1) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
               : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
}
// sum of each 16-lane row, valid in the row's lane 15
__device__ __forceinline__ float row_reduce(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  return v;
}

struct BwdPix {
  float T, Sp, Kc, g0, g1, g2, gd;
  uint32_t lastc;
};

struct BwdAcc { float v0, v1, v2, v3, v4, v5, v6, v7, v8, v9; };

__device__ __forceinline__ bool bwd_pair(BwdPix& p, BwdAcc& a, float pxf, float pyf, const float2 xy, const float4 co,
                                         const float4 cl, uint32_t i) {
  const float dx = xy.x - pxf, dy = xy.y - pyf;
  // conic pre-scaled: log2 domain; the forward's operation sequence (render_forward.hip): a dx and c dy first
  const float power = __builtin_fmaf(dx, __builtin_fmaf(co.y, dy, co.x * dx), (co.z * dy) * dy);
  const float G = __builtin_amdgcn_exp2f(power);
  const float alpha = fminf(GIP_ALPHA_MAX, co.w * G);
  const bool c = i < p.lastc && power <= 0.0f && alpha >= GIP_ALPHA_MIN;
  // predicated (branch-free): a non-contributing pixel adds exact zeros and keeps its state
  const float w = c ? alpha * p.T : 0.f;                  // blend weight of this entry at this pixel
  const float rcp1ma = __builtin_amdgcn_rcpf(1.f - alpha);
  const float kappa = cl.x * p.g0 + cl.y * p.g1 + cl.z * p.g2 + cl.w * p.gd;
  p.Sp += w * kappa;                                      // prefix including this entry
  const float dL_dalpha = c ? p.T * kappa + (p.Sp + p.Kc) * rcp1ma : 0.f;
  p.T = c ? p.T * (1.f - alpha) : p.T;
  a.v6 += w * p.g0; a.v7 += w * p.g1; a.v8 += w * p.g2; a.v9 += w * p.gd;
  // geometric terms as MOMENTS of Q = dL/dG * G over the pixel offsets; the per-Gaussian constants (conic,
  // opacity, 0.5 W/H) are applied once per Gaussian in gather_backward.hip:
  //   dL/dmean2D = -(a M1 + b M2, c M2 + b M1) * 0.5 (W, H),  dL/dconic = -0.5 (M3, M4, M5),  dL/dopacity = M0 / o
  const float Q = co.w * dL_dalpha * G;
  const float Qdx = Q * dx, Qdy = Q * dy;
  a.v5 += Q;
  a.v0 += Qdx; a.v1 += Qdy;
  a.v2 += Qdx * dx; a.v3 += Qdx * dy; a.v4 += Qdy * dy;
  return c;
}

#ifndef BWD_MINW
#define BWD_MINW 1
#endif
__global__ void __launch_bounds__(64, BWD_MINW)
gip_render_backward_kernel(GipKernelParams kp, const GipRasterHeader* __restrict__ header,
                           const uint32_t* __restrict__ seg_tile, const uint32_t* __restrict__ seg_start,
                           const uint32_t* __restrict__ ckpt_start, const float* __restrict__ checkpoints,
                           const uint32_t* __restrict__ tile_start, const unsigned long long* __restrict__ keys,
                           const GipRecord* __restrict__ records, const uint32_t* __restrict__ inst_offset,
                           const float* __restrict__ bg, const uint32_t* __restrict__ n_contrib, const float* __restrict__ final_T,
                           const float* __restrict__ color_out, const float* __restrict__ depth_out,
                           const float* __restrict__ alpha_out, const float* __restrict__ dL_dcolor,
                           const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha_in,
                           float* __restrict__ partial) {
  __shared__ float2 s_xy[64];
  __shared__ float4 s_con[64];
  __shared__ float4 s_col[64];
  __shared__ uint32_t s_row[64];
  __shared__ uint32_t s_mask[64];

  const int lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  const size_t HW = (size_t)kp.H * kp.W;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  // an overflowed forward left truncated buckets behind: nothing downstream is meaningful (the host raises when it
  // reads the header, see rasterizer.py) — leave without touching memory
  if (header->overflow) return;
  uint32_t nseg = header->num_segments;
  if (nseg > kp.seg_capacity) nseg = kp.seg_capacity;

  // XCD-aware work assignment: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share one), each XCD has
  // its own L2, and every segment of a tile re-reads that tile's per-pixel inputs (48 B / pixel) and neighbouring tiles
  // share Gaussian records.  Giving XCD k the CONTIGUOUS eighth [k * per, (k + 1) * per) of the tile-ordered segment
  // list keeps those re-reads inside one L2 (fabric fetches 544 -> ~250 MB per launch; speed only, never correctness).
  const uint32_t per = (nseg + 7u) >> 3, xcd = blockIdx.x & 7u;
  for (uint32_t local = blockIdx.x >> 3; local < per; local += gridDim.x >> 3) {
    const uint32_t seg = xcd * per + local;
    if (seg >= nseg) break;
    const uint32_t vt = seg_tile[seg];
    const uint32_t b = seg - seg_start[vt];              // segment index inside the tile
    const uint32_t start = tile_start[vt];
    uint32_t end = tile_start[vt + 1];
    if (end > kp.capacity) end = kp.capacity;
    if (end < start) end = start;
    const int first = (int)(b * GIP_SEGMENT);            // list positions [first, last) of this segment
    const int last = min((int)(end - start), first + GIP_SEGMENT);
    const uint32_t v = vt / kp.T, tile = vt - v * kp.T;
    const uint32_t tx = tile % kp.tiles_x, ty = tile / kp.tiles_x;
    const int px0 = tx * GIP_TILE + lx, py0 = ty * GIP_TILE + ly;
    const float pxf0 = (float)px0, pyf0 = (float)py0, pxf1 = pxf0 + 8.f, pyf1 = pyf0 + 8.f;
    const float tile_x0 = (float)(tx * GIP_TILE), tile_y0 = (float)(ty * GIP_TILE);
    const GipRecord* recs = records + (size_t)v * kp.P;
    const uint32_t* ioff = inst_offset + (size_t)v * kp.P;

    // per-pixel constants and the blend state at the segment start
    BwdPix p[4];
    uint32_t lmax = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int px = px0 + ((q & 1) << 3), py = py0 + ((q >> 1) << 3);
      p[q].T = 1.f; p[q].Sp = 0.f; p[q].Kc = 0.f; p[q].g0 = p[q].g1 = p[q].g2 = p[q].gd = 0.f; p[q].lastc = 0;
      if (px < kp.W && py < kp.H) {
        const size_t pix1 = (size_t)py * kp.W + px, pix = (size_t)v * HW + pix1;
        p[q].lastc = n_contrib[pix];
        if (p[q].lastc > (uint32_t)first) {
          // The fork's backward re-derives every T_j from T_final := 1 - alpha_out (not from the forward's own
          // final T), i.e. all its T_j carry the factor rho = (1 - alpha_out) / final_T.  Reproduce that.
          const float T_final = 1.f - alpha_out[pix];
          const float Tf_fwd = final_T[pix];
          const float rho = T_final / Tf_fwd;
          float ga = 0.f;
          if (dL_dcolor) {
            const float* gp = dL_dcolor + (size_t)v * 3 * HW + pix1;
            p[q].g0 = gp[0]; p[q].g1 = gp[HW]; p[q].g2 = gp[2 * HW];
          }
          if (dL_ddepth) p[q].gd = dL_ddepth[pix];
          if (dL_dalpha_in) ga = dL_dalpha_in[pix];
          const float* co = color_out + (size_t)v * 3 * HW + pix1;
          // C_tot = colour output minus the background term the forward added
          const float ct0 = co[0] - Tf_fwd * bg0, ct1 = co[HW] - Tf_fwd * bg1, ct2 = co[2 * HW] - Tf_fwd * bg2;
          const float Stot = ct0 * p[q].g0 + ct1 * p[q].g1 + ct2 * p[q].g2 + depth_out[pix] * p[q].gd;
          p[q].Kc = T_final * (ga - (bg0 * p[q].g0 + bg1 * p[q].g1 + bg2 * p[q].g2)) - rho * Stot;
          p[q].T = rho;
          if (b > 0) {
            const uint32_t slot = ckpt_start[vt] + b - 1;
            if (slot < kp.ckpt_capacity) {
              const float* cp = checkpoints + (size_t)slot * (GIP_CKPT_FLOATS * 256) + q * 64 + lane;
              p[q].T = rho * cp[0];
              p[q].Sp = rho * (cp[256] * p[q].g0 + cp[512] * p[q].g1 + cp[768] * p[q].g2 + cp[1024] * p[q].gd);
            }
          }
        }
        lmax = max(lmax, p[q].lastc);
      }
    }
    const int maxc = (int)gip_wave_max_u32(lmax);           // longest replay needed by any pixel of the tile

    // the segment's keys are fetched up front (one load per lane per 64-entry chunk) and the records of
    // chunk c+1 are in flight while chunk c is processed: the dependent key -> record gather is hidden
    uint32_t gk[GIP_SEGMENT / 64];
#pragma unroll
    for (int c = 0; c < GIP_SEGMENT / 64; c++) {
      const int i = first + c * 64 + lane;
      gk[c] = i < last ? (uint32_t)keys[start + i] : 0xffffffffu;
    }
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    uint4 r3 = make_uint4(0, 0, 0, 0);
    uint32_t rio = 0;
    if (gk[0] != 0xffffffffu) {
      const float4* rp = reinterpret_cast<const float4*>(recs + gk[0]);
      r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = reinterpret_cast<const uint4*>(rp)[3];
      rio = ioff[gk[0]];
    }

#pragma unroll
    for (int c = 0; c < GIP_SEGMENT / 64; c++) {
      const int sub = first + c * 64;
      if (sub >= last) break;
      const int cnt = min(64, last - sub);
      uint32_t mask = 0, row = 0xffffffffu;
      __syncthreads();   // single-wave workgroup: previous chunk's LDS reads are done
      if (lane < cnt) {
        const int i = sub + lane;
        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint4 q3 = r3;
        s_xy[lane] = make_float2(q0.x, q0.y);
        s_con[lane] = make_float4(q1.x * -0.72134752044448170f, q1.y * -1.4426950408889634f, q1.z * -0.72134752044448170f, q0.w);
        s_col[lane] = make_float4(q2.x, q2.y, q2.z, q0.z);
        const uint32_t rminx = q3.x & 0xffff, rminy = q3.x >> 16, rmaxx = q3.y & 0xffff, rmaxy = q3.y >> 16;
        row = rio + (uint32_t)gip_rect_rank(q3.w, (int)((rmaxx - rminx) * (rmaxy - rminy)),
                                            (int)((ty - rminy) * (rmaxx - rminx) + (tx - rminx)));
        if (row >= kp.capacity) row = 0xffffffffu;
        mask = 0xf;
        const float t2 = 2.0f * __logf(255.0f * q0.w) + 0.02f;
        const float det = q1.x * q1.z - q1.y * q1.y;
        if (t2 <= 0.f || i >= maxc) {
          mask = 0;
        } else if (det > 0.f) {
          const float inv = t2 / det;
          const float hx = sqrtf(inv * q1.z) * 1.01f + 0.05f, hy = sqrtf(inv * q1.x) * 1.01f + 0.05f;
          const float rx = q0.x - tile_x0, ry = q0.y - tile_y0;
          const bool xl = rx - hx <= 7.f, xr = rx + hx >= 8.f, yt = ry - hy <= 7.f, yb = ry + hy >= 8.f;
          // distance bound for the quadrants diagonal to the centre (render_forward.hip): q >= lambda_min |d|^2
          const float hd = 0.5f * (q1.x - q1.z);
          const float lmax = 0.5f * (q1.x + q1.z) + sqrtf(__builtin_fmaf(hd, hd, q1.y * q1.y));
          const float rad = sqrtf(inv * lmax) * 1.01f + 0.05f;
          const float rad2 = rad * rad;
          const float ex0 = fmaxf(fmaxf(-rx, rx - 7.f), 0.f), ex1 = fmaxf(fmaxf(8.f - rx, rx - 15.f), 0.f);
          const float ey0 = fmaxf(fmaxf(-ry, ry - 7.f), 0.f), ey1 = fmaxf(fmaxf(8.f - ry, ry - 15.f), 0.f);
          const float ex0s = ex0 * ex0, ex1s = ex1 * ex1, ey0s = ey0 * ey0, ey1s = ey1 * ey1;
          mask = (xl && yt && ex0s + ey0s <= rad2 ? 1u : 0u) | (xr && yt && ex1s + ey0s <= rad2 ? 2u : 0u) |
                 (xl && yb && ex0s + ey1s <= rad2 ? 4u : 0u) | (xr && yb && ex1s + ey1s <= rad2 ? 8u : 0u);
        }
        s_mask[lane] = mask;
        s_row[lane] = row;
        // entries no pixel can contribute to: the lane that staged the entry zeroes its row right away
        if (mask == 0 && row != 0xffffffffu) {
          float4* dst = reinterpret_cast<float4*>(partial + (size_t)row * GIP_PARTIAL_FLOATS);
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[0] = z; dst[1] = z; dst[2] = z;
        }
      }
      if (c + 1 < GIP_SEGMENT / 64) {
        const uint32_t gn = gk[c + 1 < GIP_SEGMENT / 64 ? c + 1 : 0];
        if (gn != 0xffffffffu) {
          const float4* rp = reinterpret_cast<const float4*>(recs + gn);
          r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = reinterpret_cast<const uint4*>(rp)[3];
          rio = ioff[gn];
        }
      }
      __syncthreads();
      unsigned long long m = __ballot(mask != 0);
      while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t mk = __builtin_amdgcn_readfirstlane(s_mask[j]);
        const uint32_t rowj = __builtin_amdgcn_readfirstlane(s_row[j]);
        const uint32_t i = (uint32_t)(sub + j);
        const float2 xy = s_xy[j];
        const float4 co = s_con[j];
        const float4 cl = s_col[j];
        BwdAcc a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bool c = false;
        if (mk & 1u) c |= bwd_pair(p[0], a, pxf0, pyf0, xy, co, cl, i);
        if (mk & 2u) c |= bwd_pair(p[1], a, pxf1, pyf0, xy, co, cl, i);
        if (mk & 4u) c |= bwd_pair(p[2], a, pxf0, pyf1, xy, co, cl, i);
        if (mk & 8u) c |= bwd_pair(p[3], a, pxf1, pyf1, xy, co, cl, i);
        if (rowj == 0xffffffffu) continue;
        float* dst = partial + (size_t)rowj * GIP_PARTIAL_FLOATS;
        if (!__any(c)) {
          if (lane < 3) reinterpret_cast<float4*>(dst)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
          continue;
        }
        // 64-lane sums of the ten terms in 27 cross-lane ops
        fold32x5(a.v0, a.v2, a.v1, a.v3, a.v4, a.v6, a.v5, a.v7, a.v8, a.v9);
        float p02 = a.v0 + a.v2, p13 = a.v1 + a.v3;   // [v0|v2], [v1|v3]
        float p46 = a.v4 + a.v6, p57 = a.v5 + a.v7;
        const float p89 = a.v8 + a.v9;
        fold16x2(p02, p13, p46, p57);
        float qa = row_reduce(p02 + p13);   // rows: v0, v1, v2, v3
        float qb = row_reduce(p46 + p57);   // rows: v4, v5, v6, v7
        float qc = row_reduce(p89);         // rows 0,1: v8 ; rows 2,3: v9
        qc = dpp_add<0x142, 0xa>(qc);       // row_bcast:15 -> lane 31 = v8, lane 63 = v9
        if ((lane & 15) == 15) {
          const int r = lane >> 4;
          dst[r] = qa;
          dst[4 + r] = qb;
          if (r & 1) dst[8 + (r >> 1)] = qc;
        }
      }
    }
  }
}

void gip_launch_render_backward(const GipKernelParams& kp, const float* bg, GipStatePtrs st, const GipRasterGradsIn& gin,
                                float* partial, hipStream_t s) {
  hipLaunchKernelGGL(gip_render_backward_kernel, dim3(BWD_GRID), dim3(64), 0, s, kp, st.header, st.seg_tile,
                     st.seg_start, st.ckpt_start, st.checkpoints, st.tile_start, st.keys, st.records, st.inst_offset, bg,
                     st.n_contrib, st.final_T, gin.color, gin.depth, gin.alpha, gin.dL_dcolor, gin.dL_ddepth,
                     gin.dL_dalpha, partial);
}
