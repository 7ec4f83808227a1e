// render_forward.hip — per-tile front-to-back alpha compositing (colour, depth, alpha) for gfx950.
//
// Replaces the fork's forward renderCUDA<3> (SURVEY.md §2.1 "fwd 6"); outputs are the 4-tuple the
// reference unpacks at gaussian_renderer/__init__.py:85: color [3,H,W], depth [1,H,W] = sum z a T,
// alpha [1,H,W] = sum a T, plus n_contrib for the backward replay.
//
// Structure (wave64-first, not a 16x16 thread block recompiled):
//   * one workgroup = 8 waves per 16x16 tile; wave w owns the 8x4 pixel block (w&1, w>>1) and its 64 lanes are
//     32 pixels x 2 LIST ENTRIES: lanes 0-31 evaluate the next surviving entry of the tile's list, lanes 32-63 the
//     one after it, for the same 32 pixels.  The two alphas are exchanged with one v_permlane32_swap and both lane
//     halves then apply the reference's sequential rule (entry e, then entry o: T, the 1e-4 stop test, the weights)
//     to identical values, so the result is the sequential blend while a wave retires two entries per trip through
//     its instruction stream.  The kernel's duration is set by the longest lists walked by a lone wave (measured:
//     ~230 cycles of issue per entry), so this halves the critical path; the finer 8x4 blocks also cull more;
//   * workgroups are launched longest-list-first (tile_order from the scan kernel);
//   * the tile's depth-sorted list is consumed in batches of 512: each lane fetches one key, gathers that
//     Gaussian's 64-byte record with dwordx4 loads, stages the 40 bytes the blend needs into LDS (conic pre-scaled
//     into the exp2 domain), and computes an 8-bit BLOCK MASK from the tight screen-space extent of the
//     alpha >= 1/255 ellipse (|dx| <= sqrt(2 ln(255 o) cov_xx), same for y; and distance <= sqrt(t / lambda_min)).  A pair outside that ellipse fails the
//     reference's alpha < 1/255 test, so skipping it cannot change any output;
//   * each wave ballots the mask bits of its block and walks only the set bits with scalar find-first-one, two per
//     step, FWD_PAIRS steps per trip (independent exp / alpha chains), and leaves as soon as all its pixels terminated;
//   * at every GIP_SEGMENT-entry boundary the per-pixel blend state (T, C, D) is checkpointed (20 B / pixel) so
//     that the backward pass needs no sequential walk over the tile.
// (A two-pass segment-parallel forward — local blend of every segment + per-tile combine with exact replay of the
//  segment a pixel terminates in — was built and measured in round 1: 0.44 ms vs 0.40 ms for the then-current kernel
//  at 100k Gaussians, because the replays of terminating pixels cost what the flat pass saves)
#include <stdlib.h>

#include "gip_internal.h"

#ifndef FWD_PAIRS
#define FWD_PAIRS 4
#endif
#define FWD_THREADS 512
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// value held by lane ^ 32 (v_permlane32_swap: a <- [a_lo | b_lo], b <- [a_hi | b_hi]; inline asm, see render_backward.hip)
__device__ __forceinline__ float xchg32(float x, int hh) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return hh ? a : b;
}

// both halves' values of x: after v_permlane32_swap(a, b) with a = b = x, a holds the LOW half's value in every lane and
// b the HIGH half's (a <- [a_lo | b_lo], b <- [a_hi | b_hi]) — no per-lane select needed
__device__ __forceinline__ void both32(float x, float& lo, float& hi) {
  lo = x; hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}

// index of the lowest set bit of the wave-uniform mask m (-1 when m is empty), cleared from m: s_ff1_i32_b64 + s_bitset0_b64
__device__ __forceinline__ int ffs_clear(unsigned long long& m) {
  int j;
  asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(m));
  return j;
}

__global__ void __launch_bounds__(FWD_THREADS)
gip_render_forward_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_order,
                          const uint32_t* __restrict__ tile_start, const unsigned long long* __restrict__ keys,
                          const GipRecord* __restrict__ records, const float* __restrict__ bg,
                          float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
                          uint32_t* __restrict__ n_contrib, float* __restrict__ final_T, const uint32_t* __restrict__ ckpt_start,
                          float* __restrict__ checkpoints) {
  // Launch position -> place in the longest-first order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8
  // share one, each XCD has its own L2), and inside a size class the order is by tile index — so consecutive positions are
  // neighbouring tiles, which share most of their Gaussians, and land on 8 different L2s: every XCD fetches its own copy of
  // the records (fabric traffic 1.87x the algorithmic bytes, profiles/r03_pmc_summary.txt).  An XCD-contiguous order inside
  // each size class was measured 20 % slower in round 4 (the long lists of a class are spatial neighbours, so contiguous
  // chunks hand some XCDs all of them): tools/experiments/render_forward_xcd_order.txt.
  const uint32_t pos = blockIdx.x;
  const uint32_t vt = tile_order[pos];   // view * T + tile
  const uint32_t v = vt / kp.T, tile = vt - v * kp.T;
  const uint32_t tx = tile % kp.tiles_x, ty = tile / kp.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // wave 0..7 = 8x4 pixel block (wave & 1, wave >> 1)
  const int hh = lane >> 5, pl = lane & 31;                        // list-entry parity, pixel inside the block
  const int lx = ((wave & 1) << 3) | (pl & 7), ly = ((wave >> 1) << 2) | (pl >> 3);
  const int px = tx * GIP_TILE + lx, py = ty * GIP_TILE + ly;
  const bool inside = px < kp.W && py < kp.H;
  const v2f pxy = {(float)px, (float)py};
  const float tile_x0 = (float)(tx * GIP_TILE), tile_y0 = (float)(ty * GIP_TILE);
  // slot of this pixel in the checkpoint rows: the backward's layout (8x8 quadrant * 64 + row * 8 + column)
  const int cpix = ((((ly >> 3) << 1) | (lx >> 3)) << 6) | ((ly & 7) << 3) | (lx & 7);

  const uint32_t start = tile_start[vt];
  uint32_t end = tile_start[vt + 1];
  if (end > kp.capacity) end = kp.capacity;
  if (end < start) end = start;
  const GipRecord* recs = records + (size_t)v * kp.P;

  // Every 64-entry chunk of the staged arrays is preceded by a NULL entry (opacity 0 => alpha 0 => skipped by the
  // alpha >= 1/255 test): chunk c lives at slots c * 65 + 1 .. c * 65 + 64, its NULL entry at c * 65.  The scalar walk
  // takes j = find-first-one(m) - which is -1 once the mask is empty - and slot (c * 65 + 1 + j) is then the NULL entry:
  // a pair whose second (or first) entry does not exist needs neither a select nor per-lane "have" flags
  // Field order chosen for PACKED fp32 math inside one (pixel, entry) evaluation — a gfx950 SIMD retires v_pk_mul /
  // v_pk_add / v_pk_fma (two fp32 per lane) at the rate of one scalar fp32 instruction: (x, y) - (px, py) is one
  // v_pk_add, (a dx, c dy) one v_pk_mul on the adjacent (a, c) pair, the colour / depth sums two v_pk_fma on (r, g), (b, d)
  constexpr int FWD_SLOTS = (FWD_THREADS / 64) * 65;
  __shared__ v2f s_xy[FWD_SLOTS];
  __shared__ v4f s_con[FWD_SLOTS];   // conic a, c, b (exp2 domain) + opacity
  __shared__ v4f s_col[FWD_SLOTS];   // r, g, b + depth
  __shared__ uint32_t s_mask[FWD_THREADS];
  if (threadIdx.x < FWD_THREADS / 64) {
    s_xy[threadIdx.x * 65] = (v2f){0.f, 0.f};
    s_con[threadIdx.x * 65] = (v4f){0.f, 0.f, 0.f, 0.f};
    s_col[threadIdx.x * 65] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
  const int sidx = threadIdx.x + (threadIdx.x >> 6) + 1;      // staging slot of this thread's entry

  bool done = !inside;
  float T = 1.0f, Wt = 0.f;
  v2f CG = {0.f, 0.f}, BD = {0.f, 0.f};   // (red, green), (blue, depth): this lane half's share of the sums
  uint32_t last_contributor = 0;
  int last_off = -1;      // byte offset (x16) of the last contributing entry inside the current batch, -1 = none yet

  // software pipeline over the batches: the key of batch b+2 and the record of batch b+1 are in flight
  // while batch b is blended, so the dependent key -> record gather never stalls the blend loop
  uint32_t g_next = 0xffffffffu, g_next2 = 0xffffffffu;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
  if (start + threadIdx.x < end) g_next = (uint32_t)keys[start + threadIdx.x];
  if (start + FWD_THREADS + threadIdx.x < end) g_next2 = (uint32_t)keys[start + FWD_THREADS + threadIdx.x];
  if (g_next != 0xffffffffu) {
    const float4* rp = reinterpret_cast<const float4*>(recs + g_next);
    r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
  }

  for (uint32_t base = start; base < end; base += FWD_THREADS) {
    if (__syncthreads_count(done) == FWD_THREADS) break;
    const uint32_t k = base + threadIdx.x;
    if (k < end) {
      const float4 q0 = r0, q1 = r1, q2 = r2;
      s_xy[sidx] = (v2f){q0.x, q0.y};
      // conic pre-scaled into the exp2 domain: power * log2(e) = A dx^2 + B dx dy + C dy^2 (sign tests are unchanged)
      s_con[sidx] = (v4f){q1.x * -0.72134752044448170f, q1.z * -0.72134752044448170f, q1.y * -1.4426950408889634f, q0.w};
      s_col[sidx] = (v4f){q2.x, q2.y, q2.z, q0.z};
      // extent of { power >= -ln(255 o) }  (conservative: +1% / +0.05 px)
      uint32_t mask = 0xff;
      const float t2 = 2.0f * __logf(255.0f * q0.w) + 0.02f;
      const float det = q1.x * q1.z - q1.y * q1.y;
      if (t2 <= 0.f) {
        mask = 0;
      } else if (det > 0.f) {
        const float inv = t2 / det;
        const float hx = sqrtf(inv * q1.z) * 1.01f + 0.05f, hy = sqrtf(inv * q1.x) * 1.01f + 0.05f;
        const float rx = q0.x - tile_x0, ry = q0.y - tile_y0;   // centre relative to the tile origin
        const uint32_t cols = (rx - hx <= 7.f ? 1u : 0u) | (rx + hx >= 8.f ? 2u : 0u);      // columns 0..7 / 8..15
        const float ylo = ry - hy, yhi = ry + hy;
        // second bound, for the blocks diagonal to the centre (the extent test keeps the whole bounding box of the
        // ellipse): the quadratic form is >= lambda_min |d|^2, so a block farther from the centre than
        // sqrt(t / lambda_min) lies outside the ellipse.  lambda_min = det / lambda_max (no cancellation).
        const float hd = 0.5f * (q1.x - q1.z);
        const float lmax = 0.5f * (q1.x + q1.z) + sqrtf(__builtin_fmaf(hd, hd, q1.y * q1.y));
        const float rad = sqrtf(inv * lmax) * 1.01f + 0.05f;
        const float rad2 = rad * rad;
        const float ex0 = fmaxf(fmaxf(-rx, rx - 7.f), 0.f), ex1 = fmaxf(fmaxf(8.f - rx, rx - 15.f), 0.f);
        const float ex0s = ex0 * ex0, ex1s = ex1 * ex1;
        mask = 0;
#pragma unroll
        for (int band = 0; band < 4; band++) {                                               // rows 4 band .. 4 band + 3
          const float ey = fmaxf(fmaxf((float)(4 * band) - ry, ry - (float)(4 * band + 3)), 0.f);
          const float eys = ey * ey;
          const uint32_t near = (ex0s + eys <= rad2 ? 1u : 0u) | (ex1s + eys <= rad2 ? 2u : 0u);
          if (ylo <= (float)(4 * band + 3) && yhi >= (float)(4 * band)) mask |= (cols & near) << (2 * band);
        }
      }
      s_mask[threadIdx.x] = mask;
    }
    // issue the loads of the following batches (consumed one / two iterations from now)
    g_next = g_next2;
    g_next2 = (k + 2 * FWD_THREADS < end) ? (uint32_t)keys[k + 2 * FWD_THREADS] : 0xffffffffu;
    if (g_next != 0xffffffffu) {
      const float4* rp = reinterpret_cast<const float4*>(recs + g_next);
      r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
    }
    __syncthreads();
    const int cnt = min((uint32_t)FWD_THREADS, end - base);
    if (!__all(done)) {
      for (int c0 = 0; c0 < cnt; c0 += 64) {
        const uint32_t rel = (base - start) + c0;
        if (rel != 0 && (rel % GIP_SEGMENT) == 0) {
          // blend state at the start of segment rel / GIP_SEGMENT (>= 1): lets the backward treat every segment of
          // this tile as an independent work item (render_backward.hip).  The two lane halves hold partial sums.
          const float s0 = CG.x + xchg32(CG.x, hh), s1 = CG.y + xchg32(CG.y, hh), s2 = BD.x + xchg32(BD.x, hh), sd = BD.y + xchg32(BD.y, hh);
          const uint32_t slot = ckpt_start[vt] + rel / GIP_SEGMENT - 1;
          // a pixel that has already terminated never reads this row back: the backward only loads the checkpoint of a segment
          // that starts before the pixel's last contributor (render_backward.hip: n_contrib > first), and a terminated pixel's
          // last contributor lies in front of this boundary — its stores are skipped (fewer bytes on opaque regions)
          if (slot < kp.ckpt_capacity && hh == 0 && !kp.forward_only && !done) {
            float* cp = checkpoints + (size_t)slot * (GIP_CKPT_FLOATS * GIP_BLOCK) + cpix;
            cp[0] = T; cp[GIP_BLOCK] = s0; cp[2 * GIP_BLOCK] = s1; cp[3 * GIP_BLOCK] = s2; cp[4 * GIP_BLOCK] = sd;
          }
        }
        const int e = c0 + lane;
        const uint32_t mk = e < cnt ? s_mask[e] : 0u;
        unsigned long long m = __ballot((mk >> wave) & 1u);
        const int cbase16 = ((c0 >> 6) * 65 + 1) << 4;
        bool wave_done = false;
        while (m) {
          // FWD_PAIRS pairs of list entries per trip: lanes 0-31 take the first entry of a pair, lanes 32-63 the second
          int ja[FWD_PAIRS];                                  // byte offset (x16) of this lane half's entry in the staged arrays
          float al[FWD_PAIRS];
          v4f cc[FWD_PAIRS];
#pragma unroll
          for (int u = 0; u < FWD_PAIRS; u++) {
            // scalar side: byte offsets of the pair's two entries; find-first-one of an empty mask is -1 = the NULL slot
            const int j1 = ffs_clear(m), j2 = ffs_clear(m);
            const int a1 = cbase16 + (j1 << 4), a2 = cbase16 + (j2 << 4);
            ja[u] = hh ? a2 : a1;
            const v2f d = *(const v2f*)((const char*)s_xy + (ja[u] >> 1)) - pxy;        // (dx, dy)
            const v4f co = *(const v4f*)((const char*)s_con + ja[u]);
            cc[u] = *(const v4f*)((const char*)s_col + ja[u]);
            const v2f t = co.xy * d;                                       // (a dx, c dy)
            // log2 domain; same operation sequence as the backward's re-evaluation (render_backward.hip: bwd_pair)
            const float power = __builtin_fmaf(d.x, __builtin_fmaf(co.z, d.y, t.x), t.y * d.y);
            const float a = fminf(GIP_ALPHA_MAX, co.w * __builtin_amdgcn_exp2f(power));
            al[u] = (power <= 0.0f && a >= GIP_ALPHA_MIN) ? a : 0.f;              // 0 = the reference skips this pair
          }
          const unsigned long long done_before = __ballot(done);
#pragma unroll
          for (int u = 0; u < FWD_PAIRS; u++) {
            float a_e, a_o;                                                      // first / second entry of the pair
            both32(al[u], a_e, a_o);
            // the reference's per-entry rule, applied to both entries in order by both lane halves; a terminated pixel
            // sees alpha 0, which leaves T and the sums unchanged.  T (1 - a) is evaluated as fma(-a, T, T).
            a_e = done ? 0.f : a_e;
            const float t1 = __builtin_fmaf(-a_e, T, T);
            const bool stop1 = t1 < GIP_T_MIN;                                   // only reachable with a_e > 0
            const float T1 = stop1 ? T : t1;
            a_o = (done || stop1) ? 0.f : a_o;
            const float t2 = __builtin_fmaf(-a_o, T1, T1);
            const bool stop2 = t2 < GIP_T_MIN;
            const float w_e = stop1 ? 0.f : a_e * T, w_o = stop2 ? 0.f : a_o * T1;
            const float w = hh ? w_o : w_e;
            CG = __builtin_elementwise_fma(cc[u].xy, (v2f){w, w}, CG);
            BD = __builtin_elementwise_fma(cc[u].zw, (v2f){w, w}, BD);
            Wt += w;
            last_off = w > 0.f ? ja[u] : last_off;                               // resolved to a list index once per batch
            T = stop2 ? T1 : t2;
            done = done || stop1 || stop2;
          }
          if (__ballot(done) != done_before) {     // wave-uniform; re-test termination only when something changed
            if (__all(done)) { wave_done = true; break; }
          }
        }
        if (wave_done) break;
      }
    }
    if (last_off >= 0) {                                // n_contrib = 1-based list position of the last contributor
      const uint32_t slot = (uint32_t)last_off >> 4;    // slot = chunk * 65 + 1 + position: list index = slot - chunk - 1
      last_contributor = (base - start) + slot - slot / 65u;
      last_off = -1;
    }
  }

  // the two lane halves hold disjoint shares of the sums and the same T
  float C0 = CG.x, C1 = CG.y, C2 = BD.x, Dp = BD.y;
  C0 += xchg32(C0, hh); C1 += xchg32(C1, hh); C2 += xchg32(C2, hh); Wt += xchg32(Wt, hh); Dp += xchg32(Dp, hh);
  const uint32_t lc_other = __float_as_uint(xchg32(__uint_as_float(last_contributor), hh));
  last_contributor = max(last_contributor, lc_other);
  if (inside && hh == 0) {
    const size_t HW = (size_t)kp.H * kp.W;
    const size_t pix = (size_t)py * kp.W + px;
    float* oc = out_color + (size_t)v * 3 * HW;
    oc[pix] = C0 + T * bg[0];
    oc[HW + pix] = C1 + T * bg[1];
    oc[2 * HW + pix] = C2 + T * bg[2];
    out_depth[(size_t)v * HW + pix] = Dp;
    out_alpha[(size_t)v * HW + pix] = Wt;
    if (!kp.forward_only) {          // the backward's inputs; a forward-only render (GipRasterConfig::forward_only) skips them
      n_contrib[(size_t)v * HW + pix] = last_contributor;
      final_T[(size_t)v * HW + pix] = T;
    }
  }
}


void gip_launch_render_forward(const GipKernelParams& kp, const float* bg, GipStatePtrs st, float* color, float* depth,
                               float* alpha, hipStream_t s) {
  hipLaunchKernelGGL(gip_render_forward_kernel, dim3(kp.V * kp.T), dim3(FWD_THREADS), 0, s, kp, st.tile_order,
                     st.tile_start, st.keys, st.records, bg, color, depth, alpha, st.n_contrib, st.final_T, st.ckpt_start,
                     st.checkpoints);
}
