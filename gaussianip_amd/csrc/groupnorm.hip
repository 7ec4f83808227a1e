// groupnorm.hip — fused GroupNorm(+SiLU) forward / backward for channels-last fp16 tensors on gfx950.
//
// See include/gip_nn.h.  Memory-bound: one statistics pass (read x) and one apply pass (read x, write y) with 16-byte
// (8 x half) accesses per lane, fp32 accumulation.  A lane owns one fixed 8-channel chunk and walks rows, so its eight
// per-channel accumulators live in registers; channel sums are folded into group sums through LDS once per workgroup
// and the per-split partials are reduced by the apply kernel's prologue (no atomics on global memory, deterministic).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gip_nn.h"

#define GN_BLOCK 256
#define GN_MAX_SPLITS 128

struct alignas(16) half8 { __half2 a, b, c, d; };

__device__ __forceinline__ void unpack8(const half8& h, float* f) {
  const float2 x = __half22float2(h.a), y = __half22float2(h.b), z = __half22float2(h.c), w = __half22float2(h.d);
  f[0] = x.x; f[1] = x.y; f[2] = y.x; f[3] = y.y; f[4] = z.x; f[5] = z.y; f[6] = w.x; f[7] = w.y;
}
__device__ __forceinline__ half8 pack8(const float* f) {
  half8 h;
  h.a = __floats2half2_rn(f[0], f[1]); h.b = __floats2half2_rn(f[2], f[3]);
  h.c = __floats2half2_rn(f[4], f[5]); h.d = __floats2half2_rn(f[6], f[7]);
  return h;
}
// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp): an IEEE `/` compiles to v_div_scale x2 + v_rcp + four fma + v_div_fmas +
// v_div_fixup — the apply kernel carried 264 of those instructions for its 64 divisions per trip, half of its vector work
// (round 4: the forward apply pass of a 268 MB tensor took 0.129 ms against 0.097 ms for a plain copy); the result is rounded
// to half anyway
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float silu_f(float v) { return v * sigmoid_f(v); }
__device__ __forceinline__ float dsilu_f(float v) { const float s = sigmoid_f(v); return s * (1.f + v * (1.f - s)); }

// geometry shared by all kernels: lanes are laid out as (chunk, row-lane); tpr = C / 8 chunks per row
struct GnGeom { int tpr, chunks_per_thread, rows_per_iter; };
__device__ __forceinline__ GnGeom geom(int C) {
  GnGeom g;
  g.tpr = C >> 3;
  g.chunks_per_thread = (g.tpr + GN_BLOCK - 1) / GN_BLOCK;                 // > 1 only for C > 2048
  const int lanes_x = (g.tpr + g.chunks_per_thread - 1) / g.chunks_per_thread;
  g.rows_per_iter = max(1, GN_BLOCK / lanes_x);
  return g;
}

// MODE 0: sum(x), sum(x^2)                         -> forward statistics
// MODE 1: sum(dxh), sum(dxh * xh), dxh = dy * dsilu?(gamma xh + beta) * gamma   -> backward reductions
template <int MODE>
__global__ void __launch_bounds__(GN_BLOCK)
gn_reduce_kernel(const half8* __restrict__ x, const half8* __restrict__ dy, const __half* __restrict__ gamma,
                 const __half* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                 float* __restrict__ partial /* [N, splits, G, 2] */, long long HW, int C, int G, int splits, int silu,
                 const __half* __restrict__ addend, int addend_stride) {
  extern __shared__ float s_ch[];   // [rows_per_iter][C][2]: one slot per (row-lane, channel) -> fixed-order sums
  const int n = blockIdx.y, split = blockIdx.x;
  const GnGeom gm = geom(C);
  const int lanes_x = (gm.tpr + gm.chunks_per_thread - 1) / gm.chunks_per_thread;
  const int cx = threadIdx.x % lanes_x, ry = threadIdx.x / lanes_x;
  const bool active = ry < gm.rows_per_iter;
  const long long rows_per_split = (HW + splits - 1) / splits;
  const long long r0 = (long long)split * rows_per_split, r1 = min(HW, r0 + rows_per_split);
  const int cg = C / G;
  for (int i = threadIdx.x; i < 2 * C * gm.rows_per_iter; i += GN_BLOCK) s_ch[i] = 0.f;
  __syncthreads();
  for (int k = 0; k < gm.chunks_per_thread; k++) {
    const int chunk = cx + k * lanes_x;
    if (!active || chunk >= gm.tpr) continue;
    float a0[8], a1[8], ga[8], be[8], mu[8], rs[8], ad[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      a0[j] = 0.f; a1[j] = 0.f;
      ad[j] = addend ? __half2float(addend[(long long)n * addend_stride + chunk * 8 + j]) : 0.f;
    }
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int c = chunk * 8 + j;
        ga[j] = __half2float(gamma[c]); be[j] = __half2float(beta[c]);
        mu[j] = mean[n * G + c / cg]; rs[j] = rstd[n * G + c / cg];
      }
    }
    const half8* xp = x + ((long long)n * HW) * gm.tpr + chunk;
    const half8* dp = MODE == 1 ? dy + ((long long)n * HW) * gm.tpr + chunk : nullptr;
#pragma unroll 4
    for (long long r = r0 + ry; r < r1; r += gm.rows_per_iter) {
      float v[8];
      unpack8(xp[r * gm.tpr], v);
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] += ad[j];
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) { a0[j] += v[j]; a1[j] += v[j] * v[j]; }
      } else {
        float d[8];
        unpack8(dp[r * gm.tpr], d);
        if (silu) {        // (wave-uniform test outside the element loop, as in the apply kernel)
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float dxh = d[j] * dsilu_f(ga[j] * xh + be[j]) * ga[j];
            a0[j] += dxh; a1[j] += dxh * xh;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float dxh = d[j] * ga[j];
            a0[j] += dxh; a1[j] += dxh * xh;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      s_ch[2 * ((size_t)ry * C + chunk * 8 + j)] = a0[j];
      s_ch[2 * ((size_t)ry * C + chunk * 8 + j) + 1] = a1[j];
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += GN_BLOCK) {
    float s0 = 0.f, s1 = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; c++)
      for (int r = 0; r < gm.rows_per_iter; r++) { s0 += s_ch[2 * ((size_t)r * C + c)]; s1 += s_ch[2 * ((size_t)r * C + c) + 1]; }
    float* o = partial + (((long long)n * splits + split) * G + g) * 2;
    o[0] = s0; o[1] = s1;
  }
}

// apply: MODE 0 forward (writes y, and mean / rstd once per sample), MODE 1 backward (writes dx)
template <int MODE>
__global__ void __launch_bounds__(GN_BLOCK)
gn_apply_kernel(const half8* __restrict__ x, const half8* __restrict__ dy, const __half* __restrict__ gamma,
                const __half* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd,
                const float* __restrict__ partial, half8* out, long long HW, int C, int G, int splits,
                int out_splits, float eps, int silu, const __half* __restrict__ addend, int addend_stride,
                const half8* accum) {          // NOT __restrict__: gip_nn.h allows dx (= out) to alias accum
  extern __shared__ float s_g[];   // [G][2]: forward mean, rstd ; backward S1/m, S2/m
  const int n = blockIdx.y, split = blockIdx.x;
  const GnGeom gm = geom(C);
  const int lanes_x = (gm.tpr + gm.chunks_per_thread - 1) / gm.chunks_per_thread;
  const int cx = threadIdx.x % lanes_x, ry = threadIdx.x / lanes_x;
  const bool active = ry < gm.rows_per_iter;
  const int cg = C / G;
  const float inv_m = 1.f / ((float)HW * (float)cg);
  // per-group totals from the per-split partials: all 256 threads take part (group = t % G, split slice = t / G),
  // fixed summation order.  splits == 0 (forward): mean / rstd were made by gn_finalize_stats_kernel, nothing to reduce
  if (MODE == 0 && splits == 0) {
    for (int gg = threadIdx.x; gg < G; gg += GN_BLOCK) { s_g[2 * gg] = mean[n * G + gg]; s_g[2 * gg + 1] = rstd[n * G + gg]; }
  } else if (MODE == 1 && splits == 0) {      // S1 / m, S2 / m made by gn_finalize_bwd_kernel from the data-gradient kernel's sums
    for (int gg = threadIdx.x; gg < G; gg += GN_BLOCK) { s_g[2 * gg] = partial[((long long)n * G + gg) * 2]; s_g[2 * gg + 1] = partial[((long long)n * G + gg) * 2 + 1]; }
  } else {
    float* s_part = s_g + 2 * G;                      // [slices][G][2]
    const int slices = GN_BLOCK / G > 0 ? GN_BLOCK / G : 1;
    const int g = threadIdx.x % G, sl = threadIdx.x / G;
    if (sl < slices) {
      float s0 = 0.f, s1 = 0.f;
      for (int s = sl; s < splits; s += slices) {
        const float* p = partial + (((long long)n * splits + s) * G + g) * 2;
        s0 += p[0]; s1 += p[1];
      }
      s_part[(sl * G + g) * 2] = s0; s_part[(sl * G + g) * 2 + 1] = s1;
    }
    __syncthreads();
    for (int gg = threadIdx.x; gg < G; gg += GN_BLOCK) {
      float s0 = 0.f, s1 = 0.f;
      for (int k = 0; k < slices; k++) { s0 += s_part[(k * G + gg) * 2]; s1 += s_part[(k * G + gg) * 2 + 1]; }
      if (MODE == 0) {
        const float mu = s0 * inv_m;
        const float var = fmaxf(s1 * inv_m - mu * mu, 0.f);
        const float rs = rsqrtf(var + eps);
        s_g[2 * gg] = mu; s_g[2 * gg + 1] = rs;
        if (split == 0) { mean[n * G + gg] = mu; rstd[n * G + gg] = rs; }
      } else {
        s_g[2 * gg] = s0 * inv_m; s_g[2 * gg + 1] = s1 * inv_m;
      }
    }
  }
  __syncthreads();
  const long long rows_per_split = (HW + out_splits - 1) / out_splits;
  const long long r0 = (long long)split * rows_per_split, r1 = min(HW, r0 + rows_per_split);
  for (int k = 0; k < gm.chunks_per_thread; k++) {
    const int chunk = cx + k * lanes_x;
    if (!active || chunk >= gm.tpr) continue;
    float sc[8], sh[8], ga[8], be[8], mu[8], rs[8], m1[8], m2[8], ad[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = chunk * 8 + j, g = c / cg;
      ga[j] = __half2float(gamma[c]); be[j] = __half2float(beta[c]);
      ad[j] = addend ? __half2float(addend[(long long)n * addend_stride + c]) : 0.f;
      if (MODE == 0) {
        sc[j] = s_g[2 * g + 1] * ga[j];
        sh[j] = be[j] - (s_g[2 * g] - ad[j]) * sc[j];      // sc * (x + ad - mean) + beta
      } else {
        mu[j] = mean[n * G + g]; rs[j] = rstd[n * G + g];
        m1[j] = s_g[2 * g]; m2[j] = s_g[2 * g + 1];
      }
    }
    const long long base = ((long long)n * HW) * gm.tpr + chunk;
    // U rows per trip: ALL of a trip's loads are issued before its first store.  `out` may alias `accum` (and the compiler must
    // assume it may alias x / dy), so a load-compute-store loop keeps one row's loads in flight per thread; batching them is
    // what lets the memory system see U x (1..3) x 16 B per lane (measured on the 268 MB tensors of the VAE's first level:
    // apply-only 0.129 ms against 0.097 ms for a plain copy before this).  In-place use stays correct: a thread reads exactly
    // the chunks it writes, and reads them first.
    constexpr int U = MODE == 0 ? 8 : 4;
    const long long stride = (long long)gm.rows_per_iter * gm.tpr;
    const long long step = (long long)U * gm.rows_per_iter;
    // one trip = U rows of this thread's chunk: load(), then finish() (math + store)
    auto load = [&](long long r, half8 (&xv)[U], half8 (&dv)[MODE == 1 ? U : 1], half8 (&av)[MODE == 1 ? U : 1]) {
      const long long off0 = base + r * gm.tpr;
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (r + (long long)u * gm.rows_per_iter < r1) {
          xv[u] = x[off0 + u * stride];
          if (MODE == 1) {
            dv[u] = dy[off0 + u * stride];
            if (accum) av[u] = accum[off0 + u * stride];
          }
        }
      }
    };
    auto finish = [&](long long r, const half8 (&xv)[U], const half8 (&dv)[MODE == 1 ? U : 1], const half8 (&av)[MODE == 1 ? U : 1]) {
      const long long off0 = base + r * gm.tpr;
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (r + (long long)u * gm.rows_per_iter >= r1) break;
        float v[8], o[8];
        unpack8(xv[u], v);
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float y = sc[j] * v[j] + sh[j];
            o[j] = silu ? silu_f(y) : y;
          }
        } else {
          float d[8];
          unpack8(dv[u], d);
          // (the wave-uniform `silu` test sits OUTSIDE the element loop: inside it the compiler emitted one scalar branch per
          // element, which also kept the eight elements' exp / rcp chains from overlapping)
          if (silu) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const float xh = (v[j] + ad[j] - mu[j]) * rs[j];
              const float dxh = d[j] * dsilu_f(ga[j] * xh + be[j]) * ga[j];
              o[j] = rs[j] * (dxh - m1[j] - xh * m2[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const float xh = (v[j] + ad[j] - mu[j]) * rs[j];
              const float dxh = d[j] * ga[j];
              o[j] = rs[j] * (dxh - m1[j] - xh * m2[j]);
            }
          }
          if (accum) {      // the other gradient that reaches x (ResnetBlock2D: the shortcut's), added before the one rounding
            float a[8];
            unpack8(av[u], a);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] += a[j];
          }
        }
        out[off0 + u * stride] = pack8(o);
      }
    };
    if constexpr (MODE == 1) {
      // (a two-register-set software pipeline — the next trip's loads issued before this trip's math — measured neutral in round 5:
      // 13.33 vs 13.34 ms for the VAE's forward + backward; the pass already moves 4.6 TB/s against 5.07 for a plain copy of a
      // 268 MB tensor.  tools/experiments/groupnorm_bwd_pipeline.txt)
      for (long long r = r0 + ry; r < r1; r += step) {
        half8 xv[U], dv[U], av[U];
        load(r, xv, dv, av);
        finish(r, xv, dv, av);
      }
    } else {
      // FORWARD: one register set (five waves per SIMD already keep the memory pipe full: 4.0-4.4 TB/s); kept as the plain loop —
      // the lambdas above cost it 45 registers (81 -> 126, four waves per SIMD)
      for (long long r = r0 + ry; r < r1; r += step) {
        half8 xv[U];
        const long long off0 = base + r * gm.tpr;
#pragma unroll
        for (int u = 0; u < U; u++)
          if (r + (long long)u * gm.rows_per_iter < r1) xv[u] = x[off0 + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (r + (long long)u * gm.rows_per_iter >= r1) break;
          float v[8], o[8];
          unpack8(xv[u], v);
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float y = sc[j] * v[j] + sh[j];
            o[j] = silu ? silu_f(y) : y;
          }
          out[off0 + u * stride] = pack8(o);
        }
      }
    }
  }
}

// GroupNorm statistics from the per-(128-row block, channel) sums the producing convolution / linear kernel wrote in its
// epilogue (csrc/conv3x3.hip, chan_stats [blocks][C][2]): one workgroup per (sample, group), fixed summation order.
// With an addend a_c (added to x on load by the apply pass): sum(x + a) = S_c + P a_c, sum((x + a)^2) = Q_c + 2 a_c S_c + P a_c^2.
// BLOCK = 256, or 1024 where a sample has >= 256 blocks (the VAE encoder's first level: 2048 blocks x 128 channels = 8 MB of sums per
// launch): a thread then walks 8 instead of 32 blocks of its channel, all of its loads in flight at once (round 6: 17 -> ~5 us per
// launch on the single-stream VAE path).  Two-step combine of the per-thread sums in LDS, fixed order.
template <int BLOCK>
__device__ __forceinline__ void gn_block_channel_sums(const float* __restrict__ chan_sums, int blocks_per_sample, int C, int cg, int n, int g,
                                                      float* s_part, float* s_mid, float& S, float& Q) {
  const int j = threadIdx.x % cg, k = threadIdx.x / cg, K = BLOCK / cg;
  S = 0.f; Q = 0.f;
  if (k < K) {
    const float* p = chan_sums + ((size_t)n * blocks_per_sample * C + (size_t)g * cg + j) * 2;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f, s2 = 0.f, q2 = 0.f, s3 = 0.f, q3 = 0.f;
    int b = k;
    for (; b + 3 * K < blocks_per_sample; b += 4 * K) {           // four independent loads per trip
      const float2 a0 = *(const float2*)(p + (size_t)b * C * 2), a1 = *(const float2*)(p + (size_t)(b + K) * C * 2);
      const float2 a2 = *(const float2*)(p + (size_t)(b + 2 * K) * C * 2), a3 = *(const float2*)(p + (size_t)(b + 3 * K) * C * 2);
      s0 += a0.x; q0 += a0.y; s1 += a1.x; q1 += a1.y; s2 += a2.x; q2 += a2.y; s3 += a3.x; q3 += a3.y;
    }
    for (; b < blocks_per_sample; b += K) { const float2 a0 = *(const float2*)(p + (size_t)b * C * 2); s0 += a0.x; q0 += a0.y; }
    S = (s0 + s1) + (s2 + s3); Q = (q0 + q1) + (q2 + q3);
  }
  s_part[2 * threadIdx.x] = S; s_part[2 * threadIdx.x + 1] = Q;
  __syncthreads();
  // step A: cg x 16 threads each add every 16th slice of their channel; step B: cg threads add those 16
  const int S2 = K < 16 ? K : 16;
  if ((int)threadIdx.x < cg * S2) {
    const int jj = threadIdx.x % cg, q = threadIdx.x / cg;
    float a = 0.f, c = 0.f;
    for (int kk = q; kk < K; kk += S2) { a += s_part[2 * (kk * cg + jj)]; c += s_part[2 * (kk * cg + jj) + 1]; }
    s_mid[2 * threadIdx.x] = a; s_mid[2 * threadIdx.x + 1] = c;
  }
  __syncthreads();
  S = 0.f; Q = 0.f;
  if ((int)threadIdx.x < cg) {
    for (int q = 0; q < S2; q++) { S += s_mid[2 * (q * cg + threadIdx.x)]; Q += s_mid[2 * (q * cg + threadIdx.x) + 1]; }
  }
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
gn_finalize_stats_kernel(const float* __restrict__ chan_stats, int blocks_per_sample, long long HW, int C, int G, float eps,
                         const __half* __restrict__ addend, int addend_stride, float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ float s_part[BLOCK * 2];
  __shared__ float s_mid[BLOCK * 2];                             // step A's cg x S2 <= BLOCK partial sums
  __shared__ float s_ch[GN_BLOCK * 2];
  const int n = blockIdx.y, g = blockIdx.x, cg = C / G;         // cg <= GN_BLOCK (checked on the host)
  float S, Q;
  gn_block_channel_sums<BLOCK>(chan_stats, blocks_per_sample, C, cg, n, g, s_part, s_mid, S, Q);
  if ((int)threadIdx.x < cg) {
    if (addend) {
      const float a = __half2float(addend[(long long)n * addend_stride + g * cg + threadIdx.x]);
      Q += 2.f * a * S + (float)HW * a * a;
      S += (float)HW * a;
    }
    s_ch[2 * threadIdx.x] = S; s_ch[2 * threadIdx.x + 1] = Q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    S = 0.f; Q = 0.f;
    for (int c = 0; c < cg; c++) { S += s_ch[2 * c]; Q += s_ch[2 * c + 1]; }
    const float inv_m = 1.f / ((float)HW * (float)cg);
    const float mu = S * inv_m;
    const float var = fmaxf(Q * inv_m - mu * mu, 0.f);
    mean[n * G + g] = mu;
    rstd[n * G + g] = rsqrtf(var + eps);
  }
}

// The two reductions of the GroupNorm backward from the per-(128-row block, channel) sums that the data-gradient convolution
// wrote in its epilogue (csrc/conv3x3.hip, GnBwdArgs): out[n][g] = (sum dxh, sum dxh xh) / (HW * C / G).  Fixed order.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
gn_finalize_bwd_kernel(const float* __restrict__ chan_sums, int blocks_per_sample, long long HW, int C, int G, float* __restrict__ out) {
  __shared__ float s_part[BLOCK * 2];
  __shared__ float s_mid[BLOCK * 2];                             // step A's cg x S2 <= BLOCK partial sums
  __shared__ float s_ch[GN_BLOCK * 2];
  const int n = blockIdx.y, g = blockIdx.x, cg = C / G;
  float S, Q;
  gn_block_channel_sums<BLOCK>(chan_sums, blocks_per_sample, C, cg, n, g, s_part, s_mid, S, Q);
  if ((int)threadIdx.x < cg) { s_ch[2 * threadIdx.x] = S; s_ch[2 * threadIdx.x + 1] = Q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    S = 0.f; Q = 0.f;
    for (int c = 0; c < cg; c++) { S += s_ch[2 * c]; Q += s_ch[2 * c + 1]; }
    const float inv_m = 1.f / ((float)HW * (float)cg);
    out[((long long)n * G + g) * 2] = S * inv_m;
    out[((long long)n * G + g) * 2 + 1] = Q * inv_m;
  }
}

static void launch_finalize_stats(int G, int N, hipStream_t s, const float* chan_stats, int blocks_per_sample, long long HW, int C, float eps,
                                  const __half* addend, int addend_stride, float* mean, float* rstd) {
  if (blocks_per_sample >= 256)
    hipLaunchKernelGGL((gn_finalize_stats_kernel<1024>), dim3(G, N), dim3(1024), 0, s, chan_stats, blocks_per_sample, HW, C, G, eps, addend, addend_stride, mean, rstd);
  else
    hipLaunchKernelGGL((gn_finalize_stats_kernel<GN_BLOCK>), dim3(G, N), dim3(GN_BLOCK), 0, s, chan_stats, blocks_per_sample, HW, C, G, eps, addend, addend_stride, mean, rstd);
}
static void launch_finalize_bwd(int G, int N, hipStream_t s, const float* chan_sums, int blocks_per_sample, long long HW, int C, float* out) {
  if (blocks_per_sample >= 256)
    hipLaunchKernelGGL((gn_finalize_bwd_kernel<1024>), dim3(G, N), dim3(1024), 0, s, chan_sums, blocks_per_sample, HW, C, G, out);
  else
    hipLaunchKernelGGL((gn_finalize_bwd_kernel<GN_BLOCK>), dim3(G, N), dim3(GN_BLOCK), 0, s, chan_sums, blocks_per_sample, HW, C, G, out);
}

static size_t reduce_lds_bytes(int C) {
  const int tpr = C >> 3, cpt = (tpr + GN_BLOCK - 1) / GN_BLOCK, lanes_x = (tpr + cpt - 1) / cpt;
  const int rpi = GN_BLOCK / lanes_x > 0 ? GN_BLOCK / lanes_x : 1;
  return (size_t)rpi * C * 2 * sizeof(float);
}

static int pick_splits(int N, long long HW, int C) {
  // enough workgroups to fill the chip (256 CUs x a few), at least 8 rows each
  long long want = (1024 + N - 1) / N;
  long long by_rows = (HW + 7) / 8;
  long long s = want < by_rows ? want : by_rows;
  if (s < 1) s = 1;
  (void)C;
  return (int)s;
}

extern "C" size_t gip_gn_workspace_bytes(int32_t N, int32_t G) {
  return (size_t)N * GN_MAX_SPLITS * G * 2 * sizeof(float);
}

static int check(const void* a, const void* b, int32_t N, long long HW, int32_t C, int32_t G, size_t ws) {
  if (!a || !b || N < 1 || HW < 1 || C < 8 || (C & 7) || G < 1 || G > GN_BLOCK || C % G || C > 8192) return 1;
  if (ws < gip_gn_workspace_bytes(N, G)) return 2;
  return 0;
}

extern "C" int gip_gn_silu_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                   int32_t N, int64_t HW, int32_t C, int32_t G, float eps, int32_t apply_silu,
                                   const void* addend, int32_t addend_stride,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check(x, y, N, HW, C, G, workspace_bytes);
  if (rc) return rc;
  if (!gamma || !beta || !mean || !rstd || !workspace) return 1;
  hipStream_t s = (hipStream_t)stream;
  int splits = pick_splits(N, HW, C);
  const int rsplits = splits > GN_MAX_SPLITS ? GN_MAX_SPLITS : splits;
  float* partial = (float*)workspace;
  hipLaunchKernelGGL((gn_reduce_kernel<0>), dim3(rsplits, N), dim3(GN_BLOCK), reduce_lds_bytes(C), s,
                     (const half8*)x, (const half8*)nullptr, (const __half*)gamma, (const __half*)beta,
                     (const float*)nullptr, (const float*)nullptr, partial, (long long)HW, C, G, rsplits, apply_silu,
                     (const __half*)addend, addend_stride);
  hipLaunchKernelGGL((gn_apply_kernel<0>), dim3(splits, N), dim3(GN_BLOCK), (size_t)(G + GN_BLOCK) * 2 * sizeof(float), s,
                     (const half8*)x, (const half8*)nullptr, (const __half*)gamma, (const __half*)beta, mean, rstd,
                     (const float*)partial, (half8*)y, (long long)HW, C, G, rsplits, splits, eps, apply_silu,
                     (const __half*)addend, addend_stride, (const half8*)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_gn_silu_forward_stats(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                         int32_t N, int64_t HW, int32_t C, int32_t G, float eps, int32_t apply_silu,
                                         const void* addend, int32_t addend_stride, const float* chan_stats,
                                         int32_t blocks_per_sample, void* stream) {
  int rc = check(x, y, N, HW, C, G, (size_t)-1);
  if (rc) return rc;
  if (!gamma || !beta || !mean || !rstd || !chan_stats || blocks_per_sample < 1 || C / G > GN_BLOCK) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int splits = pick_splits(N, HW, C);
  launch_finalize_stats(G, N, s, chan_stats, blocks_per_sample, (long long)HW, C, eps, (const __half*)addend, addend_stride, mean, rstd);
  hipLaunchKernelGGL((gn_apply_kernel<0>), dim3(splits, N), dim3(GN_BLOCK), (size_t)(G + GN_BLOCK) * 2 * sizeof(float), s,
                     (const half8*)x, (const half8*)nullptr, (const __half*)gamma, (const __half*)beta, mean, rstd,
                     (const float*)nullptr, (half8*)y, (long long)HW, C, G, 0, splits, eps, apply_silu,
                     (const __half*)addend, addend_stride, (const half8*)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_gn_stats_from_partials(float* mean, float* rstd, int32_t N, int64_t HW, int32_t C, int32_t G, float eps,
                                          const void* addend, int32_t addend_stride, const float* chan_stats, int32_t blocks_per_sample,
                                          void* stream) {
  if (!mean || !rstd || !chan_stats || N < 1 || HW < 1 || C < 8 || G < 1 || C % G || blocks_per_sample < 1 || C / G > GN_BLOCK) return 1;
  launch_finalize_stats(G, N, (hipStream_t)stream, chan_stats, blocks_per_sample, (long long)HW, C, eps, (const __half*)addend, addend_stride, mean, rstd);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

static int gn_backward(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean, const float* rstd,
                       void* dx, int32_t N, int64_t HW, int32_t C, int32_t G, int32_t apply_silu, const void* addend,
                       int32_t addend_stride, const void* accum, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check(x, dx, N, HW, C, G, workspace_bytes);
  if (rc) return rc;
  if (!dy || !gamma || !beta || !mean || !rstd || !workspace) return 1;
  hipStream_t s = (hipStream_t)stream;
  int splits = pick_splits(N, HW, C);
  const int rsplits = splits > GN_MAX_SPLITS ? GN_MAX_SPLITS : splits;
  float* partial = (float*)workspace;
  hipLaunchKernelGGL((gn_reduce_kernel<1>), dim3(rsplits, N), dim3(GN_BLOCK), reduce_lds_bytes(C), s,
                     (const half8*)x, (const half8*)dy, (const __half*)gamma, (const __half*)beta, mean, rstd, partial,
                     (long long)HW, C, G, rsplits, apply_silu, (const __half*)addend, addend_stride);
  hipLaunchKernelGGL((gn_apply_kernel<1>), dim3(splits, N), dim3(GN_BLOCK), (size_t)(G + GN_BLOCK) * 2 * sizeof(float), s,
                     (const half8*)x, (const half8*)dy, (const __half*)gamma, (const __half*)beta,
                     const_cast<float*>(mean), const_cast<float*>(rstd), (const float*)partial, (half8*)dx,
                     (long long)HW, C, G, rsplits, splits, 0.f, apply_silu, (const __half*)addend, addend_stride,
                     (const half8*)accum);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_gn_silu_backward_sums(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                                         const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G,
                                         int32_t apply_silu, const void* addend, int32_t addend_stride, const void* accum,
                                         const float* chan_sums, int32_t blocks_per_sample, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  int rc = check(x, dx, N, HW, C, G, workspace_bytes);
  if (rc) return rc;
  if (!dy || !gamma || !beta || !mean || !rstd || !workspace || !chan_sums || blocks_per_sample < 1 || C / G > GN_BLOCK) return 1;
  hipStream_t s = (hipStream_t)stream;
  float* sums = (float*)workspace;                       // [N, G, 2]
  launch_finalize_bwd(G, N, s, chan_sums, blocks_per_sample, (long long)HW, C, sums);
  const int splits = pick_splits(N, HW, C);
  hipLaunchKernelGGL((gn_apply_kernel<1>), dim3(splits, N), dim3(GN_BLOCK), (size_t)(G + GN_BLOCK) * 2 * sizeof(float), s,
                     (const half8*)x, (const half8*)dy, (const __half*)gamma, (const __half*)beta,
                     const_cast<float*>(mean), const_cast<float*>(rstd), (const float*)sums, (half8*)dx,
                     (long long)HW, C, G, 0, splits, 0.f, apply_silu, (const __half*)addend, addend_stride, (const half8*)accum);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_gn_silu_backward(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                                    const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G,
                                    int32_t apply_silu, const void* addend, int32_t addend_stride,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return gn_backward(x, dy, gamma, beta, mean, rstd, dx, N, HW, C, G, apply_silu, addend, addend_stride, nullptr, workspace,
                     workspace_bytes, stream);
}

extern "C" int gip_gn_silu_backward_accum(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                                          const float* rstd, void* dx, int32_t N, int64_t HW, int32_t C, int32_t G,
                                          int32_t apply_silu, const void* addend, int32_t addend_stride, const void* accum,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!accum) return 1;
  return gn_backward(x, dy, gamma, beta, mean, rstd, dx, N, HW, C, G, apply_silu, addend, addend_stride, accum, workspace,
                     workspace_bytes, stream);
}


// ---------------------------------------------------------------------------------------------------------------
// Pointwise companions (include/gip_nn.h): out = a + b + bias[c]   and   GEGLU out = value * gelu(gate).
// 16-byte accesses, grid-stride; C % 8 == 0.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
add_bias_residual_kernel(const half8* __restrict__ a, const half8* __restrict__ b, const __half* __restrict__ bias,
                         half8* __restrict__ out, long long n8, int tpr) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float x[8], y[8], o[8];
    unpack8(a[i], x);
    unpack8(b[i], y);
    const int c = (int)(i % tpr) * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = x[j] + y[j] + (bias ? __half2float(bias[c + j]) : 0.f);
    out[i] = pack8(o);
  }
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the half rounding of the result): one v_rcp, one v_exp and
// six fma instead of libm's erff (~40 vector instructions per value: the GEGLU kernels were bound by it, not by memory)
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  return copysignf(fmaf(-p * t, e, 1.f), x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erf_fast(v * 0.70710678118654752f)); }

__global__ void __launch_bounds__(256)
geglu_kernel(const half8* __restrict__ in /* [M, 2D] */, half8* __restrict__ out /* [M, D] */, long long M, int d8) {
  const long long total = M * d8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / d8;
    const int k = (int)(i % d8);
    float v[8], g[8], o[8];
    unpack8(in[r * 2 * d8 + k], v);
    unpack8(in[r * 2 * d8 + d8 + k], g);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = v[j] * gelu_erf(g[j]);
    out[i] = pack8(o);
  }
}

static int grid_for(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int gip_add_bias_residual(const void* a, const void* b, const void* bias, void* out, int64_t M, int32_t C,
                                     void* stream) {
  if (!a || !b || !out || M < 1 || C < 8 || (C & 7)) return 1;
  const long long n8 = (long long)M * (C >> 3);
  hipLaunchKernelGGL(add_bias_residual_kernel, dim3(grid_for(n8)), dim3(256), 0, (hipStream_t)stream, (const half8*)a,
                     (const half8*)b, (const __half*)bias, (half8*)out, n8, C >> 3);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_geglu(const void* in, void* out, int64_t M, int32_t D, void* stream) {
  if (!in || !out || M < 1 || D < 8 || (D & 7)) return 1;
  hipLaunchKernelGGL(geglu_kernel, dim3(grid_for((long long)M * (D >> 3))), dim3(256), 0, (hipStream_t)stream,
                     (const half8*)in, (half8*)out, (long long)M, D >> 3);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}


// ---------------------------------------------------------------------------------------------------------------
// Skip-connection concatenation of the U-Net decoder: out[m] = [ a[m] | b[m] (+ b_add[m]) ]  ([M, Ca + Cb] half rows), with
// the statistics of the GroupNorm that consumes it (ResnetBlock2D.norm1 over the concatenated channels) taken on the way:
// chan_stats [M / 128][Ca + Cb][2] = per 128-row block, per channel (sum, sum of squares) of the half-rounded values
// written — the same partials the convolution epilogue produces (csrc/conv3x3.hip), so gn_finalize_stats_kernel serves
// both.  b_add is the ControlNet residual of that skip (`sample + residual` of diffusers, rounded to half before the
// concatenation exactly as the separate add would).  One pass instead of add + cat + the GroupNorm's statistics pass.
// A workgroup owns 128 rows x 64 channels (8 chunks of 8 halves x 32 row-lanes x 4 rows); Ca % 64 == 0 keeps a column
// block inside one source.  Fixed summation order: deterministic.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cat2_stats_kernel(const half8* __restrict__ a, const half8* __restrict__ b, const half8* __restrict__ b_add,
                  half8* __restrict__ out, float* __restrict__ chan_stats, long long M, int Ca, int Cb) {
  __shared__ float part[32 * 64 * 2];
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int Ct = Ca + Cb, col = blockIdx.y * 64 + cx * 8;
  const long long m0 = (long long)blockIdx.x * 128;
  const bool from_a = col < Ca;
  float s8[8], q8[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { s8[j] = 0.f; q8[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const long long m = m0 + ry + 32 * i;
    if (m >= M) break;
    float v[8];
    if (from_a) {
      unpack8(a[(m * Ca + col) >> 3], v);
    } else {
      unpack8(b[(m * Cb + (col - Ca)) >> 3], v);
      if (b_add) {
        float r[8];
        unpack8(b_add[(m * Cb + (col - Ca)) >> 3], r);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] += r[j];
      }
    }
    const half8 h = pack8(v);
    out[(m * Ct + col) >> 3] = h;
    if (chan_stats) {
      unpack8(h, v);
#pragma unroll
      for (int j = 0; j < 8; j++) { s8[j] += v[j]; q8[j] = fmaf(v[j], v[j], q8[j]); }
    }
  }
  if (!chan_stats) return;                 // kernel argument: uniform
#pragma unroll
  for (int j = 0; j < 8; j++) {
    part[(ry * 64 + cx * 8 + j) * 2] = s8[j];
    part[(ry * 64 + cx * 8 + j) * 2 + 1] = q8[j];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r++) { S += part[(r * 64 + threadIdx.x) * 2]; Q += part[(r * 64 + threadIdx.x) * 2 + 1]; }
    float* o = chan_stats + ((size_t)blockIdx.x * Ct + blockIdx.y * 64 + threadIdx.x) * 2;
    o[0] = S; o[1] = Q;
  }
}

extern "C" int gip_cat2_stats_f16(const void* a, const void* b, const void* b_add, void* out, float* chan_stats, int64_t M,
                                  int32_t Ca, int32_t Cb, void* stream) {
  if (!a || !b || !out || M < 1 || Ca < 64 || Cb < 64 || (Ca & 63) || (Cb & 63)) return 1;
  if (chan_stats && (M & 127)) return 1;
  const long long blocks = (M + 127) / 128;
  if (blocks > 0x7fffffffll || M * (long long)(Ca + Cb) >= (1ll << 40)) return 1;
  hipLaunchKernelGGL(cat2_stats_kernel, dim3((unsigned)blocks, (unsigned)((Ca + Cb) / 64)), dim3(256), 0, (hipStream_t)stream,
                     (const half8*)a, (const half8*)b, (const half8*)b_add, (half8*)out, chan_stats, (long long)M, Ca, Cb);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension of [M, C] half rows (BasicTransformerBlock.norm1/2/3): y = (x - mean) * rstd * w + b.
// HBM-bound: each row is read once into registers and written once.  A row is owned by a 16-lane group (one DPP row), so
// a wave covers 4 rows and every load instruction fetches 256 contiguous bytes per row; ITER = ceil(C / 128) half8 chunks
// per lane are all issued before the first is consumed (up to 10 x 16 B in flight per lane at C = 1280).  Statistics are
// two-pass in fp32 on the register copy (exact mean first, then centred squares).
// ---------------------------------------------------------------------------------------------------------------
template <int ITER>
__global__ void __launch_bounds__(256)
layernorm_kernel(const half8* __restrict__ x, const __half* __restrict__ w, const __half* __restrict__ b,
                 half8* __restrict__ y, long long M, int C8, float invC, float eps) {
  const int sub = threadIdx.x & 15;
  const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= M) return;                       // whole 16-lane groups leave together; the shuffles below stay inside a group
  const half8* xr = x + row * C8;
  half8 v[ITER], gw[ITER], gb[ITER];
  const half8* w8 = (const half8*)w;
  const half8* b8 = (const half8*)b;
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = i * 16 + sub;
    if (c < C8) v[i] = xr[c];
  }
  // gamma / beta are requested WITH the row (they were loaded after the two reductions: a dependent L2 round trip in front of
  // the stores of every wave)
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = i * 16 + sub;
    if (c < C8) { gw[i] = w8[c]; gb[i] = b8[c]; }
  }
  float f[ITER][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = i * 16 + sub;
    if (c < C8) {
      unpack8(v[i], f[i]);
#pragma unroll
      for (int k = 0; k < 8; ++k) sum += f[i][k];
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 16);
  const float mean = sum * invC;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = i * 16 + sub;
    if (c < C8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { f[i][k] -= mean; sq += f[i][k] * f[i][k]; }
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 16);
  const float rstd = rsqrtf(sq * invC + eps);
  half8* yr = y + row * C8;
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = i * 16 + sub;
    if (c < C8) {
      float g[8], be[8], o[8];
      unpack8(gw[i], g);
      unpack8(gb[i], be);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = f[i][k] * rstd * g[k] + be[k];
      yr[c] = pack8(o);
    }
  }
}

extern "C" int gip_layernorm_f16(const void* x, const void* weight, const void* bias, void* y, int64_t M, int32_t C,
                                 float eps, void* stream) {
  if (!x || !weight || !bias || !y || M < 1 || C < 8 || (C & 7) || C > 2048 || M > (1ll << 34)) return 1;
  const int C8 = C >> 3, iter = (C8 + 15) >> 4;
  const dim3 grid((unsigned)((M + 15) >> 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define GIP_LN(I)                                                                                                       \
  hipLaunchKernelGGL((layernorm_kernel<I>), grid, block, 0, s, (const half8*)x, (const __half*)weight,                  \
                     (const __half*)bias, (half8*)y, (long long)M, C8, 1.0f / (float)C, eps)
  if (iter <= 3) GIP_LN(3);
  else if (iter <= 5) GIP_LN(5);
  else if (iter <= 10) GIP_LN(10);
  else GIP_LN(16);
#undef GIP_LN
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
