// winograd.hip — the two transforms of Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions of the U-Net's and
// ControlNet's 16 x 16 level (1280 channels and the 1920 / 2560-channel skip concatenations).  See include/gip_nn.h
// (gip_winograd_input_f16 / gip_winograd_output_f16).
//
// Why only there: Y = A^T [ (G g G^T) .* (B^T d B) ] A turns the layer into SIXTEEN GEMMs [tiles x Cin] x [Cin x Cout] over the
// 2 x 2-pixel output tiles — 16 / 4 = 4 multiplications per output pixel and channel pair instead of 9.  The transforms move 4x
// the input once and 4x the output once; that only pays where the GEMMs dominate: measured (tools/diag/winograd_feasibility.py,
// 12 samples) 1280 -> 1280 @ 16^2: implicit GEMM 118 us, the sixteen GEMMs 49 us; 2560 -> 1280: 227 vs 81 us; at 32^2 / 640
// channels and at 8^2 (weight-bound: the transformed weights are 16 / 9 as large) the transforms eat the gain.
// The sixteen GEMMs themselves are ONE batched library GEMM (plain GEMM = hipBLASLt's job).
//
//   B^T = | 1  0 -1  0 |     G = | 1    0    0  |     A^T = | 1  1  1  0 |
//         | 0  1  1  0 |         | 1/2  1/2  1/2 |           | 0  1 -1 -1 |
//         | 0 -1  1  0 |         | 1/2 -1/2  1/2 |
//         | 0  1  0 -1 |         | 0    0    1  |
// d = the 4 x 4 input patch whose top-left pixel is (2 ty - 1, 2 tx - 1) (zeros outside the image), tile (ty, tx) -> output
// pixels (2 ty .. 2 ty + 1, 2 tx .. 2 tx + 1).  The weight transform U = G g G^T is done once by the host in fp32.
// All arithmetic here in fp32; V is rounded to half once (it is a GEMM operand), the output once.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/gip_nn.h"

struct alignas(16) wg_half8 { __half2 a, b, c, d; };

__device__ __forceinline__ void wg_unpack(const wg_half8& h, float* f) {
  const float2 x = __half22float2(h.a), y = __half22float2(h.b), z = __half22float2(h.c), w = __half22float2(h.d);
  f[0] = x.x; f[1] = x.y; f[2] = y.x; f[3] = y.y; f[4] = z.x; f[5] = z.y; f[6] = w.x; f[7] = w.y;
}
__device__ __forceinline__ wg_half8 wg_pack(const float* f) {
  wg_half8 h;
  h.a = __floats2half2_rn(f[0], f[1]); h.b = __floats2half2_rn(f[2], f[3]);
  h.c = __floats2half2_rn(f[4], f[5]); h.d = __floats2half2_rn(f[6], f[7]);
  return h;
}

// GroupNorm (+ SiLU) of the convolution's INPUT applied while the patch is loaded (round 5): x is the RAW tensor, the normalised
// tensor y = silu?(GroupNorm(x + addend)) is never written or read.  The arithmetic of csrc/groupnorm.hip's apply pass (sc = rstd
// gamma, sh = beta - (mean - addend) sc, y = sc x + sh, SiLU by v_exp + v_rcp) and its ONE rounding of y to half, so V is what the
// transform of the separately normalised tensor gives; patch elements outside the image are zeros of y (the padding), not of x.
struct WgGnArgs {
  const __half* gamma;
  const __half* beta;
  const float* mean;      // [N, G]
  const float* rstd;
  const __half* addend;   // optional per-(sample, channel), row stride addend_stride (0 = one row)
  int G, silu, addend_stride;
};

// V[p][t][c], p = 4 i + j of the transformed patch, t = (n, ty, tx) tile, c = channel; one thread = one tile x 8 channels
template <bool GN>
__global__ void __launch_bounds__(256)
winograd_input_kernel(const wg_half8* __restrict__ x /* [N, H, W, C] */, wg_half8* __restrict__ V /* [16, T, C] */, int N, int H, int W, int C8,
                      long long T, WgGnArgs gn) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * C8) return;
  const int c = (int)(idx % C8);
  const long long t = idx / C8;
  const int tw = W >> 1, th = H >> 1;
  const int tx = (int)(t % tw), ty = (int)((t / tw) % th), n = (int)(t / ((long long)tw * th));
  [[maybe_unused]] float sc[8], sh[8];
  if constexpr (GN) {
    const int cg = (C8 * 8) / gn.G;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int ch = c * 8 + k, g = ch / cg;
      const float ad = gn.addend ? __half2float(gn.addend[(long long)n * gn.addend_stride + ch]) : 0.f;
      sc[k] = gn.rstd[n * gn.G + g] * __half2float(gn.gamma[ch]);
      sh[k] = __half2float(gn.beta[ch]) - (gn.mean[n * gn.G + g] - ad) * sc[k];
    }
  }
  float d[4][4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int y = 2 * ty - 1 + i, xx = 2 * tx - 1 + j;
      if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) {
        wg_unpack(x[(((long long)n * H + y) * W + xx) * C8 + c], d[i][j]);
        if constexpr (GN) {
          if (gn.silu) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
              const float yv = sc[k] * d[i][j][k] + sh[k];
              d[i][j][k] = __half2float(__float2half_rn(yv * __builtin_amdgcn_rcpf(1.f + __expf(-yv))));
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; k++) d[i][j][k] = __half2float(__float2half_rn(sc[k] * d[i][j][k] + sh[k]));
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) d[i][j][k] = 0.f;
      }
    }
  // rows: B^T d
  float r[4][4][8];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      r[0][j][k] = d[0][j][k] - d[2][j][k];
      r[1][j][k] = d[1][j][k] + d[2][j][k];
      r[2][j][k] = d[2][j][k] - d[1][j][k];
      r[3][j][k] = d[1][j][k] - d[3][j][k];
    }
  // columns: (B^T d) B
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float o[4][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      o[0][k] = r[i][0][k] - r[i][2][k];
      o[1][k] = r[i][1][k] + r[i][2][k];
      o[2][k] = r[i][2][k] - r[i][1][k];
      o[3][k] = r[i][1][k] - r[i][3][k];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) V[((long long)(4 * i + j) * T + t) * C8 + c] = wg_pack(o[j]);
  }
}

// out[n, 2 ty + a, 2 tx + b, co] = (A^T M A)[a][b] + bias[co] (+ residual): one thread = one tile x 8 output channels
__global__ void __launch_bounds__(256)
winograd_output_kernel(const wg_half8* __restrict__ Mm /* [16, T, C] */, const __half* __restrict__ bias, const wg_half8* __restrict__ residual,
                       wg_half8* __restrict__ out /* [N, H, W, C] */, int N, int H, int W, int C8, long long T) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * C8) return;
  const int c = (int)(idx % C8);
  const long long t = idx / C8;
  const int tw = W >> 1, th = H >> 1;
  const int tx = (int)(t % tw), ty = (int)((t / tw) % th), n = (int)(t / ((long long)tw * th));
  float m[4][4][8];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) wg_unpack(Mm[((long long)(4 * i + j) * T + t) * C8 + c], m[i][j]);
  float b8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) b8[k] = bias ? __half2float(bias[c * 8 + k]) : 0.f;
  // rows: A^T M   (2 x 4)
  float r[2][4][8];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      r[0][j][k] = m[0][j][k] + m[1][j][k] + m[2][j][k];
      r[1][j][k] = m[1][j][k] - m[2][j][k] - m[3][j][k];
    }
#pragma unroll
  for (int a = 0; a < 2; a++) {
    float o[2][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      o[0][k] = r[a][0][k] + r[a][1][k] + r[a][2][k] + b8[k];
      o[1][k] = r[a][1][k] - r[a][2][k] - r[a][3][k] + b8[k];
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const long long pix = (((long long)n * H + 2 * ty + a) * W + 2 * tx + b) * C8 + c;
      if (residual) {
        // half(conv + bias) first, then the residual: the implicit GEMM's (and diffusers') two roundings
        float h[8], rr[8];
        wg_unpack(wg_pack(o[b]), h);
        wg_unpack(residual[pix], rr);
#pragma unroll
        for (int k = 0; k < 8; k++) o[b][k] = h[k] + rr[k];
      }
      out[pix] = wg_pack(o[b]);
    }
  }
}

// The same output transform organised for the GroupNorm that follows the convolution: a workgroup owns 32 consecutive tiles x
// 64 channels — for W = 16 or 32 exactly 128 consecutive NHWC pixels — and also leaves chan_stats [pixels / 128][C][2] = the
// per-channel sum and sum of squares of the half-rounded values it wrote (the partials gip_gn_silu_forward_stats takes).
__global__ void __launch_bounds__(256)
winograd_output_stats_kernel(const wg_half8* __restrict__ Mm, const __half* __restrict__ bias, const wg_half8* __restrict__ residual,
                             wg_half8* __restrict__ out, float* __restrict__ chan_stats, int N, int H, int W, int C8, long long T) {
  __shared__ float part[32 * 64 * 2];
  const int cx = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int c = blockIdx.y * 8 + cx;
  const long long t = (long long)blockIdx.x * 32 + tl;
  const int tw = W >> 1, th = H >> 1;
  float s8[8], q8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { s8[k] = 0.f; q8[k] = 0.f; }
  if (c < C8 && t < T) {
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th), n = (int)(t / ((long long)tw * th));
    float m[4][4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) wg_unpack(Mm[((long long)(4 * i + j) * T + t) * C8 + c], m[i][j]);
    float b8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) b8[k] = bias ? __half2float(bias[c * 8 + k]) : 0.f;
    float r[2][4][8];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int k = 0; k < 8; k++) {
        r[0][j][k] = m[0][j][k] + m[1][j][k] + m[2][j][k];
        r[1][j][k] = m[1][j][k] - m[2][j][k] - m[3][j][k];
      }
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float o[2][8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        o[0][k] = r[a][0][k] + r[a][1][k] + r[a][2][k] + b8[k];
        o[1][k] = r[a][1][k] - r[a][2][k] - r[a][3][k] + b8[k];
      }
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const long long pix = (((long long)n * H + 2 * ty + a) * W + 2 * tx + b) * C8 + c;
        if (residual) {
          float h[8], rr[8];
          wg_unpack(wg_pack(o[b]), h);
          wg_unpack(residual[pix], rr);
#pragma unroll
          for (int k = 0; k < 8; k++) o[b][k] = h[k] + rr[k];
        }
        const wg_half8 hv = wg_pack(o[b]);
        out[pix] = hv;
        float f[8];
        wg_unpack(hv, f);
#pragma unroll
        for (int k = 0; k < 8; k++) { s8[k] += f[k]; q8[k] = fmaf(f[k], f[k], q8[k]); }
      }
    }
  }
  if (!chan_stats) return;                 // kernel argument: uniform
#pragma unroll
  for (int k = 0; k < 8; k++) {
    part[(tl * 64 + cx * 8 + k) * 2] = s8[k];
    part[(tl * 64 + cx * 8 + k) * 2 + 1] = q8[k];
  }
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < C8 * 8) {
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int rr = 0; rr < 32; rr++) { S += part[(rr * 64 + threadIdx.x) * 2]; Q += part[(rr * 64 + threadIdx.x) * 2 + 1]; }
    float* o = chan_stats + ((size_t)blockIdx.x * (C8 * 8) + blockIdx.y * 64 + threadIdx.x) * 2;
    o[0] = S; o[1] = Q;
  }
}

extern "C" int gip_winograd_output_stats_f16(const void* Mm, const void* bias, const void* residual, void* out, float* chan_stats,
                                             int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!Mm || !out || N < 1 || H < 2 || (H & 1) || (W != 16 && W != 32) || C < 8 || (C & 7) || ((long long)H * W) % 128) return 1;
  const long long T = (long long)N * (H / 2) * (W / 2);       // a multiple of 32: H * W % 128 == 0
  hipLaunchKernelGGL(winograd_output_stats_kernel, dim3((unsigned)(T / 32), (unsigned)((C / 8 + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const wg_half8*)Mm, (const __half*)bias, (const wg_half8*)residual, (wg_half8*)out, chan_stats, N, H, W, C / 8, T);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_winograd_input_f16(const void* x, void* V, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !V || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || (C & 7)) return 1;
  const long long T = (long long)N * (H / 2) * (W / 2), total = T * (C / 8);
  if (total > 0x7fffffffll * 256) return 1;
  hipLaunchKernelGGL((winograd_input_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const wg_half8*)x, (wg_half8*)V, N, H, W, C / 8, T, WgGnArgs{});
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_winograd_input_gn_f16(const void* x, void* V, int32_t N, int32_t H, int32_t W, int32_t C, const void* gamma, const void* beta,
                                         const float* mean, const float* rstd, int32_t G, int32_t apply_silu, const void* addend,
                                         int32_t addend_stride, void* stream) {
  if (!x || !V || !gamma || !beta || !mean || !rstd || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || (C & 7) || G < 1 || C % G) return 1;
  const long long T = (long long)N * (H / 2) * (W / 2), total = T * (C / 8);
  if (total > 0x7fffffffll * 256) return 1;
  WgGnArgs gn{(const __half*)gamma, (const __half*)beta, mean, rstd, (const __half*)addend, G, apply_silu, addend_stride};
  hipLaunchKernelGGL((winograd_input_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const wg_half8*)x, (wg_half8*)V, N, H, W, C / 8, T, gn);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_winograd_output_f16(const void* Mm, const void* bias, const void* residual, void* out, int32_t N, int32_t H, int32_t W,
                                       int32_t C, void* stream) {
  if (!Mm || !out || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || (C & 7)) return 1;
  const long long T = (long long)N * (H / 2) * (W / 2), total = T * (C / 8);
  if (total > 0x7fffffffll * 256) return 1;
  hipLaunchKernelGGL(winograd_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const wg_half8*)Mm, (const __half*)bias, (const wg_half8*)residual, (wg_half8*)out, N, H, W, C / 8, T);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
