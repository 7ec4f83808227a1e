// softmax.hip — row softmax forward / backward, in place, for the VAE encoder's single-head 512-channel mid attention
// (diffusers AttnBlock of AutoencoderKL.encode; reference call ipa_guidance.py:522-531).  See include/gip_nn.h.
//
// The mid attention is two dense GEMMs forward (S = Q K^T, O = P V) and four backward at the FLOP minimum; a flash-style kernel
// would re-compute S in the backward (+17 % FLOPs) with a 512-wide head that does not fit one wave's accumulators — so the
// products stay GEMMs and only the softmax between them is this file's business (round 4 ran torch's softmax kernels plus a
// `q * scale` pass and a second 134 MB score tensor):
//   forward   P[r, :] = softmax(scale * S[r, :])                  in place over S      (one read, one write)
//   backward  dS[r, :] = scale * P[r, :] * (dP[r, :] - sum_j dP[r, j] P[r, j])   in place over dP   (two reads, one write)
// One wave per row, the row in registers between the passes (N <= 64 * 8 * SM_MAX_CHUNKS halves), fp32 arithmetic, exp2 domain.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gip_nn.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SM_MAX_CHUNKS 16        // 16-byte chunks per lane: rows of up to 8192 halves

__device__ __forceinline__ float sm_wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float sm_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <int CH>
__global__ void __launch_bounds__(256)
softmax_rows_kernel(_Float16* __restrict__ s, long long rows, int n, float c /* scale * log2(e) */) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f16x8* p = reinterpret_cast<f16x8*>(s + row * n);
  f16x8 v[CH];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < CH; k++) {
    const int ch = k * 64 + lane;
    if (ch * 8 < n) {
      v[k] = p[ch];
#pragma unroll
      for (int j = 0; j < 8; j++) m = fmaxf(m, (float)v[k][j]);
    }
  }
  m = sm_wave_max(m);
  const float mc = m * c;
  float e[CH][8], sum = 0.f;
#pragma unroll
  for (int k = 0; k < CH; k++) {
    const int ch = k * 64 + lane;
    if (ch * 8 < n) {
#pragma unroll
      for (int j = 0; j < 8; j++) { e[k][j] = __builtin_amdgcn_exp2f(__builtin_fmaf((float)v[k][j], c, -mc)); sum += e[k][j]; }
    }
  }
  sum = sm_wave_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < CH; k++) {
    const int ch = k * 64 + lane;
    if (ch * 8 < n) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = (_Float16)(e[k][j] * inv);
      p[ch] = o;
    }
  }
}

template <int CH>
__global__ void __launch_bounds__(256)
softmax_rows_bwd_kernel(const _Float16* __restrict__ pr, _Float16* __restrict__ dp, long long rows, int n, float scale) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f16x8* p = reinterpret_cast<const f16x8*>(pr + row * n);
  f16x8* g = reinterpret_cast<f16x8*>(dp + row * n);
  f16x8 pv[CH], gv[CH];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < CH; k++) {
    const int ch = k * 64 + lane;
    if (ch * 8 < n) {
      pv[k] = p[ch];
      gv[k] = g[ch];
#pragma unroll
      for (int j = 0; j < 8; j++) dot += (float)pv[k][j] * (float)gv[k][j];
    }
  }
  dot = sm_wave_sum(dot);
#pragma unroll
  for (int k = 0; k < CH; k++) {
    const int ch = k * 64 + lane;
    if (ch * 8 < n) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = (_Float16)(scale * (float)pv[k][j] * ((float)gv[k][j] - dot));
      g[ch] = o;
    }
  }
}

template <int CH>
static void launch_fwd(hipStream_t s, void* x, long long rows, int n, float c) {
  hipLaunchKernelGGL((softmax_rows_kernel<CH>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (_Float16*)x, rows, n, c);
}
template <int CH>
static void launch_bwd(hipStream_t s, const void* p, void* dp, long long rows, int n, float scale) {
  hipLaunchKernelGGL((softmax_rows_bwd_kernel<CH>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const _Float16*)p, (_Float16*)dp, rows, n, scale);
}

static int chunks_for(int n) { return (n / 8 + 63) / 64; }

extern "C" int gip_softmax_rows_f16(void* s, int64_t rows, int32_t n, float scale, void* stream) {
  if (!s || rows < 1 || n < 8 || (n & 7) || chunks_for(n) > SM_MAX_CHUNKS || (rows + 3) / 4 > 0x7fffffffll) return 1;
  const float c = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  switch (chunks_for(n)) {
    case 1: launch_fwd<1>(st, s, rows, n, c); break;
    case 2: launch_fwd<2>(st, s, rows, n, c); break;
    case 3: case 4: launch_fwd<4>(st, s, rows, n, c); break;
    case 5: case 6: case 7: case 8: launch_fwd<8>(st, s, rows, n, c); break;
    default: launch_fwd<SM_MAX_CHUNKS>(st, s, rows, n, c); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_softmax_rows_backward_f16(const void* p, void* dp, int64_t rows, int32_t n, float scale, void* stream) {
  if (!p || !dp || rows < 1 || n < 8 || (n & 7) || chunks_for(n) > SM_MAX_CHUNKS || (rows + 3) / 4 > 0x7fffffffll) return 1;
  hipStream_t st = (hipStream_t)stream;
  switch (chunks_for(n)) {
    case 1: launch_bwd<1>(st, p, dp, rows, n, scale); break;
    case 2: launch_bwd<2>(st, p, dp, rows, n, scale); break;
    case 3: case 4: launch_bwd<4>(st, p, dp, rows, n, scale); break;
    case 5: case 6: case 7: case 8: launch_bwd<8>(st, p, dp, rows, n, scale); break;
    default: launch_bwd<SM_MAX_CHUNKS>(st, p, dp, rows, n, scale); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
