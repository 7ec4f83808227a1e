// lpips.hip — one LPIPS layer term against cached target features, forward and data gradient (include/gip_nn.h).
//
//   d[n] = sum_{hw} sum_c lin[c] * ( f[n,hw,c] / (|f[n,hw,:]| + eps)  -  t[n,hw,c] )^2          (the caller divides by HW)
//
// f: raw VGG feature map of the rendered image (fp16, NHWC), t: channel-normalised features of the fixed target image
// (fp16, computed once), lin: the layer's non-negative 1x1 weights.  In PyTorch this is ~12 elementwise / reduction
// passes over fp32 copies of a 60 MB tensor per layer and direction; here a pixel's channel row is read once into
// registers by a 16-lane group (one DPP row; 4 pixels per wave, each load 256 contiguous bytes per pixel) and everything
// else is register math: HBM-bound at one read (forward) / one read + one write (backward) of the fp16 tensors.
//
// The gradient is written in fp16 for the fp16 VGG graph behind it, pre-multiplied by coef[n] = gout[n] / HW * S with the
// caller's power-of-two loss scale S (raw values are 1e-8..1e-6, below fp16's subnormals) and saturated to +-65504 so a
// pixel whose features are all zero (|f| = 0: the 1/eps branch of the reference formula) cannot inject inf.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/gip_nn.h"

namespace {

struct alignas(16) h8 { __half2 a, b, c, d; };

__device__ __forceinline__ void unpack(const h8& h, float* f) {
  const float2 p0 = __half22float2(h.a), p1 = __half22float2(h.b), p2 = __half22float2(h.c), p3 = __half22float2(h.d);
  f[0] = p0.x; f[1] = p0.y; f[2] = p1.x; f[3] = p1.y; f[4] = p2.x; f[5] = p2.y; f[6] = p3.x; f[7] = p3.y;
}

__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}

constexpr int LP_BLOCK = 256;            // 16 pixel rows per workgroup pass
constexpr float LP_EPS = 1e-10f;         // lpips.normalize_tensor

// MODE 0: partial[n][block] = sum over this block's pixels of the weighted squared difference
// MODE 1: grad[n, hw, :] = d(sum_hw ...)/df * coef[n], saturated fp16
template <int ITER, int MODE>
__global__ void __launch_bounds__(LP_BLOCK)
lpips_layer_kernel(const h8* __restrict__ feat, const h8* __restrict__ target, const float* __restrict__ lin,
                   float* __restrict__ partial, const float* __restrict__ coef, h8* __restrict__ grad, long long HW, int C8) {
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int n = blockIdx.y;
  float w[ITER][8];
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int c = i * 16 + sub;
#pragma unroll
    for (int k = 0; k < 8; ++k) w[i][k] = c < C8 ? lin[c * 8 + k] : 0.f;
  }
  const float cf = MODE == 1 ? coef[n] : 0.f;
  const h8* fn = feat + (long long)n * HW * C8;
  const h8* tn = target + (long long)n * HW * C8;
  float acc = 0.f;
  for (long long row = (long long)blockIdx.x * 16 + grp; row < HW; row += (long long)gridDim.x * 16) {
    h8 fv[ITER], tv[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int c = i * 16 + sub;
      if (c < C8) { fv[i] = fn[row * C8 + c]; tv[i] = tn[row * C8 + c]; }
    }
    float f[ITER][8], t[ITER][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int c = i * 16 + sub;
      if (c < C8) {
        unpack(fv[i], f[i]);
        unpack(tv[i], t[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) ss += f[i][k] * f[i][k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { f[i][k] = 0.f; t[i][k] = 0.f; }
      }
    }
    const float s = sqrtf(group_sum(ss));
    const float inv = 1.f / (s + LP_EPS);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = f[i][k] * inv - t[i][k];
          acc += w[i][k] * d * d;
        }
    } else {
      // g_u = 2 lin (u - t) coef;   g_f = g_u / (s + eps) - f (f . g_u) / (s (s + eps)^2)
      float gu[ITER][8];
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          gu[i][k] = 2.f * w[i][k] * (f[i][k] * inv - t[i][k]) * cf;
          dot += f[i][k] * gu[i][k];
        }
      dot = group_sum(dot);
      const float back = s > 0.f ? dot * inv * inv / s : 0.f;
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int c = i * 16 + sub;
        if (c < C8) {
          float o[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = fminf(fmaxf(gu[i][k] * inv - f[i][k] * back, -65504.f), 65504.f);
          h8 r;
          r.a = __floats2half2_rn(o[0], o[1]); r.b = __floats2half2_rn(o[2], o[3]);
          r.c = __floats2half2_rn(o[4], o[5]); r.d = __floats2half2_rn(o[6], o[7]);
          grad[((long long)n * HW + row) * C8 + c] = r;
        }
      }
    }
  }
  if (MODE == 0) {
    // fixed-order workgroup reduction (no float atomics): lanes -> wave -> LDS -> thread 0
    __shared__ float s_part[LP_BLOCK / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)n * gridDim.x + blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  }
}

template <int MODE>
int launch(const void* feat, const void* target, const float* lin, float* partial, const float* coef, void* grad, int32_t N,
           int64_t HW, int32_t C, int32_t blocks, hipStream_t s) {
  const int C8 = C >> 3, iter = (C8 + 15) >> 4;
  const dim3 grid((unsigned)blocks, (unsigned)N), block(LP_BLOCK);
#define GIP_LP(I)                                                                                                    \
  hipLaunchKernelGGL((lpips_layer_kernel<I, MODE>), grid, block, 0, s, (const h8*)feat, (const h8*)target, lin, partial, \
                     coef, (h8*)grad, (long long)HW, C8)
  if (iter <= 1) GIP_LP(1);
  else if (iter <= 2) GIP_LP(2);
  else if (iter <= 4) GIP_LP(4);
  else return 1;
#undef GIP_LP
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // namespace

extern "C" int32_t gip_lpips_layer_blocks(int32_t N, int64_t HW) {
  // enough workgroups to fill the chip, at least one 16-row pass each
  long long want = (2048 + N - 1) / (N > 0 ? N : 1);
  const long long by_rows = (HW + 15) / 16;
  if (want > by_rows) want = by_rows;
  return (int32_t)(want < 1 ? 1 : want);
}

extern "C" int gip_lpips_layer_forward(const void* feat, const void* target_unit, const float* lin, float* partial, int32_t N,
                                       int64_t HW, int32_t C, int32_t blocks, void* stream) {
  if (!feat || !target_unit || !lin || !partial || N < 1 || HW < 1 || C < 8 || (C & 7) || C > 512 || blocks < 1) return 1;
  return launch<0>(feat, target_unit, lin, partial, nullptr, nullptr, N, HW, C, blocks, (hipStream_t)stream);
}

extern "C" int gip_lpips_layer_backward(const void* feat, const void* target_unit, const float* lin, const float* coef,
                                        void* grad_feat, int32_t N, int64_t HW, int32_t C, void* stream) {
  if (!feat || !target_unit || !lin || !coef || !grad_feat || N < 1 || HW < 1 || C < 8 || (C & 7) || C > 512) return 1;
  return launch<1>(feat, target_unit, lin, nullptr, coef, grad_feat, N, HW, C, gip_lpips_layer_blocks(N, HW), (hipStream_t)stream);
}
