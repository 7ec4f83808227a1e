// guidance_glue.hip — the small tensor algebra AROUND the denoiser as one launch per stage (include/gip_nn.h, "guidance glue").
//
// The reference writes this algebra as chains of PyTorch element-wise ops (threestudio/models/guidance/ipa_guidance.py:
// interpolate + `imgs * 2 - 1` :602-614 / :522-531, latent_dist.sample() * scaling_factor :529, add_noise :395, the ANPG
// combination / weighting / per-"pixel" clip :411-431, nan_to_num + the detached-target MSE :645-653).  On the MI355X every
// one of those ops is a 4-6 us launch on a 64k-element tensor, ~120 of them per step on the critical path between the VAE
// encoder and the U-Net.  The kernels here evaluate the SAME expressions in the same order with the same intermediate
// roundings (a half tensor op rounds to half after every operation: H() below), so their outputs equal the op chain's
// bit for bit except where a reduction order is PyTorch's own (the row norm of the clip, the two scalar sums).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gip_nn.h"

namespace {

__device__ __forceinline__ float H(float x) { return (float)(_Float16)x; }      // round to half and back (one half tensor op)
__device__ __forceinline__ float ldh(const _Float16* p) { return (float)*p; }

struct Strides4 { int64_t b, c, h, w; };

__device__ __forceinline__ float block_sum(float v, float* s_red) {
  // fixed-order sum over the workgroup (deterministic): wave sums by shuffles, then lane 0 adds the waves in order
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; i++) t += s_red[i];
  return t;
}

// ---- ANPG combination -> weighting -> row clip -> nan_to_num -> detached-target MSE ------------------------------------------
// The clip's L2 norm runs over the LAST axis (ipa_guidance.py:427-431): one WAVE per (b, h) image row, lane = w, every lane walks
// the C channels; a channel's row norm is a wave reduction (fixed butterfly order).  Per-wave (loss, norm^2) partials, added up in
// index order by anpg_finish_kernel.  (A first version with one thread per row in a single workgroup took 330 us.)
#define ANPG_MAXC 8
__global__ void __launch_bounds__(256)
anpg_loss_kernel(const _Float16* __restrict__ noise_pred, Strides4 ns, const _Float16* __restrict__ latents, Strides4 ls,
                 const int64_t* __restrict__ t, const float* __restrict__ acp, int B, int C, int Hh, int W, float guidance_scale,
                 int t_switch, int weighting, float clip_threshold, float* __restrict__ grad_out, float* __restrict__ diff_out,
                 float* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wv >= B * Hh) return;
  const int b = wv / Hh, h = wv - b * Hh;
  const int64_t tb = t[b];
  const float a = acp[tb];
  const float wgt = weighting == 0 ? 1.f - a : (weighting == 1 ? 1.f : sqrtf(a) * (1.f - a));
  const float mf = tb < (int64_t)t_switch ? 1.f : 0.f;
  float loss = 0.f, nrm = 0.f;
  for (int w0 = 0; w0 < W; w0 += 64) {            // W <= 64 at the training shape: one trip; the row norm below needs W <= 64
    const int w = w0 + lane;
    const bool on = w < W;
    for (int c = 0; c < C; c++) {
      float g = 0.f, l32 = 0.f;
      if (on) {
        // noise_pred rows (neg | text | null) of sample b: b, B + b, 2B + b
        const _Float16* pn = noise_pred + (int64_t)b * ns.b + (int64_t)c * ns.c + (int64_t)h * ns.h + (int64_t)w * ns.w;
        const float en = ldh(pn), et = ldh(pn + (int64_t)B * ns.b), eu = ldh(pn + 2 * (int64_t)B * ns.b);
        l32 = ldh(latents + (int64_t)b * ls.b + (int64_t)c * ls.c + (int64_t)h * ls.h + (int64_t)w * ls.w);
        const float dc = H(guidance_scale * H(et - eu));                  // guidance_scale * (eps_text - eps_null)
        const float dd = H(H(mf * eu) + H((1.f - mf) * H(eu - en)));      // mask * eps_null + (1 - mask) * (eps_null - eps_neg): 0 * NaN stays NaN
        g = wgt * H(dc + dd);                                             // float32 from here on (w(t) is a float32 tensor)
      }
      if (clip_threshold > 0.f) {
        float ss = g * g;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d, 64);
        const float n = sqrtf(ss) + 1e-8f;
        g = fminf(n, clip_threshold) * g / n;                             // n.clamp(max = threshold) * grad / n
      }
      if (g != g) g = 0.f;                                                // nan_to_num
      else if (g == INFINITY) g = 3.4028234663852886e38f;
      else if (g == -INFINITY) g = -3.4028234663852886e38f;
      if (on) {
        const int64_t o = (((int64_t)b * C + c) * Hh + h) * W + w;
        const float d = l32 - (l32 - g);                                  // lat32 - target, target = (lat32 - grad).detach()
        grad_out[o] = g;
        diff_out[o] = d;
        loss += d * d;
        nrm += g * g;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { loss += __shfl_xor(loss, d, 64); nrm += __shfl_xor(nrm, d, 64); }
  if (lane == 0) { partials[2 * wv] = loss; partials[2 * wv + 1] = nrm; }
}

__global__ void __launch_bounds__(256) anpg_finish_kernel(const float* __restrict__ partials, int n, int B, float* __restrict__ scalars) {
  __shared__ float s_red[16];
  float l = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { l += partials[2 * i]; q += partials[2 * i + 1]; }
  l = block_sum(l, s_red);
  q = block_sum(q, s_red);
  if (threadIdx.x == 0) {
    scalars[0] = 0.5f * l / (float)B;
    scalars[1] = sqrtf(q);
  }
}

// out_half[i] = half(x[i] * (*scale) * mult)
__global__ void scale_cast_kernel(const float* __restrict__ x, const float* __restrict__ scale, float mult, _Float16* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (_Float16)(x[i] * (scale[0] * mult));
}

// ---- latent_dist.sample() * scaling_factor, then add_noise, tiled `replicas` times ------------------------------------------------
__global__ void latent_sample_kernel(const _Float16* __restrict__ moments, Strides4 ms, const _Float16* __restrict__ eps,
                                     const _Float16* __restrict__ noise, const int64_t* __restrict__ t, const float* __restrict__ acp,
                                     float scaling, int B, int C, int Hh, int W, int replicas, _Float16* __restrict__ latents,
                                     _Float16* __restrict__ noisy) {
  const int64_t n = (int64_t)B * C * Hh * W, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % W), h = (int)((i / W) % Hh), c = (int)((i / ((int64_t)W * Hh)) % C), b = (int)(i / ((int64_t)W * Hh * C));
  const _Float16* pm = moments + (int64_t)b * ms.b + (int64_t)h * ms.h + (int64_t)w * ms.w;
  const float mean = ldh(pm + (int64_t)c * ms.c), logvar = ldh(pm + (int64_t)(C + c) * ms.c);
  const float lv = fminf(fmaxf(logvar, -30.f), 20.f);
  const float sd = H(expf(H(0.5f * lv)));                               // torch.exp(0.5 * logvar.clamp(-30, 20))
  const float lat = H(H(mean + H(sd * (float)eps[i])) * scaling);       // (mean + std * noise) * scaling_factor
  latents[i] = (_Float16)lat;
  // add_noise with the half alphas table: sqrt(a) x0 + sqrt(1 - a) eps
  const float a = H(acp[t[b]]);
  const float sa = H(sqrtf(a)), sb = H(sqrtf(H(1.f - a)));
  const _Float16 xn = (_Float16)(H(sa * lat) + H(sb * (float)noise[i]));
  for (int r = 0; r < replicas; r++) noisy[(int64_t)r * n + i] = xn;
}

// gradient of the latents with respect to the moments (mean | logvar), in the moments' own memory layout
__global__ void latent_sample_bwd_kernel(const _Float16* __restrict__ moments, Strides4 ms, const _Float16* __restrict__ eps,
                                         const _Float16* __restrict__ g_lat, float scaling, int B, int C, int Hh, int W,
                                         _Float16* __restrict__ g_mom) {
  const int64_t n = (int64_t)B * C * Hh * W, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % W), h = (int)((i / W) % Hh), c = (int)((i / ((int64_t)W * Hh)) % C), b = (int)(i / ((int64_t)W * Hh * C));
  const int64_t om = (int64_t)b * ms.b + (int64_t)h * ms.h + (int64_t)w * ms.w;
  const float logvar = ldh(moments + om + (int64_t)(C + c) * ms.c);
  const bool inside = logvar >= -30.f && logvar <= 20.f;                // clamp's gradient mask
  const float sd = H(expf(H(0.5f * fminf(fmaxf(logvar, -30.f), 20.f))));
  const float gs = H((float)g_lat[i] * scaling);                        // d / d(mean + std * noise)
  const float gsd = H(gs * (float)eps[i]);                              // d / d std
  const float ghl = H(gsd * sd);                                        // exp backward: grad * result
  g_mom[om + (int64_t)c * ms.c] = (_Float16)gs;
  g_mom[om + (int64_t)(C + c) * ms.c] = (_Float16)(inside ? H(ghl * 0.5f) : 0.f);
}

// ---- F.interpolate(rgb, (H/2, W/2), "bilinear", align_corners=False) -> half -> * 2 - 1, as the channels-last tensor the VAE reads --
// an exact 2x reduction with align_corners=False samples at source 2 i + 0.5: weights (0.5, 0.5) per axis = the 2x2 box mean
__global__ void image_prep_kernel(const float* __restrict__ rgb, int B, int C, int Ho, int Wo, _Float16* __restrict__ out) {
  const int64_t n = (int64_t)B * Ho * Wo, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((int64_t)Wo * Ho));
  const int Wi = 2 * Wo, Hi = 2 * Ho;
  for (int c = 0; c < C; c++) {
    const float* p = rgb + (((int64_t)b * C + c) * Hi + 2 * y) * Wi + 2 * x;
    const float2 r0 = *(const float2*)p, r1 = *(const float2*)(p + Wi);
    // upsample_bilinear2d: h0lambda * (w0lambda * p00 + w1lambda * p01) + h1lambda * (w0lambda * p10 + w1lambda * p11), lambdas = 0.5
    const float v = 0.5f * (0.5f * r0.x + 0.5f * r0.y) + 0.5f * (0.5f * r1.x + 0.5f * r1.y);
    out[i * C + c] = (_Float16)(H(H(v) * 2.f) - 1.f);                   // .to(half); imgs * 2.0 - 1.0
  }
}

__global__ void image_prep_bwd_kernel(const _Float16* __restrict__ g, int B, int C, int Ho, int Wo, float* __restrict__ g_rgb) {
  const int64_t n = (int64_t)B * Ho * Wo, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((int64_t)Wo * Ho));
  const int Wi = 2 * Wo, Hi = 2 * Ho;
  for (int c = 0; c < C; c++) {
    const float v = 0.25f * H((float)g[i * C + c] * 2.f);               // mul backward in half, cast to float32, four taps of weight 1/4
    float* p = g_rgb + (((int64_t)b * C + c) * Hi + 2 * y) * Wi + 2 * x;
    *(float2*)p = make_float2(v, v);
    *(float2*)(p + Wi) = make_float2(v, v);
  }
}

// ---- diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(max_period) k / (dim/2)) ----
// (arange, two scalar ops, exp, cast, outer product, cos, sin, concatenation, cast in the op-chain spelling: ten launches per network)
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, int B, int half_dim, float neg_log_period, _Float16* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half_dim) return;
  const int b = i / half_dim, k = i - b * half_dim;
  const float freq = expf((neg_log_period * (float)k) / (float)half_dim);
  const float arg = (float)t[b] * freq;
  out[(size_t)b * 2 * half_dim + k] = (_Float16)cosf(arg);
  out[(size_t)b * 2 * half_dim + half_dim + k] = (_Float16)sinf(arg);
}

Strides4 strides(const int64_t* s) { return Strides4{s[0], s[1], s[2], s[3]}; }

}  // namespace

extern "C" int gip_anpg_loss_f16(const void* noise_pred, const int64_t* np_strides, const void* latents, const int64_t* lat_strides,
                                 const int64_t* t, const float* acp, int32_t B, int32_t C, int32_t H, int32_t W, float guidance_scale,
                                 int32_t t_switch, int32_t weighting, float clip_threshold, float* grad_out, float* diff_out,
                                 float* scalars, float* partials, void* stream) {
  if (!noise_pred || !np_strides || !latents || !lat_strides || !t || !acp || !grad_out || !diff_out || !scalars) return 1;
  if (B < 1 || C < 1 || H < 1 || W < 1 || weighting < 0 || weighting > 2) return 1;
  if (clip_threshold > 0.f && W > 64) return 1;                 // the row norm is one wave reduction
  if (!partials) return 1;
  const int waves = B * H;
  hipLaunchKernelGGL(anpg_loss_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const _Float16*)noise_pred,
                     strides(np_strides), (const _Float16*)latents, strides(lat_strides), t, acp, B, C, H, W, guidance_scale, t_switch,
                     weighting, clip_threshold, grad_out, diff_out, partials);
  hipLaunchKernelGGL(anpg_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, waves, B, scalars);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_scale_cast_f16(const float* x, const float* scale, float mult, void* out, int64_t n, void* stream) {
  if (!x || !scale || !out || n < 1) return 1;
  hipLaunchKernelGGL(scale_cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale, mult, (_Float16*)out, n);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_latent_sample_f16(const void* moments, const int64_t* m_strides, const void* eps, const void* noise, const int64_t* t,
                                     const float* acp, float scaling, int32_t B, int32_t C, int32_t H, int32_t W, int32_t replicas,
                                     void* latents, void* noisy, void* stream) {
  if (!moments || !m_strides || !eps || !noise || !t || !acp || !latents || !noisy || B < 1 || C < 1 || H < 1 || W < 1 || replicas < 1) return 1;
  const int64_t n = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(latent_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)moments,
                     strides(m_strides), (const _Float16*)eps, (const _Float16*)noise, t, acp, scaling, B, C, H, W, replicas,
                     (_Float16*)latents, (_Float16*)noisy);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_latent_sample_backward_f16(const void* moments, const int64_t* m_strides, const void* eps, const void* g_latents,
                                              float scaling, int32_t B, int32_t C, int32_t H, int32_t W, void* g_moments, void* stream) {
  if (!moments || !m_strides || !eps || !g_latents || !g_moments || B < 1 || C < 1 || H < 1 || W < 1) return 1;
  const int64_t n = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(latent_sample_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)moments,
                     strides(m_strides), (const _Float16*)eps, (const _Float16*)g_latents, scaling, B, C, H, W, (_Float16*)g_moments);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_image_prep_f16(const float* rgb, int32_t B, int32_t C, int32_t Hout, int32_t Wout, void* out, void* stream) {
  if (!rgb || !out || B < 1 || C < 1 || Hout < 1 || Wout < 1) return 1;
  const int64_t n = (int64_t)B * Hout * Wout;
  hipLaunchKernelGGL(image_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb, B, C, Hout, Wout, (_Float16*)out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_image_prep_backward_f16(const void* g_out, int32_t B, int32_t C, int32_t Hout, int32_t Wout, float* g_rgb, void* stream) {
  if (!g_out || !g_rgb || B < 1 || C < 1 || Hout < 1 || Wout < 1) return 1;
  const int64_t n = (int64_t)B * Hout * Wout;
  hipLaunchKernelGGL(image_prep_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)g_out, B, C,
                     Hout, Wout, g_rgb);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_timestep_embedding_f16(const int64_t* t, int32_t B, int32_t dim, float max_period, void* out, void* stream) {
  if (!t || !out || B < 1 || dim < 2 || (dim & 1) || !(max_period > 1.f)) return 1;
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, B, dim / 2,
                     (float)(-log((double)max_period)), (_Float16*)out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
