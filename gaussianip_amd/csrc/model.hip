// model.hip — multi-tensor row gather for densify / prune (include/gip_model.h).  HBM-bound byte movement: one launch,
// blockIdx.y = tensor, 4-byte words, consecutive lanes on consecutive words of the output (coalesced stores; loads are
// coalesced within a row and rows are mostly consecutive because survivors keep their order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gip_model.h"

struct GatherArgs {
  GipGatherTensor t[GIP_GATHER_MAX_TENSORS];
};

__global__ void __launch_bounds__(256)
gip_gather_rows_kernel(GatherArgs args, const int64_t* __restrict__ index, long long n_out, long long n_old) {
  const GipGatherTensor t = args.t[blockIdx.y];
  const int wpr = t.row_bytes >> 2;
  const long long total = n_out * wpr;
  const uint32_t* __restrict__ old_rows = (const uint32_t*)t.old_rows;
  const uint32_t* __restrict__ new_rows = (const uint32_t*)t.new_rows;
  uint32_t* __restrict__ dst = (uint32_t*)t.dst;
  for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < total; w += (long long)gridDim.x * 256) {
    const long long j = w / wpr;
    const int c = (int)(w - j * wpr);
    const long long src = index[j];
    uint32_t v = 0u;
    if (src < n_old) v = old_rows[src * wpr + c];
    else if (new_rows) v = new_rows[(src - n_old) * wpr + c];
    dst[w] = v;
  }
}

extern "C" int gip_gather_rows(const GipGatherTensor* tensors, int32_t n_tensors, const int64_t* index, int64_t n_out,
                               int64_t n_old, void* stream) {
  if (!tensors || n_tensors < 1 || n_tensors > GIP_GATHER_MAX_TENSORS || n_out < 0 || n_old < 0) return 1;
  if (n_out == 0) return 0;
  if (!index) return 1;
  GatherArgs a;
  int max_wpr = 1;
  for (int i = 0; i < n_tensors; i++) {
    if (!tensors[i].dst || (n_old > 0 && !tensors[i].old_rows) || tensors[i].row_bytes < 4 || (tensors[i].row_bytes & 3)) return 1;
    a.t[i] = tensors[i];
    if ((tensors[i].row_bytes >> 2) > max_wpr) max_wpr = tensors[i].row_bytes >> 2;
  }
  for (int i = n_tensors; i < GIP_GATHER_MAX_TENSORS; i++) a.t[i] = tensors[0];
  long long blocks = (n_out * max_wpr + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gip_gather_rows_kernel, dim3((unsigned)blocks, n_tensors), dim3(256), 0, (hipStream_t)stream, a, index,
                     (long long)n_out, (long long)n_old);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}


// ---------------------------------------------------------------------------------------------------------------
// Exchange bucket: pack the gradient tensors (+ the view-space gradient norms, computed here) into one flat buffer and
// scatter them back after the all-reduce.  A workgroup row (blockIdx.y) per segment; float4 where the alignment allows.
// ---------------------------------------------------------------------------------------------------------------
struct PackArgs {
  float* seg[GIP_PACK_MAX_SEGS + 1];
  long long count[GIP_PACK_MAX_SEGS + 1];
  long long offset[GIP_PACK_MAX_SEGS + 1];
};

template <bool UNPACK>
__global__ void __launch_bounds__(256)
gip_bucket_kernel(PackArgs a, int n_segs, const float* __restrict__ g2d, int V, long long P, float* __restrict__ flat, float scale) {
  const int sidx = blockIdx.y;
  const long long n = a.count[sidx];
  float* __restrict__ seg = a.seg[sidx];
  float* __restrict__ fl = flat + a.offset[sidx];
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nthreads = (long long)gridDim.x * 256;
  if (!UNPACK && sidx == n_segs) {                       // the tail: sum over the local views of |grad_xy|
    for (long long p = tid; p < P; p += nthreads) {
      float acc = 0.f;
      for (int v = 0; v < V; v++) {
        const float gx = g2d[((long long)v * P + p) * 3], gy = g2d[((long long)v * P + p) * 3 + 1];
        acc += sqrtf(gx * gx + gy * gy);
      }
      fl[p] = acc;
    }
    return;
  }
  const float sc = (UNPACK && sidx < n_segs) ? scale : 1.f;
  const bool vec = ((((uintptr_t)seg) | ((uintptr_t)fl)) & 15) == 0;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long i = tid; i < n4; i += nthreads) {
      if (UNPACK) { float4 x = ((const float4*)fl)[i]; x.x *= sc; x.y *= sc; x.z *= sc; x.w *= sc; ((float4*)seg)[i] = x; }
      else ((float4*)fl)[i] = ((const float4*)seg)[i];
    }
    for (long long i = (n4 << 2) + tid; i < n; i += nthreads) { if (UNPACK) seg[i] = fl[i] * sc; else fl[i] = seg[i]; }
  } else {
    for (long long i = tid; i < n; i += nthreads) { if (UNPACK) seg[i] = fl[i] * sc; else fl[i] = seg[i]; }
  }
}

static int bucket_args(void* const* segs, const int64_t* counts, int32_t n_segs, PackArgs* a, long long* total, long long* biggest) {
  if (!segs || !counts || n_segs < 0 || n_segs > GIP_PACK_MAX_SEGS) return 1;
  long long off = 0, big = 0;
  for (int i = 0; i < n_segs; i++) {
    if (counts[i] < 0 || (counts[i] > 0 && !segs[i])) return 1;
    a->seg[i] = (float*)segs[i]; a->count[i] = counts[i]; a->offset[i] = off;
    off += counts[i];
    if (counts[i] > big) big = counts[i];
  }
  *total = off; *biggest = big;
  return 0;
}

extern "C" int gip_pack_bucket(const void* const* segs, const int64_t* counts, int32_t n_segs, const void* g2d, int32_t V,
                               int64_t P, void* flat, void* stream) {
  PackArgs a;
  long long total = 0, big = 0;
  if (!flat || bucket_args((void* const*)segs, counts, n_segs, &a, &total, &big)) return 1;
  if (g2d && (V < 1 || P < 0)) return 1;
  const int rows = n_segs + (g2d ? 1 : 0);
  if (rows == 0) return 0;
  a.seg[n_segs] = nullptr; a.count[n_segs] = g2d ? P : 0; a.offset[n_segs] = total;
  if (g2d && P > big) big = P;
  long long blocks = ((big >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL((gip_bucket_kernel<false>), dim3((unsigned)blocks, rows), dim3(256), 0, (hipStream_t)stream, a, n_segs,
                     (const float*)g2d, V, (long long)P, (float*)flat, 1.f);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_unpack_bucket(void* const* segs, const int64_t* counts, int32_t n_segs, void* tail_dst, int64_t tail_count,
                                 const void* flat, float scale, void* stream) {
  PackArgs a;
  long long total = 0, big = 0;
  if (!flat || bucket_args(segs, counts, n_segs, &a, &total, &big)) return 1;
  if (tail_count < 0 || (tail_count > 0 && !tail_dst)) return 1;
  const bool tail = tail_dst && tail_count > 0;
  const int rows = n_segs + (tail ? 1 : 0);
  if (rows == 0) return 0;
  a.seg[n_segs] = (float*)tail_dst; a.count[n_segs] = tail ? tail_count : 0; a.offset[n_segs] = total;
  if (tail && tail_count > big) big = tail_count;
  long long blocks = ((big >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL((gip_bucket_kernel<true>), dim3((unsigned)blocks, rows), dim3(256), 0, (hipStream_t)stream, a, n_segs,
                     (const float*)nullptr, 0, 0ll, const_cast<float*>((const float*)flat), scale);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}


// ---------------------------------------------------------------------------------------------------------------
// MAX bucket: radii maximum over the local views + depth maximum (as its int32 bit pattern, atomicMax on a zeroed slot)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gip_max_bucket_kernel(const int32_t* __restrict__ radii, int V, long long P, const float4* __restrict__ depth4,
                      const float* __restrict__ depth, long long n_depth, int32_t* __restrict__ out) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nthreads = (long long)gridDim.x * 256;
  for (long long p = tid; p < P; p += nthreads) {
    int32_t m = radii[p];
    for (int v = 1; v < V; v++) { const int32_t r = radii[(long long)v * P + p]; m = r > m ? r : m; }
    out[p] = m;
  }
  float mx = 0.f;
  const long long n4 = depth4 ? n_depth >> 2 : 0;
  for (long long i = tid; i < n4; i += nthreads) {
    const float4 d = depth4[i];
    mx = fmaxf(fmaxf(mx, fmaxf(d.x, d.y)), fmaxf(d.z, d.w));
  }
  for (long long i = (n4 << 2) + tid; i < n_depth; i += nthreads) mx = fmaxf(mx, depth[i]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __shared__ float s_mx[4];
  if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    if (m > 0.f) atomicMax(out + P, __float_as_int(m));
  }
}

extern "C" int gip_max_bucket(const int32_t* radii, int32_t V, int64_t P, const float* depth, int64_t n_depth, int32_t* out,
                              void* stream) {
  if (!out || P < 0 || V < 1 || (P > 0 && !radii) || n_depth < 0 || (n_depth > 0 && !depth)) return 1;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out + P, 0, sizeof(int32_t), s) != hipSuccess) return 3;
  long long work = P > (n_depth >> 2) ? P : (n_depth >> 2);
  long long blocks = (work + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  const bool vec = (((uintptr_t)depth) & 15) == 0;
  hipLaunchKernelGGL(gip_max_bucket_kernel, dim3((unsigned)blocks), dim3(256), 0, s, radii, (int)V, (long long)P,
                     vec ? (const float4*)depth : (const float4*)nullptr, depth, (long long)n_depth, out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}



// ---------------------------------------------------------------------------------------------------------------------
// gip_adam_step: all parameter groups in one launch (include/gip_model.h)
// ---------------------------------------------------------------------------------------------------------------------
struct AdamArgs {
  GipAdamGroup g[GIP_ADAM_MAX_GROUPS];
  long long start[GIP_ADAM_MAX_GROUPS + 1];     // element offsets of the groups in the flat index space
  int n_groups;
  float beta1, beta2, eps;
  float w1, w2;                                  // 1 - beta1, 1 - beta2 rounded from DOUBLE like torch's host-side constants
  double beta1d, beta2d;
};

__global__ void __launch_bounds__(256)
gip_adam_kernel(AdamArgs a, const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;                         // GradScaler: skipped step, nothing moves
  // bias corrections per group, in double like torch's `1 - beta ** step` on Python floats; one thread per group
  __shared__ float s_bc1[GIP_ADAM_MAX_GROUPS], s_bc2s[GIP_ADAM_MAX_GROUPS];
  if (threadIdx.x < a.n_groups) {
    const double t = (double)*a.g[threadIdx.x].step + 1.0;            // every block reads the OLD count; gip_adam_count_kernel stores the new one
    s_bc1[threadIdx.x] = (float)(1.0 - pow(a.beta1d, t));
    s_bc2s[threadIdx.x] = (float)sqrt(1.0 - pow(a.beta2d, t));
  }
  __syncthreads();
  const long long total = a.start[a.n_groups];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < GIP_ADAM_MAX_GROUPS; j++) k += (j < a.n_groups && i >= a.start[j]) ? 1 : 0;
    const GipAdamGroup& G = a.g[k];
    const long long e = i - a.start[k];
    const float g = ((const float*)G.grad)[e];
    float m = ((float*)G.exp_avg)[e], v = ((float*)G.exp_avg_sq)[e];
    m = m + a.w1 * (g - m);                                           // exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + a.w2 * g * g;                                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / s_bc2s[k] + a.eps;
    float* p = (float*)G.param;
    p[e] = p[e] - (G.lr / s_bc1[k]) * (m / denom);                    // param.addcdiv_(exp_avg, denom, value = -step_size)
    ((float*)G.exp_avg)[e] = m;
    ((float*)G.exp_avg_sq)[e] = v;
  }
}

// the step counters are advanced by a second, one-thread-per-group launch AFTER the update kernel has read them (same stream)
__global__ void gip_adam_count_kernel(AdamArgs a, const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;
  if (threadIdx.x < a.n_groups) *a.g[threadIdx.x].step += 1.f;
}

extern "C" int gip_adam_step(const GipAdamGroup* groups, int32_t n_groups, double beta1, double beta2, double eps, const float* found_inf,
                             void* stream) {
  if (!groups || n_groups < 1 || n_groups > GIP_ADAM_MAX_GROUPS) return 1;
  if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0)) return 1;
  AdamArgs a;
  // betas / eps arrive as the caller's DOUBLES (Python floats): 1 - beta and the bias corrections are formed in double
  // exactly like torch's host-side constants, whatever the number of significant digits of the betas
  a.n_groups = n_groups; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
  a.beta1d = beta1; a.beta2d = beta2;
  a.w1 = (float)(1.0 - a.beta1d); a.w2 = (float)(1.0 - a.beta2d);
  long long off = 0;
  for (int i = 0; i < n_groups; i++) {
    if (!groups[i].param || !groups[i].grad || !groups[i].exp_avg || !groups[i].exp_avg_sq || !groups[i].step || groups[i].n < 0) return 1;
    a.g[i] = groups[i];
    a.start[i] = off;
    off += groups[i].n;
  }
  for (int i = n_groups; i <= GIP_ADAM_MAX_GROUPS; i++) a.start[i] = off;
  if (off == 0) return 0;
  long long blocks = (off + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gip_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, found_inf);
  hipLaunchKernelGGL(gip_adam_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, found_inf);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
