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


// ---- the sparsity term of the stage-1 loss: mean(sqrt((depth / (max(depth) + 1e-5))^2 + 0.01))  (GaussianIP.py:225, :377-380) -------
// The reference spells it as max -> add -> div -> pow -> add -> sqrt -> mean on the [B, H, W, 1] depth maps (4 M elements at the
// training shape): seven forward and about twice as many backward launches.  Here: three launches forward, two backward, fixed
// summation orders (bitwise reproducible), no inter-workgroup fences (a first version that let the last workgroup of a launch add up
// the partials behind `__threadfence()` took 130 us per kernel: an agent-scope release writes the L2 back, once per workgroup).
// Every workgroup of a consumer kernel re-derives the scalar it needs from the producer's per-workgroup partials (4 KB, same order
// everywhere).  `ws` (float32): [0] max, [1] loss, [2] count of maxima, [4, 4 + SP_BLOCKS) maxima / backward sums per workgroup,
// [4 + SP_BLOCKS, 4 + 3 SP_BLOCKS) (loss sum, tie count as uint32 bits) per workgroup.  A NaN depth propagates into the maximum
// (and so into the loss and every gradient) like torch.max(); the tie count is integer arithmetic until its one final rounding.
#define SP_BLOCKS 1024
#define SP_THREADS 256

// maximum that PROPAGATES NaN like torch.max() (fmaxf alone drops it: a poisoned depth map must poison the loss, as in the op chain)
__device__ __forceinline__ float sp_max(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

__device__ __forceinline__ float sp_block_reduce(float v, bool is_max, float* s_red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float o = __shfl_xor(v, d, 64);
    v = is_max ? sp_max(v, o) : v + o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = s_red[0];
  for (int i = 1; i < SP_THREADS / 64; i++) t = is_max ? sp_max(t, s_red[i]) : t + s_red[i];
  return t;
}

// the tie count of the maximum is an INTEGER all the way (a float32 count stops growing at 2^24 tied elements — an all-zero or
// saturated batch of 4 x 2048^2 depths — and the maximum's gradient share would be divided by a rounded count)
__device__ __forceinline__ unsigned long long sp_block_count(unsigned long long v, unsigned long long* s_cnt) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long t = s_cnt[0];
  for (int i = 1; i < SP_THREADS / 64; i++) t += s_cnt[i];
  return t;
}

// the same value in every workgroup: partial[0 .. nb) combined in one fixed order
__device__ __forceinline__ float sp_combine(const float* __restrict__ part, int nb, int stride, bool is_max, float* s_red) {
  float t = is_max ? -INFINITY : 0.f;
  for (int i = threadIdx.x; i < nb; i += SP_THREADS) t = is_max ? sp_max(t, part[(size_t)i * stride]) : t + part[(size_t)i * stride];
  return sp_block_reduce(t, is_max, s_red);
}

__global__ void __launch_bounds__(SP_THREADS) sp_max_kernel(const float* __restrict__ d, int64_t n, float* __restrict__ ws) {
  __shared__ float s_red[SP_THREADS / 64];
  float m = -INFINITY;
  bool nan = false;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const float x = d[i];
    m = fmaxf(m, x);
    nan |= x != x;
  }
  if (nan) m = __builtin_nanf("");
  m = sp_block_reduce(m, true, s_red);
  if (threadIdx.x == 0) ws[4 + blockIdx.x] = m;
}

__global__ void __launch_bounds__(SP_THREADS) sp_loss_kernel(const float* __restrict__ d, int64_t n, float* __restrict__ ws) {
  __shared__ float s_red[SP_THREADS / 64];
  __shared__ unsigned long long s_cnt[SP_THREADS / 64];
  const float dmax = sp_combine(ws + 4, (int)gridDim.x, 1, true, s_red), den = dmax + 1e-5f;
  float sum = 0.f;
  uint32_t cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const float x = d[i], o = x / den;
    sum += sqrtf(o * o + 0.01f);
    cnt += x == dmax ? 1u : 0u;
  }
  sum = sp_block_reduce(sum, false, s_red);
  const unsigned long long c = sp_block_count(cnt, s_cnt);
  float* part = ws + 4 + SP_BLOCKS;
  // a workgroup sees at most n / gridDim.x + 256 elements: its count fits 32 bits for every n < 2^41; stored as integer bits
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = sum; reinterpret_cast<uint32_t*>(part)[2 * blockIdx.x + 1] = (uint32_t)c; }
}

__global__ void __launch_bounds__(SP_THREADS) sp_finish_kernel(int nb, int64_t n, float* __restrict__ ws) {
  __shared__ float s_red[SP_THREADS / 64];
  __shared__ unsigned long long s_cnt[SP_THREADS / 64];
  const float dmax = sp_combine(ws + 4, nb, 1, true, s_red);
  const float sum = sp_combine(ws + 4 + SP_BLOCKS, nb, 2, false, s_red);
  const uint32_t* cp = reinterpret_cast<const uint32_t*>(ws + 4 + SP_BLOCKS);
  unsigned long long c = 0;
  for (int i = threadIdx.x; i < nb; i += SP_THREADS) c += cp[2 * i + 1];
  c = sp_block_count(c, s_cnt);
  // ws[2]: the exact integer count, rounded to float ONCE (it only ever divides the maximum's gradient share)
  if (threadIdx.x == 0) { ws[0] = dmax; ws[1] = sum / (float)n; ws[2] = (float)c; }
}

// pass 1 of the backward: g_d[i] = (g / n) (o / sqrt(o^2 + 0.01)) / den; per workgroup the sum of (d loss / d o_i) o_i, whose
// negative over den is what reaches the maximum through the denominator
__global__ void __launch_bounds__(SP_THREADS) sp_bwd_kernel(const float* __restrict__ d, int64_t n, const float* __restrict__ g_loss,
                                                              float mult, float* __restrict__ ws, float* __restrict__ g_d) {
  __shared__ float s_red[SP_THREADS / 64];
  const float dmax = ws[0], den = dmax + 1e-5f, gn = g_loss[0] * mult / (float)n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    const float o = d[i] / den;
    const float go = gn * (o / sqrtf(o * o + 0.01f));          // d loss / d opacity_i
    g_d[i] = go / den;
    acc += go * o;
  }
  acc = sp_block_reduce(acc, false, s_red);
  if (threadIdx.x == 0) ws[4 + blockIdx.x] = acc;
}

// pass 2: the maximum's share, split evenly among the elements that equal it (torch.max()'s backward)
__global__ void __launch_bounds__(SP_THREADS) sp_bwd_max_kernel(const float* __restrict__ d, int64_t n, const float* __restrict__ ws,
                                                                  float* __restrict__ g_d) {
  __shared__ float s_red[SP_THREADS / 64];
  const float dmax = ws[0], den = dmax + 1e-5f;
  const float add = -(sp_combine(ws + 4, (int)gridDim.x, 1, false, s_red) / den) / ws[2];
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS)
    if (d[i] == dmax) g_d[i] += add;
}

static int sp_blocks(int64_t n) { const int64_t b = (n + SP_THREADS - 1) / SP_THREADS; return (int)(b < SP_BLOCKS ? b : SP_BLOCKS); }

extern "C" size_t gip_sparsity_workspace_bytes(void) { return (4 + 3 * SP_BLOCKS) * sizeof(float); }

extern "C" int gip_sparsity_loss_forward(const float* depth, int64_t n, void* workspace, void* stream) {
  if (!depth || !workspace || n < 1) return 1;
  const int blocks = sp_blocks(n);
  float* ws = (float*)workspace;
  hipLaunchKernelGGL(sp_max_kernel, dim3(blocks), dim3(SP_THREADS), 0, (hipStream_t)stream, depth, n, ws);
  hipLaunchKernelGGL(sp_loss_kernel, dim3(blocks), dim3(SP_THREADS), 0, (hipStream_t)stream, depth, n, ws);
  hipLaunchKernelGGL(sp_finish_kernel, dim3(1), dim3(SP_THREADS), 0, (hipStream_t)stream, blocks, n, ws);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_sparsity_loss_backward(const float* depth, int64_t n, const float* g_loss, float mult, void* workspace, float* g_depth,
                                          void* stream) {
  if (!depth || !workspace || !g_loss || !g_depth || n < 1) return 1;
  const int blocks = sp_blocks(n);
  float* ws = (float*)workspace;
  hipLaunchKernelGGL(sp_bwd_kernel, dim3(blocks), dim3(SP_THREADS), 0, (hipStream_t)stream, depth, n, g_loss, mult, ws, g_depth);
  hipLaunchKernelGGL(sp_bwd_max_kernel, dim3(blocks), dim3(SP_THREADS), 0, (hipStream_t)stream, depth, n, ws, g_depth);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ---- the three parameter activations of GaussianModel (gaussian_model.py:36-41, getters :72-89) in one launch each way ---------------
// opacity = sigmoid(o), scaling = exp(s), rotation = q / max(||q||, 1e-12): three op chains forward (sigmoid; exp; norm, clamp,
// div) and about a dozen launches backward in the reference's spelling; one thread per Gaussian here.
__global__ void activate_kernel(const float* __restrict__ o, const float* __restrict__ sc, const float* __restrict__ q, int64_t P,
                                float* __restrict__ oo, float* __restrict__ so, float* __restrict__ qo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  oo[i] = 1.f / (1.f + expf(-o[i]));
#pragma unroll
  for (int k = 0; k < 3; k++) so[3 * i + k] = expf(sc[3 * i + k]);
  const float4 v = *(const float4*)(q + 4 * i);
  const float n = sqrtf(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w), den = fmaxf(n, 1e-12f);
  *(float4*)(qo + 4 * i) = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
}

__global__ void activate_bwd_kernel(const float* __restrict__ oo, const float* __restrict__ so, const float* __restrict__ q,
                                    const float* __restrict__ g_o, const float* __restrict__ g_s, const float* __restrict__ g_q, int64_t P,
                                    float* __restrict__ d_o, float* __restrict__ d_s, float* __restrict__ d_q) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  if (d_o) { const float y = oo[i]; d_o[i] = g_o ? (g_o[i] * (1.f - y)) * y : 0.f; }                       // sigmoid_backward
  if (d_s) {
#pragma unroll
    for (int k = 0; k < 3; k++) d_s[3 * i + k] = g_s ? g_s[3 * i + k] * so[3 * i + k] : 0.f;              // exp: grad * result
  }
  if (d_q) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_q) {
      const float4 v = *(const float4*)(q + 4 * i), g = *(const float4*)(g_q + 4 * i);
      const float n = sqrtf(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w), den = fmaxf(n, 1e-12f);
      // y = v / den: d v = g / den + [n >= eps] * (-(g . v) / den^2) * v / n   (div, clamp_min and norm backward in autograd's order)
      const float gden = -(((g.x * v.x + g.y * v.y) + g.z * v.z) + g.w * v.w) / (den * den);
      const float k = n >= 1e-12f && n > 0.f ? gden / n : 0.f;
      r = make_float4(g.x / den + k * v.x, g.y / den + k * v.y, g.z / den + k * v.z, g.w / den + k * v.w);
    }
    *(float4*)(d_q + 4 * i) = r;
  }
}

extern "C" int gip_activate_gaussians(const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, int64_t P,
                                      float* opacity, float* scaling, float* rotation, void* stream) {
  if (!opacity_raw || !scaling_raw || !rotation_raw || !opacity || !scaling || !rotation || P < 1) return 1;
  hipLaunchKernelGGL(activate_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, opacity_raw, scaling_raw,
                     rotation_raw, P, opacity, scaling, rotation);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_activate_gaussians_backward(const float* opacity, const float* scaling, const float* rotation_raw, const float* g_opacity,
                                               const float* g_scaling, const float* g_rotation, int64_t P, float* d_opacity_raw,
                                               float* d_scaling_raw, float* d_rotation_raw, void* stream) {
  if (!opacity || !scaling || !rotation_raw || P < 1) return 1;
  hipLaunchKernelGGL(activate_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, opacity, scaling,
                     rotation_raw, g_opacity, g_scaling, g_rotation, P, d_opacity_raw, d_scaling_raw, d_rotation_raw);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}


// ---- densification statistics of a step (GaussianIP.on_before_optimizer_step :451-457 + gaussian_model.py:420-422) in one launch ---------
//   grad = sum_v viewspace_grad[v]           (V = 1: the exchange of a multi-GPU step already summed it)
//   max_radii2D[vis] = max(max_radii2D, radii)[vis];  xyz_gradient_accum += ||grad[:, :2]|| * vis;  denom += vis
// the reference's chain: a view sum, a cast, max, where, slice + norm, mask cast, mul and two in-place adds.
__global__ void densify_stats_kernel(const float* __restrict__ vgrad, int V, int64_t P, const uint8_t* __restrict__ vis,
                                     const int32_t* __restrict__ radii, float* __restrict__ max_radii, float* __restrict__ accum,
                                     float* __restrict__ denom) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float gx = 0.f, gy = 0.f;
  for (int v = 0; v < V; v++) {
    const float* g = vgrad + ((int64_t)v * P + i) * 3;
    gx = v == 0 ? g[0] : gx + g[0];
    gy = v == 0 ? g[1] : gy + g[1];
  }
  const float m = vis[i] ? 1.f : 0.f;
  if (vis[i]) max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
  accum[i] += sqrtf(gx * gx + gy * gy) * m;
  denom[i] += m;
}

extern "C" int gip_densify_stats(const float* viewspace_grad, int32_t V, int64_t P, const uint8_t* visible, const int32_t* radii,
                                 float* max_radii2D, float* xyz_gradient_accum, float* denom, void* stream) {
  if (!viewspace_grad || !visible || !radii || !max_radii2D || !xyz_gradient_accum || !denom || V < 1 || P < 1) return 1;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, viewspace_grad, V, P, visible,
                     radii, max_radii2D, xyz_gradient_accum, denom);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
