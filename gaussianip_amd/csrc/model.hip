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
