// binning.hip — tile binning for gfx950: offsets scan, bucket fill, per-tile depth sort.
//
// Replaces the fork's InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs(u64 keys) +
// identifyTileRanges (SURVEY.md §2.1 "fwd 2-5").  The reference sorts R = sum(tiles_touched) pairs
// globally on 32+log2(T) bits; the result it needs is, per tile, the list of Gaussian indices ordered
// by (depth, index).  MI355X-first formulation (no global multi-pass radix sort, no host read-back):
//
//   1. tile histogram               (done inside preprocess, integer atomics)
//   2. gip_scan_kernel              exclusive scan of the V*T tile counts -> ranges; scan of the
//                                   per-workgroup tiles_touched sums -> instance offsets; header
//   3. gip_scatter_kernel           every (Gaussian, tile) instance takes a slot in its tile's bucket
//                                   (one returning integer atomic) and stores key = depth_bits<<32 | index
//   4. gip_tile_sort_kernel         one workgroup per tile sorts its bucket in LDS (bitonic network on
//                                   u64, all-ascending "flip" form so any length works without padding)
//
// Keys are unique, so the sorted order — and therefore every downstream buffer — is independent of
// the order in which the atomics resolved: the tile / index buffers are deterministic and equal to
// the reference's (tile | depth) stable radix sort (ties on depth resolved by Gaussian index).
#include "gip_internal.h"

// ------------------------------------------------------------------------------------------------
// scan: one workgroup of 1024 threads, each thread owns a contiguous chunk
// ------------------------------------------------------------------------------------------------
#define SCAN_THREADS 1024

__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* s_wave /*[16]*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = gip_wave_incl_scan_u32(v);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    uint32_t x = s_wave[w];
    if (w < wave) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__device__ void scan_array(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n, uint32_t* s_wave,
                           uint32_t* total_out, uint32_t* max_out) {
  const int chunk = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  const int lo = threadIdx.x * chunk, hi = min(n, lo + chunk);
  uint32_t sum = 0, mx = 0;
  for (int i = lo; i < hi; i++) { uint32_t x = in[i]; sum += x; mx = x > mx ? x : mx; }
  uint32_t total;
  uint32_t run = block_excl_scan_1024(sum, s_wave, &total);
  for (int i = lo; i < hi; i++) { uint32_t x = in[i]; out[i] = run; run += x; }
  if (threadIdx.x == 0) out[n] = total;
  *total_out = total;
  if (max_out) {
    mx = gip_wave_max_u32(mx);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = mx;
    __syncthreads();
    uint32_t m = 0;
    for (int w = 0; w < SCAN_THREADS / 64; w++) m = s_wave[w] > m ? s_wave[w] : m;
    __syncthreads();
    *max_out = m;
  }
}

// Launch order for the per-tile kernels: tiles bucketed by floor(log2(count)) and emitted longest
// bucket first, so the long lists start early and the short ones fill the tail.  Order inside a bucket
// is arbitrary (it only decides which workgroup id renders which tile; results do not depend on it).
__device__ void heavy_first_order(const uint32_t* __restrict__ counts, uint32_t* __restrict__ order, int n,
                                  uint32_t* s_bucket /*[33]*/) {
  for (int i = threadIdx.x; i < 33; i += SCAN_THREADS) s_bucket[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
    const uint32_t c = counts[i];
    atomicAdd(&s_bucket[c ? 32 - __clz(c) : 0], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int b = 32; b >= 0; b--) { uint32_t c = s_bucket[b]; s_bucket[b] = run; run += c; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
    const uint32_t c = counts[i];
    order[atomicAdd(&s_bucket[c ? 32 - __clz(c) : 0], 1u)] = (uint32_t)i;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
gip_scan_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_start,
                const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_offset,
                uint32_t* __restrict__ tile_order, GipRasterHeader* __restrict__ header) {
  __shared__ uint32_t s_wave[SCAN_THREADS / 64];
  __shared__ uint32_t s_bucket[33];
  uint32_t total_tiles, max_tile, total_inst;
  scan_array(tile_count, tile_start, kp.V * kp.T, s_wave, &total_tiles, &max_tile);
  scan_array(block_sums, block_offset, kp.V * kp.nblk, s_wave, &total_inst, nullptr);
  heavy_first_order(tile_count, tile_order, kp.V * kp.T, s_bucket);
  if (threadIdx.x == 0) {
    header->abi_version = GIP_ABI_VERSION;
    header->num_rendered = total_tiles;   // == total_inst
    header->overflow = (total_tiles > kp.capacity) ? 1u : 0u;
    header->max_tile_count = max_tile;
  }
}

void gip_launch_scan(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  hipLaunchKernelGGL(gip_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, kp, st.tile_count, st.tile_start,
                     st.block_sums, st.block_offset, st.tile_order, st.header);
}

// ------------------------------------------------------------------------------------------------
// scatter: fill the tile buckets
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GIP_BLOCK)
gip_scatter_kernel(GipKernelParams kp, const GipRecord* __restrict__ records, const uint32_t* __restrict__ tile_start,
                   uint32_t* __restrict__ tile_cursor, const uint32_t* __restrict__ block_offset,
                   uint32_t* __restrict__ inst_offset, unsigned long long* __restrict__ keys) {
  const int v = blockIdx.y;
  const int idx = blockIdx.x * GIP_BLOCK + threadIdx.x;
  uint32_t tiles = 0, rmin = 0, rmax = 0, dbits = 0;
  if (idx < kp.P) {
    const uint4* rp = reinterpret_cast<const uint4*>(records + (size_t)v * kp.P + idx);
    const uint4 q0 = rp[0], q1 = rp[1], q3 = rp[3];
    dbits = q0.z; tiles = q1.w; rmin = q3.x; rmax = q3.y;
  }
  // exclusive prefix of tiles_touched inside the workgroup
  __shared__ uint32_t s_wave[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = gip_wave_incl_scan_u32(tiles);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t base = block_offset[(size_t)v * kp.nblk + blockIdx.x];
  for (int w = 0; w < wave; w++) base += s_wave[w];
  if (idx < kp.P) inst_offset[(size_t)v * kp.P + idx] = base + incl - tiles;
  if (tiles == 0) return;
  const int rminx = rmin & 0xffff, rminy = rmin >> 16, rmaxx = rmax & 0xffff, rmaxy = rmax >> 16;
  const unsigned long long key = ((unsigned long long)dbits << 32) | (uint32_t)idx;
  const size_t tbase = (size_t)v * kp.T;
  for (int ty = rminy; ty < rmaxy; ty++)
    for (int tx = rminx; tx < rmaxx; tx++) {
      const size_t t = tbase + ty * kp.tiles_x + tx;
      const uint32_t slot = atomicAdd(&tile_cursor[t], 1u);
      const uint32_t pos = tile_start[t] + slot;
      if (pos < kp.capacity) keys[pos] = key;
    }
}

void gip_launch_scatter(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  hipLaunchKernelGGL(gip_scatter_kernel, dim3(kp.nblk, kp.V), dim3(GIP_BLOCK), 0, s, kp, st.records, st.tile_start,
                     st.tile_cursor, st.block_offset, st.inst_offset, st.keys);
}

// ------------------------------------------------------------------------------------------------
// per-tile sort.  All-ascending bitonic network ("flip" first sub-stage): comparators only ever move
// the larger element to the higher index, so virtual +inf padding at indices >= n never moves and
// comparators that touch it can simply be skipped -> correct for any n.
// ------------------------------------------------------------------------------------------------
template <typename PtrT>
__device__ __forceinline__ void bitonic_any_n(PtrT a, uint32_t n) {
  uint32_t m = 1;
  while (m < n) m <<= 1;
  const uint32_t half = m >> 1;
  for (uint32_t k = 2; k <= m; k <<= 1) {
    const uint32_t hk = k >> 1;
    for (uint32_t t = threadIdx.x; t < half; t += GIP_BLOCK) {   // flip stage
      const uint32_t blk = t / hk, off = t - blk * hk;
      const uint32_t lo = blk * k + off, hi = blk * k + k - 1 - off;
      if (hi < n) {
        unsigned long long x = a[lo], y = a[hi];
        if (x > y) { a[lo] = y; a[hi] = x; }
      }
    }
    __syncthreads();
    for (uint32_t j = k >> 2; j >= 1; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < half; t += GIP_BLOCK) {
        const uint32_t lo = 2 * j * (t / j) + (t % j), hi = lo + j;
        if (hi < n) {
          unsigned long long x = a[lo], y = a[hi];
          if (x > y) { a[lo] = y; a[hi] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// CAP > 0: sort in LDS, handles tiles with LO < n <= CAP.  CAP == 0: in place in global memory, n > LO.
template <int CAP, int LO>
__global__ void __launch_bounds__(GIP_BLOCK)
gip_tile_sort_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_start, unsigned long long* __restrict__ keys) {
  const uint32_t t = blockIdx.x;
  const uint32_t start = tile_start[t];
  uint32_t end = tile_start[t + 1];
  if (end > kp.capacity) end = kp.capacity;
  if (end <= start) return;
  const uint32_t n = end - start;
  if (n <= (uint32_t)LO) return;
  if (CAP > 0) {
    if (n > (uint32_t)CAP) return;
    __shared__ unsigned long long s_keys[CAP > 0 ? CAP : 1];
    for (uint32_t i = threadIdx.x; i < n; i += GIP_BLOCK) s_keys[i] = keys[start + i];
    __syncthreads();
    bitonic_any_n(s_keys, n);
    for (uint32_t i = threadIdx.x; i < n; i += GIP_BLOCK) keys[start + i] = s_keys[i];
  } else {
    __threadfence_block();
    bitonic_any_n(keys + start, n);
  }
}

void gip_launch_tile_sort(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  const dim3 grid(kp.V * kp.T), block(GIP_BLOCK);
  hipLaunchKernelGGL((gip_tile_sort_kernel<1024, 1>), grid, block, 0, s, kp, st.tile_start, st.keys);
  hipLaunchKernelGGL((gip_tile_sort_kernel<8192, 1024>), grid, block, 0, s, kp, st.tile_start, st.keys);
  hipLaunchKernelGGL((gip_tile_sort_kernel<0, 8192>), grid, block, 0, s, kp, st.tile_start, st.keys);
}
