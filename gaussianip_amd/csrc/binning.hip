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
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_LDS_TILES 32768   // tile counts staged in LDS when V*T fits (128 KB of the 160 KB)

struct U3 { uint32_t a, b, c; };

// exclusive scan of three quantities at once across the 1024 threads
__device__ __forceinline__ U3 block_excl_scan3(U3 v, uint32_t (*s_wave)[SCAN_WAVES], U3* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  U3 incl = {gip_wave_incl_scan_u32(v.a), gip_wave_incl_scan_u32(v.b), gip_wave_incl_scan_u32(v.c)};
  if (lane == 63) { s_wave[0][wave] = incl.a; s_wave[1][wave] = incl.b; s_wave[2][wave] = incl.c; }
  __syncthreads();
  U3 base = {0, 0, 0}, tot = {0, 0, 0};
#pragma unroll
  for (int w = 0; w < SCAN_WAVES; w++) {
    const uint32_t xa = s_wave[0][w], xb = s_wave[1][w], xc = s_wave[2][w];
    if (w < wave) { base.a += xa; base.b += xb; base.c += xc; }
    tot.a += xa; tot.b += xb; tot.c += xc;
  }
  __syncthreads();
  *total = tot;
  return {base.a + incl.a - v.a, base.b + incl.b - v.b, base.c + incl.c - v.c};
}

__device__ __forceinline__ uint32_t nseg_of(uint32_t count) { return (count + GIP_SEGMENT - 1) / GIP_SEGMENT; }
__device__ __forceinline__ int bucket_of(uint32_t c) { return c ? 32 - __clz(c) : 0; }

// Launch order for the per-tile kernels: tiles bucketed by floor(log2(count)) and emitted longest bucket
// first, so the long lists start early and the short ones fill the tail.  Order inside a bucket is arbitrary
// (it only decides which workgroup renders which tile; results do not depend on it).  Wave-aggregated:
// one LDS atomic per (wave, bucket present) instead of one per tile.
__device__ void heavy_first_order(const uint32_t* __restrict__ counts, uint32_t* __restrict__ order, int n,
                                  uint32_t* s_bucket /*[33]*/, uint32_t* class_end /*[4] out, thread 0*/) {
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 33; i += SCAN_THREADS) s_bucket[i] = 0;
  __syncthreads();
  const int rounds = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  for (int r = 0; r < rounds; r++) {
    const int i = r * SCAN_THREADS + threadIdx.x;
    const int b = i < n ? bucket_of(counts[i]) : -1;
    unsigned long long pending = __ballot(b >= 0);
    while (pending) {
      const int leader = __builtin_ctzll(pending);
      const int lb = __shfl(b, leader, 64);
      const unsigned long long m = __ballot(b == lb);
      if (lane == leader) atomicAdd(&s_bucket[lb], (uint32_t)__popcll(m));
      pending &= ~m;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int b = 32; b >= 0; b--) {
      const uint32_t c = s_bucket[b];
      s_bucket[b] = run;
      run += c;
      // sort classes: >= 8192 (bucket >= 14) | 2048..8191 (12,13) | 1024..2047 (11) | 1..1023 (1..10)
      if (b == 14) class_end[0] = run;
      if (b == 12) class_end[1] = run;
      if (b == 11) class_end[2] = run;
      if (b == 1) class_end[3] = run;
    }
  }
  __syncthreads();
  for (int r = 0; r < rounds; r++) {
    const int i = r * SCAN_THREADS + threadIdx.x;
    const int b = i < n ? bucket_of(counts[i]) : -1;
    unsigned long long pending = __ballot(b >= 0);
    while (pending) {
      const int leader = __builtin_ctzll(pending);
      const int lb = __shfl(b, leader, 64);
      const unsigned long long m = __ballot(b == lb);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&s_bucket[lb], (uint32_t)__popcll(m));
      base = __shfl(base, leader, 64);
      if (b == lb) order[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
      pending &= ~m;
    }
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
gip_scan_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_start,
                uint32_t* __restrict__ seg_start, uint32_t* __restrict__ ckpt_start, uint32_t* __restrict__ seg_tile,
                const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_offset,
                uint32_t* __restrict__ tile_order, GipRasterHeader* __restrict__ header) {
  __shared__ uint32_t s_wave[3][SCAN_WAVES];
  __shared__ uint32_t s_bucket[33];
  __shared__ uint32_t s_class[4];
  // ---- tiles: instance ranges, segment ranges, checkpoint slots (one pass, three running sums) ----
  // counts are first staged in LDS with coalesced loads (each thread then walks its contiguous chunk
  // out of LDS instead of issuing serial dependent global loads)
  extern __shared__ uint32_t s_cnt[];
  const int n = kp.V * kp.T;
  const bool in_lds = n <= SCAN_LDS_TILES;
  if (in_lds) {
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) s_cnt[i] = tile_count[i];
    __syncthreads();
  }
  const uint32_t* cnt = in_lds ? s_cnt : tile_count;
  const int chunk = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  const int lo = threadIdx.x * chunk, hi = min(n, lo + chunk);
  U3 sum = {0, 0, 0};
  uint32_t mx = 0;
  for (int i = lo; i < hi; i++) {
    const uint32_t c = cnt[i], a = nseg_of(c);
    sum.a += c; sum.b += a; sum.c += a ? a - 1 : 0;
    mx = c > mx ? c : mx;
  }
  U3 total;
  U3 run = block_excl_scan3(sum, s_wave, &total);
  for (int i = lo; i < hi; i++) {
    const uint32_t c = cnt[i], a = nseg_of(c);
    tile_start[i] = run.a; seg_start[i] = run.b; ckpt_start[i] = run.c;
    for (uint32_t b = 0; b < a; b++)
      if (run.b + b < kp.seg_capacity) seg_tile[run.b + b] = (uint32_t)i;
    run.a += c; run.b += a; run.c += a ? a - 1 : 0;
  }
  if (threadIdx.x == 0) { tile_start[n] = total.a; seg_start[n] = total.b; ckpt_start[n] = total.c; }
  mx = gip_wave_max_u32(mx);
  if ((threadIdx.x & 63) == 0) s_wave[0][threadIdx.x >> 6] = mx;
  __syncthreads();
  uint32_t max_tile = 0;
  for (int w = 0; w < SCAN_WAVES; w++) max_tile = s_wave[0][w] > max_tile ? s_wave[0][w] : max_tile;
  __syncthreads();
  // ---- Gaussians: per-workgroup sums of tiles_touched -> instance offsets ----
  {
    const int nb = kp.V * kp.nblk;
    const int ch = (nb + SCAN_THREADS - 1) / SCAN_THREADS;
    const int l2 = threadIdx.x * ch, h2 = min(nb, l2 + ch);
    U3 s2 = {0, 0, 0};
    for (int i = l2; i < h2; i++) s2.a += block_sums[i];
    U3 t2;
    U3 r2 = block_excl_scan3(s2, s_wave, &t2);
    for (int i = l2; i < h2; i++) { block_offset[i] = r2.a; r2.a += block_sums[i]; }
    if (threadIdx.x == 0) block_offset[nb] = t2.a;
  }
  heavy_first_order(cnt, tile_order, n, s_bucket, s_class);
  __syncthreads();
  if (threadIdx.x == 0) {
    header->abi_version = GIP_ABI_VERSION;
    header->num_rendered = total.a;
    header->overflow = (total.a > kp.capacity) ? 1u : 0u;
    header->max_tile_count = max_tile;
    header->num_segments = total.b;
    header->num_checkpoints = total.c;
    header->class_end[0] = s_class[0]; header->class_end[1] = s_class[1];
    header->class_end[2] = s_class[2]; header->class_end[3] = s_class[3];
  }
}

void gip_launch_scan(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  const int n = kp.V * kp.T;
  const size_t lds = n <= SCAN_LDS_TILES ? (size_t)n * 4 : 0;
  // > 64 KB of dynamic LDS needs the per-function opt-in (idempotent, set once per process)
  static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(gip_scan_kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_LDS_TILES * 4);
  (void)attr_once;
  hipLaunchKernelGGL(gip_scan_kernel, dim3(1), dim3(SCAN_THREADS), lds, s, kp, st.tile_count, st.tile_start,
                     st.seg_start, st.ckpt_start, st.seg_tile, st.block_sums, st.block_offset, st.tile_order, st.header);
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

// One class of tile sizes per launch; the classes are contiguous ranges of tile_order (written by the scan
// kernel into header->class_end), walked grid-stride by a small persistent grid so that no workgroup is
// launched for tiles outside the class.  CAP > 0: sort in LDS.  CAP == 0: in place in global memory.
template <int CAP, int CLS>
__global__ void __launch_bounds__(GIP_BLOCK)
gip_tile_sort_kernel(GipKernelParams kp, const GipRasterHeader* __restrict__ header, const uint32_t* __restrict__ tile_order,
                     const uint32_t* __restrict__ tile_start, unsigned long long* __restrict__ keys) {
  const uint32_t pos_lo = CLS == 0 ? 0u : header->class_end[CLS - 1];
  const uint32_t pos_hi = header->class_end[CLS];
  if (pos_lo + blockIdx.x >= pos_hi) return;
  __shared__ unsigned long long s_keys[CAP > 0 ? CAP : 1];
  for (uint32_t pos = pos_lo + blockIdx.x; pos < pos_hi; pos += gridDim.x) {
    const uint32_t t = tile_order[pos];
    const uint32_t start = tile_start[t];
    uint32_t end = tile_start[t + 1];
    if (end > kp.capacity) end = kp.capacity;
    if (end <= start + 1) continue;
    const uint32_t n = end - start;
    if (CAP > 0) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n; i += GIP_BLOCK) s_keys[i] = keys[start + i];
      __syncthreads();
      bitonic_any_n(s_keys, n);
      for (uint32_t i = threadIdx.x; i < n; i += GIP_BLOCK) keys[start + i] = s_keys[i];
    } else {
      __threadfence_block();
      bitonic_any_n(keys + start, n);
      __syncthreads();
    }
  }
}

void gip_launch_tile_sort(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  const dim3 block(GIP_BLOCK);
  hipLaunchKernelGGL((gip_tile_sort_kernel<0, 0>), dim3(64), block, 0, s, kp, st.header, st.tile_order, st.tile_start, st.keys);
  hipLaunchKernelGGL((gip_tile_sort_kernel<8192, 1>), dim3(128), block, 0, s, kp, st.header, st.tile_order, st.tile_start, st.keys);
  hipLaunchKernelGGL((gip_tile_sort_kernel<2048, 2>), dim3(512), block, 0, s, kp, st.header, st.tile_order, st.tile_start, st.keys);
  hipLaunchKernelGGL((gip_tile_sort_kernel<1024, 3>), dim3(2048), block, 0, s, kp, st.header, st.tile_order, st.tile_start, st.keys);
}
