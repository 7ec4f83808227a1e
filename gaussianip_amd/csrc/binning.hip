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
// scan: three workgroups of 1024 threads (one per independent job), each thread owns a contiguous chunk
// ------------------------------------------------------------------------------------------------
#ifndef SCAN_SKIP
#define SCAN_SKIP 0
#endif
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

// Launch order for the per-tile kernels: tiles grouped into 6 size classes and emitted longest class first,
// so the long lists start early and the short ones fill the tail.  Order inside a class is arbitrary (it only
// decides which workgroup handles which tile; results do not depend on it).  Wave-aggregated: one LDS atomic
// per (wave, class, round) — the empty-tile class would otherwise serialise 64 lanes on one LDS word.
//   class 0: >= 2048 | 1: 1024..2047 | 2: 512..1023 | 3: 128..511 | 4: 1..127 | 5: empty
#define ORDER_CLASSES 6
__device__ __forceinline__ int class_of(uint32_t c) {
  return c >= 2048 ? 0 : c >= 1024 ? 1 : c >= 512 ? 2 : c >= 128 ? 3 : c >= 1 ? 4 : 5;
}
__device__ void heavy_first_order(const uint32_t* lds_counts, const uint32_t* ca, const uint32_t* cb,
                                  uint32_t* __restrict__ order, int n, uint32_t* s_bucket /*[33]*/,
                                  uint32_t* class_end /*[4] out, thread 0*/) {
  auto count_of = [&](int i) -> uint32_t { return lds_counts ? lds_counts[i] : ca[i] + cb[i]; };
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int i = threadIdx.x; i < 33; i += SCAN_THREADS) s_bucket[i] = 0;
  __syncthreads();
  const int rounds = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  for (int r = 0; r < rounds; r++) {
    const int i = r * SCAN_THREADS + threadIdx.x;
    const int cls = i < n ? class_of(count_of(i)) : -1;
#pragma unroll
    for (int c = 0; c < ORDER_CLASSES; c++) {
      const unsigned long long m = __ballot(cls == c);
      if (m && lane == 0) atomicAdd(&s_bucket[c], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int c = 0; c < ORDER_CLASSES; c++) {
      const uint32_t k = s_bucket[c];
      s_bucket[c] = run;
      run += k;
      if (c == 0) class_end[0] = class_end[1] = run;   // sort role A: >= 2048
      if (c == 1) class_end[2] = run;                  // sort role B: 1024..2047
      if (c == 4) class_end[3] = run;                  // sort role C: 1..1023 ; beyond: empty tiles
    }
  }
  __syncthreads();
  for (int r = 0; r < rounds; r++) {
    const int i = r * SCAN_THREADS + threadIdx.x;
    const int cls = i < n ? class_of(count_of(i)) : -1;
#pragma unroll
    for (int c = 0; c < ORDER_CLASSES; c++) {
      const unsigned long long m = __ballot(cls == c);
      if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_bucket[c], (uint32_t)__popcll(m));
        base = __shfl(base, 0, 64);
        if (cls == c) order[base + __popcll(m & lt)] = (uint32_t)i;
      }
    }
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
gip_scan_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_count_b,
                uint32_t* __restrict__ tile_start,
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
  const int role = blockIdx.x;
  if (role == 2) {
    const int nb = kp.V * kp.nblk;
    const int ch = (nb + SCAN_THREADS - 1) / SCAN_THREADS;
    const int l2 = threadIdx.x * ch, h2 = min(nb, l2 + ch);
    U3 s2 = {0, 0, 0};
    for (int i = l2; i < h2; i++) s2.a += block_sums[i];
    U3 t2;
    U3 r2 = block_excl_scan3(s2, s_wave, &t2);
    for (int i = l2; i < h2; i++) { block_offset[i] = r2.a; r2.a += block_sums[i]; }
    if (threadIdx.x == 0) block_offset[nb] = t2.a;
    return;
  }
  const bool in_lds = n <= SCAN_LDS_TILES;
  if (in_lds) {
    // 8 independent load pairs in flight per thread (a plain loop would wait for each pair in turn)
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * SCAN_THREADS) {
      uint32_t va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * SCAN_THREADS;
        va[u] = i < n ? tile_count[i] : 0u;
        vb[u] = i < n ? tile_count_b[i] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u * SCAN_THREADS;
        if (i < n) s_cnt[i] = va[u] + vb[u];
      }
    }
    __syncthreads();
  }
  // (beyond SCAN_LDS_TILES tiles the two partial counts are summed on the fly from global memory)
  auto cnt_at = [&](int i) -> uint32_t { return in_lds ? s_cnt[i] : tile_count[i] + tile_count_b[i]; };
  // Three independent jobs, one workgroup each (blockIdx.x = role), so that their latency chains overlap:
  //   role 0: tile prefixes (ranges / segments / checkpoint slots) + totals in the header
  //   role 1: longest-first launch order + class boundaries in the header
  //   role 2: per-workgroup tiles_touched sums -> instance offsets
  if (role == 1) {
#if !(SCAN_SKIP & 1)
    heavy_first_order(in_lds ? s_cnt : nullptr, tile_count, tile_count_b, tile_order, n, s_bucket, s_class);
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
      header->class_end[0] = s_class[0]; header->class_end[1] = s_class[1];
      header->class_end[2] = s_class[2]; header->class_end[3] = s_class[3];
    }
    return;
  }
  const int chunk = (n + SCAN_THREADS - 1) / SCAN_THREADS;
  const int lo = threadIdx.x * chunk, hi = min(n, lo + chunk);
  U3 sum = {0, 0, 0};
  uint32_t mx = 0;
  for (int i = lo; i < hi; i++) {
    const uint32_t c = cnt_at(i), a = nseg_of(c);
    sum.a += c; sum.b += a; sum.c += a ? a - 1 : 0;
    mx = c > mx ? c : mx;
  }
  U3 total;
  const U3 run0 = block_excl_scan3(sum, s_wave, &total);
  if (in_lds && 2 * n <= SCAN_LDS_TILES) {
    // prefixes go through a second LDS array so that the global stores are coalesced (a thread's own chunk is
    // 16 consecutive words: written directly it costs 64 separate cache lines per wave store)
    uint32_t* s_out = s_cnt + n;
    for (int which = 0; which < 3; which++) {
      uint32_t r = which == 0 ? run0.a : which == 1 ? run0.b : run0.c;
      for (int i = lo; i < hi; i++) {
        const uint32_t c = s_cnt[i], a = nseg_of(c);
        s_out[i] = r;
        r += which == 0 ? c : which == 1 ? a : (a ? a - 1 : 0);
      }
      __syncthreads();
      uint32_t* dst = which == 0 ? tile_start : which == 1 ? seg_start : ckpt_start;
      for (int i = threadIdx.x; i < n; i += SCAN_THREADS) dst[i] = s_out[i];
      __syncthreads();
    }
  } else {
    U3 run = run0;
    for (int i = lo; i < hi; i++) {
      const uint32_t c = cnt_at(i), a = nseg_of(c);
      tile_start[i] = run.a; seg_start[i] = run.b; ckpt_start[i] = run.c;
      run.a += c; run.b += a; run.c += a ? a - 1 : 0;
    }
  }
  if (threadIdx.x == 0) { tile_start[n] = total.a; seg_start[n] = total.b; ckpt_start[n] = total.c; }
  mx = gip_wave_max_u32(mx);
  if ((threadIdx.x & 63) == 0) s_wave[0][threadIdx.x >> 6] = mx;
  __syncthreads();
  uint32_t max_tile = 0;
  for (int w = 0; w < SCAN_WAVES; w++) max_tile = s_wave[0][w] > max_tile ? s_wave[0][w] : max_tile;
  if (threadIdx.x == 0) {
    header->abi_version = GIP_ABI_VERSION;
    header->num_rendered = total.a;
    header->overflow = (total.a > kp.capacity) ? 1u : 0u;
    header->max_tile_count = max_tile;
    header->num_segments = total.b;
    header->num_checkpoints = total.c;
  }
}

void gip_launch_scan(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  const int n = kp.V * kp.T;
  const size_t lds = n <= SCAN_LDS_TILES ? (size_t)(2 * n <= SCAN_LDS_TILES ? 2 * n : n) * 4 : 0;
  // > 64 KB of dynamic LDS needs the per-function opt-in (idempotent, set once per process)
  static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(gip_scan_kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_LDS_TILES * 4);
  (void)attr_once;
  hipLaunchKernelGGL(gip_scan_kernel, dim3(3), dim3(SCAN_THREADS), lds, s, kp, st.tile_count, st.tile_count_b, st.tile_start,
                     st.seg_start, st.ckpt_start, st.seg_tile, st.block_sums, st.block_offset, st.tile_order, st.header);
}

// ------------------------------------------------------------------------------------------------
// scatter: fill the tile buckets
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GIP_BLOCK)
gip_scatter_kernel(GipKernelParams kp, const GipRecord* __restrict__ records, const uint32_t* __restrict__ tile_start,
                   const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_cursor,
                   const uint32_t* __restrict__ inst_slot, const uint32_t* __restrict__ block_offset,
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
  // slots drawn by preprocess (first GIP_SLOTS instances): no atomic here
  const uint4* sp = reinterpret_cast<const uint4*>(inst_slot + ((size_t)v * kp.P + idx) * GIP_SLOTS);
  const uint4 s0 = sp[0];
  uint4 s1 = make_uint4(0, 0, 0, 0);
  if (tiles > 4) s1 = sp[1];
  const uint32_t slots[GIP_SLOTS] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  int k = 0;
  for (int ty = rminy; ty < rmaxy; ty++)
    for (int tx = rminx; tx < rmaxx; tx++, k++) {
      const size_t t = tbase + ty * kp.tiles_x + tx;
      uint32_t slot;
      if (k < GIP_SLOTS) {
        slot = 0;
#pragma unroll
        for (int kk = 0; kk < GIP_SLOTS; kk++) if (kk == k) slot = slots[kk];
      } else {
        slot = tile_count[t] + atomicAdd(&tile_cursor[t], 1u);   // after the remembered-slot part of the bucket
      }
      const uint32_t pos = tile_start[t] + slot;
      if (pos < kp.capacity) keys[pos] = key;
    }
}

void gip_launch_scatter(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  hipLaunchKernelGGL(gip_scatter_kernel, dim3(kp.nblk, kp.V), dim3(GIP_BLOCK), 0, s, kp, st.records, st.tile_start,
                     st.tile_count, st.tile_cursor, st.inst_slot, st.block_offset, st.inst_offset, st.keys);
}

// ------------------------------------------------------------------------------------------------
// per-tile sort.  All-ascending bitonic network ("flip" first sub-stage): comparators only ever move
// the larger element to the higher index, so virtual +inf padding at indices >= n never moves and
// comparators that touch it can simply be skipped -> correct for any n.
// ------------------------------------------------------------------------------------------------
// Register-blocked small strides: thread t owns the 8 consecutive keys [8t, 8t+8) (256 threads x 8 = the 2048-key LDS
// chunk), so every comparator with both ends inside an aligned 8-block — the stages k = 2, 4, 8 and the merge strides
// 4, 2, 1 of every later stage — runs on registers between ONE 64-byte LDS read and write, without barriers in between.
// Keys past n are +inf in registers (a comparator against +inf never swaps: the same as skipping it).
#define CE(x, y) { const unsigned long long lo_ = x < y ? x : y, hi_ = x < y ? y : x; x = lo_; y = hi_; }
__device__ __forceinline__ void regs_load(const unsigned long long* a, uint32_t n, uint32_t b, unsigned long long* r) {
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = b + i < n ? a[b + i] : ~0ull;
}
__device__ __forceinline__ void regs_store(unsigned long long* a, uint32_t n, uint32_t b, const unsigned long long* r) {
#pragma unroll
  for (int i = 0; i < 8; i++) if (b + i < n) a[b + i] = r[i];
}
__device__ __forceinline__ void regs_merge421(unsigned long long* r) {
  CE(r[0], r[4]) CE(r[1], r[5]) CE(r[2], r[6]) CE(r[3], r[7])
  CE(r[0], r[2]) CE(r[1], r[3]) CE(r[4], r[6]) CE(r[5], r[7])
  CE(r[0], r[1]) CE(r[2], r[3]) CE(r[4], r[5]) CE(r[6], r[7])
}
__device__ __forceinline__ void regs_sort8(unsigned long long* r) {     // stages k = 2, 4, 8 of the flip network
  CE(r[0], r[1]) CE(r[2], r[3]) CE(r[4], r[5]) CE(r[6], r[7])                                   // k = 2: flip
  CE(r[0], r[3]) CE(r[1], r[2]) CE(r[4], r[7]) CE(r[5], r[6])                                   // k = 4: flip
  CE(r[0], r[1]) CE(r[2], r[3]) CE(r[4], r[5]) CE(r[6], r[7])                                   //        stride 1
  CE(r[0], r[7]) CE(r[1], r[6]) CE(r[2], r[5]) CE(r[3], r[4])                                   // k = 8: flip
  CE(r[0], r[2]) CE(r[1], r[3]) CE(r[4], r[6]) CE(r[5], r[7])                                   //        stride 2
  CE(r[0], r[1]) CE(r[2], r[3]) CE(r[4], r[5]) CE(r[6], r[7])                                   //        stride 1
}

// strides j = 4, 2, 1 of one merge on registers (all threads; ends with a barrier)
__device__ __forceinline__ void merge_low_regs(unsigned long long* a, uint32_t n) {
  for (uint32_t b = threadIdx.x * 8; b < n; b += GIP_BLOCK * 8) {
    unsigned long long r[8];
    regs_load(a, n, b, r);
    regs_merge421(r);
    regs_store(a, n, b, r);
  }
  __syncthreads();
}

__device__ __forceinline__ void bitonic_any_n(unsigned long long* a, uint32_t n) {
  uint32_t m = 1;
  while (m < n) m <<= 1;
  const uint32_t half = m >> 1;
  for (uint32_t b = threadIdx.x * 8; b < n; b += GIP_BLOCK * 8) {             // stages k = 2, 4, 8
    unsigned long long r[8];
    regs_load(a, n, b, r);
    regs_sort8(r);
    regs_store(a, n, b, r);
  }
  __syncthreads();
  for (uint32_t k = 16; k <= m; k <<= 1) {
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
    for (uint32_t j = k >> 2; j >= 8; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < half; t += GIP_BLOCK) {
        const uint32_t lo = 2 * j * (t / j) + (t % j), hi = lo + j;
        if (hi < n) {
          unsigned long long x = a[lo], y = a[hi];
          if (x > y) { a[lo] = y; a[hi] = x; }
        }
      }
      __syncthreads();
    }
    merge_low_regs(a, n);
  }
}

// Per-tile sort, ONE launch.  The size classes are contiguous ranges of tile_order (written by the scan kernel
// into header->class_end); workgroups take a role by blockIdx so that the few long lists (many bitonic stages,
// latency-bound) sort concurrently with the many short ones instead of in separate back-to-back launches:
//   role A  [0, SORT_WG_BIG)              tiles with >= 2048 entries: all-ascending network, strides < 2048 in LDS
//                                          one 2048-key chunk at a time, longer strides in place in global memory
//   role B  [SORT_WG_BIG, +SORT_WG_MID)   1024..2047 entries, fully in LDS
//   role C  the rest                       1..1023 entries, fully in LDS
#define BIG_CHUNK 2048
#define SORT_WG_BIG 256
#define SORT_WG_MID 512
#define SORT_WG_SMALL 2048
__device__ __forceinline__ void ce_global(unsigned long long* a, uint32_t lo, uint32_t hi, uint32_t n) {
  if (hi < n) {
    const unsigned long long x = a[lo], y = a[hi];
    if (x > y) { a[lo] = y; a[hi] = x; }
  }
}

__device__ void sort_big_tile(unsigned long long* a, uint32_t n, unsigned long long* s_keys) {
  uint32_t m = 1;
  while (m < n) m <<= 1;
  // phase 1: sort every BIG_CHUNK-aligned chunk completely in LDS
  for (uint32_t c0 = 0; c0 < n; c0 += BIG_CHUNK) {
    const uint32_t cn = min((uint32_t)BIG_CHUNK, n - c0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cn; i += GIP_BLOCK) s_keys[i] = a[c0 + i];
    __syncthreads();
    bitonic_any_n(s_keys, cn);
    for (uint32_t i = threadIdx.x; i < cn; i += GIP_BLOCK) a[c0 + i] = s_keys[i];
  }
  __threadfence_block();
  __syncthreads();
  // phase 2: merges of size k > BIG_CHUNK
  for (uint32_t k = 2 * BIG_CHUNK; k <= m; k <<= 1) {
    const uint32_t hk = k >> 1;
    for (uint32_t tt = threadIdx.x; tt < (m >> 1); tt += GIP_BLOCK) {       // flip stage, global
      const uint32_t blk = tt / hk, off = tt - blk * hk;
      ce_global(a, blk * k + off, blk * k + k - 1 - off, n);
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t j = k >> 2; j >= BIG_CHUNK; j >>= 1) {                    // long strides, global
      for (uint32_t tt = threadIdx.x; tt < (m >> 1); tt += GIP_BLOCK) {
        const uint32_t lo = 2 * j * (tt / j) + (tt % j);
        ce_global(a, lo, lo + j, n);
      }
      __threadfence_block();
      __syncthreads();
    }
    for (uint32_t c0 = 0; c0 < n; c0 += BIG_CHUNK) {                        // strides < BIG_CHUNK, in LDS
      const uint32_t cn = min((uint32_t)BIG_CHUNK, n - c0);
      for (uint32_t i = threadIdx.x; i < cn; i += GIP_BLOCK) s_keys[i] = a[c0 + i];
      __syncthreads();
      for (uint32_t j = BIG_CHUNK >> 1; j >= 8; j >>= 1) {
        for (uint32_t tt = threadIdx.x; tt < (BIG_CHUNK >> 1); tt += GIP_BLOCK) {
          const uint32_t lo = 2 * j * (tt / j) + (tt % j), hi = lo + j;
          if (hi < cn) {
            const unsigned long long x = s_keys[lo], y = s_keys[hi];
            if (x > y) { s_keys[lo] = y; s_keys[hi] = x; }
          }
        }
        __syncthreads();
      }
      merge_low_regs(s_keys, cn);
      for (uint32_t i = threadIdx.x; i < cn; i += GIP_BLOCK) a[c0 + i] = s_keys[i];
      __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
  }
}

__global__ void __launch_bounds__(GIP_BLOCK)
gip_tile_sort_kernel(GipKernelParams kp, const GipRasterHeader* __restrict__ header, const uint32_t* __restrict__ tile_order,
                     const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ seg_start,
                     uint32_t* __restrict__ seg_tile, unsigned long long* __restrict__ keys) {
  __shared__ unsigned long long s_keys[BIG_CHUNK];
  uint32_t pos_lo, pos_hi, first, stride;
  bool big = false;
  if (blockIdx.x < SORT_WG_BIG) {
    pos_lo = 0; pos_hi = header->class_end[1]; first = blockIdx.x; stride = SORT_WG_BIG; big = true;
  } else if (blockIdx.x < SORT_WG_BIG + SORT_WG_MID) {
    pos_lo = header->class_end[1]; pos_hi = header->class_end[2]; first = blockIdx.x - SORT_WG_BIG; stride = SORT_WG_MID;
  } else {
    pos_lo = header->class_end[2]; pos_hi = header->class_end[3];
    first = blockIdx.x - SORT_WG_BIG - SORT_WG_MID; stride = SORT_WG_SMALL;
  }
  for (uint32_t pos = pos_lo + first; pos < pos_hi; pos += stride) {
    const uint32_t t = tile_order[pos];
    const uint32_t start = tile_start[t];
    uint32_t end = tile_start[t + 1];
    if (end > kp.capacity) end = kp.capacity;
    if (end <= start) continue;
    const uint32_t n = end - start;
    // segment -> tile map of this tile's segments (work list of the backward kernel)
    {
      const uint32_t s0 = seg_start[t], ns = (n + GIP_SEGMENT - 1) / GIP_SEGMENT;
      for (uint32_t b = threadIdx.x; b < ns; b += GIP_BLOCK)
        if (s0 + b < kp.seg_capacity) seg_tile[s0 + b] = t;
    }
    if (n <= 1) continue;
    if (big) {
      sort_big_tile(keys + start, n, s_keys);
    } else {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n; i += GIP_BLOCK) s_keys[i] = keys[start + i];
      __syncthreads();
      bitonic_any_n(s_keys, n);
      for (uint32_t i = threadIdx.x; i < n; i += GIP_BLOCK) keys[start + i] = s_keys[i];
    }
  }
}

void gip_launch_tile_sort(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  hipLaunchKernelGGL(gip_tile_sort_kernel, dim3(SORT_WG_BIG + SORT_WG_MID + SORT_WG_SMALL), dim3(GIP_BLOCK), 0, s, kp,
                     st.header, st.tile_order, st.tile_start, st.seg_start, st.seg_tile, st.keys);
}
