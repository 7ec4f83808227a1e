// binning.hip — tile binning for gfx950: offsets scan, bucket fill, per-tile depth sort.
//
// Replaces the fork's InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs(u64 keys) +
// identifyTileRanges (SURVEY.md §2.1 "fwd 2-5").  The reference sorts R = sum(tiles_touched) pairs
// globally on 32+log2(T) bits; the result it needs is, per tile, the list of Gaussian indices ordered
// by (depth, index).  MI355X-first formulation (no global multi-pass radix sort, no host read-back):
//
//   1. tile histogram               (done inside preprocess, integer atomics)
//   2. gip_scan_kernel              exclusive scan of the V*T tile counts -> ranges; longest-first launch order; scan of
//                                   the per-workgroup tiles_touched sums -> instance offsets; header
//   3. gip_scatter_kernel           every (Gaussian, tile) instance takes a slot in its tile's bucket
//                                   (one returning integer atomic) and stores key = depth_bits<<32 | index
//   4. gip_tile_sort_kernel /       per-tile sort of the bucket (bitonic network on u64, all-ascending "flip" form so any
//      gip_tile_sort_long_kernel    length works without padding): a wave sorts lists < 512 entirely in registers with
//                                   DPP / ds_bpermute lane exchanges; longer lists combine those wave blocks through LDS
//
// Keys are unique, so the sorted order — and therefore every downstream buffer — is independent of
// the order in which the atomics resolved: the tile / index buffers are deterministic and equal to
// the reference's (tile | depth) stable radix sort (ties on depth resolved by Gaussian index).
#include "gip_internal.h"

// ------------------------------------------------------------------------------------------------
// scan: ceil(V*T / 1024) + ceil(V*nblk / 1024) INDEPENDENT workgroups of 1024 threads, one launch, no inter-workgroup communication
// ------------------------------------------------------------------------------------------------
// Round 5.  The round-2..4 kernel ran three workgroups (tile prefixes / launch order / instance offsets) whose threads each
// walked a 16-tile chunk out of LDS: 17.5 us at V*T = 16384 with 83 % of the launch idle.  Now every
// workgroup owns ONE 1024-tile chunk (a tile per thread) and derives everything it needs from the raw counts itself:
//   pass 1  all V*T counts, coalesced, 8 loads in flight per thread: sums of (count, segments, checkpoint slots) over the
//           tiles BEFORE its chunk and over all tiles, the six size-class counts before its chunk and in total (wave ballots:
//           a 64-tile group lies entirely before, inside or behind a 1024-aligned chunk), the longest list;
//   pass 2  its own chunk: block-wide exclusive scan of the three quantities (tile_start / seg_start / ckpt_start = before +
//           scan), class ranks by ballot (tile_order position = class base + class-before + rank).
// The redundant pass 1 costs V*T * 8 bytes of L2 reads per workgroup (128 KB at four 1024^2 views) and removes every
// dependency between workgroups.  Workgroup 0 also writes the header (+ the pinned host mirror); the workgroups behind the tile
// chunks turn the per-256-Gaussian sums of tiles_touched into instance offsets the same way (a chunk each, own prefix).
#define SCAN_THREADS 1024
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_CHUNK SCAN_THREADS        // tiles per workgroup

__device__ __forceinline__ uint32_t nseg_of(uint32_t count) { return (count + GIP_SEGMENT - 1) / GIP_SEGMENT; }

// Launch order for the per-tile kernels: tiles grouped into 6 size classes and emitted longest class first,
// so the long lists start early and the short ones fill the tail.  Inside a class: tile index order.
//   class 0: >= 2048 | 1: 1024..2047 | 2: 512..1023 | 3: 128..511 | 4: 1..127 | 5: empty
#define ORDER_CLASSES 6
__device__ __forceinline__ int class_of(uint32_t c) {
  return c >= 2048 ? 0 : c >= 1024 ? 1 : c >= 512 ? 2 : c >= 128 ? 3 : c >= 1 ? 4 : 5;
}

// exclusive scan of v over the 1024 threads; *total = the sum.  s_w: SCAN_WAVES words of LDS (reused across calls: two barriers)
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = gip_wave_incl_scan_u32(v);
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_WAVES; w++) { const uint32_t x = s_w[w]; base += w < wave ? x : 0u; tot += x; }
  *total = tot;
  return base + incl - v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
  return v;
}

__global__ void __launch_bounds__(SCAN_THREADS)
gip_scan_kernel(GipKernelParams kp, const uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_count_b,
                uint32_t* __restrict__ tile_start,
                uint32_t* __restrict__ seg_start, uint32_t* __restrict__ ckpt_start,
                const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ block_offset,
                uint32_t* __restrict__ tile_order, GipRasterHeader* __restrict__ header, uint32_t* __restrict__ host_header) {
  __shared__ uint32_t s_w[SCAN_WAVES];
  __shared__ uint32_t s_acc[2][3 + ORDER_CLASSES];       // [before | total][count, segments, checkpoints, classes 0..5]
  __shared__ uint32_t s_max;
  __shared__ uint32_t s_cls[ORDER_CLASSES][SCAN_WAVES];
  const int n = kp.V * kp.T;
  const int n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= n_chunks) {
    // ---- instance offsets: exclusive scan of the per-256-Gaussian sums of tiles_touched (V * nblk entries), the same way: a
    //      workgroup per 1024-entry chunk sums everything BEFORE its chunk itself (coalesced, 8 loads in flight), then scans its
    //      chunk.  (Until round 5 ONE workgroup walked a contiguous slice per thread with dependent loads: 46 + 46 serial round
    //      trips at 1M Gaussians x 12 views — that, not the tile prefixes, was the 75 us of the 1M scan.)
    const int nb = kp.V * kp.nblk;
    const int c0 = ((int)blockIdx.x - n_chunks) * SCAN_CHUNK;
    uint32_t before = 0;
    for (int i0 = tid; i0 < c0; i0 += 8 * SCAN_THREADS) {
      uint32_t v8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u * SCAN_THREADS; v8[u] = i < c0 ? block_sums[i] : 0u; }
#pragma unroll
      for (int u = 0; u < 8; u++) before += v8[u];
    }
    before = wave_sum_u32(before);
    if (tid == 0) s_max = 0u;
    __syncthreads();
    if (lane == 0) atomicAdd(&s_max, before);
    __syncthreads();
    const uint32_t base = s_max;
    const int i = c0 + tid;
    const uint32_t v = i < nb ? block_sums[i] : 0u;
    uint32_t total;
    const uint32_t e = block_excl_scan(v, s_w, &total);
    if (i < nb) block_offset[i] = base + e;
    if (c0 + SCAN_CHUNK >= nb && tid == 0) block_offset[nb] = base + total;      // the last chunk closes the array
    return;
  }
  const int chunk0 = (int)blockIdx.x * SCAN_CHUNK;
  if (tid < 2 * (3 + ORDER_CLASSES)) (&s_acc[0][0])[tid] = 0u;
  if (tid == 0) s_max = 0u;
  __syncthreads();
  // ---- pass 1: every count once; wave-uniform "before my chunk" test (64-tile groups never straddle a 1024-aligned chunk) ----
  uint32_t b_c = 0, b_s = 0, b_k = 0, t_c = 0, t_s = 0, t_k = 0, mx = 0;      // per-lane partial sums
  uint32_t b_cls[ORDER_CLASSES], t_cls[ORDER_CLASSES];                         // wave-uniform class counts
#pragma unroll
  for (int c = 0; c < ORDER_CLASSES; c++) { b_cls[c] = 0; t_cls[c] = 0; }
  for (int i0 = wave * 64; i0 < n; i0 += 8 * SCAN_THREADS) {
    uint32_t va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + u * SCAN_THREADS + lane;
      va[u] = i < n ? tile_count[i] : 0u;
      vb[u] = i < n ? tile_count_b[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int g0 = i0 + u * SCAN_THREADS;          // first tile of this wave's 64-tile group (wave-uniform)
      if (g0 >= n) break;
      const int i = g0 + lane;
      const uint32_t c = va[u] + vb[u], a = nseg_of(c), k = a ? a - 1 : 0;
      const int cls = i < n ? class_of(c) : -1;
      const bool before = g0 < chunk0;
      t_c += c; t_s += a; t_k += k;
      mx = c > mx ? c : mx;
      if (before) { b_c += c; b_s += a; b_k += k; }
#pragma unroll
      for (int q = 0; q < ORDER_CLASSES; q++) {
        const uint32_t m = (uint32_t)__popcll(__ballot(cls == q));
        t_cls[q] += m;
        if (before) b_cls[q] += m;
      }
    }
  }
  b_c = wave_sum_u32(b_c); b_s = wave_sum_u32(b_s); b_k = wave_sum_u32(b_k);
  t_c = wave_sum_u32(t_c); t_s = wave_sum_u32(t_s); t_k = wave_sum_u32(t_k);
  mx = gip_wave_max_u32(mx);
  if (lane == 0) {
    atomicAdd(&s_acc[0][0], b_c); atomicAdd(&s_acc[0][1], b_s); atomicAdd(&s_acc[0][2], b_k);
    atomicAdd(&s_acc[1][0], t_c); atomicAdd(&s_acc[1][1], t_s); atomicAdd(&s_acc[1][2], t_k);
#pragma unroll
    for (int q = 0; q < ORDER_CLASSES; q++) { atomicAdd(&s_acc[0][3 + q], b_cls[q]); atomicAdd(&s_acc[1][3 + q], t_cls[q]); }
    atomicMax(&s_max, mx);
  }
  __syncthreads();
  // ---- pass 2: this workgroup's chunk, one tile per thread ----
  const int i = chunk0 + tid;
  uint32_t c = 0;
  if (i < n) c = tile_count[i] + tile_count_b[i];
  const uint32_t a = nseg_of(c), k = a ? a - 1 : 0;
  uint32_t tot;
  const uint32_t e_c = block_excl_scan(c, s_w, &tot);
  const uint32_t e_s = block_excl_scan(a, s_w, &tot);
  const uint32_t e_k = block_excl_scan(k, s_w, &tot);
  if (i < n) {
    tile_start[i] = s_acc[0][0] + e_c;
    seg_start[i] = s_acc[0][1] + e_s;
    ckpt_start[i] = s_acc[0][2] + e_k;
  }
  // class ranks: ballot rank inside the wave + the counts of the chunk's earlier waves
  const int cls = i < n ? class_of(c) : -1;
  uint32_t rank = 0;
#pragma unroll
  for (int q = 0; q < ORDER_CLASSES; q++) {
    const unsigned long long m = __ballot(cls == q);
    if (cls == q) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_cls[q][wave] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (cls >= 0) {
    uint32_t pos = rank + s_acc[0][3 + cls];                    // + this class's tiles in earlier chunks
#pragma unroll
    for (int q = 0; q < ORDER_CLASSES; q++) pos += q < cls ? s_acc[1][3 + q] : 0u;     // + all tiles of the longer classes
    for (int w = 0; w < wave; w++) pos += s_cls[cls][w];        // + this class's tiles in earlier waves of the chunk
    tile_order[pos] = (uint32_t)i;
  }
  if (blockIdx.x == 0 && tid == 0) {
    const uint32_t total = s_acc[1][0];
    tile_start[n] = total; seg_start[n] = s_acc[1][1]; ckpt_start[n] = s_acc[1][2];
    const uint32_t n0 = s_acc[1][3], n1 = s_acc[1][4], n2 = s_acc[1][5], n3 = s_acc[1][6], n4 = s_acc[1][7];
    header->class_end[1] = n0;                          // sort phase A: lists >= 2048 = [0, class_end[1])
    header->class_end[2] = n0 + n1;                     // (>= 1024)
    header->class_end[0] = n0 + n1 + n2;                // sort phase M: 512..2047 = [class_end[1], class_end[0])
    header->class_end[3] = n0 + n1 + n2 + n3 + n4;      // sort phase B: 1..511 = [class_end[0], class_end[3]); beyond: empty tiles
    header->abi_version = GIP_ABI_VERSION;
    header->num_rendered = total;
    header->overflow = (total > kp.capacity) ? 1u : 0u;
    header->max_tile_count = s_max;
    header->num_segments = s_acc[1][1];
    header->num_checkpoints = s_acc[1][2];
    if (host_header) {      // pinned host mirror: the caller's capacity check needs no device-to-host copy
      // (no system-scope fence here: the host reads the mirror only after an event recorded behind this kernel, and the
      // kernel's completion releases its stores; the fence held the last workgroup for a PCIe round trip)
      host_header[0] = GIP_ABI_VERSION; host_header[1] = total;
      host_header[2] = (total > kp.capacity) ? 1u : 0u; host_header[3] = s_max;
    }
  }
}

void gip_launch_scan(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  const int n = kp.V * kp.T, nb = kp.V * kp.nblk;
  const int n_chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK, b_chunks = nb > 0 ? (nb + SCAN_CHUNK - 1) / SCAN_CHUNK : 1;
  hipLaunchKernelGGL(gip_scan_kernel, dim3(n_chunks + b_chunks), dim3(SCAN_THREADS), 0, s, kp, st.tile_count, st.tile_count_b, st.tile_start,
                     st.seg_start, st.ckpt_start, st.block_sums, st.block_offset, st.tile_order, st.header, st.host_header);
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
  uint32_t tiles = 0, rmin = 0, rmax = 0, dbits = 0, tmask = 0;
  if (idx < kp.P) {
    const uint4* rp = reinterpret_cast<const uint4*>(records + (size_t)v * kp.P + idx);
    const uint4 q0 = rp[0], q1 = rp[1], q3 = rp[3];
    dbits = q0.z; tiles = q1.w; rmin = q3.x; rmax = q3.y; tmask = q3.w;
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
  const int area = (rmaxx - rminx) * (rmaxy - rminy);
  int k = -1, kt = 0;                                  // k: instance number, kt: tile of the rectangle
  for (int ty = rminy; ty < rmaxy; ty++)
    for (int tx = rminx; tx < rmaxx; tx++, kt++) {
      if (!gip_rect_has(tmask, area, kt)) continue;
      k++;
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
// ------------------------------------------------------------------------------------------------
// Wave-level part of the network.  A wave holds a 64*R-key block as R registers per lane (key i of the block = row
// i / 64, lane i % 64).  Every comparator of the all-ascending network pairs index x with x ^ mask (flip: mask = k - 1,
// stride: mask = j), so inside such a block
//   * masks < 64 are lane exchanges (__shfl_xor) — no LDS traffic, no barrier;
//   * strides >= 64 pair two rows of the same lane — plain register compare-exchange;
//   * flips with k >= 128 pair row r with row r ^ (k/64 - 1) of the mirrored lane (lane ^ 63).
// A wave therefore runs ALL stages k <= 64*R of its block (wave_sort) or all strides < 64*R of a later stage
// (wave_tail) between one load and one store of the block.  Keys past n are +inf and never move.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long u64min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned long long u64max(unsigned long long a, unsigned long long b) { return a < b ? b : a; }

// value of `x` held by lane ^ MASK.  VALU-only forms where the hardware has them (no LDS round trip, no wait):
//   1, 2, 3 -> DPP quad_perm;  4 -> row_shl:4 / row_shr:4 on alternate banks;  7 -> row_half_mirror;  8 -> row_ror:8
//   (== xor 8 inside a 16-lane row);  15 -> row_mirror;  16, 31, 32, 63 -> ds_bpermute through __shfl_xor (measured: v_permlane16/32_swap
//   forms are slower — the network is VALU-bound and the LDS crossbar is otherwise idle).
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t x) {
  if constexpr (MASK == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
  else if constexpr (MASK == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
  else if constexpr (MASK == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x1B, 0xF, 0xF, false);
  else if constexpr (MASK == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);
  else if constexpr (MASK == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);
  else if constexpr (MASK == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);
  else if constexpr (MASK == 4) {       // lanes with bit 2 clear take lane + 4 (row_shl:4, banks 0 and 2), the others lane - 4
    const int t = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0x5, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);
  } else return (uint32_t)__shfl_xor((int)x, MASK, 64);
}
template <int MASK>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long x) {
  const uint32_t lo = lane_xor32<MASK>((uint32_t)x), hi = lane_xor32<MASK>((uint32_t)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// one comparator layer x <-> x ^ MASK inside the rows: a lane keeps the minimum if it is the lower index of its pair.
// (keys are unique, so "keep v unless the partner wins" is one 64-bit compare and two selects)
template <int R, int MASK>
__device__ __forceinline__ void wave_xor_step(unsigned long long (&v)[R], bool keep_min) {
#pragma unroll
  for (int r = 0; r < R; r++) {
    const unsigned long long p = lane_xor64<MASK>(v[r]);
    v[r] = ((v[r] < p) == keep_min) ? v[r] : p;
  }
}
template <int R, int FROM>
__device__ __forceinline__ void wave_lane_strides(unsigned long long (&v)[R], int lane) {     // strides FROM .. 1
  if constexpr (FROM >= 32) wave_xor_step<R, 32>(v, !(lane & 32));
  if constexpr (FROM >= 16) wave_xor_step<R, 16>(v, !(lane & 16));
  if constexpr (FROM >= 8) wave_xor_step<R, 8>(v, !(lane & 8));
  if constexpr (FROM >= 4) wave_xor_step<R, 4>(v, !(lane & 4));
  if constexpr (FROM >= 2) wave_xor_step<R, 2>(v, !(lane & 2));
  if constexpr (FROM >= 1) wave_xor_step<R, 1>(v, !(lane & 1));
}
template <int R>
__device__ __forceinline__ void wave_row_strides(unsigned long long (&v)[R], int from) {
#pragma unroll
  for (int jr = R >> 1; jr >= 1; jr >>= 1) {
    if (jr > from) continue;
#pragma unroll
    for (int r = 0; r < R; r++)
      if (!(r & jr)) { const unsigned long long lo = u64min(v[r], v[r + jr]), hi = u64max(v[r], v[r + jr]); v[r] = lo; v[r + jr] = hi; }
  }
}
template <int R, int ROWS>
__device__ __forceinline__ void wave_row_stage(unsigned long long (&v)[R], int lane) {         // stage k = 64 * ROWS
  if constexpr (ROWS <= R) {
#pragma unroll
    for (int r = 0; r < R; r++)
      if (!(r & (ROWS >> 1))) {
        const int q = r ^ (ROWS - 1);
        const unsigned long long a = v[r], b = v[q];
        const unsigned long long br = lane_xor64<63>(b), ar = lane_xor64<63>(a);
        v[r] = a < br ? a : br;
        v[q] = b < ar ? ar : b;
      }
    wave_row_strides<R>(v, ROWS >> 2);
    wave_lane_strides<R, 32>(v, lane);
  }
}
template <int R>
__device__ __forceinline__ void wave_sort(unsigned long long (&v)[R], int lane) {       // stages k = 2 .. 64*R
  wave_xor_step<R, 1>(v, !(lane & 1));                                                  // k = 2
  wave_xor_step<R, 3>(v, !(lane & 2));  wave_lane_strides<R, 1>(v, lane);               // k = 4
  wave_xor_step<R, 7>(v, !(lane & 4));  wave_lane_strides<R, 2>(v, lane);               // k = 8
  wave_xor_step<R, 15>(v, !(lane & 8)); wave_lane_strides<R, 4>(v, lane);               // k = 16
  wave_xor_step<R, 31>(v, !(lane & 16)); wave_lane_strides<R, 8>(v, lane);              // k = 32
  wave_xor_step<R, 63>(v, !(lane & 32)); wave_lane_strides<R, 16>(v, lane);             // k = 64
  wave_row_stage<R, 2>(v, lane);
  wave_row_stage<R, 4>(v, lane);
  wave_row_stage<R, 8>(v, lane);
}
template <int R>
__device__ __forceinline__ void wave_tail(unsigned long long (&v)[R], int lane) {       // strides 32*R .. 1
  wave_row_strides<R>(v, R >> 1);
  wave_lane_strides<R, 32>(v, lane);
}
template <int R>
__device__ __forceinline__ void wave_load(const unsigned long long* a, uint32_t n, uint32_t base, int lane, unsigned long long (&v)[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) { const uint32_t i = base + r * 64 + lane; v[r] = i < n ? a[i] : ~0ull; }
}
template <int R>
__device__ __forceinline__ void wave_store(unsigned long long* a, uint32_t n, uint32_t base, int lane, const unsigned long long (&v)[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) { const uint32_t i = base + r * 64 + lane; if (i < n) a[i] = v[r]; }
}
// one tile of n <= 64*R keys, one wave, registers only
template <int R>
__device__ __forceinline__ void wave_sort_tile(unsigned long long* a, uint32_t n, int lane) {
  unsigned long long v[R];
  wave_load<R>(a, n, 0, lane, v);
  wave_sort<R>(v, lane);
  wave_store<R>(a, n, 0, lane, v);
}

// LDS image of a key chunk: one pad slot per 16 keys, so that a thread writing its own 8 / 16 consecutive keys (stride
// 8.5 / 17 slots across lanes) and a wave writing 64 consecutive keys are both (nearly) bank-conflict free
__device__ __forceinline__ uint32_t padk(uint32_t i) { return i + (i >> 4); }
#define PADK_SLOTS(n) ((n) + ((n) >> 4) + 1)

// One merge level on the chunk: sorted runs of length L -> sorted runs of 2L (the last run of a list may be short or
// missing).  Merge path: group thread `tid` produces the E consecutive outputs [tid * E, tid * E + E): a binary search
// along its diagonal finds how many of the preceding outputs come from each run, then it merges sequentially into
// registers; after a barrier the registers go back in place.  ~20 instructions per key and level, where the bitonic
// merge of the same two runs costs log2(2L) comparator layers of ~6 instructions each.
// Call with the same L by every thread of the workgroup (two barriers inside); n == 0 only runs the barriers.
template <int E>
__device__ __forceinline__ void merge_level(unsigned long long* s, uint32_t n, uint32_t L, uint32_t tid) {
  unsigned long long out[E];
  const uint32_t o0 = tid * E;
  uint32_t cnt = 0;
  if (o0 < n) {
    const uint32_t a0 = o0 & ~(2u * L - 1u);
    const uint32_t b0 = min(a0 + L, n), b1 = min(a0 + 2u * L, n);
    const uint32_t lenA = b0 - a0, lenB = b1 - b0, d = o0 - a0;
    uint32_t lo = d > lenB ? d - lenB : 0u, hi = min(d, lenA);
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s[padk(a0 + mid)] < s[padk(b0 + d - 1u - mid)]) lo = mid + 1u; else hi = mid;
    }
    uint32_t i = lo, j = d - lo;
    cnt = min((uint32_t)E, b1 - o0);
    unsigned long long va = i < lenA ? s[padk(a0 + i)] : ~0ull, vb = j < lenB ? s[padk(b0 + j)] : ~0ull;
#pragma unroll
    for (int k = 0; k < E; k++) {
      const bool take_a = va < vb;                      // an exhausted run reads as +inf; real keys are unique and smaller
      out[k] = take_a ? va : vb;
      if (take_a) { ++i; va = i < lenA ? s[padk(a0 + i)] : ~0ull; }
      else { ++j; vb = j < lenB ? s[padk(b0 + j)] : ~0ull; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < E; k++)
    if ((uint32_t)k < cnt) s[padk(o0 + k)] = out[k];
  __syncthreads();
}

// n keys (512 < n <= NT * 16) from global `src` to global `dst`, sorted, by the NT threads of the workgroup:
//   pass 0   every wave sorts 512-key blocks straight from global memory in registers (wave_sort) and parks them in LDS;
//   levels   L = 512, 1024, ... : merge_level (8 outputs per thread while the list fits NT * 8 keys, else 16);
//   copy     LDS -> dst, coalesced.
template <int NT>
__device__ void lds_sort(unsigned long long* s, uint32_t n, const unsigned long long* src, unsigned long long* dst) {
  constexpr int WAVES = NT / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                                   // the previous tile's readers are done with s
  if (n <= 512) {                                    // one block: registers only
    if (wave == 0) {
      unsigned long long v[8];
      wave_load<8>(src, n, 0, lane, v);
      wave_sort<8>(v, lane);
      wave_store<8>(dst, n, 0, lane, v);
    }
    return;
  }
  for (uint32_t base = wave * 512; base < n; base += WAVES * 512) {
    unsigned long long v[8];
    wave_load<8>(src, n, base, lane, v);
    wave_sort<8>(v, lane);
#pragma unroll
    for (int r = 0; r < 8; r++) { const uint32_t i = base + r * 64 + lane; if (i < n) s[padk(i)] = v[r]; }
  }
  __syncthreads();
  if (n <= (uint32_t)NT * 8u) {
    for (uint32_t L = 512; L < n; L <<= 1) merge_level<8>(s, n, L, threadIdx.x);
  } else {
    for (uint32_t L = 512; L < n; L <<= 1) merge_level<16>(s, n, L, threadIdx.x);
  }
  for (uint32_t i = threadIdx.x; i < n; i += NT) dst[i] = s[padk(i)];
}

// the strides hi_stride .. 1 (hi_stride < chunk) of one merge stage on an n-key chunk: global -> LDS, strides >= 512 as LDS
// passes, strides 256 .. 1 in the waves' registers, -> global
template <int NT>
__device__ void lds_merge(unsigned long long* s, uint32_t n, const unsigned long long* src, unsigned long long* dst,
                          uint32_t hi_stride, uint32_t half) {
  constexpr int WAVES = NT / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += NT) s[i] = src[i];
  __syncthreads();
  for (uint32_t j = hi_stride; j >= 512; j >>= 1) {
    for (uint32_t t = threadIdx.x; t < half; t += NT) {
      const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo + j;
      if (hi < n) {
        const unsigned long long x = s[lo], y = s[hi];
        if (x > y) { s[lo] = y; s[hi] = x; }
      }
    }
    __syncthreads();
  }
  for (uint32_t base = wave * 512; base < n; base += WAVES * 512) {
    unsigned long long v[8];
    wave_load<8>(s, n, base, lane, v);
    wave_tail<8>(v, lane);
    wave_store<8>(dst, n, base, lane, v);
  }
}

// Per-tile sort, ONE launch of one 1024-thread workgroup per CU with a 16384-key (128 KB) LDS chunk.  The scan kernel
// wrote tile_order longest class first and the class boundaries into header->class_end; every workgroup runs three
// phases; A and M draw their work from atomic cursors in the header, so a workgroup held up in A simply draws less in M
// (as separate launches the few long lists of a training view cost 25 serial microseconds):
//   A  lists >= 2048 entries, one tile per workgroup: up to 16384 entries are read once, sorted entirely on chip
//      (lds_sort) and written once; longer lists sort their 16384-key chunks that way and run only the merge strides
//      >= 16384 in place in global memory.  (1M-Gaussian orbit views put ~6000 entries, up to 22k, in every occupied
//      tile: phase A IS the sort there.)
//   M  lists of 512..2047 entries, FOUR tiles per workgroup at a time: each 256-thread quarter sorts its tile in its own
//      2048-key slice of the chunk (one 512-key register block per wave, then the stages 1024 and 2048).  All quarters
//      run the same fixed stage sequence so that the workgroup barriers line up; comparators beyond a list's length are
//      skipped, a stage a short list does not need leaves it unchanged.
//   B  lists of 1..511 entries, one tile per WAVE, entirely in registers (wave_sort_tile: 64 / 128 / 256 / 512-key
//      networks by length), no LDS and no barrier; static round-robin over the grid's waves (a 1024^2 view has at most
//      4096 tiles: about one per wave).
#define LONG_CHUNK 16384
#define LONG_THREADS 1024
__device__ __forceinline__ void ce_global(unsigned long long* a, uint32_t lo, uint32_t hi, uint32_t n) {
  if (hi < n) {
    const unsigned long long x = a[lo], y = a[hi];
    if (x > y) { a[lo] = y; a[hi] = x; }
  }
}

// n > LONG_CHUNK: chunks in LDS, strides >= LONG_CHUNK in place in global memory
__device__ void sort_beyond_lds(unsigned long long* a, uint32_t n, unsigned long long* s_keys) {
  constexpr int NT = LONG_THREADS;
  uint32_t m = 1;
  while (m < n) m <<= 1;
  // phase 1: sort every LONG_CHUNK-aligned chunk completely in LDS
  for (uint32_t c0 = 0; c0 < n; c0 += LONG_CHUNK) {
    const uint32_t cn = min((uint32_t)LONG_CHUNK, n - c0);
    lds_sort<NT>(s_keys, cn, a + c0, a + c0);
  }
  __threadfence_block();
  __syncthreads();
  // phase 2: merges of size k > LONG_CHUNK
  for (uint32_t k = 2 * LONG_CHUNK; k <= m; k <<= 1) {
    const uint32_t hk = k >> 1;
    for (uint32_t tt = threadIdx.x; tt < (m >> 1); tt += NT) {       // flip stage, global
      const uint32_t base = (tt & ~(hk - 1)) << 1, off = tt & (hk - 1);
      ce_global(a, base + off, base + k - 1 - off, n);
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t j = k >> 2; j >= LONG_CHUNK; j >>= 1) {                    // long strides, global
      for (uint32_t tt = threadIdx.x; tt < (m >> 1); tt += NT) {
        const uint32_t lo = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
        ce_global(a, lo, lo + j, n);
      }
      __threadfence_block();
      __syncthreads();
    }
    for (uint32_t c0 = 0; c0 < n; c0 += LONG_CHUNK) {                        // strides < LONG_CHUNK, on chip
      const uint32_t cn = min((uint32_t)LONG_CHUNK, n - c0);
      lds_merge<NT>(s_keys, cn, a + c0, a + c0, LONG_CHUNK >> 1, LONG_CHUNK >> 1);
    }
    __threadfence_block();
    __syncthreads();
  }
}

// segment -> tile map of one tile's segments (work list of the backward kernel)
template <int NT>
__device__ __forceinline__ void write_seg_tiles(const GipKernelParams& kp, const uint32_t* __restrict__ seg_start,
                                                uint32_t* __restrict__ seg_tile, uint32_t t, uint32_t n) {
  const uint32_t s0 = seg_start[t], ns = (n + GIP_SEGMENT - 1) / GIP_SEGMENT;
  for (uint32_t b = threadIdx.x; b < ns; b += NT)
    if (s0 + b < kp.seg_capacity) seg_tile[s0 + b] = t;
}

// phase M worker: the calling 256-thread quarter sorts a[0, n) (n < 2048; n == 0: no tile, barriers only) in its LDS slice:
// one 512-key register block per wave, then the two merge levels 512 -> 1024 -> 2048, then the coalesced copy out.
// Fixed sequence for every quarter, so the workgroup barriers inside merge_level line up.
__device__ void quarter_sort(unsigned long long* s, uint32_t n, unsigned long long* a) {
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3;
  const uint32_t ltid = threadIdx.x & 255, base = w * 512;
  const bool multi = n > 512;
  if (base < n) {
    unsigned long long v[8];
    wave_load<8>(a, n, base, lane, v);
    wave_sort<8>(v, lane);
    if (multi) {
#pragma unroll
      for (int r = 0; r < 8; r++) { const uint32_t i = base + r * 64 + lane; if (i < n) s[padk(i)] = v[r]; }
    } else {
      wave_store<8>(a, n, base, lane, v);
    }
  }
  __syncthreads();
  const uint32_t nm = multi ? n : 0u;
  merge_level<8>(s, nm, 512, ltid);
  merge_level<8>(s, nm, 1024, ltid);
  for (uint32_t i = ltid; i < nm; i += 256) a[i] = s[padk(i)];
}

__global__ void __launch_bounds__(LONG_THREADS)
gip_tile_sort_kernel(GipKernelParams kp, GipRasterHeader* __restrict__ header, const uint32_t* __restrict__ tile_order,
                     const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ seg_start,
                     uint32_t* __restrict__ seg_tile, unsigned long long* __restrict__ keys) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_long[];
  __shared__ uint32_t s_pos;
  const uint32_t end_a = header->class_end[1], end_m = header->class_end[0], end_b = header->class_end[3];
  // ---- phase A: workgroup per tile ----
  if (end_a > 0) {
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_pos = atomicAdd(&header->sort_cursor, 1u);
      __syncthreads();
      const uint32_t pos = s_pos;
      if (pos >= end_a) break;
      const uint32_t t = tile_order[pos];
      const uint32_t start = tile_start[t];
      uint32_t end = tile_start[t + 1];
      if (end > kp.capacity) end = kp.capacity;
      if (end <= start) continue;
      const uint32_t n = end - start;
      write_seg_tiles<LONG_THREADS>(kp, seg_start, seg_tile, t, n);
      if (n <= 1) continue;
      if (n <= LONG_CHUNK) lds_sort<LONG_THREADS>(s_long, n, keys + start, keys + start);
      else sort_beyond_lds(keys + start, n, s_long);
    }
  }
  // ---- phase M: quarter (256 threads) per tile, four tiles per draw ----
  if (end_m > end_a) {
    const uint32_t q = threadIdx.x >> 8;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_pos = atomicAdd(&header->sort_cursor_m, 4u);
      __syncthreads();
      const uint32_t pos = end_a + s_pos + q;
      if (end_a + s_pos >= end_m) break;
      uint32_t n = 0;
      unsigned long long* a = keys;
      if (pos < end_m) {
        const uint32_t t = tile_order[pos];
        const uint32_t start = tile_start[t];
        uint32_t end = tile_start[t + 1];
        if (end > kp.capacity) end = kp.capacity;
        if (end > start) {
          n = end - start;
          a = keys + start;
          const uint32_t s0 = seg_start[t], ns = (n + GIP_SEGMENT - 1) / GIP_SEGMENT;
          for (uint32_t b = threadIdx.x & 255; b < ns; b += 256)
            if (s0 + b < kp.seg_capacity) seg_tile[s0 + b] = t;
        }
      }
      quarter_sort(s_long + q * PADK_SLOTS(2048), n > 1 ? n : 0, a);
    }
  }
  // ---- phase B: wave per tile, static round-robin over all waves of the grid (no barrier from here on) ----
  const int lane = threadIdx.x & 63;
  const uint32_t waves_per_wg = LONG_THREADS / 64;
  // consecutive positions (similar lengths: the order is by size class) go to different CUs, not to the waves of one
  for (uint32_t pos = end_m + (threadIdx.x >> 6) * gridDim.x + blockIdx.x; pos < end_b; pos += gridDim.x * waves_per_wg) {
    const uint32_t t = tile_order[pos];
    const uint32_t start = tile_start[t];
    uint32_t end = tile_start[t + 1];
    if (end > kp.capacity) end = kp.capacity;
    if (end <= start) continue;
    const uint32_t n = end - start;
    {
      const uint32_t s0 = seg_start[t], ns = (n + GIP_SEGMENT - 1) / GIP_SEGMENT;
      for (uint32_t b = lane; b < ns; b += 64)
        if (s0 + b < kp.seg_capacity) seg_tile[s0 + b] = t;
    }
    unsigned long long* a = keys + start;
    if (n <= 1) continue;
    if (n <= 64) wave_sort_tile<1>(a, n, lane);
    else if (n <= 128) wave_sort_tile<2>(a, n, lane);
    else if (n <= 256) wave_sort_tile<4>(a, n, lane);
    else wave_sort_tile<8>(a, n, lane);
  }
}

void gip_launch_tile_sort(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s) {
  // > 64 KB of dynamic LDS needs the per-function opt-in (idempotent, set once per process); one workgroup per CU
  static int wgs = 0;
  constexpr size_t lds = (size_t)PADK_SLOTS(LONG_CHUNK) * sizeof(unsigned long long);
  if (!wgs) {
    (void)hipFuncSetAttribute((const void*)gip_tile_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;
    wgs = cus;
  }
  hipLaunchKernelGGL(gip_tile_sort_kernel, dim3(wgs), dim3(LONG_THREADS), lds, s, kp, st.header, st.tile_order,
                     st.tile_start, st.seg_start, st.seg_tile, st.keys);
}

