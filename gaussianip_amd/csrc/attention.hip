// attention.hip — softmax(Q K^T * scale) V forward for the denoiser's self-attention (no mask, no dropout, no
// gradient: the U-Net / ControlNet are frozen and run under no_grad), fp16 in / fp16 out, fp32 softmax and accumulation,
// on the gfx950 matrix cores.  See include/gip_nn.h (gip_attention_fwd_f16).
//
// Layout: q, k, v, o are the [B, N, H*D] tensors the to_q / to_k / to_v projections produce and to_out consumes — the
// head split / merge transposes of the reference (attention_processor_faceid.py:300-318) never materialise.
//
// One workgroup = 128 query rows of one (batch, head); 4 waves x 32 rows; keys / values stream through LDS in blocks of
// 64 rows (two stages; the next block's global loads are in flight during the current block's math).  Per block and wave:
//   S^T = K Q^T      v_mfma_f32_32x32x16_f16, A = K rows (ds_read_b128 from a chunk-swizzled row image), B = Q^T kept
//                    in registers for the whole kernel.  The result has the QUERY on the lane and the 32 keys of the
//                    tile in the 16 registers of the two lane halves, so the online softmax (max, exp2, sum, rescale)
//                    is lane-local except for one exchange with lane ^ 32 per block.
//   O^T += V^T P^T   P^T is consumed straight from the S^T accumulators as the B operand (register pairs -> half); the
//                    matching k-permuted A operand V^T comes from the row-major V image by ds_read_b64_tr_b16, the
//                    hardware transposing LDS read.  O^T again has the query on the lane: rescaling is lane-local.
// The running maximum is only raised when a block exceeds it by more than 2^8 (in the exp2 domain), which removes
// almost all accumulator rescales; probabilities stay <= 256, well inside half range.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gip_nn.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

#define AT_BQ 128
#define AT_BKV 64
// LDS rows are 128 bytes (D <= 64) or 256 bytes (D <= 128); chunk swizzles per row length:
//   K image (row reads, ds_read_b128):          chunk ^ ((row >> 1) & 7)      |  chunk ^ (row & 15)
//   V image (transposed reads, 4-row blocks):   chunk ^ (((row >> 1) & 1) << 2) |  chunk ^ ((row & 3) << 2)
// both make a 16-lane row read / a half-wave transposed read cover all 64 banks once.
template <int ROWB> __device__ __forceinline__ int kswz(int row) { return ROWB == 128 ? ((row >> 1) & 7) : (row & 15); }
template <int ROWB> __device__ __forceinline__ int vswz(int row) { return ROWB == 128 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); }
#define AT_DEFER 8.0f
// max over the two lane halves (the v_permlane32_swap form measured the same as this ds_bpermute: DESIGN.md section 4d)
__device__ __forceinline__ float at_half_max(float v) {
  return fmaxf(v, __shfl_xor(v, 32));
}

union Frag8 {
  f16x8 v;
  s16x4 h[2];
  uint4 u;
};

// TWO: 0 = one key set.  1 = a second key set of at most AT_BKV keys (the decoupled cross-attention's 4 image-prompt tokens,
// attention_processor_faceid.py:462-466): the first set's result is normalised in place and the second set's probabilities
// are scaled by w2 / l2 BEFORE their PV product, so both sets share ONE accumulator (the two-accumulator form spilled
// 68-124 bytes per lane at D = 80 / 160).  2 = a second key set of any length (two accumulators).
template <int D, int TWO, int NW = 4>
__global__ void __launch_bounds__(64 * NW, D > 128 ? 1 : 2)   // (threads, waves per SIMD)
attn_fwd_kernel(const _Float16* __restrict__ q, const _Float16* __restrict__ k, const _Float16* __restrict__ v,
                _Float16* __restrict__ o, int Nq, int Nkv, int H, float c /* scale * log2(e) */,
                const _Float16* __restrict__ k2, const _Float16* __restrict__ v2, int Nkv2, float w2, int ld_kv, int ld_kv2, int ld_q,
                int xcd_remap = 0) {
  // ld_kv / ld_kv2: elements between consecutive key rows of (k, v) / (k2, v2).  H * D for packed projections; larger
  // when a layer's keys are a column range of one wide matrix holding the key / value projections of MANY layers
  static_assert(D % 8 == 0 && D <= 160, "head dim");
  // D = 160 (the 16x16 and 8x8 levels of the U-Net: 1280 channels / 8 heads): 512-byte rows, one workgroup per CU with the
  // whole 512-register file per lane (O^T alone is 5 accumulator tiles) — tiny layers, but they were the last ones on
  // torch SDPA (an AOTriton kernel on ROCm)
  constexpr int AT_ROW = D <= 64 ? 128 : (D <= 128 ? 256 : 512);          // bytes per LDS row
  constexpr int AT_TILE = AT_BKV * AT_ROW;             // one K or V stage
  constexpr int NS = (D + 15) / 16;      // k-steps of the S^T product
  constexpr int ND = (D + 31) / 32;      // 32-row tiles of O^T
  constexpr int CH = D / 8;              // 16-byte chunks per row
  constexpr int NT = 64 * NW;            // threads: NW waves of 32 queries share the K / V stages (the stream from L2 per query ~ 1 / NW)
  constexpr int PER = (AT_BKV * CH + NT - 1) / NT;
  // D < 64 leaves unused rows in the last O^T tile: row D of V^T is set to ONES, so that O^T[D][q] = sum_k p[k][q] — the
  // softmax denominator comes out of the PV matrix product and the 32 scalar adds per block disappear (VALU-bound kernel)
  constexpr bool L_FROM_MFMA = (D % 32) != 0;
  constexpr int L_TILE = D / 32, L_ROW = D % 32;          // position of that row in the O^T tiles
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 2 * AT_TILE];   // [stage][K | V]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  // workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest): with xcd_remap every (batch, head)'s query
  // blocks land on ONE XCD, so its keys / values are filled into one L2 instead of eight (speed only)
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_remap) {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if ((total & 7) == 0) {
      const int lin = by * gx + bx, nl = (lin & 7) * (total >> 3) + (lin >> 3);
      by = nl / gx;
      bx = nl - by * gx;
    }
  }
  const int b = by / H, h = by - b * H;
  const int C = H * D;
  const int q0 = bx * (32 * NW) + wave * 32;
  const bool q_valid = q0 < Nq;          // Nq % 32 == 0: a wave's 32 query rows exist or not as a whole (the 8x8 level has 64)

  // Q^T fragments (B operand): lane = query column, element j = feature 16 s + 8 hh + j; zero beyond D
  f16x8 qf[NS];
  {
    const _Float16* qp = q + ((size_t)b * Nq + q0 + r) * ld_q + h * D;      // ld_q > C: q is a column range of a fused q | k | v projection
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int d0 = 16 * s + 8 * hh;
      Frag8 f;
      f.u = make_uint4(0, 0, 0, 0);
      if (d0 < D && q_valid) f.u = *(const uint4*)(qp + d0);
      qf[s] = f.v;
    }
  }
  // the padding columns of the images must be finite zeros (0 * garbage would poison S^T): clear everything once
  for (int i = tid * 16; i < 2 * 2 * AT_TILE; i += NT * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);
  if constexpr (L_FROM_MFMA) {
    __syncthreads();
    // feature D of every V row (both stages) = 1.0; the staging only ever rewrites chunks < D / 8
    if (tid < 2 * AT_BKV) {
      const int stage = tid / AT_BKV, row = tid % AT_BKV, lch = D / 8;
      unsigned char* p = smem + stage * 2 * AT_TILE + AT_TILE + row * AT_ROW + ((lch ^ vswz<AT_ROW>(row)) << 4) + (D % 8) * 2;
      *(_Float16*)p = (_Float16)1.0f;
    }
  }

  // staging descriptors: every thread moves two chunks of K and of V per block; the tail indices wrap around (a few
  // chunks are moved twice with identical data) so that no load or LDS store is predicated; rows past the end of a
  // ragged key set are clamped to its last row (they are masked to -inf before the softmax)
  static_assert(PER >= 1 && PER <= 5 && NT * PER < 3 * AT_BKV * CH, "staging chunks per thread (two wrap-arounds at most)");
  // one set of NAMED scalars per staged chunk (e = 0 .. PER - 1), enumerated by macros under `if constexpr`: arrays (even
  // with fully unrolled loops, even inside inlined lambdas) were placed in scratch memory by hipcc, which made the D = 40
  // kernel 2x slower
#define AT_SLOT(e)                                                                                        \
  int idx##e = tid + NT * e;                                                                              \
  if (idx##e >= AT_BKV * CH) idx##e -= AT_BKV * CH;                                                       \
  if (idx##e >= AT_BKV * CH) idx##e -= AT_BKV * CH;                                                       \
  const int row##e = idx##e / CH, ch##e = idx##e - row##e * CH;                                           \
  const int kl##e = row##e * AT_ROW + ((ch##e ^ kswz<AT_ROW>(row##e)) << 4);                              \
  const int vl##e = AT_TILE + row##e * AT_ROW + ((ch##e ^ vswz<AT_ROW>(row##e)) << 4);                    \
  uint4 kr##e = make_uint4(0, 0, 0, 0), vr##e = make_uint4(0, 0, 0, 0);
  AT_SLOT(0) AT_SLOT(1) AT_SLOT(2) AT_SLOT(3) AT_SLOT(4)
#undef AT_SLOT
#define AT_FETCH1(e, blk_)                                                      \
  if constexpr (PER > e) {                                                      \
    const int ra_ = min((blk_) * AT_BKV + row##e, n_keys - 1);                  \
    kr##e = *(const uint4*)(kp + (size_t)ra_ * ld + ch##e * 8);                 \
    vr##e = *(const uint4*)(vp + (size_t)ra_ * ld + ch##e * 8);                 \
  }
#define AT_FETCH(blk_) { AT_FETCH1(0, blk_) AT_FETCH1(1, blk_) AT_FETCH1(2, blk_) AT_FETCH1(3, blk_) AT_FETCH1(4, blk_) }
#define AT_DEPOSIT1(e, st_)                                                     \
  if constexpr (PER > e) {                                                      \
    *(uint4*)(st_ + kl##e) = kr##e;                                             \
    *(uint4*)(st_ + vl##e) = vr##e;                                             \
  }
#define AT_DEPOSIT(stage_)                                                      \
  {                                                                             \
    unsigned char* stp_ = smem + (stage_) * 2 * AT_TILE;                        \
    AT_DEPOSIT1(0, stp_) AT_DEPOSIT1(1, stp_) AT_DEPOSIT1(2, stp_) AT_DEPOSIT1(3, stp_) AT_DEPOSIT1(4, stp_)   \
  }

  // fragment addressing
  const int k_row_off = r * AT_ROW, k_swz = kswz<AT_ROW>(r);       // tile bases are multiples of 32: swizzle term of row = of r
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, half16 = (lane >> 4) & 1;

  f32x16 O[ND], Oacc[TWO == 2 ? ND : 1];
  if constexpr (TWO == 2) {
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) Oacc[dt][i] = 0.f;
  }
  _Float16* op = o + ((size_t)b * Nq + q0 + r) * C + h * D;

  for (int seg = 0; seg < (TWO == 2 ? 2 : 1); seg++) {
    const int n_keys = seg == 0 ? Nkv : Nkv2;
    const size_t ld = (size_t)(seg == 0 ? ld_kv : ld_kv2);
    const _Float16* kp = (seg == 0 ? k : k2) + (size_t)b * n_keys * ld + h * D;
    const _Float16* vp = (seg == 0 ? v : v2) + (size_t)b * n_keys * ld + h * D;
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) O[dt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    AT_FETCH(0);
    __syncthreads();        // the clear (first segment) / every read of the previous segment's stages is done
    AT_DEPOSIT(0);
    __syncthreads();
    const int NB = (n_keys + AT_BKV - 1) / AT_BKV;
    for (int blk = 0; blk < NB; blk++) {
      const int stage = blk & 1;
      const unsigned char* sk = smem + stage * 2 * AT_TILE;
      const int nblk = blk + 1 < NB ? blk + 1 : blk;      // the last iteration re-fetches its own block (never deposited)
      AT_FETCH(nblk);

      // ---- S^T = K Q^T for the two 32-key tiles ----
      f32x16 S[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int i = 0; i < 16; i++) S[t][i] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++) {
          const f16x8 a = *(const f16x8*)(sk + t * 32 * AT_ROW + k_row_off + (((2 * s + hh) ^ k_swz) << 4));
          S[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], S[t], 0, 0, 0);
        }
      }

      // ---- V^T fragments: issued now, they land while the softmax runs ----
      const unsigned char* sv = sk + AT_TILE;
      Frag8 vt[ND][2][2];
#pragma unroll
      for (int dt = 0; dt < ND; dt++) {
        const int col = dt * 32 + 16 * half16 + 4 * p4;          // first feature this lane addresses
        const int lch = col >> 3, sub = (p4 & 1) * 8;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++) {
            const int row = t * 32 + 16 * s2 + 4 * hh + q4;
            const int off = row * AT_ROW + ((lch ^ vswz<AT_ROW>(row)) << 4) + sub;
            vt[dt][t][s2].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + off));
            vt[dt][t][s2].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + off + 8 * AT_ROW));
          }
      }
      __builtin_amdgcn_sched_barrier(0);

      // ---- ragged tail: keys past the end never win the softmax (register i of tile t = key 32 t + (i&3) + 8 (i>>2) + 4 hh)
      if (blk * AT_BKV + AT_BKV > n_keys) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int i = 0; i < 16; i++)
            if (blk * AT_BKV + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= n_keys) S[t][i] = -INFINITY;
      }

      // ---- online softmax on the lane's query column ----
      float mloc = S[0][0];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) mloc = fmaxf(mloc, S[t][i]);
      mloc = at_half_max(mloc);
      const bool raise = (mloc - m_run) * c > AT_DEFER;     // true on the first block (m_run = -inf)
      if (__any(raise)) {
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        if constexpr (!L_FROM_MFMA) l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
          for (int i = 0; i < 16; i++) O[dt][i] *= alpha;
        m_run = m_new;
      }
      const float mc = m_run * c;
      Frag8 P[2][2];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[t][8 * s2 + j], c, -mc));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[t][8 * s2 + j + 1], c, -mc));
            if constexpr (!L_FROM_MFMA) l_run += p0 + p1;
            P[t][s2].v[j] = (_Float16)p0;
            P[t][s2].v[j + 1] = (_Float16)p1;
          }

      // ---- O^T += V^T P^T ----
#pragma unroll
      for (int dt = 0; dt < ND; dt++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++)
            O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vt[dt][t][s2].v, P[t][s2].v, O[dt], 0, 0, 0);

      if (blk + 1 < NB) { AT_DEPOSIT(stage ^ 1); }
      __syncthreads();
    }

    // ---- normalise this segment ----
    float l_tot;
    if constexpr (L_FROM_MFMA) {
      // row L_ROW of tile L_TILE: register (L_ROW & 3) + 4 * (L_ROW >> 3) of the lane half (L_ROW >> 2) & 1
      constexpr int reg = (L_ROW & 3) + 4 * (L_ROW >> 3), half = (L_ROW >> 2) & 1;
      const float mine = O[L_TILE][reg];
      const float other = __shfl_xor(mine, 32);
      l_tot = (hh == half) ? mine : other;
    } else {
      l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = (seg == 0 ? 1.f : w2) / l_tot;
    if constexpr (TWO == 2) {
#pragma unroll
      for (int dt = 0; dt < ND; dt++)
#pragma unroll
        for (int i = 0; i < 16; i++) Oacc[dt][i] += O[dt][i] * inv;
    } else {
#pragma unroll
      for (int dt = 0; dt < ND; dt++)
#pragma unroll
        for (int i = 0; i < 16; i++) O[dt][i] *= inv;
    }
  }

  if constexpr (TWO == 1) {
    // ---- second key set, one block: O (already normalised) += (w2 / l2) * sum_k p2[k] v2[k] ----
    const int n_keys = Nkv2;
    const size_t ld = (size_t)ld_kv2;
    const _Float16* kp = k2 + (size_t)b * n_keys * ld + h * D;
    const _Float16* vp = v2 + (size_t)b * n_keys * ld + h * D;
    AT_FETCH(0);
    __syncthreads();        // every read of the first set's stages is done
    AT_DEPOSIT(0);
    __syncthreads();
    const unsigned char* sk = smem;
    f32x16 S[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int i = 0; i < 16; i++) S[t][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const f16x8 a = *(const f16x8*)(sk + t * 32 * AT_ROW + k_row_off + (((2 * s + hh) ^ k_swz) << 4));
        S[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], S[t], 0, 0, 0);
      }
    }
    const unsigned char* sv = sk + AT_TILE;
    Frag8 vt[ND][2][2];
#pragma unroll
    for (int dt = 0; dt < ND; dt++) {
      const int col = dt * 32 + 16 * half16 + 4 * p4;
      const int lch = col >> 3, sub = (p4 & 1) * 8;
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          const int row = t * 32 + 16 * s2 + 4 * hh + q4;
          const int off = row * AT_ROW + ((lch ^ vswz<AT_ROW>(row)) << 4) + sub;
          vt[dt][t][s2].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + off));
          vt[dt][t][s2].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + off + 8 * AT_ROW));
        }
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int i = 0; i < 16; i++)
        if (t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= n_keys) S[t][i] = -INFINITY;
    float mloc = S[0][0];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) mloc = fmaxf(mloc, S[t][i]);
    const float mc = at_half_max(mloc) * c;
    float pf[2][16], l2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        pf[t][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[t][i], c, -mc));
        l2 += pf[t][i];
      }
    l2 += __shfl_xor(l2, 32);
    const float f = w2 / l2;
    Frag8 P[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int j = 0; j < 8; j++) P[t][s2].v[j] = (_Float16)(pf[t][8 * s2 + j] * f);
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
          O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vt[dt][t][s2].v, P[t][s2].v, O[dt], 0, 0, 0);
  }

  // ---- store: lane holds O^T[d = 32 dt + 8 i + 4 hh + 0..3][query r] in registers 4i..4i+3 ----
#pragma unroll
  for (int dt = 0; dt < ND; dt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int d = dt * 32 + 8 * i + 4 * hh;
      if (d < D && q_valid) {
        f16x4 w;
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = (_Float16)(TWO == 2 ? Oacc[dt][4 * i + j] : O[dt][4 * i + j]);
        *(f16x4*)(op + d) = w;
      }
    }
}

// ---- two query tiles per wave (long self-attention, one key set) -----------------------------------------------------------------
// The loop above runs its S^T MFMAs, its softmax and its PV MFMAs one after the other, and co-resident waves do not fill the
// gaps (counters + ablation builds: tools/experiments/attention_pipelined_qk.md).  Here a wave owns TWO 32-query tiles: the K
// and V^T fragments of a block are read from LDS once for both, and the two tiles' chains are independent, so inside ONE wave
// the matrix pipe works on one tile's products while the vector unit does the other tile's softmax.  Same arithmetic per
// (query, key): bit-identical outputs.  64 * NW queries per workgroup.
template <int D, int NW>
__global__ void __launch_bounds__(64 * NW, 2)
attn_fwd_q2_kernel(const _Float16* __restrict__ q, const _Float16* __restrict__ k, const _Float16* __restrict__ v,
                   _Float16* __restrict__ o, int Nq, int Nkv, int H, float c, int ld_kv, int ld_q, int xcd_remap) {
  static_assert(D % 8 == 0 && D <= 64, "head dim");
  constexpr int AT_ROW = 128;
  constexpr int AT_TILE = AT_BKV * AT_ROW;
  constexpr int NS = (D + 15) / 16, ND = (D + 31) / 32, CH = D / 8;
  constexpr int NT = 64 * NW;
  constexpr int PER = (AT_BKV * CH + NT - 1) / NT;
  constexpr bool L_FROM_MFMA = (D % 32) != 0;
  constexpr int L_TILE = D / 32, L_ROW = D % 32;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 2 * AT_TILE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_remap) {
    const int gx = gridDim.x, total = gx * gridDim.y;
    if ((total & 7) == 0) {
      const int lin = by * gx + bx, nl = (lin & 7) * (total >> 3) + (lin >> 3);
      by = nl / gx;
      bx = nl - by * gx;
    }
  }
  const int b = by / H, h = by - b * H;
  const int C = H * D;
  const int q0 = bx * (64 * NW) + wave * 64;
  bool q_valid[2];
  f16x8 qf[2][NS];
#pragma unroll
  for (int qt = 0; qt < 2; qt++) {
    q_valid[qt] = q0 + 32 * qt < Nq;
    const _Float16* qp = q + ((size_t)b * Nq + q0 + 32 * qt + r) * ld_q + h * D;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int d0 = 16 * s + 8 * hh;
      Frag8 f;
      f.u = make_uint4(0, 0, 0, 0);
      if (d0 < D && q_valid[qt]) f.u = *(const uint4*)(qp + d0);
      qf[qt][s] = f.v;
    }
  }
  for (int i = tid * 16; i < 2 * 2 * AT_TILE; i += NT * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);
  if constexpr (L_FROM_MFMA) {
    __syncthreads();
    if (tid < 2 * AT_BKV) {
      const int stage = tid / AT_BKV, row = tid % AT_BKV, lch = D / 8;
      unsigned char* p = smem + stage * 2 * AT_TILE + AT_TILE + row * AT_ROW + ((lch ^ vswz<AT_ROW>(row)) << 4) + (D % 8) * 2;
      *(_Float16*)p = (_Float16)1.0f;
    }
  }
  static_assert(PER >= 1 && PER <= 5 && NT * PER < 3 * AT_BKV * CH, "staging chunks per thread");
#define AT_SLOT(e)                                                                                        \
  int idx##e = tid + NT * e;                                                                              \
  if (idx##e >= AT_BKV * CH) idx##e -= AT_BKV * CH;                                                       \
  if (idx##e >= AT_BKV * CH) idx##e -= AT_BKV * CH;                                                       \
  const int row##e = idx##e / CH, ch##e = idx##e - row##e * CH;                                           \
  const int kl##e = row##e * AT_ROW + ((ch##e ^ kswz<AT_ROW>(row##e)) << 4);                              \
  const int vl##e = AT_TILE + row##e * AT_ROW + ((ch##e ^ vswz<AT_ROW>(row##e)) << 4);                    \
  uint4 kr##e = make_uint4(0, 0, 0, 0), vr##e = make_uint4(0, 0, 0, 0);
  AT_SLOT(0) AT_SLOT(1) AT_SLOT(2) AT_SLOT(3) AT_SLOT(4)
#undef AT_SLOT
  const int k_row_off = r * AT_ROW, k_swz = kswz<AT_ROW>(r);
  const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, half16 = (lane >> 4) & 1;

  f32x16 O[2][ND];
#pragma unroll
  for (int qt = 0; qt < 2; qt++)
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) O[qt][dt][i] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const int n_keys = Nkv;
  const size_t ld = (size_t)ld_kv;
  const _Float16* kp = k + (size_t)b * n_keys * ld + h * D;
  const _Float16* vp = v + (size_t)b * n_keys * ld + h * D;

  AT_FETCH(0);
  __syncthreads();
  AT_DEPOSIT(0);
  __syncthreads();
  const int NB = (n_keys + AT_BKV - 1) / AT_BKV;
  for (int blk = 0; blk < NB; blk++) {
    const int stage = blk & 1;
    const unsigned char* sk = smem + stage * 2 * AT_TILE;
    const int nblk = blk + 1 < NB ? blk + 1 : blk;
    AT_FETCH(nblk);

    // ---- S^T = K Q^T: every K fragment feeds both query tiles ----
    f32x16 S[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; qt++)
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) S[qt][t][i] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const f16x8 a = *(const f16x8*)(sk + t * 32 * AT_ROW + k_row_off + (((2 * s + hh) ^ k_swz) << 4));
#pragma unroll
        for (int qt = 0; qt < 2; qt++) S[qt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[qt][s], S[qt][t], 0, 0, 0);
      }

    // ---- V^T fragments (shared by both tiles) ----
    const unsigned char* sv = sk + AT_TILE;
    Frag8 vt[ND][2][2];
#pragma unroll
    for (int dt = 0; dt < ND; dt++) {
      const int col = dt * 32 + 16 * half16 + 4 * p4;
      const int lch = col >> 3, sub = (p4 & 1) * 8;
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          const int row = t * 32 + 16 * s2 + 4 * hh + q4;
          const int off = row * AT_ROW + ((lch ^ vswz<AT_ROW>(row)) << 4) + sub;
          vt[dt][t][s2].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + off));
          vt[dt][t][s2].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sv + off + 8 * AT_ROW));
        }
    }

    if (blk * AT_BKV + AT_BKV > n_keys) {
#pragma unroll
      for (int qt = 0; qt < 2; qt++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int i = 0; i < 16; i++)
            if (blk * AT_BKV + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= n_keys) S[qt][t][i] = -INFINITY;
    }

    // ---- per tile: online softmax, then O^T += V^T P^T (tile 1's softmax has tile 0's PV MFMAs to run beside) ----
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
      float mloc = S[qt][0][0];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) mloc = fmaxf(mloc, S[qt][t][i]);
      mloc = at_half_max(mloc);
      const bool raise = (mloc - m_run[qt]) * c > AT_DEFER;
      if (__any(raise)) {
        const float m_new = fmaxf(m_run[qt], mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * c);
        if constexpr (!L_FROM_MFMA) l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
          for (int i = 0; i < 16; i++) O[qt][dt][i] *= alpha;
        m_run[qt] = m_new;
      }
      const float mc = m_run[qt] * c;
      Frag8 P[2][2];
#pragma unroll
      for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(S[qt][t][8 * s2 + j], c, -mc));
            if constexpr (!L_FROM_MFMA) l_run[qt] += p;
            P[t][s2].v[j] = (_Float16)p;
          }
#pragma unroll
      for (int dt = 0; dt < ND; dt++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++)
            O[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vt[dt][t][s2].v, P[t][s2].v, O[qt][dt], 0, 0, 0);
    }

    if (blk + 1 < NB) { AT_DEPOSIT(stage ^ 1); }
    __syncthreads();
  }

#pragma unroll
  for (int qt = 0; qt < 2; qt++) {
    float l_tot;
    if constexpr (L_FROM_MFMA) {
      constexpr int reg = (L_ROW & 3) + 4 * (L_ROW >> 3), half = (L_ROW >> 2) & 1;
      const float mine = O[qt][L_TILE][reg];
      const float other = __shfl_xor(mine, 32);
      l_tot = (hh == half) ? mine : other;
    } else {
      l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
    }
    const float inv = 1.f / l_tot;
    _Float16* op = o + ((size_t)b * Nq + q0 + 32 * qt + r) * C + h * D;
#pragma unroll
    for (int dt = 0; dt < ND; dt++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int d = dt * 32 + 8 * i + 4 * hh;
        if (d < D && q_valid[qt]) {
          f16x4 w;
#pragma unroll
          for (int j = 0; j < 4; j++) w[j] = (_Float16)(O[qt][dt][4 * i + j] * inv);
          *(f16x4*)(op + d) = w;
        }
      }
  }
}

template <int D, int NW>
static void launch_attn_q2(hipStream_t s, const void* q, const void* k, const void* v, void* o, int BH, int Nq, int Nkv, int H, float c,
                           int ld_kv, int ld_q) {
  const int xcd = 1;      // every (batch, head)'s query blocks on one XCD (see launch_attn_wide)
  const dim3 grid((Nq + 64 * NW - 1) / (64 * NW), BH);
  hipLaunchKernelGGL((attn_fwd_q2_kernel<D, NW>), grid, dim3(64 * NW), 0, s, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                     (_Float16*)o, Nq, Nkv, H, c, ld_kv, ld_q, xcd);
}

template <int D>
static void launch_attn(dim3 grid, hipStream_t s, const void* q, const void* k, const void* v, void* o, int Nq, int Nkv, int H,
                        float c, const void* k2, const void* v2, int Nkv2, float w2, int ld_kv, int ld_kv2, int ld_q) {
  if (k2 && Nkv2 <= AT_BKV)
    hipLaunchKernelGGL((attn_fwd_kernel<D, 1>), grid, dim3(256), 0, s, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                       (_Float16*)o, Nq, Nkv, H, c, (const _Float16*)k2, (const _Float16*)v2, Nkv2, w2, ld_kv, ld_kv2, ld_q);
  else if (k2)
    hipLaunchKernelGGL((attn_fwd_kernel<D, 2>), grid, dim3(256), 0, s, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                       (_Float16*)o, Nq, Nkv, H, c, (const _Float16*)k2, (const _Float16*)v2, Nkv2, w2, ld_kv, ld_kv2, ld_q);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<D, 0>), grid, dim3(256), 0, s, (const _Float16*)q, (const _Float16*)k, (const _Float16*)v,
                       (_Float16*)o, Nq, Nkv, H, c, (const _Float16*)nullptr, (const _Float16*)nullptr, 0, 0.f, ld_kv, ld_kv, ld_q);
}

// long self-attention (one key set): NW waves of 32 queries per workgroup share the K / V stages
template <int D, int NW>
static void launch_attn_wide(hipStream_t s, const void* q, const void* k, const void* v, void* o, int BH, int Nq, int Nkv, int H, float c,
                             int ld_kv, int ld_q) {
  const dim3 grid((Nq + 32 * NW - 1) / (32 * NW), BH);
  // every (batch, head)'s query blocks on ONE XCD (its keys / values fill one L2 instead of eight): 0.421 -> 0.4105 ms at batch 12,
  // 0.140 -> 0.137 at batch 4, same box, two alternating runs
  const int xcd = 1;
  hipLaunchKernelGGL((attn_fwd_kernel<D, 0, NW>), grid, dim3(64 * NW), 0, s, (const _Float16*)q, (const _Float16*)k,
                     (const _Float16*)v, (_Float16*)o, Nq, Nkv, H, c, (const _Float16*)nullptr, (const _Float16*)nullptr, 0, 0.f, ld_kv,
                     ld_kv, ld_q, xcd);
}

extern "C" int gip_attention_fwd_strided2_f16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H,
                                              int32_t Nq, int32_t Nkv, int32_t D, float scale, const void* k2, const void* v2,
                                              int32_t Nkv2, float weight2, int32_t ld_q, int32_t ld_kv, int32_t ld_kv2, void* stream) {
  if (!q || !k || !v || !o || B < 1 || H < 1 || Nq < 32 || Nq % 32 || Nkv < 1 || ld_q < H * D || ld_q % 8) return 1;
  if ((k2 != nullptr) != (v2 != nullptr) || (k2 && Nkv2 < 1)) return 1;
  if (ld_kv < H * D || ld_kv % 8 || (k2 && (ld_kv2 < H * D || ld_kv2 % 8))) return 1;        // 16-byte row loads
  const float c = scale * 1.4426950408889634f;
  const dim3 grid((Nq + AT_BQ - 1) / AT_BQ, B * H);
  hipStream_t s = (hipStream_t)stream;
  // GIP_ATTN_NW (6 / 8; default 4): waves per workgroup of the long self-attention layers (D = 40 / 80, >= 1024 keys); A/B switch
  // long D = 40 self-attention (the 64 x 64 level: 4096 keys): 8 waves per workgroup share the K / V stages.  The plain loop
  // streams every (batch, head)'s keys / values from L2 once per 128 queries — 2 GB per launch at batch 12, and a build without
  // any MFMA or exp2 still takes 2/3 of the time (tools/experiments/attention_pipelined_qk.md, round 4) — 256 queries per workgroup
  // halve that stream: 0.44 -> 0.41 ms at batch 12, 0.155 -> 0.138 at batch 4 (one workgroup per CU, so only where >= 512
  // workgroups remain; D = 80 at 1024 keys measured slower).  GIP_ATTN_NW = 4: the plain loop everywhere (A/B), 8: wherever supported
  static const int nw = [] { const char* e = getenv("GIP_ATTN_NW"); return e && *e ? atoi(e) : 0; }();
  // Two query tiles per wave, 8 waves (512 queries) per workgroup for the long D = 40 layers that leave >= 192 workgroups
  // (attn_fwd_q2_kernel): one wave's matrix and vector work overlap across its two tiles and the K / V stream per query is a
  // quarter of the plain loop's.  Same box, bit-identical outputs: 0.44 -> 0.39 ms at batch 12 on a slow box (0.365 on a fast
  // one), 0.150 -> 0.141 at batch 4, 1.11 -> 0.99 at 16384 keys (refine).  GIP_ATTN_QT = 1: off (A/B); 2: wherever supported
  static const int qt2 = [] { const char* e = getenv("GIP_ATTN_QT"); return e && *e ? atoi(e) : 0; }();
  if (qt2 != 1 && !k2 && D == 40 && Nkv >= 1024 && Nq % 64 == 0 && (qt2 == 2 || ((Nq + 511) / 512) * B * H >= 192)) {
    if (nw == 4) launch_attn_q2<40, 4>(s, q, k, v, o, B * H, Nq, Nkv, H, c, ld_kv, ld_q);
    else launch_attn_q2<40, 8>(s, q, k, v, o, B * H, Nq, Nkv, H, c, ld_kv, ld_q);
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  if (!k2 && D == 40 && Nkv >= 1024 && Nq % 256 == 0 && nw != 4 && (nw == 8 || (Nq / 256) * B * H >= 512)) {
    launch_attn_wide<40, 8>(s, q, k, v, o, B * H, Nq, Nkv, H, c, ld_kv, ld_q);
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  switch (D) {
    case 40: launch_attn<40>(grid, s, q, k, v, o, Nq, Nkv, H, c, k2, v2, Nkv2, weight2, ld_kv, ld_kv2, ld_q); break;
    case 64: launch_attn<64>(grid, s, q, k, v, o, Nq, Nkv, H, c, k2, v2, Nkv2, weight2, ld_kv, ld_kv2, ld_q); break;
    case 80: launch_attn<80>(grid, s, q, k, v, o, Nq, Nkv, H, c, k2, v2, Nkv2, weight2, ld_kv, ld_kv2, ld_q); break;
    case 160: launch_attn<160>(grid, s, q, k, v, o, Nq, Nkv, H, c, k2, v2, Nkv2, weight2, ld_kv, ld_kv2, ld_q); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_attention_fwd_strided_f16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H,
                                             int32_t Nq, int32_t Nkv, int32_t D, float scale, const void* k2, const void* v2,
                                             int32_t Nkv2, float weight2, int32_t ld_kv, int32_t ld_kv2, void* stream) {
  return gip_attention_fwd_strided2_f16(q, k, v, o, B, H, Nq, Nkv, D, scale, k2, v2, Nkv2, weight2, H * D, ld_kv, ld_kv2, stream);
}

extern "C" int gip_attention_fwd_f16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H,
                                     int32_t Nq, int32_t Nkv, int32_t D, float scale, const void* k2, const void* v2,
                                     int32_t Nkv2, float weight2, void* stream) {
  return gip_attention_fwd_strided_f16(q, k, v, o, B, H, Nq, Nkv, D, scale, k2, v2, Nkv2, weight2, H * D, H * D, stream);
}
