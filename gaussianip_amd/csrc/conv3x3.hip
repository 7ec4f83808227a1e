// conv3x3.hip — 3x3 / stride 1 / pad 1 convolution, NHWC fp16 -> fp16 with fp32 accumulation, as an implicit GEMM on
// the gfx950 matrix cores (v_mfma_f32_16x16x32_f16).  See include/gip_nn.h (gip_conv3x3_nhwc_f16).
//
// GEMM view: D[co][m] = sum_k W[co][k] * X[m][k],  m = (n, y, x) output pixel, k = (tap, ci), K = 9 * Cin.
//   * workgroup tile 128 pixels x BN output channels (BN = 128, or 160 so that Cout = 320 splits without waste),
//     4 waves as 2 (pixel halves) x 2 (channel halves), K step 64 = one tap x 64 input channels;
//   * both operands are staged global -> LDS by 16-byte LDS-DMA (buffer_load_dwordx4 ... lds): a pixel row of the K step
//     is the 128 contiguous bytes x[n, y+dy-1, x+dx-1, c0:c0+64] of the NHWC tensor.  The lane's byte offset of the
//     CENTRE pixel is a loop-invariant VGPR, the tap / channel-block displacement is the instruction's SGPR offset, and a
//     tap that falls outside the image gets an out-of-range offset, for which the buffer unit returns zeros — the
//     zero padding costs one v_cndmask per load and no memory;
//   * the LDS image keeps 128-byte rows; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 ON THE SOURCE
//     ADDRESS, which makes every 16-lane group of a ds_read_b128 fragment read cover all 64 banks once;
//   * two LDS stages: the DMA of K step t+1 is in flight while the MFMAs of step t run; one barrier per step;
//   * the MFMA's A operand is the weight tile (rows = output channels) and B the pixel tile, so a lane's four
//     accumulator registers are four consecutive output channels of one pixel = one 8-byte NHWC store;
//   * blockIdx -> tile is remapped so that the tiles sharing a pixel block (same activations, different channel
//     blocks) are consecutive on ONE XCD and hit in its L2;
//   * the epilogue stages the half-rounded tile through the (free) stage buffers and writes whole 16-byte row chunks; the
//     residual rows it adds are requested before that staging; optionally it also leaves the per-channel sums the NEXT
//     GroupNorm needs (forward statistics, or — as a data-gradient kernel — that GroupNorm's two backward reductions).
// Also in this file: conv_big_kernel (256 x 256 tile, 8 waves, one workgroup per CU, for the VAE's 256 / 512-channel levels),
// TAPS = 1 = nn.Linear / 1x1 convolution (+ GEGLU) on the same machinery, split-K with its reduce kernels, and the
// TAP-SUBSET mode (`tapsel`): a launch visits only some of the nine taps and scatters its output to one parity class of a
// tensor of twice the size — the data gradient of a stride-2 convolution and nearest-2x-upsample + convolution, each as
// four small convolutions over the source grid (gip_conv3x3s2_dgrad_nhwc_f16, gip_upsample2x_conv3x3_nhwc_f16).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gip_nn.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CV_BM 128
#define CV_BK 64
#define CV_THREADS 256

#define CV_OOB 0xFFFF0000u                // voffset beyond any num_records: the buffer load returns 0
#define CV_RSRC_FLAGS 0x00020000         // raw buffer, 32-bit data format (gfx9 family)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7; the same function as csrc/groupnorm.hip's GEGLU kernel uses): the GEGLU
// epilogue evaluates 32 gelu per thread and tile, and libm's erff (~40 vector instructions each) made it longer than the K loop
// of the K = 320 projection it ends
__device__ __forceinline__ float cv_erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  return copysignf(fmaf(-p * t, e, 1.f), x);
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voffset, soffset, 0, 0);
}

// Optional second meaning of the epilogue statistics (gnb.x != NULL): the kernel is the DATA GRADIENT of a convolution whose
// input was y = silu?(GroupNorm(x + addend)); its output `out` is dL/dy, and the per-channel sums it leaves in chan_stats are
// the two reductions of that GroupNorm's backward, sum(dxh) and sum(dxh * xh) with xh = (x + addend - mean) * rstd,
// dxh = dL/dy * dsilu?(gamma xh + beta) * gamma — what gn_reduce_kernel<1> (csrc/groupnorm.hip) computes in a separate pass
// over x and dL/dy.  Needs HW % (rows of a tile) == 0: a tile lies inside one sample.
struct GnBwdArgs {
  const _Float16* x;        // the GroupNorm's input [N, HW, C] (C = this kernel's Cout)
  const _Float16* gamma;
  const _Float16* beta;
  const _Float16* addend;   // optional per-(sample, channel) addend, row stride addend_stride (0 = one row)
  const float* mean;        // [N, G]
  const float* rstd;
  int G, silu, addend_stride, HW;
  // --- LayerNorm folded into a TAPS = 1 GEMM (gip_linear_ln_f16): out = LN(x) W^T + b computed as
  //       out[m][n] = rstd_m (x_m . (W gamma)_n) - rstd_m mu_m s_n + t_n,   s_n = sum_k (W gamma)[n][k],  t_n = sum_k W[n][k] beta_k + b_n
  //     with x read RAW (the LayerNorm kernel and its round trip through memory disappear); mu_m / rstd_m come from the per-row
  //     partial sums [M][ln_parts][2] = (sum, sum of squares) that the kernel which PRODUCED x left in its epilogue (rows_out)
  const float* ln_rows;     // NULL = no LayerNorm fold
  const float* ln_s;        // [Nout] (GEGLU: [2 Nout]) float32
  const float* ln_t;
  float* rows_out;          // [M][n_tiles][2] per-row (sum, sum of squares) of this kernel's final output over its channel tile; NULL = none
  int ln_parts;
  float ln_inv_c, ln_eps;
};

struct GnBwdLane {          // one lane's eight channels
  float ga[8], be[8], mu[8], rs[8], ad[8];
};

__device__ __forceinline__ void gnb_load(const GnBwdArgs& a, int n, int c0, int C, GnBwdLane& L) {
  const int cg = C / a.G;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int c = c0 + j, g = c / cg;
    L.ga[j] = (float)a.gamma[c];
    L.be[j] = (float)a.beta[c];
    L.mu[j] = a.mean[n * a.G + g];
    L.rs[j] = a.rstd[n * a.G + g];
    L.ad[j] = a.addend ? (float)a.addend[(size_t)n * a.addend_stride + c] : 0.f;
  }
}

__device__ __forceinline__ void gnb_accumulate(const GnBwdArgs& a, const GnBwdLane& L, const f16x8& dy, const f16x8& xv, float* s8, float* q8) {
  // (NOT hoisting the wave-uniform `silu` test out of this loop: the two-loop form made the register allocator spill 144-192 bytes per
  // lane in the convolution kernels this is inlined into — VAE forward + backward 13.3 -> 14.1 ms, round 5)
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float xh = ((float)xv[j] + L.ad[j] - L.mu[j]) * L.rs[j];
    float g = (float)dy[j];
    if (a.silu) {
      const float v = L.ga[j] * xh + L.be[j];
      const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-v));      // v_rcp_f32, not the IEEE division sequence
      g *= sg * (1.f + v * (1.f - sg));
    }
    const float dxh = g * L.ga[j];
    s8[j] += dxh;
    q8[j] = fmaf(dxh, xh, q8[j]);
  }
}

// TAPS = 9: the 3x3 convolution.  TAPS = 1: the same machinery as a plain GEMM out[m][co] = sum_k x[m][k] w[co][k]
// (nn.Linear / 1x1 convolution on the NHWC token view; H = 1, W = M).  GEGLU (TAPS = 1 only): w has 2 * Cout rows
// [value | gate]; a workgroup computes 64 value and the matching 64 gate columns and writes value * gelu(gate).
// HALO (TAPS = 9, Cin = 128, stride 1, pad 1, all nine taps, no split-K, H % 8 == 0, W % 16 == 0): the pixel tile is a 16 x 8
// BLOCK of the image instead of 128 consecutive pixels, and its 18 x 10 x 128-channel halo is brought into LDS ONCE (46 KB);
// the nine taps then read their fragments from that one image at shifted rows.  The K loop only streams the weights: 16 KB of
// LDS-DMA per K step instead of 32, plus 46 KB once instead of 9 x 32 KB of pixels.  Why: at Cin = 128 (the VAE encoder's first
// level, 4 x 512^2: 8 of its convolutions, 4.1 ms of the training step at 675 TFLOP/s) the K loop is only 18 steps and the
// kernel is bound by its L2 -> LDS traffic (tools/exp_conv_ablate.py: the DMA stream alone = 70 % of the kernel).
#define CVH_ROWS 184                      // halo rows held per channel block (180 used: 10 x 18; waves 0-2 of the sixth round)
#define CVH_KC_BYTES (CVH_ROWS * 128)
// KG = 2 (round 6, TAPS = 1 GEMMs on grids of at most one workgroup per CU): TWO K groups of four waves inside one workgroup.  A lone
// workgroup's K step is the issue of its LDS-DMA instructions + its MFMAs in series inside each wave (tools/experiments/
// conv3x3_four_stage.diff.txt), which two co-resident workgroups interleave — but a small grid has no second workgroup.  Here group
// g stages and multiplies its half of the K steps in its own pair of stage buffers, the halves run interleaved on the CU's four
// SIMDs, group 1 hands its accumulators over through LDS and group 0 runs the (unchanged) epilogue: half the serial K steps per
// tile, no split-K slabs, no reduce launch.  The sum is (first half) + (second half) instead of one chain: deterministic, not
// bit-identical to KG = 1.  (Extended to whole-K 3x3 convolutions of 171..256 tiles — the shared prefix at batch 4, a shard's
// 64 x 64 level — in place of split-K = 2, it measured SLOWER in the step: full step 33.25 -> 33.38 ms, and it gave back 0.09 of the
// shard's 0.22 ms: eight waves holding 147 KB of LDS keep the OTHER stream's workgroups off the CU in the two-stream denoise.)
template <int BN, int STAGES, int TAPS, bool GEGLU, bool HALO = false, int KG = 1>
__global__ void __launch_bounds__(CV_THREADS * KG, 2)
conv3x3_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const _Float16* __restrict__ bias,
               const _Float16* __restrict__ residual, _Float16* __restrict__ out, int N, int H, int W, int Cin, int Cout,
               int m_tiles, int n_tiles, int ksplit, float* __restrict__ partial, int Hin, int Win, int geom,
               float* __restrict__ chan_stats, GnBwdArgs gnb, int tapsel_in, long long bs_x = 0, long long bs_w = 0, long long bs_o = 0) {
  // BATCHED GEMM (TAPS = 1, gip_linear_batched_f16): blockIdx.y = batch entry; x / w / out advance by bs_x / bs_w / bs_o elements
  // per entry (the sixteen products of a Winograd F(2x2, 3x3) convolution in ONE launch)
  if constexpr (TAPS == 1) {
    x += (size_t)blockIdx.y * bs_x; w += (size_t)blockIdx.y * bs_w; out += (size_t)blockIdx.y * bs_o;
  }
  // geom = stride | pad_top << 8 | pad_left << 16; H, W are the OUTPUT dims, Hin, Win the input dims (equal at stride 1)
  // tapsel (TAPS = 9): bits 0-8 = the taps the K loop visits (0x1ff: all nine); bit 11 = scatter: output pixel (n, a, b) is
  // written to row (n, 2 a + pi, 2 b + pj) of a [N, 2 H, 2 W, Cout] tensor, pi = bit 9, pj = bit 10 — one parity class of
  // the DATA GRADIENT of a stride-2 convolution (gip_conv3x3s2_dgrad_nhwc_f16 below).  Bit 12 = all four parity classes in
  // ONE launch: class c = blockIdx.x / (tiles * ksplit) picks its tap set from a table (bit 13 = which: 0 the stride-2 data
  // gradient, 1 nearest-2x-upsample + convolution) and its weight block w + c * Cout * 9 * Cin.
  int tapsel = tapsel_in;
  unsigned bid = blockIdx.x;
  if (TAPS == 9 && ((tapsel_in >> 12) & 1)) {
    const unsigned per = (unsigned)(m_tiles * n_tiles * ksplit), cls = bid / per;
    bid -= cls * per;
    const int pi = (int)(cls >> 1), pj = (int)(cls & 1);
    int mask;
    if ((tapsel_in >> 13) & 1) mask = pi ? (pj ? 0x1b0 : 0x0d8) : (pj ? 0x036 : 0x01b);      // rows {0,1} | {1,2} x columns {0,1} | {1,2}
    else mask = pi ? (pj ? 0x010 : 0x018) : (pj ? 0x012 : 0x01b);
    tapsel = mask | (pi << 9) | (pj << 10) | (1 << 11);
    w += (size_t)cls * Cout * 9 * Cin;
  }
  const int cstride = geom & 0xff, pad_t = (geom >> 8) & 0xff, pad_l = (geom >> 16) & 0xff;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int A_BYTES = HALO ? 0 : CV_BM * 128;   // pixel tile: 128 rows x 64 halves (HALO: the pixels live in the halo image)
  constexpr int B_BYTES = BN * 128;             // weight tile: BN rows x 64 halves
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int HALO_BYTES = HALO ? 2 * CVH_KC_BYTES : 0;       // [2 channel blocks][184 halo rows][128 B], in front of the stages
  static_assert(!HALO || (TAPS == 9 && !GEGLU), "halo mode: 3x3 convolution");
  static_assert(KG == 1 || (KG == 2 && TAPS == 1 && !HALO && !GEGLU), "two K groups: plain GEMMs only");
  constexpr int B_ROUNDS = BN / 32;
  constexpr int NI = BN / 32;                   // 16-channel MFMA tiles per wave (its half of BN)
  const int kg = KG > 1 ? (int)threadIdx.x / CV_THREADS : 0;           // K group of this wave (wave-uniform)
  const int tid = KG > 1 ? (int)threadIdx.x % CV_THREADS : (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned char* const gsm = smem + (KG > 1 ? kg * (STAGES * STAGE) : 0);      // this group's stage buffers
  const int wm = wave & 1, wn = wave >> 1;

  // tile of this workgroup (bijective XCD remap: ids congruent mod 8 share an XCD)
  // split-K (small pixel counts: too few output tiles to fill the chip): blockIdx.x = split * total + tile, every split
  // accumulates its share of the K steps and stores an fp32 slab; conv_splitk_reduce_kernel sums the slabs
  const int total = m_tiles * n_tiles, id = (int)(bid % (unsigned)total), split = (int)(bid / (unsigned)total);
  const int q = total >> 3, r = total & 7, xcd = id & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  // tile order inside an XCD's contiguous chunk (geom bit 24): m-major keeps the tiles that share a PIXEL block together
  // (the activations are the big operand: VAE, 64^2 / 32^2 levels), n-major the tiles that share a WEIGHT block (16^2 /
  // 8^2 levels: 30-60 MB of weights against a few MB of activations) — whichever operand is larger is then fetched into
  // each XCD's L2 once instead of eight times
  const bool nmajor = (geom >> 24) & 1;
  const int mt = nmajor ? t % m_tiles : t / n_tiles, nt = nmajor ? t / m_tiles : t - (t / n_tiles) * n_tiles;
  const unsigned M = (unsigned)N * H * W;          // < 2^31 (checked on the host): 32-bit index arithmetic throughout
  const unsigned m0 = (unsigned)mt * CV_BM;
  // HALO: tile mt = block (n, by, bx) of 8 rows x 16 columns; tile row r = pixel (by * 8 + (r >> 4), bx * 16 + (r & 15)).
  // Tiles stay sample-major (mt / (HW / 128) = n), so the per-128-row statistics blocks still never straddle two samples.
  unsigned h_n = 0, h_y0 = 0, h_x0 = 0;
  if constexpr (HALO) {
    const unsigned tiles_x = (unsigned)W >> 4, per_img = tiles_x * ((unsigned)H >> 3);
    h_n = (unsigned)mt / per_img;
    const unsigned rem = (unsigned)mt - h_n * per_img, by = rem / tiles_x;
    h_y0 = by * 8u; h_x0 = (rem - by * tiles_x) * 16u;
  }
  auto row_m = [&](int row) -> unsigned {          // linear NHWC pixel index of tile row `row`
    if constexpr (HALO) return (h_n * (unsigned)H + h_y0 + ((unsigned)row >> 4)) * (unsigned)W + h_x0 + ((unsigned)row & 15u);
    else return m0 + (unsigned)row;
  };
  const int co0 = nt * (GEGLU ? BN / 2 : BN);      // first OUTPUT channel of the tile
  const int HW = H * W;

  // ---- per-thread DMA descriptors (fixed over the K loop) ----
  const int sub_row = wave * 8 + (lane >> 3);            // row inside a 32-row round
  const int pchunk = lane & 7;                           // physical 16-byte chunk this lane fills
  unsigned a_off[4];                                     // byte offset of the centre pixel's channel 0 (+ swizzled chunk)
  unsigned a_mask[4];                                    // 9 validity bits, one per tap
#pragma unroll
  for (int i = 0; i < (HALO ? 0 : 4); i++) {
    const int row = i * 32 + sub_row;
    const unsigned m = m0 + row;
    const int lchunk = pchunk ^ ((row >> 1) & 7);
    unsigned mask = 0;
    unsigned off = 0;
    if (m < M) {
      if constexpr (TAPS == 9) {
        const unsigned n = m / (unsigned)HW, rem = m - n * (unsigned)HW;
        const int y = (int)(rem / (unsigned)W), xx = (int)(rem - (unsigned)y * W);
        const int iy0 = y * cstride - pad_t, ix0 = xx * cstride - pad_l;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++)
            if ((unsigned)(iy0 + dy) < (unsigned)Hin && (unsigned)(ix0 + dx) < (unsigned)Win) mask |= 1u << (dy * 3 + dx);
        off = (((n * (unsigned)Hin + (unsigned)(y * cstride)) * (unsigned)Win + (unsigned)(xx * cstride)) * (unsigned)Cin + lchunk * 8) * 2u;
      } else {
        mask = 1u;
        off = (m * (unsigned)Cin + lchunk * 8) * 2u;
      }
    }
    a_off[i] = off;
    a_mask[i] = mask;
  }
  unsigned b_off[B_ROUNDS];
#pragma unroll
  for (int i = 0; i < B_ROUNDS; i++) {
    const int row = i * 32 + sub_row;
    const int lchunk = pchunk ^ ((row >> 1) & 7);
    int wrow = co0 + row;                                  // row of w this tile row holds
    bool ok = wrow < Cout;
    if constexpr (GEGLU) {                                 // per channel half: [32 value rows | 32 gate rows]
      const int half = row / (BN / 2), j = row - half * (BN / 2);
      const int ch = co0 + half * (BN / 4) + (j % (BN / 4));
      ok = ch < Cout;
      wrow = j < BN / 4 ? ch : Cout + ch;
    }
    b_off[i] = ok ? (unsigned)(wrow * TAPS * Cin + lchunk * 8) * 2u : CV_OOB;
  }
  // buffer resources: activations based one row + one pixel BEFORE x so that every tap displacement is >= 0
  const unsigned shift = TAPS == 9 ? (unsigned)(pad_t * Win + pad_l) * Cin * 2u : 0u;
  const unsigned Min = TAPS == 9 ? (unsigned)N * Hin * Win : M;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)x - shift), 0, (int)(Min * (unsigned)Cin * 2u + (unsigned)(2 * Win + 2) * Cin * 2u + shift), CV_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t wr =
      __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((unsigned)Cout * (GEGLU ? 2u : 1u) * TAPS * Cin * 2u), CV_RSRC_FLAGS);

  const int cblocks = Cin / CV_BK;
  const int KT_all = (TAPS == 9 ? __builtin_popcount(tapsel & 0x1ff) : TAPS) * cblocks;
  const int kpart = split * KG + kg, kparts = ksplit * KG;       // K range of this (split, group)
  const int kt_begin = (int)((long long)kpart * KT_all / kparts), kt_end = (int)((long long)(kpart + 1) * KT_all / kparts);
  const int KT = kt_end - kt_begin;
  // the groups of a workgroup share its barriers: both run the longer group's trip count (KG = 2: the second half has the odd step)
  const int KT_loop = KG > 1 ? KT_all - (int)((long long)(KG - 1) * KT_all / KG) : KT;

  auto stage = [&](int tap, int cb, int buf) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const unsigned tap_off = (unsigned)((TAPS == 9 ? (dy * Win + dx) * Cin : 0) + cb * CV_BK) * 2u;   // relative to the shifted base
    const unsigned wtap_off = (unsigned)(tap * Cin + cb * CV_BK) * 2u;
    if constexpr (!HALO) {
      unsigned char* sa = gsm + buf * STAGE + wave * 1024;
      if (!((geom >> 26) & 1)) {
#pragma unroll
        for (int i = 0; i < 4; i++) dma16(xr, ((a_mask[i] >> tap) & 1u) ? a_off[i] : CV_OOB, tap_off, sa + i * 4096);
      }
    }
    unsigned char* sb = gsm + HALO_BYTES + buf * STAGE + A_BYTES + wave * 1024;
    if (!((geom >> 27) & 1)) {
#pragma unroll
      for (int i = 0; i < B_ROUNDS; i++) dma16(wr, b_off[i], wtap_off, sb + i * 4096);
    }
  };

  f32x4 acc[NI][4];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addressing: row = tile base + (lane & 15); swizzle term (row >> 1) & 7 == (lane >> 1) & 7 for every tile
  const int frag_row = lane & 15, swz = (lane >> 1) & 7, kq = lane >> 4;
  const int pix_base = (wm * 64 + frag_row) * 128;
  const int ch_base = A_BYTES + (wn * (BN / 2) + frag_row) * 128;
  // HALO: halo row of this lane's pixel of fragment tile mi at the (0, 0) tap: (wm * 4 + mi) * 18 + (lane & 15); tap (dy, dx)
  // adds dy * 18 + dx.  The chunk swizzle is that of the halo row the fragment row sits in.
  const int hrow0 = wm * 4 * 18 + frag_row;

  auto compute = [&](const unsigned char* sbuf, int tap_c, int cb_c) {
    int hoff[4], hswz[4];
    if constexpr (HALO) {
      const int shift_rows = (tap_c / 3) * 18 + (tap_c - (tap_c / 3) * 3);
#pragma unroll
      for (int mi = 0; mi < 4; mi++) {
        const int hr = hrow0 + mi * 18 + shift_rows;
        hoff[mi] = cb_c * CVH_KC_BYTES + hr * 128;
        hswz[mi] = (hr >> 1) & 7;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int pc = ((ks * 4 + kq) ^ swz) * 16;
      f16x8 pix[4], wt[NI];
#pragma unroll
      for (int mi = 0; mi < 4; mi++) {
        if constexpr (HALO) pix[mi] = *(const f16x8*)(smem + hoff[mi] + (((ks * 4 + kq) ^ hswz[mi]) << 4));
        else pix[mi] = *(const f16x8*)(sbuf + pix_base + mi * 2048 + pc);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ni++) wt[ni] = *(const f16x8*)(sbuf + ch_base + ni * 2048 + pc);
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[ni], pix[mi], acc[ni][mi], 0, 0, 0);
    }
  };

  // K step the NEXT stage() call loads: (tap, channel block); the taps run over the set bits of tapsel in ascending order
  int taps_left = TAPS == 9 ? (tapsel & 0x1ff) : 1;
  for (int i = 0; i < kt_begin / cblocks; i++) taps_left &= taps_left - 1;
  int tap = TAPS == 9 ? __builtin_ctz(taps_left | 0x200) : 0, cb = kt_begin - (kt_begin / cblocks) * cblocks;
  auto advance = [&]() {
    if (++cb == cblocks) {
      cb = 0;
      taps_left &= taps_left - 1;
      tap = TAPS == 9 ? __builtin_ctz(taps_left | 0x200) : 0;
    }
  };
  // two LDS stages: the DMA of step t + 1 is in flight while the MFMAs of step t run; one wait + barrier per step.  (A
  // three-stage single-workgroup-per-CU variant and BK = 32 variants with 3 / 4 stages and counted vmcnt were measured
  // slower: tools/experiments/conv3x3_bk32_multistage.hip.txt holds both.)
  static_assert(STAGES == 2, "one pipeline depth is built");
  if constexpr (HALO) {
    // the halo image, once: 6 rounds of 32 rows per channel block (round 5: waves 0-2 only, rows 160..183; rows >= 180 and
    // pixels outside the image read zeros from an out-of-range offset).  Row r = halo pixel (r / 18, r % 18) = image pixel
    // (y0 - 1 + r / 18, x0 - 1 + r % 18); chunk swizzle (r >> 1) & 7 on the source address as in the stage images.
    const __amdgpu_buffer_rsrc_t xh = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(M * (unsigned)Cin * 2u), CV_RSRC_FLAGS);
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (i == 5 && wave == 3) break;
      const int hr = i * 32 + sub_row;
      const int hy = hr / 18, hx = hr - hy * 18;
      const int iy = (int)h_y0 - 1 + hy, ix = (int)h_x0 - 1 + hx;
      const int lchunk = pchunk ^ ((hr >> 1) & 7);
      const bool ok = hr < 180 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const unsigned off = ok ? (((h_n * (unsigned)H + (unsigned)iy) * (unsigned)W + (unsigned)ix) * (unsigned)Cin + lchunk * 8) * 2u : CV_OOB;
      dma16(xh, off, 0u, smem + i * 4096 + wave * 1024);
      dma16(xh, off, 128u, smem + CVH_KC_BYTES + i * 4096 + wave * 1024);
    }
  }
  int tap_c = tap, cb_c = cb;                      // (tap, channel block) of the step being COMPUTED (stage() runs one ahead)
  if (KG == 1 || KT > 0) {
    stage(tap, cb, 0);
    advance();
  }
  // the bias of this lane's output channels is requested here, under the first operand DMA, not at the head of the epilogue (where its
  // round trip was exposed once per tile: 1 us of a 22 us tile at the VAE's first level)
  [[maybe_unused]] f16x4 bias4[NI];
  if constexpr (!GEGLU) {
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      const int cl = wn * (BN / 2) + ni * 16 + (lane >> 4) * 4;
      bias4[ni] = (bias && co0 + cl < Cout) ? *(const f16x4*)(bias + co0 + cl) : (f16x4){0, 0, 0, 0};
    }
  }
  // GN-IN: this thread's eight (scale, shift) pairs are requested while the halo DMA is in flight
  [[maybe_unused]] const int gn_cbk = (tid >> 3) & 1, gn_lch = tid & 7;
  [[maybe_unused]] float sc8[8], sh8[8];
  if constexpr (HALO) {
    if ((geom >> 30) & 1) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int c = gn_cbk * 64 + gn_lch * 8 + j, g = c / (128 / gnb.G);
        const float ad = gnb.addend ? (float)gnb.addend[(size_t)h_n * gnb.addend_stride + c] : 0.f;
        sc8[j] = gnb.rstd[h_n * gnb.G + g] * (float)gnb.gamma[c];
        sh8[j] = (float)gnb.beta[c] - (gnb.mean[h_n * gnb.G + g] - ad) * sc8[j];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (HALO) {
    if ((geom >> 30) & 1) {
      // GN-IN (round 5): the convolution's input is y = silu?(GroupNorm(x + addend)) and y is never materialised — the halo image
      // holds RAW x and is normalised in place here, once per tile (gnb = that GroupNorm: gamma, beta, mean, rstd, addend over
      // the Cin = 128 input channels).  Same arithmetic as csrc/groupnorm.hip's apply pass (sc = rstd gamma, sh = beta - (mean -
      // addend) sc, y = sc x + sh, SiLU through v_exp + v_rcp, one rounding to half), so the result equals convolving the
      // separately normalised tensor; halo pixels outside the image stay zero (the padding applies to y, not to x).  Saves
      // the apply pass' read + write of the activation (0.12 ms per 268 MB tensor of the VAE encoder's first level) for ~12
      // 16-byte chunks of vector work per thread and tile.
      // a thread keeps ONE 8-channel chunk (its scale / shift in registers: tid & 15 = channel block x logical chunk) and walks the
      // halo rows tid >> 4, + 16, ...: 11-12 chunks of 16 bytes per thread, no table in LDS, one LDS round trip per chunk
      const int lch = gn_lch;
      unsigned char* hb = smem + gn_cbk * CVH_KC_BYTES;
#pragma unroll 4
      for (int hr = tid >> 4; hr < 180; hr += CV_THREADS / 16) {
        const int hy = hr / 18, hx = hr - hy * 18;
        const int iy = (int)h_y0 - 1 + hy, ix = (int)h_x0 - 1 + hx;
        if (!((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)) continue;      // outside the image: y is zero-padded
        f16x8* ptr = (f16x8*)(hb + hr * 128 + ((lch ^ ((hr >> 1) & 7)) << 4));
        const f16x8 v = *ptr;
        f16x8 o;
        if (gnb.silu) {        // wave-uniform, outside the element loop: the eight exp / rcp chains overlap
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const float y = sc8[j] * (float)v[j] + sh8[j];
            o[j] = (_Float16)(y * __builtin_amdgcn_rcpf(1.f + __expf(-y)));
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] = (_Float16)(sc8[j] * (float)v[j] + sh8[j]);
        }
        *ptr = o;
      }
      __syncthreads();
    }
  }
  for (int kt = 0; kt < KT_loop; kt++) {
    const int buf = kt & 1;
    const int tap_n = tap, cb_n = cb;
    if (kt + 1 < KT) { stage(tap, cb, buf ^ 1); advance(); }
    if (!((geom >> 28) & 1) && (KG == 1 || kt < KT)) compute(gsm + HALO_BYTES + buf * STAGE, tap_c, cb_c);
    tap_c = tap_n; cb_c = cb_n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if constexpr (KG == 2) {
    // group 1's partial products -> LDS (every lane its own f32x4 slots: conflict-free), group 0 adds them and goes on alone
    f32x4* red = (f32x4*)smem;                     // NI * 4 * 256 x 16 B <= the two groups' stage buffers (checked at the launch)
    if (kg == 1) {
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++) red[(ni * 4 + mi) * CV_THREADS + tid] = acc[ni][mi];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int mi = 0; mi < 4; mi++) {
        const f32x4 o = red[(ni * 4 + mi) * CV_THREADS + tid];
        acc[ni][mi][0] += o[0]; acc[ni][mi][1] += o[1]; acc[ni][mi][2] += o[2]; acc[ni][mi][3] += o[3];
      }
    asm volatile("s_barrier" ::: "memory");        // group 0's four waves: `red` is about to be reused by the epilogue's tile image
  }

  // ---- LayerNorm fold: (rstd, -rstd * mu) of the tile's 128 rows from the producer's per-row partial sums, kept in the last KB
  //      of the (now free) stage buffers — behind everything the epilogues below stage there
  const bool ln_fold = TAPS == 1 && gnb.ln_rows != nullptr;
  float* lnbuf = (float*)(smem + HALO_BYTES + STAGES * STAGE - 1024);
  if (ln_fold) {
    if (tid < CV_BM) {
      const unsigned m = row_m(tid);
      float rs = 0.f, a = 0.f;
      if (m < M) {
        float S = 0.f, Q = 0.f;
        const float* pr = gnb.ln_rows + (size_t)m * gnb.ln_parts * 2;
        for (int pp = 0; pp < gnb.ln_parts; pp++) { S += pr[2 * pp]; Q += pr[2 * pp + 1]; }      // fixed order
        const float mu = S * gnb.ln_inv_c;
        const float var = fmaxf(Q * gnb.ln_inv_c - mu * mu, 0.f);
        rs = rsqrtf(var + gnb.ln_eps);
        a = -rs * mu;
      }
      lnbuf[2 * tid] = rs; lnbuf[2 * tid + 1] = a;
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds out[pixel = lane & 15][co = (lane >> 4) * 4 + 0..3] of each 16x16 tile ----
  const bool scatter = TAPS == 9 && ((tapsel >> 11) & 1);
  auto out_row = [&](unsigned m) -> size_t {      // row of `out` that output pixel m is written to
    if (!scatter) return (size_t)m;
    const unsigned n = m / (unsigned)HW, rem = m - n * (unsigned)HW, a = rem / (unsigned)W, b = rem - a * (unsigned)W;
    return ((size_t)n * (2u * H) + 2u * a + ((tapsel >> 9) & 1)) * (2u * W) + 2u * b + ((tapsel >> 10) & 1);
  };
  auto add4 = [](f32x4& v, const _Float16* p) {
    const f16x4 b = *(const f16x4*)p;
    v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
  };
  if (ksplit > 1) {
    if constexpr (!GEGLU) {
      float* slab = partial + (size_t)split * M * Cout;
#pragma unroll
      for (int mi = 0; mi < 4; mi++) {
        const unsigned m = m0 + wm * 64 + mi * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
          const int co = co0 + wn * (BN / 2) + ni * 16 + (lane >> 4) * 4;
          if (co < Cout) *(f32x4*)(slab + (size_t)m * Cout + co) = acc[ni][mi];
        }
      }
    }
    return;
  }
  if constexpr (!GEGLU) {
    if (((geom >> 25) & 1) && !(Cout & 7)) {
      // ---- coalesced epilogue through LDS (the stage buffers are free: every wave has passed the loop's last barrier).
      // The accumulator layout gives a lane 4 channels of one pixel = 8-byte stores 32 contiguous bytes apiece (and the
      // same shape for the residual loads): the memory pipe sees eight times the instructions a full-row access needs.
      // Here the tile is rounded to half once (+ bias), laid out [128 pixels][BN channels] in LDS, and written with
      // 16 bytes per lane, 16 (20) lanes per 256 (320)-byte row; the residual is read the same way and added to the
      // half-rounded convolution output — fp16(fp16(conv + bias) + residual), diffusers' own arithmetic for
      // `input_tensor + hidden_states` in ResnetBlock2D.
      constexpr int ROWB = BN * 2 + 16;                 // padded row: 16-byte aligned, 8-byte writes at most 2-way conflicted
      constexpr int CH = BN / 8, RPP = CV_THREADS / CH;     // 16-byte chunks per row, rows per pass
      constexpr int RPT = (CV_BM + RPP - 1) / RPP;          // rows per thread
      const int chunk = tid % CH, r0 = tid / CH;
      const int co = co0 + chunk * 8;
      const bool mine = tid < RPP * CH && co < Cout;
      // the residual rows this thread will add are requested FIRST (geom bit 29): their HBM latency then overlaps the
      // accumulator -> LDS staging and its barrier instead of starting after them
      // (the data-gradient + GroupNorm-backward mode has no residual: its early rows are the GroupNorm input it reads instead)
      const _Float16* early_src = residual ? residual : ((chan_stats && gnb.x) ? gnb.x : nullptr);
      const bool res_early = early_src && ((geom >> 29) & 1);
      f16x8 rres[RPT];
      if (res_early && mine) {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const int row = r0 + k * RPP;
          const unsigned m = row_m(row);
          rres[k] = (row < CV_BM && m < M) ? *(const f16x8*)(early_src + (size_t)m * Cout + co) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ni++) {
        const int cl = wn * (BN / 2) + ni * 16 + (lane >> 4) * 4;
        const f32x4 b4 = (f32x4){(float)bias4[ni][0], (float)bias4[ni][1], (float)bias4[ni][2], (float)bias4[ni][3]};
        f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f}, t4 = s4;
        if (ln_fold && co0 + cl < Cout) { s4 = *(const f32x4*)(gnb.ln_s + co0 + cl); t4 = *(const f32x4*)(gnb.ln_t + co0 + cl); }
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
          const int p = wm * 64 + mi * 16 + (lane & 15);
          const f32x4 v = acc[ni][mi];
          f16x4 o;
          if (ln_fold) {       // rstd (x . W') - rstd mu s + t  (uniform branch; the plain path below is unchanged bit for bit)
            const float rs = lnbuf[2 * p], a = lnbuf[2 * p + 1];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (_Float16)fmaf(rs, v[j], fmaf(a, s4[j], t4[j]));
          } else {
            o[0] = (_Float16)(v[0] + b4[0]); o[1] = (_Float16)(v[1] + b4[1]); o[2] = (_Float16)(v[2] + b4[2]); o[3] = (_Float16)(v[3] + b4[3]);
          }
          *(f16x4*)(smem + p * ROWB + cl * 2) = o;
        }
      }
      __syncthreads();
      // per-channel sum / sum of squares of the FINAL half-rounded outputs of this tile (chan_stats != NULL): the
      // statistics pass of the GroupNorm that consumes this tensor (gip_gn_finalize_stats) — it never re-reads the tensor
      float s8[8], q8[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { s8[j] = 0.f; q8[j] = 0.f; }
      // per-ROW sums of the final output over this tile's channels (rows_out: the LayerNorm that consumes this tensor is folded
      // into its consumer GEMM): each thread leaves the sums of its 8 channels per row in LDS [row][chunk] (the region the
      // per-channel statistics would use: the two are never requested together), 128 threads then add a row's chunks in order
      float* rowpart = (float*)(smem + CV_BM * ROWB);
      const bool rows_wanted = gnb.rows_out != nullptr && !chan_stats;
      if (rows_wanted && tid < RPP * CH && !mine) {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const int row = r0 + k * RPP;
          if (row < CV_BM) { rowpart[(row * CH + chunk) * 2] = 0.f; rowpart[(row * CH + chunk) * 2 + 1] = 0.f; }
        }
      }
      if (mine) {
        GnBwdLane gl;
        if (chan_stats && gnb.x) gnb_load(gnb, (int)(HALO ? h_n : m0 / (unsigned)gnb.HW), co, Cout, gl);
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const int row = r0 + k * RPP;
          const unsigned m = row_m(row);
          if (row >= CV_BM || m >= M) break;
          f16x8 v = *(const f16x8*)(smem + row * ROWB + chunk * 16);
          if (residual) {
            const f16x8 rr = res_early ? rres[k] : *(const f16x8*)(residual + (size_t)m * Cout + co);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (_Float16)((float)v[j] + (float)rr[j]);
          }
          *(f16x8*)(out + out_row(m) * Cout + co) = v;
          if (rows_wanted) {
            float ps = 0.f, pq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) { const float f = (float)v[j]; ps += f; pq = fmaf(f, f, pq); }
            rowpart[(row * CH + chunk) * 2] = ps; rowpart[(row * CH + chunk) * 2 + 1] = pq;
          }
          if (chan_stats) {
            if (gnb.x) {
              gnb_accumulate(gnb, gl, v, res_early ? rres[k] : *(const f16x8*)(gnb.x + (size_t)m * Cout + co), s8, q8);
            } else {
#pragma unroll
              for (int j = 0; j < 8; j++) { const float f = (float)v[j]; s8[j] += f; q8[j] = fmaf(f, f, q8[j]); }
            }
          }
        }
      }
      if (rows_wanted) {                                     // kernel argument: uniform over the workgroup
        if (mine) {                                          // rows past M in a ragged last tile: zeros (the loop above broke before them)
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            const int row = r0 + k * RPP;
            if (row < CV_BM && row_m(row) >= M) { rowpart[(row * CH + chunk) * 2] = 0.f; rowpart[(row * CH + chunk) * 2 + 1] = 0.f; }
          }
        }
        __syncthreads();
        if (tid < CV_BM) {
          const unsigned m = row_m(tid);
          if (m < M) {
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int c = 0; c < CH; c++) { S += rowpart[(tid * CH + c) * 2]; Q += rowpart[(tid * CH + c) * 2 + 1]; }      // fixed order
            float* o = gnb.rows_out + ((size_t)m * n_tiles + nt) * 2;
            o[0] = S; o[1] = Q;
          }
        }
      }
      if (chan_stats) {                                      // kernel argument: uniform over the workgroup
        float* part = (float*)(smem + CV_BM * ROWB);         // [RPP][BN][2] behind the tile image (fits: see launch())
        if (tid < RPP * CH) {
#pragma unroll
          for (int j = 0; j < 8; j++) {
            part[((r0 * BN) + chunk * 8 + j) * 2] = s8[j];
            part[((r0 * BN) + chunk * 8 + j) * 2 + 1] = q8[j];
          }
        }
        __syncthreads();
        if (tid < BN && co0 + tid < Cout) {
          float S = 0.f, Q = 0.f;
#pragma unroll
          for (int r = 0; r < RPP; r++) { S += part[(r * BN + tid) * 2]; Q += part[(r * BN + tid) * 2 + 1]; }   // fixed order
          float* o = chan_stats + ((size_t)mt * Cout + co0 + tid) * 2;
          o[0] = S; o[1] = Q;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; mi++) {
    const unsigned m = m0 + wm * 64 + mi * 16 + (lane & 15);
    if (m >= M) continue;
    if constexpr (GEGLU) {
#pragma unroll
      for (int ni = 0; ni < NI / 2; ni++) {
        const int co = co0 + wn * (BN / 4) + ni * 16 + (lane >> 4) * 4;
        if (co >= Cout) continue;
        f32x4 v = acc[ni][mi], g = acc[ni + NI / 2][mi];
        if (ln_fold) {
          const int p = wm * 64 + mi * 16 + (lane & 15);
          const float rs = lnbuf[2 * p], a = lnbuf[2 * p + 1];
          const f32x4 sv = *(const f32x4*)(gnb.ln_s + co), tv = *(const f32x4*)(gnb.ln_t + co);
          const f32x4 sg = *(const f32x4*)(gnb.ln_s + Cout + co), tg = *(const f32x4*)(gnb.ln_t + Cout + co);
#pragma unroll
          for (int j = 0; j < 4; j++) { v[j] = fmaf(rs, v[j], fmaf(a, sv[j], tv[j])); g[j] = fmaf(rs, g[j], fmaf(a, sg[j], tg[j])); }
        } else if (bias) { add4(v, bias + co); add4(g, bias + Cout + co); }
        f16x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (_Float16)(v[j] * (0.5f * g[j] * (1.f + cv_erf_fast(g[j] * 0.70710678118654752f))));
        *(f16x4*)(out + out_row(m) * Cout + co) = o;
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ni++) {
        const int co = co0 + wn * (BN / 2) + ni * 16 + (lane >> 4) * 4;
        if (co >= Cout) continue;
        f32x4 v = acc[ni][mi];
        if (bias) add4(v, bias + co);
        if (residual) {       // fp16(fp16(conv + bias) + residual), like the LDS epilogue and the split-K reduce kernels
          v[0] = (float)(_Float16)v[0]; v[1] = (float)(_Float16)v[1]; v[2] = (float)(_Float16)v[2]; v[3] = (float)(_Float16)v[3];
          add4(v, residual + m * Cout + co);
        }
        f16x4 o;
        o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
        *(f16x4*)(out + out_row(m) * Cout + co) = o;
      }
    }
  }
}


// -----------------------------------------------------------------------------------------------------------------------
// conv_big_kernel — the same implicit GEMM with a 256-pixel x BN-channel tile (BN = 256 or 128) owned by ONE 8-wave
// workgroup per CU instead of two (four) independent 128 x 128 workgroups.
//
// Why (tools/exp_conv_ablate.py, round 3): in the 128 x 128 kernel the LDS-DMA stream alone takes as long as the LDS-read +
// MFMA stream alone (each ~70 % of the full kernel; 23 TB/s of L2 -> LDS traffic at 512 -> 512 @ 128^2, two thirds of the
// aggregate L2 peak), so the two overlap imperfectly and neither can shrink.  A 256 x 256 tile moves HALF the operand
// bytes per MFMA (64 KB per K step for 64 MFMAs per wave instead of 2 x 32 KB for 2 x 32), and a wave's 128 x 64 output
// needs 24 fragment reads per 64 MFMAs instead of 16 per 32.  Structure otherwise as above: two LDS stages of a full K
// step (2 x 64 KB), the DMA of step t + 1 issued before the MFMAs of step t, ONE barrier per K step (64 MFMAs per wave
// between barriers), weights as the MFMA A operand, XOR chunk swizzle on the source address, XCD-aware tile order,
// coalesced epilogue through LDS (in two 128-row halves for BN = 256) with the optional per-channel statistics.
// Used where the 256-row tiles still fill the chip (launch_big below); no split-K, no GEGLU.
// -----------------------------------------------------------------------------------------------------------------------
#define CVB_BM 256
#define CVB_THREADS 512

// Tile shapes (BM pixels x BN channels, 8 waves as WM x WN):
//   256 x 256  WM 2 x WN 4   a wave owns 128 pixels x 64 channels (MI 8 x NI 4)          VAE 256 / 512-channel levels
//   256 x 128  WM 4 x WN 2   64 x 64                                                     (built, not dispatched: see big_tile_width)
//   128 x 256  WM 2 x WN 4   64 x 64 (MI 4 x NI 4)                                         (round 6, built and measured for the VAE's 512 channels
//       at 4 x 64^2 — 256 tiles, one 8-wave workgroup per CU sharing one stage instead of two 128 x 128 workgroups with their own —
//       78.8 -> 79.7 us per layer, VAE forward + backward 13.18 -> 13.40 ms: not dispatched, not instantiated)
//   384 x 160  WM 4 x WN 2   96 pixels x 80 channels (MI 6 x NI 5)                        round 6: Cout = 320 at 12 x 64^2 = 49 152 pixels is
//       exactly 128 x 2 = 256 tiles, one per CU, and moves (1/384 + 1/160) operand bytes per MAC against (1/128 + 1/160) for the
//       128 x 160 tile it replaces (768 workgroups = 1.5 rounds of two per CU).  160 weight rows are staged as 2.5 DMA rounds
//       (the third one by waves 0-3; its LDS rows 160..191 are padding), the pixel fragments go in groups of three.
template <int BM, int BN, int TAPS>
__global__ void __launch_bounds__(CVB_THREADS, 2)
conv_big_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const _Float16* __restrict__ bias,
                const _Float16* __restrict__ residual, _Float16* __restrict__ out, int N, int H, int W, int Cin, int Cout,
                int m_tiles, int n_tiles, int Hin, int Win, int geom, float* __restrict__ chan_stats, GnBwdArgs gnb,
                long long bs_x = 0, long long bs_w = 0, long long bs_o = 0) {
  // BATCHED GEMM (TAPS = 1, gip_linear_batched_f16 on the 256 x 256 tile, round 6): blockIdx.y = batch entry, as in conv3x3_kernel
  if constexpr (TAPS == 1) {
    x += (size_t)blockIdx.y * bs_x; w += (size_t)blockIdx.y * bs_w; out += (size_t)blockIdx.y * bs_o;
  }
  const int cstride = geom & 0xff, pad_t = (geom >> 8) & 0xff, pad_l = (geom >> 16) & 0xff;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int WN = BN == 256 ? 4 : 2, WM = 8 / WN;      // wave grid: WM (pixel direction) x WN (channel direction)
  constexpr int MI = BM / WM / 16;                         // 16-pixel MFMA tiles per wave: 8, 4 or 6
  constexpr int NI = BN / WN / 16;                         // 16-channel MFMA tiles per wave: 4 or 5
  constexpr int PG = MI % 4 == 0 ? 4 : 3;                  // pixel fragments per MFMA group
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && MI % PG == 0 && (MI / PG == 2 || MI / PG == 1), "tile shape");
  constexpr int A_ROUNDS = BM / 64, B_ROUNDS = (BN + 63) / 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = B_ROUNDS * 64 * 128, STAGE = A_BYTES + B_BYTES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WN, wc = wave % WN;

  const int total = m_tiles * n_tiles, id = (int)blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = id & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  const int mt = t / n_tiles, nt = t - mt * n_tiles;
  const unsigned M = (unsigned)N * H * W;
  const unsigned m0 = (unsigned)mt * BM;
  const int co0 = nt * BN;
  const int HW = H * W;

  // ---- per-thread DMA descriptors: a round is 64 rows (8 waves x 8 rows), a lane fills one 16-byte chunk ----
  const int sub_row = wave * 8 + (lane >> 3);
  const int pchunk = lane & 7;
  unsigned a_off[A_ROUNDS], a_mask[A_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; i++) {
    const int row = i * 64 + sub_row;
    const unsigned m = m0 + row;
    const int lchunk = pchunk ^ ((row >> 1) & 7);
    unsigned mask = 0, off = 0;
    if (m < M) {
      if constexpr (TAPS == 9) {
        const unsigned n = m / (unsigned)HW, rem = m - n * (unsigned)HW;
        const int y = (int)(rem / (unsigned)W), xx = (int)(rem - (unsigned)y * W);
        const int iy0 = y * cstride - pad_t, ix0 = xx * cstride - pad_l;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++)
            if ((unsigned)(iy0 + dy) < (unsigned)Hin && (unsigned)(ix0 + dx) < (unsigned)Win) mask |= 1u << (dy * 3 + dx);
        off = (((n * (unsigned)Hin + (unsigned)(y * cstride)) * (unsigned)Win + (unsigned)(xx * cstride)) * (unsigned)Cin + lchunk * 8) * 2u;
      } else {
        mask = 1u;
        off = (m * (unsigned)Cin + lchunk * 8) * 2u;
      }
    }
    a_off[i] = off;
    a_mask[i] = mask;
  }
  unsigned b_off[B_ROUNDS];
#pragma unroll
  for (int i = 0; i < B_ROUNDS; i++) {
    const int row = i * 64 + sub_row;
    const int lchunk = pchunk ^ ((row >> 1) & 7);
    const int wrow = co0 + row;
    b_off[i] = (row < BN && wrow < Cout) ? (unsigned)(wrow * TAPS * Cin + lchunk * 8) * 2u : CV_OOB;
  }
  const unsigned shift = TAPS == 9 ? (unsigned)(pad_t * Win + pad_l) * Cin * 2u : 0u;
  const unsigned Min = TAPS == 9 ? (unsigned)N * Hin * Win : M;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)x - shift), 0, (int)(Min * (unsigned)Cin * 2u + (unsigned)(2 * Win + 2) * Cin * 2u + shift), CV_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((unsigned)Cout * TAPS * Cin * 2u), CV_RSRC_FLAGS);

  const int cblocks = Cin / CV_BK;
  const int KT = TAPS * cblocks;

  auto stage = [&](int tap, int cb, int buf) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const unsigned tap_off = (unsigned)((TAPS == 9 ? (dy * Win + dx) * Cin : 0) + cb * CV_BK) * 2u;
    const unsigned wtap_off = (unsigned)(tap * Cin + cb * CV_BK) * 2u;
    unsigned char* sa = smem + buf * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; i++) dma16(xr, ((a_mask[i] >> tap) & 1u) ? a_off[i] : CV_OOB, tap_off, sa + i * 8192);
    unsigned char* sb = smem + buf * STAGE + A_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < B_ROUNDS; i++) {
      if ((i + 1) * 64 > BN && wave * 8 >= BN - i * 64) break;      // a partial last round: only the waves whose rows exist (wave-uniform)
      dma16(wrs, b_off[i], wtap_off, sb + i * 8192);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; a++)
#pragma unroll
    for (int b = 0; b < MI; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, swz = (lane >> 1) & 7, kq = lane >> 4;
  const int pix_base = (wr * (BM / WM) + frag_row) * 128;
  const int ch_base = A_BYTES + (wc * (BN / WN) + frag_row) * 128;

  // A K step = 2 k-halves x (MI / PG) groups of PG pixel fragments = PG x NI MFMAs per group.  The fragments of group g + 1 are
  // read while the MFMAs of group g run (two register sets, order pinned with sched_barrier: left alone the compiler
  // serialises "2 reads, wait, 8 MFMAs" with the LDS latency exposed every 8 MFMAs).
  constexpr int GROUPS = 2 * (MI / PG);
  auto ldw = [&](f16x8* wt, const unsigned char* sbuf, int ks) {
    const int pc = ((ks * 4 + kq) ^ swz) * 16;
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wt[ni] = *(const f16x8*)(sbuf + ch_base + ni * 2048 + pc);
  };
  auto ldp = [&](f16x8* px, const unsigned char* sbuf, int g) {
    const int ks = g / (MI / PG), mg = (g % (MI / PG)) * PG;
    const int pc = ((ks * 4 + kq) ^ swz) * 16;
#pragma unroll
    for (int mi = 0; mi < PG; mi++) px[mi] = *(const f16x8*)(sbuf + pix_base + (mg + mi) * 2048 + pc);
  };
  auto mma = [&](const f16x8* wt, const f16x8* px, int g) {
    const int mg = (g % (MI / PG)) * PG;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < PG; mi++)
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
        acc[ni][mg + mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[ni], px[mi], acc[ni][mg + mi], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  f16x8 w0[NI], w1[NI], pa[PG], pb[PG];
  // first half of a K step (k 0..31): reads everything it needs, and the second half's first fragments while its MFMAs run
  auto half1 = [&](const unsigned char* sbuf) {
    ldw(w0, sbuf, 0);
    ldp(pa, sbuf, 0);
    if constexpr (GROUPS == 4) {
      ldp(pb, sbuf, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(w0, pa, 0);
      __builtin_amdgcn_sched_barrier(0);
      ldw(w1, sbuf, 1);
      ldp(pa, sbuf, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(w0, pb, 1);
    } else {
      __builtin_amdgcn_sched_barrier(0);
      ldw(w1, sbuf, 1);
      ldp(pb, sbuf, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(w0, pa, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // second half (k 32..63): starts on fragments that are already in registers
  auto half2 = [&](const unsigned char* sbuf) {
    if constexpr (GROUPS == 4) {
      ldp(pb, sbuf, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(w1, pa, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(w1, pb, 3);
    } else {
      mma(w1, pb, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  int tap = 0, cb = 0;
  auto advance = [&]() {
    if (++cb == cblocks) { cb = 0; ++tap; }
  };
  stage(tap, cb, 0);
  advance();
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; kt++) {
      const int buf = kt & 1;
      if (kt + 1 < KT) { stage(tap, cb, buf ^ 1); advance(); }
      half1(smem + buf * STAGE);
      half2(smem + buf * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: half-rounded tile image [rows][BN] in LDS, 16-byte row accesses (see conv3x3_kernel) ----
  auto add4 = [](f32x4& v, const _Float16* p) {
    const f16x4 b = *(const f16x4*)p;
    v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
  };
  constexpr int ROWB = BN * 2 + 16;
  constexpr int PASSES = BN == 256 ? 2 : 1;               // image rows per pass: 128 (BN = 256) or the whole tile
  constexpr int PROWS = BM / PASSES;
  constexpr int CH = BN / 8, RPP = CVB_THREADS / CH;      // 32 chunks x 16 row lanes, 16 x 32, or 20 x 25 (12 threads idle)
  constexpr int SB = BM / 128;                            // 128-row statistics blocks per tile
  static_assert(PROWS * ROWB <= 2 * STAGE && RPP * BN * 8 <= 2 * STAGE, "epilogue image / partials must fit the stage buffers");
  const int chunk = tid % CH, r0 = tid / CH;
  const int co = co0 + chunk * 8;
  const bool mine = r0 < RPP && co < Cout;
  const _Float16* early_src = residual ? residual : ((chan_stats && gnb.x) ? gnb.x : nullptr);
  const bool res_early = early_src && ((geom >> 29) & 1);
  float s8[SB][8], q8[SB][8];                             // statistics of the tile's 128-row blocks
#pragma unroll
  for (int b = 0; b < SB; b++)
#pragma unroll
    for (int j = 0; j < 8; j++) { s8[b][j] = 0.f; q8[b][j] = 0.f; }
  GnBwdLane gl;
  if (chan_stats && gnb.x && mine) gnb_load(gnb, (int)(m0 / (unsigned)gnb.HW), co, Cout, gl);
#pragma unroll
  for (int pass = 0; pass < PASSES; pass++) {
    // the residual rows of this pass are requested before the image is staged (geom bit 29, see conv3x3_kernel)
    constexpr int RPT = (PROWS + RPP - 1) / RPP;
    constexpr int RPF = RPT / 2;                          // half of them: the register file is full (229 of 256 in the main loop)
    f16x8 rres[RPF];
    if (res_early && mine) {
#pragma unroll
      for (int k = 0; k < RPF; k++) {
        const int row = r0 + k * RPP;
        const unsigned m = m0 + pass * PROWS + row;
        rres[k] = (row < PROWS && m < M) ? *(const f16x8*)(early_src + (size_t)m * Cout + co) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    if (PASSES == 1 || wr == pass) {
#pragma unroll
      for (int ni = 0; ni < NI; ni++) {
        const int cl = wc * (BN / WN) + ni * 16 + (lane >> 4) * 4;
        f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias && co0 + cl < Cout) add4(b4, bias + co0 + cl);      // (not preloaded before the K loop: this kernel's register file is full)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
          const int p = (PASSES == 1 ? wr * (BM / WM) : 0) + mi * 16 + (lane & 15);
          const f32x4 v = acc[ni][mi];
          f16x4 o;
          o[0] = (_Float16)(v[0] + b4[0]); o[1] = (_Float16)(v[1] + b4[1]); o[2] = (_Float16)(v[2] + b4[2]); o[3] = (_Float16)(v[3] + b4[3]);
          *(f16x4*)(smem + p * ROWB + cl * 2) = o;
        }
      }
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int row = r0 + k * RPP;
        const unsigned m = m0 + pass * PROWS + row;
        if (row >= PROWS || m >= M) break;
        f16x8 v = *(const f16x8*)(smem + row * ROWB + chunk * 16);
        if (residual) {
          const f16x8 rr = (res_early && k < RPF) ? rres[k < RPF ? k : 0] : *(const f16x8*)(residual + (size_t)m * Cout + co);
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = (_Float16)((float)v[j] + (float)rr[j]);
        }
        *(f16x8*)(out + (size_t)m * Cout + co) = v;
        if (chan_stats) {
          const int blk = (pass * PROWS + row) / 128;
          if (gnb.x) {
            const f16x8 xv = (res_early && k < RPF) ? rres[k < RPF ? k : 0] : *(const f16x8*)(gnb.x + (size_t)m * Cout + co);
#pragma unroll
            for (int b = 0; b < SB; b++) if (b == blk) gnb_accumulate(gnb, gl, v, xv, s8[b], q8[b]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const float f = (float)v[j];
#pragma unroll
              for (int b = 0; b < SB; b++) if (b == blk) { s8[b][j] += f; q8[b][j] = fmaf(f, f, q8[b][j]); }
            }
          }
        }
      }
    }
    __syncthreads();                                      // the image is rewritten by the next pass / the partials below
  }
  if (chan_stats) {
    float* part = (float*)smem;                           // [RPP][BN][2]
#pragma unroll
    for (int b = 0; b < SB; b++) {
      if ((size_t)(mt * SB + b) * 128 >= M) break;        // uniform
      if (r0 < RPP) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          part[((r0 * BN) + chunk * 8 + j) * 2] = s8[b][j];
          part[((r0 * BN) + chunk * 8 + j) * 2 + 1] = q8[b][j];
        }
      }
      __syncthreads();
      if (tid < BN && co0 + tid < Cout) {
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int rr = 0; rr < RPP; rr++) { S += part[(rr * BN + tid) * 2]; Q += part[(rr * BN + tid) * 2 + 1]; }
        float* o = chan_stats + ((size_t)(mt * SB + b) * Cout + co0 + tid) * 2;
        o[0] = S; o[1] = Q;
      }
      __syncthreads();
    }
  }
}

// out[m][co] = sum_s slab[s][m][co] + bias[co] + residual[m][co], 4 channels per lane, fixed summation order
__global__ void __launch_bounds__(256)
conv_splitk_reduce_kernel(const float* __restrict__ partial, const _Float16* __restrict__ bias, const _Float16* __restrict__ residual,
                          _Float16* __restrict__ out, unsigned n4, int c4, int ksplit, size_t slab) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = *(const f32x4*)(partial + (size_t)i * 4);
  for (int s = 1; s < ksplit; s++) {
    const f32x4 p = *(const f32x4*)(partial + (size_t)s * slab + (size_t)i * 4);
    v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
  }
  if (bias) {
    const f16x4 b = *(const f16x4*)(bias + (i % (unsigned)c4) * 4);
    v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
  }
  if (residual) {
    // fp16(fp16(conv + bias) + residual): the LDS epilogue of the whole-K tiles rounds the convolution output to half before
    // it adds the residual (diffusers' own arithmetic), so the same layer must not round differently when it runs split-K
    const f16x4 r = *(const f16x4*)(residual + (size_t)i * 4);
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (float)(_Float16)v[j] + (float)r[j];
  }
  f16x4 o;
  o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
  *(f16x4*)(out + (size_t)i * 4) = o;
}

// The same reduction for a tensor whose consumer is a GroupNorm: it also leaves the per-(RB-row block, channel) sum and sum of
// squares of the half-rounded outputs in chan_stats [M / RB][Cout][2] (what the whole-K epilogue writes per 128-row block), so
// the split-K layers — the 16 x 16 and 8 x 8 levels of the U-Net / ControlNet — no longer need the GroupNorm's own statistics
// pass.  RB = 128, or 64 where a sample has only 64 pixels (a block must not straddle two samples).  A workgroup owns RB rows
// x 64 channels: 16 lanes x 4 channels across, 16 row-lanes down; fixed summation order.
template <int RB>
__global__ void __launch_bounds__(256)
conv_splitk_reduce_stats_kernel(const float* __restrict__ partial, const _Float16* __restrict__ bias, const _Float16* __restrict__ residual,
                                _Float16* __restrict__ out, float* __restrict__ chan_stats, unsigned M, int Cout, int ksplit, size_t slab) {
  __shared__ float part[16 * 64 * 2];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int co = blockIdx.y * 64 + cx * 4;
  const unsigned m0 = blockIdx.x * RB;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  if (co < Cout) {
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias) { const f16x4 b = *(const f16x4*)(bias + co); b4 = (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]}; }
#pragma unroll
    for (int k = 0; k < RB / 16; k++) {
      const unsigned m = m0 + ry + 16 * k;
      if (m >= M) break;
      const size_t i = (size_t)m * Cout + co;
      f32x4 v = *(const f32x4*)(partial + i);
      for (int sp = 1; sp < ksplit; sp++) {
        const f32x4 p = *(const f32x4*)(partial + (size_t)sp * slab + i);
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
      }
      v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
      if (residual) {       // two roundings like the whole-K LDS epilogue (see conv_splitk_reduce_kernel)
        const f16x4 r = *(const f16x4*)(residual + i);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = (float)(_Float16)v[j] + (float)r[j];
      }
      f16x4 o;
      o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
      *(f16x4*)(out + i) = o;
#pragma unroll
      for (int j = 0; j < 4; j++) { const float f = (float)o[j]; s4[j] += f; q4[j] = fmaf(f, f, q4[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    part[(ry * 64 + cx * 4 + j) * 2] = s4[j];
    part[(ry * 64 + cx * 4 + j) * 2 + 1] = q4[j];
  }
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < Cout) {
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) { S += part[(r * 64 + threadIdx.x) * 2]; Q += part[(r * 64 + threadIdx.x) * 2 + 1]; }
    float* o = chan_stats + ((size_t)blockIdx.x * Cout + blockIdx.y * 64 + threadIdx.x) * 2;
    o[0] = S; o[1] = Q;
  }
}

// Debug / A-B knobs (tools/exp_conv*.py set them through ctypes; -1 = the shape heuristic below decides)
extern "C" { int gip_dbg_conv_order = -1; int gip_dbg_conv_epilogue = -1; int gip_dbg_conv_ksplit = 0; int gip_dbg_conv_ablate = 0;
             int gip_dbg_conv_big = -1; int gip_dbg_linear_narrow = -1; int gip_dbg_linear_kg = -1; }
// same-box A/B of a whole training step (tools/ab_ahds.sh); read once.  (Rounds 3-5 also had GIP_CONV_EPILOGUE / _RES_EARLY / _BIG /
// _KSPLIT_R2: the per-lane 8-byte epilogue, residual rows requested late, no 256-row tile, the round-2 split-K factor — each measured
// slower in DESIGN §4c and retired in round 6; the gip_dbg_* knobs above still reach them from tools/exp_conv*.py.)
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

template <int BM, int BN, int TAPS>
static int launch_big(const void* x, const void* w, const void* bias, const void* residual, void* out, int N, int H, int W, int Cin,
                      int Cout, hipStream_t s, int Hin, int Win, int geom, float* chan_stats, const GnBwdArgs& gnb, int batch = 1,
                      long long bs_x = 0, long long bs_w = 0, long long bs_o = 0) {
  const long long M = (long long)N * H * W;
  const int m_tiles = (int)((M + BM - 1) / BM), n_tiles = (Cout + BN - 1) / BN;
  const size_t lds = 2 * ((size_t)BM + (size_t)((BN + 63) / 64) * 64) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_big_kernel<BM, BN, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 3;
    attr_set = true;
  }
  geom |= 1 << 29;                                  // residual rows requested before the epilogue staging
  hipLaunchKernelGGL((conv_big_kernel<BM, BN, TAPS>), dim3(m_tiles * n_tiles, batch), dim3(CVB_THREADS), lds, s, (const _Float16*)x,
                     (const _Float16*)w, (const _Float16*)bias, (const _Float16*)residual, (_Float16*)out, N, H, W, Cin, Cout,
                     m_tiles, n_tiles, Hin, Win, geom, chan_stats, gnb, bs_x, bs_w, bs_o);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// The 256-row tile is used where its tiles still fill the chip at ONE workgroup per CU (>= 7/8 of a round of 256) and
// Cout is a multiple of its channel width; everything else stays on the 128-row kernel (two workgroups per CU, split-K).
static int big_tile_width(long long M, int Cout) {
  int use = 1;
  if (gip_dbg_conv_big >= 0) use = gip_dbg_conv_big;
  if (!use) return 0;
  // 384 x 160 (round 6): Cout = 320 / 960-style widths whose 384-row tiles make whole rounds of one workgroup per CU (at least 7/8 of
  // the last one): the 12 x 64^2 layers of the U-Net / ControlNet (128 x 2 = 256 tiles)
  if (Cout % 160 == 0 && (Cout & 127)) {
    const long long tiles = ((M + 383) / 384) * (Cout / 160), rem = tiles % 256;
    return (tiles >= 224 && (rem == 0 || rem >= 224)) ? 160 : 0;
  }
  if (Cout & 127) return 0;
  const long long m_tiles = (M + CVB_BM - 1) / CVB_BM;
  if (!(Cout & 255) && m_tiles * (Cout / 256) >= 224) return 256;

  // a 256 x 128 tile (Cout = 128, 640) measured equal or slower than two 128 x 128 workgroups per CU: not dispatched
  return 0;
}

static inline bool workspace_is_forced_splitk() { return gip_dbg_conv_ksplit > 0; }      // tools/exp_conv5.py forces a split-K factor

template <int BN, int STAGES, int TAPS, bool GEGLU>
static int launch(const void* x, const void* w, const void* bias, const void* residual, void* out, int N, int H, int W,
                  int Cin, int Cout, hipStream_t s, void* workspace = nullptr, size_t workspace_bytes = 0, int Hin = 0, int Win = 0,
                  int geom = 1 | (1 << 8) | (1 << 16), float* chan_stats = nullptr, const GnBwdArgs* gnb_in = nullptr,
                  int tapsel = 0x1ff, int stats_rows = 128, int batch = 1, long long bs_x = 0, long long bs_w = 0, long long bs_o = 0,
                  bool gn_in = false) {
  // gn_in: *gnb_in describes the GroupNorm (+ SiLU) applied to this convolution's INPUT (halo-resident kernel only: anything else
  // returns 1 and the caller runs the separate apply pass)
  GnBwdArgs gnb = {};
  if (gnb_in) gnb = *gnb_in;
  if (Hin == 0) { Hin = H; Win = W; }
  const long long M = (long long)N * H * W;
  if (batch > 1 && (TAPS != 1 || GEGLU || workspace || chan_stats || gnb_in)) return 1;
  if constexpr (!GEGLU) {
    if (!(Cout & 7) && batch == 1 && !gn_in) {
      const int bw = big_tile_width(M, Cout);
      if (bw == 256 && !(gnb.x && gnb.HW % CVB_BM) && tapsel == 0x1ff && !gnb.ln_rows && !gnb.rows_out)
        return launch_big<256, 256, TAPS>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, Hin, Win, geom, chan_stats, gnb);
      {
        // a 384-row tile straddles samples (HW % 384 != 0): fine for the implicit GEMM and for the 128-row statistics blocks, not for
        // the per-sample constants of the GroupNorm-backward sums (gnb.x) — those layers keep the 128-row kernel.  As a GEMM
        // (TAPS = 1) only with a K loop long enough to amortise the one-workgroup-per-CU prologue / epilogue (K >= 1280: ff_out).
        // Same-box A/B of the whole step, four runs per setting (round 6): 33.55 -> 33.48 ms with the 3x3 layers, -> 33.35 with ff_out too
        if (bw == 160 && !gnb.x && tapsel == 0x1ff && !gnb.ln_rows && !gnb.rows_out && !workspace_is_forced_splitk() &&
            (TAPS == 9 || Cin >= 1280))
          return launch_big<384, 160, TAPS>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, Hin, Win, geom, chan_stats, gnb);
      }
    }
  }
  const int m_tiles = (int)((M + CV_BM - 1) / CV_BM), n_tiles = (Cout + (GEGLU ? BN / 2 : BN) - 1) / (GEGLU ? BN / 2 : BN);
  if constexpr (BN == 128 && TAPS == 9 && !GEGLU) {
    // halo-resident pixel tile (see conv3x3_kernel): Cin = 128, plain 3x3 / stride 1 / pad 1, whole-K tiles that fill the chip
    static const int env_halo = env_int("GIP_CONV_HALO", 1);
    if (env_halo && gip_dbg_conv_epilogue != 0 && gip_dbg_conv_ksplit <= 0 && Cin == 128 && (geom & 0xffffff) == (1 | (1 << 8) | (1 << 16)) &&
        tapsel == 0x1ff && !(H & 7) && !(W & 15) && Hin == H && Win == W && !(Cout & 7) && stats_rows == 128 &&
        (long long)m_tiles * n_tiles >= 256 && !(gnb.x && gnb.HW != H * W)) {
      constexpr size_t lds_h = 2 * (size_t)CVH_KC_BYTES + STAGES * (size_t)BN * 128;
      static_assert((size_t)CV_BM * (BN * 2 + 16) + (size_t)(CV_THREADS / (BN / 8)) * BN * 8 <= lds_h, "epilogue image must fit");
      static bool attr_h = false;
      if (!attr_h) {
        if (hipFuncSetAttribute((const void*)conv3x3_kernel<BN, STAGES, TAPS, GEGLU, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_h) != hipSuccess)
          return 3;
        attr_h = true;
      }
      const int geom_h = geom | (1 << 25) | ((gip_dbg_conv_ablate & 7) << 26) | (1 << 29) | ((gn_in ? 1 : 0) << 30);
      hipLaunchKernelGGL((conv3x3_kernel<BN, STAGES, TAPS, GEGLU, true>), dim3(m_tiles * n_tiles), dim3(CV_THREADS), lds_h, s,
                         (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (const _Float16*)residual, (_Float16*)out,
                         N, H, W, Cin, Cout, m_tiles, n_tiles, 1, (float*)nullptr, Hin, Win, geom_h, chan_stats, gnb, tapsel);
      return hipGetLastError() == hipSuccess ? 0 : 3;
    }
  }
  if (gn_in) return 1;
  const size_t lds = STAGES * (size_t)(CV_BM + BN) * 128;
  static_assert((size_t)CV_BM * (BN * 2 + 16) + (size_t)(CV_THREADS / (BN / 8)) * BN * 8 <= STAGES * (size_t)(CV_BM + BN) * 128,
                "epilogue tile image + statistics partials must fit in the stage buffers");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv3x3_kernel<BN, STAGES, TAPS, GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return 3;
    attr_set = true;
  }
  // split-K when the output tiles cannot fill the chip and a workspace was handed in: the largest factor that still
  // fits ONE round of 2 workgroups per CU (240 tiles: 2 -> 480 workgroups; 3 -> 720 = 1.4 rounds measured 16 % slower;
  // 60 tiles: 8; tools/exp_conv5.py)
  int ksplit = 1;
  const int tiles = m_tiles * n_tiles, KT = (TAPS == 9 ? __builtin_popcount(tapsel & 0x1ff) : TAPS) * (Cin / CV_BK);
  if (!GEGLU && workspace && tiles < 256 && !((tapsel >> 11) & 1)) {
    ksplit = 512 / tiles;
    if (ksplit > KT / 8) ksplit = KT / 8;
    if (ksplit > 16) ksplit = 16;
    while (ksplit > 1 && (size_t)ksplit * M * Cout * sizeof(float) > workspace_bytes) ksplit--;
    if (ksplit < 2) ksplit = 1;
  }
  if (gip_dbg_conv_ksplit > 0 && !GEGLU && workspace) {
    ksplit = gip_dbg_conv_ksplit;
    while (ksplit > 1 && ((size_t)ksplit * M * Cout * sizeof(float) > workspace_bytes || ksplit > KT)) ksplit--;
  }
  // tile order: n-major only where the pixel count is tiny against the weights (the 8x8 level: 6 pixel blocks, 30-60 MB
  // of weights: 55 -> 50 us); measured slower everywhere else, also at 16x16 (tools/exp_conv5.py)
  int nmajor = m_tiles <= 8 && n_tiles > 1 ? 1 : 0;
  if (gip_dbg_conv_order >= 0) nmajor = gip_dbg_conv_order;
  int lds_epi = 1;
  if (gip_dbg_conv_epilogue >= 0) lds_epi = gip_dbg_conv_epilogue;
  const bool stats_in_reduce = chan_stats && ksplit > 1 && !gnb.x;      // split-K: the reduce kernel makes the statistics
  if (chan_stats && !stats_in_reduce) {            // statistics come out of the LDS epilogue of whole-K tiles (128-row blocks)
    if (GEGLU || (Cout & 7) || stats_rows != 128) return 1;
    ksplit = 1;
    lds_epi = 1;
  }
  if (gnb.ln_rows || gnb.rows_out) {               // LayerNorm fold / row statistics: whole-K tiles with the LDS epilogue
    if (TAPS != 1 || ksplit != 1 || (!GEGLU && (Cout & 7)) || (gnb.rows_out && (GEGLU || chan_stats))) return 1;
    lds_epi = 1;
  }
  geom |= (nmajor << 24) | (lds_epi << 25) | ((gip_dbg_conv_ablate & 7) << 26) | (1 << 29);   // bits 26-28: timing ablations (WRONG results)
  const int classes = (tapsel >> 12) & 1 ? 4 : 1;
  if constexpr (TAPS == 1 && !GEGLU) {
    // two K groups per workgroup (KG = 2, see conv3x3_kernel) where the grid leaves at most one workgroup per CU and the K loop is long
    // enough to split: the GEMMs of the 8 x 8 level at batch 12, most GEMMs of a 1-view shard (batch 3)
    const bool kg2 = gip_dbg_linear_kg >= 0 ? gip_dbg_linear_kg != 0 : true;
    if (kg2 && ksplit == 1 && batch == 1 && tiles <= 256 && KT >= 8 && lds_epi) {
      constexpr size_t lds2 = 2 * STAGES * (size_t)(CV_BM + BN) * 128;
      static_assert((size_t)(BN / 32) * 4 * CV_THREADS * 16 <= lds2, "the handed-over accumulators must fit the stage buffers");
      static bool attr2 = false;
      if (!attr2) {
        if (hipFuncSetAttribute((const void*)conv3x3_kernel<BN, STAGES, TAPS, GEGLU, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds2) != hipSuccess)
          return 3;
        attr2 = true;
      }
      hipLaunchKernelGGL((conv3x3_kernel<BN, STAGES, TAPS, GEGLU, false, 2>), dim3(tiles), dim3(CV_THREADS * 2), lds2, s,
                         (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (const _Float16*)residual, (_Float16*)out,
                         N, H, W, Cin, Cout, m_tiles, n_tiles, 1, (float*)nullptr, Hin, Win, geom, chan_stats, gnb, tapsel, bs_x, bs_w, bs_o);
      return hipGetLastError() == hipSuccess ? 0 : 3;
    }
  }
  hipLaunchKernelGGL((conv3x3_kernel<BN, STAGES, TAPS, GEGLU>), dim3(tiles * ksplit * classes, batch), dim3(CV_THREADS), lds, s,
                     (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (const _Float16*)residual, (_Float16*)out,
                     N, H, W, Cin, Cout, m_tiles, n_tiles, ksplit, (float*)workspace, Hin, Win, geom,
                     stats_in_reduce ? nullptr : chan_stats, gnb, tapsel, bs_x, bs_w, bs_o);
  if (stats_in_reduce) {
    if ((Cout & 3) || M % stats_rows) return 1;
    const dim3 grid((unsigned)(M / stats_rows), (unsigned)((Cout + 63) / 64));
    if (stats_rows == 128)
      hipLaunchKernelGGL((conv_splitk_reduce_stats_kernel<128>), grid, dim3(256), 0, s, (const float*)workspace, (const _Float16*)bias,
                         (const _Float16*)residual, (_Float16*)out, chan_stats, (unsigned)M, Cout, ksplit, (size_t)M * Cout);
    else if (stats_rows == 64)
      hipLaunchKernelGGL((conv_splitk_reduce_stats_kernel<64>), grid, dim3(256), 0, s, (const float*)workspace, (const _Float16*)bias,
                         (const _Float16*)residual, (_Float16*)out, chan_stats, (unsigned)M, Cout, ksplit, (size_t)M * Cout);
    else
      return 1;
  } else if (ksplit > 1) {
    const unsigned n4 = (unsigned)(M * Cout / 4);
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, (const float*)workspace,
                       (const _Float16*)bias, (const _Float16*)residual, (_Float16*)out, n4, Cout / 4, ksplit, (size_t)M * Cout);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// GEMMs whose 128 x 128 (160) tiles leave most CUs with at most ONE workgroup run on 128 x 64 tiles: twice the workgroups, and a
// lone workgroup's K step (8 LDS-DMA instructions per wave + 32 MFMAs, in series inside the wave: tools/experiments/
// conv3x3_four_stage.diff.txt) becomes 6 + 16.  GIP_LINEAR_NARROW = the largest 128-wide grid that takes the narrow tile (0: off).
static bool narrow_tiles(long long M, int Nout, int bn) {
  static const int env_narrow = env_int("GIP_LINEAR_NARROW", 256);
  const int lim = gip_dbg_linear_narrow >= 0 ? gip_dbg_linear_narrow : env_narrow;
  return lim > 0 && !(Nout & 63) && ((M + CV_BM - 1) / CV_BM) * ((Nout + bn - 1) / bn) <= lim;
}

// channel-tile width of the plain (no GEGLU) linear for an [M, Nout] output: 160 where Nout is a multiple of 160 but not of 128
// (320, 960), 128 otherwise, 64 on small grids (narrow_tiles)
static int linear_width(long long M, int Nout) {
  const int wide = (Nout % 160 == 0 && Nout % 128 != 0) ? 160 : 128;
  return narrow_tiles(M, Nout, wide) ? 64 : wide;
}
template <class F>
static int with_linear_width(long long M, int Nout, F f) {
  switch (linear_width(M, Nout)) {
    case 64: return f(std::integral_constant<int, 64>{});
    case 160: return f(std::integral_constant<int, 160>{});
    default: return f(std::integral_constant<int, 128>{});
  }
}

static bool fits32(long long M, int Cin, int Cout, int wrows, int taps) {
  return M * (long long)(Cin > Cout ? Cin : Cout) * 2 < (1ll << 31) && (long long)wrows * taps * Cin * 2 < (1ll << 31);
}

extern "C" int gip_conv3x3_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                    int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  if (!x || !w || !out || N < 1 || H < 1 || W < 1 || Cin < CV_BK || Cin % CV_BK || Cout < 4 || (Cout & 3)) return 1;
  if (!fits32((long long)N * H * W, Cin, Cout, Cout, 9)) return 1;   // 32-bit byte offsets
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  return wide ? launch<160, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, workspace, workspace_bytes)
              : launch<128, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, workspace, workspace_bytes);
}

extern "C" int gip_conv3x3_stats_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                          int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* chan_stats,
                                          void* stream) {
  if (!x || !w || !out || !chan_stats || N < 1 || H < 1 || W < 1 || Cin < CV_BK || Cin % CV_BK || Cout < 8 || (Cout & 7)) return 1;
  if (!fits32((long long)N * H * W, Cin, Cout, Cout, 9)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  const int geom = 1 | (1 << 8) | (1 << 16);
  return wide ? launch<160, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, nullptr, 0, 0, 0, geom, chan_stats)
              : launch<128, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, nullptr, 0, 0, 0, geom, chan_stats);
}

extern "C" int gip_conv3x3_stats_ws_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out,
                                             int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* chan_stats,
                                             int32_t stats_rows, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !w || !out || !chan_stats || N < 1 || H < 1 || W < 1 || Cin < CV_BK || Cin % CV_BK || Cout < 8 || (Cout & 7)) return 1;
  if ((stats_rows != 128 && stats_rows != 64) || ((long long)H * W) % stats_rows) return 1;
  if (!fits32((long long)N * H * W, Cin, Cout, Cout, 9)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  const int geom = 1 | (1 << 8) | (1 << 16);
  return wide ? launch<160, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, workspace, workspace_bytes, 0, 0, geom, chan_stats,
                                         nullptr, 0x1ff, stats_rows)
              : launch<128, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, s, workspace, workspace_bytes, 0, 0, geom, chan_stats,
                                         nullptr, 0x1ff, stats_rows);
}

extern "C" int gip_conv3x3_gnbwd_nhwc_f16(const void* dy_in, const void* w, void* out, int32_t N, int32_t H, int32_t W, int32_t Cin,
                                          int32_t Cout, const void* gn_x, const void* gamma, const void* beta, const float* mean,
                                          const float* rstd, int32_t G, int32_t apply_silu, const void* addend,
                                          int32_t addend_stride, float* chan_sums, void* stream) {
  if (!dy_in || !w || !out || !gn_x || !gamma || !beta || !mean || !rstd || !chan_sums || N < 1 || H < 1 || W < 1 || Cin < CV_BK ||
      Cin % CV_BK || Cout < 8 || (Cout & 7) || G < 1 || Cout % G || ((long long)H * W) % CV_BM)
    return 1;
  if (!fits32((long long)N * H * W, Cin, Cout, Cout, 9)) return 1;
  GnBwdArgs g = {};
  g.x = (const _Float16*)gn_x; g.gamma = (const _Float16*)gamma; g.beta = (const _Float16*)beta; g.addend = (const _Float16*)addend;
  g.mean = mean; g.rstd = rstd; g.G = G; g.silu = apply_silu; g.addend_stride = addend_stride; g.HW = H * W;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  const int geom = 1 | (1 << 8) | (1 << 16);
  return wide ? launch<160, 2, 9, false>(dy_in, w, nullptr, nullptr, out, N, H, W, Cin, Cout, s, nullptr, 0, 0, 0, geom, chan_sums, &g)
              : launch<128, 2, 9, false>(dy_in, w, nullptr, nullptr, out, N, H, W, Cin, Cout, s, nullptr, 0, 0, 0, geom, chan_sums, &g);
}

extern "C" int gip_linear_stats_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                                    int32_t K, int32_t Nout, float* chan_stats, void* stream) {
  if (!x || !w || !out || !chan_stats || M < 1 || M >= (1ll << 31) || K < CV_BK || K % CV_BK || Nout < 8 || (Nout & 7)) return 1;
  if (!fits32(M, K, Nout, Nout, 1)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int geom = 1 | (1 << 8) | (1 << 16);
  return with_linear_width(M, Nout, [&](auto bn) {
    return launch<decltype(bn)::value, 2, 1, false>(x, w, bias, residual, out, 1, 1, (int)M, K, Nout, s, nullptr, 0, 0, 0, geom, chan_stats);
  });
}

// Number of per-row partial sums a [M, Nout] output of the linear kernel carries in rows_out (= its channel tiles).
extern "C" int32_t gip_linear_row_parts(int64_t M, int32_t Nout) {
  const int bn = linear_width(M, Nout);
  return (Nout + bn - 1) / bn;
}

// gip_linear_f16 (no GEGLU) that also leaves, per output row, the (sum, sum of squares) of the final half-rounded output over
// each channel tile: rows_out [M][gip_linear_row_parts(M, Nout)][2] float32 — what the LayerNorm of that row needs.
extern "C" int gip_linear_rows_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                                   int32_t K, int32_t Nout, float* rows_out, void* stream) {
  if (!x || !w || !out || !rows_out || M < 1 || M >= (1ll << 31) || K < CV_BK || K % CV_BK || Nout < 8 || (Nout & 7)) return 1;
  if (!fits32(M, K, Nout, Nout, 1)) return 1;
  GnBwdArgs g = {};
  g.rows_out = rows_out;
  hipStream_t s = (hipStream_t)stream;
  const int geom = 1 | (1 << 8) | (1 << 16);
  return with_linear_width(M, Nout, [&](auto bn) {
    return launch<decltype(bn)::value, 2, 1, false>(x, w, bias, residual, out, 1, 1, (int)M, K, Nout, s, nullptr, 0, 0, 0, geom, nullptr, &g);
  });
}

// LayerNorm(x) W^T + b (optionally GEGLU of it) WITHOUT a LayerNorm pass: x is read raw, `wg` = W * gamma (half), s_n = sum_k wg[n][k],
// t_n = sum_k W[n][k] beta_k + b_n (float32; [Nout], GEGLU: [2 Nout] = [value | gate] like the weight), and the per-row statistics
// come from ln_rows [M][ln_parts][2], the partial sums its producer left (gip_linear_rows_f16): see GnBwdArgs.
extern "C" int gip_linear_ln_f16(const void* x, const void* wg, const float* s_vec, const float* t_vec, void* out, int64_t M, int32_t K,
                                 int32_t Nout, int32_t geglu, const float* ln_rows, int32_t ln_parts, float eps, void* stream) {
  if (!x || !wg || !s_vec || !t_vec || !out || !ln_rows || ln_parts < 1 || M < 1 || M >= (1ll << 31) || K < CV_BK || K % CV_BK || Nout < 8 ||
      (Nout & 7))
    return 1;
  if (geglu && (Nout % 64)) return 1;
  if (!fits32(M, K, geglu ? 2 * Nout : Nout, geglu ? 2 * Nout : Nout, 1)) return 1;
  GnBwdArgs g = {};
  g.ln_rows = ln_rows; g.ln_s = s_vec; g.ln_t = t_vec; g.ln_parts = ln_parts; g.ln_inv_c = 1.0f / (float)K; g.ln_eps = eps;
  hipStream_t s = (hipStream_t)stream;
  const int geom = 1 | (1 << 8) | (1 << 16);
  if (geglu) return launch<128, 2, 1, true>(x, wg, nullptr, nullptr, out, 1, 1, (int)M, K, Nout, s, nullptr, 0, 0, 0, geom, nullptr, &g);
  return with_linear_width(M, Nout, [&](auto bn) {
    return launch<decltype(bn)::value, 2, 1, false>(x, wg, nullptr, nullptr, out, 1, 1, (int)M, K, Nout, s, nullptr, 0, 0, 0, geom, nullptr, &g);
  });
}

extern "C" int gip_conv3x3s2_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t Hin,
                                      int32_t Win, int32_t Cin, int32_t Cout, int32_t pad_top, int32_t pad_left,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !w || !out || N < 1 || Hin < 2 || Win < 2 || (Hin & 1) || (Win & 1) || Cin < CV_BK || Cin % CV_BK || Cout < 4 ||
      (Cout & 3) || pad_top < 0 || pad_top > 1 || pad_left < 0 || pad_left > 1)
    return 1;
  if (!fits32((long long)N * Hin * Win, Cin, Cout, Cout, 9)) return 1;
  const int H = Hin / 2, W = Win / 2, geom = 2 | (pad_top << 8) | (pad_left << 16);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  return wide ? launch<160, 2, 9, false>(x, w, bias, nullptr, out, N, H, W, Cin, Cout, s, workspace, workspace_bytes, Hin, Win, geom)
              : launch<128, 2, 9, false>(x, w, bias, nullptr, out, N, H, W, Cin, Cout, s, workspace, workspace_bytes, Hin, Win, geom);
}

// The same stride-2 convolution whose epilogue also leaves the NEXT GroupNorm's statistics (per 128 output pixels and channel,
// like gip_conv3x3_stats_nhwc_f16): the ResnetBlock2D behind every Downsample2D then reads its input once.  Whole-K tiles only.
extern "C" int gip_conv3x3s2_stats_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t Hin,
                                            int32_t Win, int32_t Cin, int32_t Cout, int32_t pad_top, int32_t pad_left,
                                            float* chan_stats, void* stream) {
  if (!x || !w || !out || !chan_stats || N < 1 || Hin < 2 || Win < 2 || (Hin & 1) || (Win & 1) || Cin < CV_BK || Cin % CV_BK ||
      Cout < 8 || (Cout & 7) || pad_top < 0 || pad_top > 1 || pad_left < 0 || pad_left > 1)
    return 1;
  if (!fits32((long long)N * Hin * Win, Cin, Cout, Cout, 9)) return 1;
  const int H = Hin / 2, W = Win / 2, geom = 2 | (pad_top << 8) | (pad_left << 16);
  if (((long long)H * W) % 128) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  return wide ? launch<160, 2, 9, false>(x, w, bias, nullptr, out, N, H, W, Cin, Cout, s, nullptr, 0, Hin, Win, geom, chan_stats)
              : launch<128, 2, 9, false>(x, w, bias, nullptr, out, N, H, W, Cin, Cout, s, nullptr, 0, Hin, Win, geom, chan_stats);
}

// Data gradient of the 3x3 / stride 2 convolution y[oy][ox] = sum x[2 oy + ky][2 ox + kx] w[ky][kx] (input zero beyond its
// last row / column: the VAE's F.pad(x, (0, 1, 0, 1)) + padding = 0 form).  dx[2 a + pi][2 b + pj] only receives the taps
// with ky = pi, kx = pj (mod 2): four independent small convolutions over dy's own grid — 4, 2, 2 and 1 taps — each a
// launch of the implicit GEMM with a tap subset and the parity scatter in its epilogue.  Exactly the minimal FLOPs (the
// zero-dilated form costs 4x) and no dilated copy of dy.  wt4 [4][Cout][3][3][Cin]: class c = 2 pi + pj holds, at tap (dy,
// dx) (input offset dy - 1, dx - 1), w[co][ci][ky][kx] transposed, with ky = 2 for dy = 0 and ky = pi for dy = 1 (same in x).
extern "C" int gip_conv3x3s2_dgrad_nhwc_f16(const void* dy, const void* wt4, void* dx, int32_t N, int32_t Ho, int32_t Wo,
                                            int32_t Cin, int32_t Cout, void* stream) {
  if (!dy || !wt4 || !dx || N < 1 || Ho < 1 || Wo < 1 || Cin < CV_BK || Cin % CV_BK || Cout < 8 || (Cout & 7)) return 1;
  if (!fits32((long long)N * Ho * Wo * 4, Cin, Cout, Cout, 9)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  const int geom = 1 | (1 << 8) | (1 << 16);
  const int tapsel = 0x1ff | (1 << 11) | (1 << 12);          // four classes in one launch, the stride-2 data-gradient tap sets
  return wide ? launch<160, 2, 9, false>(dy, wt4, nullptr, nullptr, dx, N, Ho, Wo, Cin, Cout, s, nullptr, 0, 0, 0, geom, nullptr, nullptr, tapsel)
              : launch<128, 2, 9, false>(dy, wt4, nullptr, nullptr, dx, N, Ho, Wo, Cin, Cout, s, nullptr, 0, 0, 0, geom, nullptr, nullptr, tapsel);
}

// nearest-neighbour 2x upsampling followed by the 3x3 / pad 1 convolution (diffusers Upsample2D) WITHOUT the upsampled tensor:
// out[2 a + pi][2 b + pj] reads x rows {a - 1, a} (pi = 0) or {a, a + 1} (pi = 1), because the three taps of the upsampled
// image fall on two source pixels — so each parity class is a 2 x 2-tap convolution over x's own grid with the weights of
// the coinciding taps summed (by the host: wt4 [4][Cout][3][3][Cin], class c = 2 pi + pj, the summed weight stored at the tap
// whose input offset it applies to).  4 tap-GEMMs per output pixel instead of 9, and x is read at its own size.
extern "C" int gip_upsample2x_conv3x3_nhwc_f16(const void* x, const void* wt4, const void* bias, void* out, int32_t N, int32_t Hin,
                                               int32_t Win, int32_t Cin, int32_t Cout, void* stream) {
  if (!x || !wt4 || !out || N < 1 || Hin < 1 || Win < 1 || Cin < CV_BK || Cin % CV_BK || Cout < 8 || (Cout & 7)) return 1;
  if (!fits32((long long)N * Hin * Win * 4, Cin, Cout, Cout, 9)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Cout % 160 == 0 && Cout % 128 != 0;
  const int geom = 1 | (1 << 8) | (1 << 16);
  const int tapsel = 0x1ff | (1 << 11) | (1 << 12) | (1 << 13);
  return wide ? launch<160, 2, 9, false>(x, wt4, bias, nullptr, out, N, Hin, Win, Cin, Cout, s, nullptr, 0, 0, 0, geom, nullptr, nullptr, tapsel)
              : launch<128, 2, 9, false>(x, wt4, bias, nullptr, out, N, Hin, Win, Cin, Cout, s, nullptr, 0, 0, 0, geom, nullptr, nullptr, tapsel);
}

extern "C" int gip_conv3x3_gnin_nhwc_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int32_t N,
                                        int32_t H, int32_t W, int32_t Cin, int32_t Cout, const void* gamma, const void* beta,
                                        const float* mean, const float* rstd, int32_t G, int32_t apply_silu, const void* addend,
                                        int32_t addend_stride, float* chan_stats, void* stream) {
  if (!x || !w || !out || !gamma || !beta || !mean || !rstd || N < 1 || Cin != 128 || Cout < 8 || (Cout & 7) || G < 1 || 128 % G) return 1;
  if (!fits32((long long)N * H * W, Cin, Cout, Cout, 9)) return 1;
  GnBwdArgs gnb = {};
  gnb.gamma = (const _Float16*)gamma; gnb.beta = (const _Float16*)beta; gnb.addend = (const _Float16*)addend;
  gnb.mean = mean; gnb.rstd = rstd; gnb.G = G; gnb.silu = apply_silu; gnb.addend_stride = addend_stride; gnb.HW = H * W;
  return launch<128, 2, 9, false>(x, w, bias, residual, out, N, H, W, Cin, Cout, (hipStream_t)stream, nullptr, 0, 0, 0,
                                  1 | (1 << 8) | (1 << 16), chan_stats, &gnb, 0x1ff, 128, 1, 0, 0, 0, true);
}

extern "C" int gip_linear_batched_f16(const void* x, const void* w, void* out, int32_t B, int64_t M, int32_t K, int32_t Nout,
                                      int64_t bs_x, int64_t bs_w, int64_t bs_o, void* stream) {
  if (!x || !w || !out || B < 1 || B > 65535 || M < 1 || M >= (1ll << 31) || K < CV_BK || K % CV_BK || Nout < 4 || (Nout & 3)) return 1;
  if (!fits32(M, K, Nout, Nout, 1) || bs_x < 0 || bs_w < 0 || bs_o < 0 || (bs_x & 7) || (bs_w & 7) || (bs_o & 3)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = Nout % 160 == 0 && Nout % 128 != 0;
  const int geom = 1 | (1 << 8) | (1 << 16);
  // 256 x 256 tiles (one 8-wave workgroup per CU, half the operand bytes per MAC) where the batch of products fills the chip with
  // them: the sixteen [768, K] x [K, 1280] products of a Winograd convolution at the 16 x 16 level are 3 x 5 x 16 = 240 tiles
  if (!(Nout & 255) && gip_dbg_conv_big != 0) {
    const long long tiles = ((M + 255) / 256) * (Nout / 256) * B;
    if (tiles >= 224 && (M % 256 == 0 || M >= 2048)) {
      GnBwdArgs none = {};
      return launch_big<256, 256, 1>(x, w, nullptr, nullptr, out, 1, 1, (int)M, K, Nout, s, 1, (int)M, geom, nullptr, none, B, bs_x, bs_w, bs_o);
    }
  }
  return wide ? launch<160, 2, 1, false>(x, w, nullptr, nullptr, out, 1, 1, (int)M, K, Nout, s, nullptr, 0, 0, 0, geom, nullptr, nullptr, 0x1ff,
                                         128, B, bs_x, bs_w, bs_o)
              : launch<128, 2, 1, false>(x, w, nullptr, nullptr, out, 1, 1, (int)M, K, Nout, s, nullptr, 0, 0, 0, geom, nullptr, nullptr, 0x1ff,
                                         128, B, bs_x, bs_w, bs_o);
}

extern "C" int gip_linear_f16(const void* x, const void* w, const void* bias, const void* residual, void* out, int64_t M,
                              int32_t K, int32_t Nout, int32_t geglu, void* stream) {
  if (!x || !w || !out || M < 1 || M >= (1ll << 31) || K < CV_BK || K % CV_BK || Nout < 4 || (Nout & 3)) return 1;
  if (geglu && (residual || Nout % 64)) return 1;
  if (!fits32(M, K, Nout, geglu ? 2 * Nout : Nout, 1)) return 1;
  hipStream_t s = (hipStream_t)stream;
  if (geglu) return launch<128, 2, 1, true>(x, w, bias, nullptr, out, 1, 1, (int)M, K, Nout, s);
  return with_linear_width(M, Nout, [&](auto bn) {
    return launch<decltype(bn)::value, 2, 1, false>(x, w, bias, residual, out, 1, 1, (int)M, K, Nout, s);
  });
}
