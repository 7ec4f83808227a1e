// conv_small.hip — the 3x3 convolutions that the 128-channel implicit GEMM (conv3x3.hip) does not fit, because one side of
// them is narrow.  See include/gip_nn.h (gip_conv3x3_c3_fwd_nhwc_f16 / _dgrad_, gip_conv3x3_fewch_nhwc_f16).
//
//   conv_c3_fwd_kernel<CO>   3 input channels -> 16 / 128 (VAE encoder conv_in, ControlNet stem conv_in): 16 x 16 pixel tile,
//             the 18 x 18 x 3 input patch in LDS, K = 27 (padded to 32) in ONE v_mfma_f32_16x16x32_f16 per 16 pixels x 16
//             channels, weights held in registers as the A operand, bias (+ SiLU) in the epilogue, 16-byte coalesced
//             stores through an LDS transpose.  Bound by one pass over the output tensor (268 MB at 4 x 512^2 x 128).
//   conv_c3_dgrad_kernel     128 -> 3 (the gradient that leaves the VAE towards the rasterizer): 8 x 16 pixel tile, its
//             10 x 18 pixel halo of dy (46 KB) brought in ONCE by LDS-DMA (16-byte chunks XOR-swizzled on the source
//             address so that the 16-pixel fragment reads are conflict-free) and reused by all nine taps; the rearranged
//             weight [3][9][128] sits in LDS; 36 K steps of 32 per 16 pixels; 6-byte stores.
//   conv_fewch_kernel<CIN, COUT, STRIDE>   16 / 32 / 96-channel layers of the ControlNet's conditioning stem (stride 1 and
//             2, bias + SiLU in the epilogue), the 8-channel inputs (conv_in of the U-Net / ControlNet: 4 latent channels padded
//             to 8 -> 320; the VAE's moment gradient 8 -> 512) and the two narrow OUTPUT convolutions 320 -> 4, 512 -> 8: input halo once
//             in LDS, weights straight from L2 one K step ahead, the weight as the MFMA A operand (documented at the kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gip_nn.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CS_OOB 0xFFFF0000u
#define CS_RSRC_FLAGS 0x00020000

// ---------------------------------------------------------------------------------------------------------------------
// forward: x [N, H, W, 3] -> out [N, H, W, 128] (+ bias)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ _Float16 silu_h(_Float16 h) {      // torch's half SiLU: fp32 on the half-rounded value, one more rounding
  const float v = (float)h;
  return (_Float16)(v * __builtin_amdgcn_rcpf(1.f + __expf(-v)));      // v_rcp_f32 instead of the IEEE division sequence
}

// chan_stats (optional): per 16 x 8 half tile (128 pixels: waves 0-1 / 2-3 of the workgroup) and channel the (sum, sum of squares)
// of the half-rounded outputs, [N * H * W / 128][CO][2] float32 in the convention of conv3x3.hip's statistics epilogue (a
// block = 128 pixels of ONE sample; consumers only add a sample's blocks): the GroupNorm behind the VAE's conv_in then needs
// no statistics pass over the 268 MB tensor.
template <int CO, bool ACT>
__global__ void __launch_bounds__(256)
conv_c3_fwd_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w /* [CO][3][3][3] */, const _Float16* __restrict__ bias,
                   _Float16* __restrict__ out, int H, int W, float* __restrict__ chan_stats = nullptr) {
  constexpr int NI = CO / 16;
  constexpr int IN_W = 18 * 3;                               // halves per patch row
  constexpr int ROWB = CO * 2 + 16;                          // padded output staging row
  __shared__ _Float16 s_in[18 * IN_W + 8];
  __shared__ __attribute__((aligned(16))) unsigned char s_out[4][16 * ROWB];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_x = W / 16;
  const int n = blockIdx.y, tile = blockIdx.x, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
  // ---- the input patch (zeros outside the image)
  for (int i = tid; i < 18 * IN_W; i += 256) {
    const int hy = i / IN_W, r = i - hy * IN_W, hx = r / 3, c = r - hx * 3;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    s_in[i] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? x[(((size_t)n * H + iy) * W + ix) * 3 + c] : (_Float16)0.f;
  }
  // ---- weights as the MFMA A operand: lane (row = co % 16, kq) holds k = 8 kq .. 8 kq + 7 of w[co][k], k = (ky, kx, ci) < 27
  const int frow = lane & 15, kq = lane >> 4;
  f16x8 wa[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 8 * kq + j;
      wa[ni][j] = k < 27 ? w[(ni * 16 + frow) * 27 + k] : (_Float16)0.f;
    }
  }
  // per-lane patch offsets of its eight k (relative to the patch's top-left element of the pixel)
  int koff[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = 8 * kq + j;
    koff[j] = k < 27 ? (k / 9) * IN_W + (k % 9) : -1;
  }
  f32x4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const f16x4 b = bias ? *(const f16x4*)(bias + ni * 16 + kq * 4) : (f16x4){0, 0, 0, 0};
    bv[ni] = (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
  __syncthreads();
  unsigned char* so = s_out[wave];
  f32x4 st_s[NI], st_q[NI];                 // this lane's pixel column x its 4 channels per tile, over the wave's 4 rows
#pragma unroll
  for (int ni = 0; ni < NI; ni++) { st_s[ni] = (f32x4){0.f, 0.f, 0.f, 0.f}; st_q[ni] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const int ty = wave * 4 + i;
    f16x8 pb;
    const int base = ty * IN_W + frow * 3;
#pragma unroll
    for (int j = 0; j < 8; j++) pb[j] = koff[j] >= 0 ? s_in[base + koff[j]] : (_Float16)0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], pb, bv[ni], 0, 0, 0);
      f16x4 o;
      o[0] = (_Float16)acc[0]; o[1] = (_Float16)acc[1]; o[2] = (_Float16)acc[2]; o[3] = (_Float16)acc[3];
      if (ACT) { o[0] = silu_h(o[0]); o[1] = silu_h(o[1]); o[2] = silu_h(o[2]); o[3] = silu_h(o[3]); }
      *(f16x4*)(so + frow * ROWB + (ni * 16 + kq * 4) * 2) = o;
      if (chan_stats) {
#pragma unroll
        for (int j = 0; j < 4; j++) { const float v = (float)o[j]; st_s[ni][j] += v; st_q[ni][j] += v * v; }
      }
    }
    // the wave's own 16 pixels x CO channels: 16-byte chunks, CO / 8 lanes per pixel row (only this wave touches `so`)
    _Float16* orow = out + (((size_t)n * H + y0 + ty) * W + x0) * CO;
    constexpr int CPP = CO / 8;
#pragma unroll
    for (int q = lane; q < 16 * CPP; q += 64) {
      const int px = q / CPP, ch = q % CPP;
      *(uint4*)(orow + (size_t)px * CO + ch * 8) = *(const uint4*)(so + px * ROWB + ch * 16);
    }
  }
  if (chan_stats) {
    // the 16 pixel columns of a channel quadruple sit on the 16 lanes of one DPP row (lane = 16 kq + column): row sums by four
    // shifts, then the two waves of a half tile through LDS (fixed order: wave 2 h, then 2 h + 1)
    __shared__ float s_st[4][CO][2];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float a = st_s[ni][j], q = st_q[ni][j];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); q += __shfl_xor(q, d, 64); }
        if (frow == 0) { s_st[wave][ni * 16 + kq * 4 + j][0] = a; s_st[wave][ni * 16 + kq * 4 + j][1] = q; }
      }
    __syncthreads();
    const size_t blk0 = ((size_t)n * gridDim.x + tile) * 2;
    for (int e = tid; e < 2 * CO; e += 256) {
      const int hb = e / CO, c = e - hb * CO;
      float* dst = chan_stats + ((blk0 + hb) * CO + c) * 2;
      dst[0] = s_st[2 * hb][c][0] + s_st[2 * hb + 1][c][0];
      dst[1] = s_st[2 * hb][c][1] + s_st[2 * hb + 1][c][1];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Few-channel 3x3 / pad 1 convolutions, stride 1 or 2, bias (+ SiLU) in the epilogue: the ControlNet's conditioning stem
// (controlnet_cond_embedding: 16 -> 16, 16 -> 32 /2, 32 -> 32, 32 -> 96 /2, 96 -> 96, 96 -> 256 /2).  Too narrow for the
// 128-wide implicit GEMM; on the library each layer was a MIOpen kernel + a bias kernel + a layout copy + a SiLU kernel.
//   * a workgroup owns TR rows x 16 columns of output pixels; its input halo ((TR - 1) S + 3) x (15 S + 3) pixels x CIN
//     channels is read ONCE into LDS (16-byte chunks, pixel stride CIN * 2 + 16 bytes: the 16-pixel fragment reads of a
//     stride-1 layer hit 16 distinct bank groups) and reused by all nine taps and all output channels;
//   * GEMM view per 16-pixel output row: [COUT x 9 CIN] x [9 CIN x 16]; K runs in steps of 32 = (tap, 8-channel chunk) x 4
//     lane groups, zero weights past 9 CIN (CIN = 16: 4.5 steps); the WEIGHT is the MFMA A operand (rows = output
//     channels) so that a lane ends up with 4 consecutive channels of one pixel = 8-byte stores;
//   * weights come straight from global memory (L2-resident, <= 442 KB), one step ahead of the MFMAs that use them;
//     waves split the output channels WN ways and the tile rows 4 / WN ways.
// ---------------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int STRIDE, int WN, int TR>
__global__ void __launch_bounds__(256)
conv_fewch_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w /* [COUT][3][3][CIN] */, const _Float16* __restrict__ bias,
                  _Float16* __restrict__ out, int Hin, int Win, int Hout, int Wout, int act, int cout_real) {
  // cout_real < COUT: the weight has COUT = 16 rows of which only the first cout_real are real (the 320 -> 4 / 512 -> 8 output
  // convolutions of the U-Net / VAE encoder); bias and out have cout_real channels
  constexpr int WM = 4 / WN, MI = TR / WM, NI = COUT / 16 / WN;
  constexpr int HR = (TR - 1) * STRIDE + 3, HC = 15 * STRIDE + 3;
  constexpr int PSB = CIN * 2 + 16, CPP = CIN / 8;                         // pixel stride (bytes), 16-byte chunks per pixel
  constexpr int K = 9 * CIN, KS = (K + 31) / 32;
  static_assert(TR % WM == 0 && (COUT / 16) % WN == 0 && CIN % 8 == 0, "tile split");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_halo[];   // [HR * HC][PSB]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wn = wave % WN, wm = wave / WN;
  const int tiles_x = Wout / 16;
  const int n = blockIdx.y, tile = blockIdx.x, y0 = (tile / tiles_x) * TR, x0 = (tile % tiles_x) * 16;
  const int iy0 = y0 * STRIDE - 1, ix0 = x0 * STRIDE - 1;
  for (int i = tid; i < HR * HC * CPP; i += 256) {
    const int hp = i / CPP, ch = i - hp * CPP;
    const int hy = hp / HC, hx = hp - hy * HC;
    const int iy = iy0 + hy, ix = ix0 + hx;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win)
      v = *(const uint4*)(x + (((size_t)n * Hin + iy) * Win + ix) * CIN + ch * 8);
    *(uint4*)(s_halo + hp * PSB + ch * 16) = v;
  }
  const int frow = lane & 15, kq = lane >> 4;
  const int n0 = wn * NI * 16;
  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const f16x4 b = (bias && n0 + ni * 16 + kq * 4 < cout_real) ? *(const f16x4*)(bias + n0 + ni * 16 + kq * 4) : (f16x4){0, 0, 0, 0};
#pragma unroll
    for (int mi = 0; mi < MI; mi++) acc[ni][mi] = (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
  const _Float16* wl = w + (size_t)(n0 + frow) * K + kq * 8;               // this lane's weight row, its 8-wide K slot
  f16x8 wa[NI], wnx[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) wa[ni] = (kq * 8 < K) ? *(const f16x8*)(wl + (size_t)ni * 16 * K) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
#pragma unroll 3
  for (int s = 0; s < KS; s++) {
    const int kk = s * 32 + kq * 8;
    if (s + 1 < KS) {
      const bool okn = kk + 32 < K;
#pragma unroll
      for (int ni = 0; ni < NI; ni++)
        wnx[ni] = okn ? *(const f16x8*)(wl + (size_t)ni * 16 * K + (s + 1) * 32) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const int kc = kk < K ? kk : 0;                      // past the end: weights are zero, read any finite pixel data
    const int tap = kc / CIN, c = kc - tap * CIN, ty = tap / 3, tx = tap - ty * 3;
    const unsigned char* hb = s_halo + ((wm * MI * STRIDE + ty) * HC + frow * STRIDE + tx) * PSB + c * 2;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
      const f16x8 pb = *(const f16x8*)(hb + mi * STRIDE * HC * PSB);
#pragma unroll
      for (int ni = 0; ni < NI; ni++) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ni], pb, acc[ni][mi], 0, 0, 0);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ni++) wa[ni] = wnx[ni];
  }
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    _Float16* o = out + (((size_t)n * Hout + y0 + wm * MI + mi) * Wout + x0 + frow) * cout_real + n0 + kq * 4;
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      const f32x4 a = acc[ni][mi];
      f16x4 h;
      h[0] = (_Float16)a[0]; h[1] = (_Float16)a[1]; h[2] = (_Float16)a[2]; h[3] = (_Float16)a[3];
      if (act) { h[0] = silu_h(h[0]); h[1] = silu_h(h[1]); h[2] = silu_h(h[2]); h[3] = silu_h(h[3]); }
      if (n0 + ni * 16 + kq * 4 < cout_real) *(f16x4*)(o + ni * 16) = h;
    }
  }
}

template <int CIN, int COUT, int STRIDE, int WN, int TR>
static int launch_fewch(const void* x, const void* w, const void* bias, void* out, int N, int Hin, int Win, int act, hipStream_t s,
                        int cout_real = COUT) {
  const int Hout = Hin / STRIDE, Wout = Win / STRIDE;
  if ((Hin % STRIDE) || (Win % STRIDE) || (Hout % TR) || (Wout & 15)) return 1;
  constexpr int HR = (TR - 1) * STRIDE + 3, HC = 15 * STRIDE + 3;
  constexpr int lds = HR * HC * (CIN * 2 + 16);
  static bool attr_set = false;
  if (!attr_set) {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_fewch_kernel<CIN, COUT, STRIDE, WN, TR>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 3;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_fewch_kernel<CIN, COUT, STRIDE, WN, TR>), dim3((Hout / TR) * (Wout / 16), N), dim3(256), lds, s,
                     (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, Hin, Win, Hout, Wout, act, cout_real);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_conv3x3_fewch_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t Hin, int32_t Win,
                                          int32_t Cin, int32_t Cout, int32_t stride, int32_t act, void* stream) {
  if (!x || !w || !out || N < 1 || Hin < 1 || Win < 1 || (stride != 1 && stride != 2)) return 1;
  if ((long long)N * Hin * Win * (Cin > Cout ? Cin : Cout) * 2 >= (1ll << 40)) return 1;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 3 && stride == 1 && (Cout == 16 || Cout == 128)) {
    if ((Hin & 15) || (Win & 15)) return 1;
    const dim3 grid((Hin / 16) * (Win / 16), N);
    if (Cout == 16 && act)
      hipLaunchKernelGGL((conv_c3_fwd_kernel<16, true>), grid, dim3(256), 0, s, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, Hin, Win);
    else if (Cout == 16)
      hipLaunchKernelGGL((conv_c3_fwd_kernel<16, false>), grid, dim3(256), 0, s, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, Hin, Win);
    else if (act)
      hipLaunchKernelGGL((conv_c3_fwd_kernel<128, true>), grid, dim3(256), 0, s, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, Hin, Win);
    else
      hipLaunchKernelGGL((conv_c3_fwd_kernel<128, false>), grid, dim3(256), 0, s, (const _Float16*)x, (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, Hin, Win);
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  if (Cin == 16 && Cout == 16 && stride == 1) return launch_fewch<16, 16, 1, 1, 8>(x, w, bias, out, N, Hin, Win, act, s);
  if (Cin == 16 && Cout == 32 && stride == 2) return launch_fewch<16, 32, 2, 2, 8>(x, w, bias, out, N, Hin, Win, act, s);
  if (Cin == 32 && Cout == 32 && stride == 1) return launch_fewch<32, 32, 1, 2, 8>(x, w, bias, out, N, Hin, Win, act, s);
  if (Cin == 32 && Cout == 96 && stride == 2) return launch_fewch<32, 96, 2, 2, 8>(x, w, bias, out, N, Hin, Win, act, s);
  if (Cin == 96 && Cout == 96 && stride == 1) return launch_fewch<96, 96, 1, 2, 8>(x, w, bias, out, N, Hin, Win, act, s);
  if (Cin == 96 && Cout == 256 && stride == 2) return launch_fewch<96, 256, 2, 4, 4>(x, w, bias, out, N, Hin, Win, act, s);
  // 8 input channels: conv_in of the U-Net / ControlNet (4 latent channels, padded to 8 by the host: 8 -> 320) and the data
  // gradient of the VAE encoder's (conv_out . quant_conv) (8 moment channels -> 512)
  if (Cin == 8 && Cout == 320 && stride == 1) return launch_fewch<8, 320, 1, 4, 8>(x, w, bias, out, N, Hin, Win, act, s);
  if (Cin == 8 && Cout == 512 && stride == 1) return launch_fewch<8, 512, 1, 4, 4>(x, w, bias, out, N, Hin, Win, act, s);
  // the narrow OUTPUT convolutions (conv_out of the U-Net: 320 -> 4, of the VAE encoder: 512 -> 8): w holds 16 rows, rows >= Cout zero
  if (Cin == 320 && Cout == 4 && stride == 1) return launch_fewch<320, 16, 1, 1, 8>(x, w, bias, out, N, Hin, Win, act, s, 4);
  if (Cin == 512 && Cout == 8 && stride == 1) return launch_fewch<512, 16, 1, 1, 4>(x, w, bias, out, N, Hin, Win, act, s, 8);
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient: dy [N, H, W, 128] -> dx [N, H, W, 3];  wt [3][9][128] = w[co][2 - ty][2 - tx][c] rearranged by the host
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv_c3_dgrad_kernel(const _Float16* __restrict__ dy, const _Float16* __restrict__ wt, _Float16* __restrict__ dx, int N, int H, int W) {
  constexpr int C = 128;
  constexpr int HALO_W = 18, HALO_H = 10, HALO = HALO_W * HALO_H;          // 180 pixels x 256 bytes
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // [HALO][256] | weights [3][9 * 128] halves
  unsigned char* s_halo = smem;
  _Float16* s_w = (_Float16*)(smem + HALO * 256);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_x = W / 16;
  const int n = blockIdx.y, tile = blockIdx.x, y0 = (tile / tiles_x) * 8, x0 = (tile % tiles_x) * 16;
  // ---- halo of dy by LDS-DMA: one instruction = 4 pixels (64 lanes x 16 bytes); chunk c of pixel hp lands at chunk c ^ (hp & 15)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)N * H * W * C * 2u), CS_RSRC_FLAGS);
  for (int i = wave; i < HALO / 4; i += 4) {
    const int hp = i * 4 + (lane >> 4), pchunk = lane & 15, lchunk = pchunk ^ (hp & 15);
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const unsigned off = ok ? (unsigned)((((unsigned)n * H + iy) * W + ix) * C + lchunk * 8) * 2u : CS_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(s_halo + i * 1024), 16, off, 0, 0, 0);
  }
  for (int i = tid; i < 3 * 9 * C / 8; i += 256) ((uint4*)s_w)[i] = ((const uint4*)wt)[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int frow = lane & 15, kq = lane >> 4;
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
  for (int t = 0; t < 9; t++) {
    const int ty = t / 3, tx = t - ty * 3;
    int hp[2], sw[2];
#pragma unroll
    for (int cs = 0; cs < 2; cs++) {
      hp[cs] = (wave * 2 + cs + ty) * HALO_W + frow + tx;
      sw[cs] = hp[cs] & 15;
    }
#pragma unroll
    for (int cb = 0; cb < 4; cb++) {
      f16x8 a = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (frow < 3) a = *(const f16x8*)(s_w + (frow * 9 + t) * C + cb * 32 + kq * 8);
#pragma unroll
      for (int cs = 0; cs < 2; cs++) {
        const f16x8 b = *(const f16x8*)(s_halo + hp[cs] * 256 + (((cb * 4 + kq) ^ sw[cs]) << 4));
        acc[cs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[cs], 0, 0, 0);
      }
    }
  }
  // lanes 0-15 hold output channels 0..3 (3 used) of pixel frow
  if (kq == 0) {
#pragma unroll
    for (int cs = 0; cs < 2; cs++) {
      _Float16* o = dx + (((size_t)n * H + y0 + wave * 2 + cs) * W + x0 + frow) * 3;
      o[0] = (_Float16)acc[cs][0]; o[1] = (_Float16)acc[cs][1]; o[2] = (_Float16)acc[cs][2];
    }
  }
}

extern "C" int gip_conv3x3_c3_fwd_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t H, int32_t W,
                                           int32_t Cout, void* stream) {
  if (!x || !w || !out || N < 1 || H < 16 || W < 16 || (H & 15) || (W & 15) || Cout != 128) return 1;
  if ((long long)N * H * W * Cout * 2 >= (1ll << 32)) return 1;
  hipLaunchKernelGGL((conv_c3_fwd_kernel<128, false>), dim3((H / 16) * (W / 16), N), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                     (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, H, W);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_conv3x3_c3_fwd_stats_nhwc_f16(const void* x, const void* w, const void* bias, void* out, int32_t N, int32_t H, int32_t W,
                                                 int32_t Cout, float* chan_stats, void* stream) {
  if (!x || !w || !out || !chan_stats || N < 1 || H < 16 || W < 16 || (H & 15) || (W & 15) || Cout != 128) return 1;
  if ((long long)N * H * W * Cout * 2 >= (1ll << 32)) return 1;
  hipLaunchKernelGGL((conv_c3_fwd_kernel<128, false>), dim3((H / 16) * (W / 16), N), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                     (const _Float16*)w, (const _Float16*)bias, (_Float16*)out, H, W, chan_stats);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

extern "C" int gip_conv3x3_c3_dgrad_nhwc_f16(const void* dy, const void* wt, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                             void* stream) {
  if (!dy || !wt || !dx || N < 1 || H < 8 || W < 16 || (H & 7) || (W & 15) || C != 128) return 1;
  if ((long long)N * H * W * C * 2 >= (1ll << 31)) return 1;          // buffer resource: 32-bit byte offsets
  const size_t lds = 180 * 256 + 3 * 9 * 128 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_c3_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 3;
    attr_set = true;
  }
  hipLaunchKernelGGL(conv_c3_dgrad_kernel, dim3((H / 8) * (W / 16), N), dim3(256), lds, (hipStream_t)stream, (const _Float16*)dy,
                     (const _Float16*)wt, (_Float16*)dx, N, H, W);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
