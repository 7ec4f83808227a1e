// gip_internal.h — shared declarations of the HIP rasterizer (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gip_raster.h"

#define GIP_BLOCK 256           // threads per workgroup everywhere (4 x wave64)
#define GIP_WAVE 64
#define GIP_NEAR 0.2f           // view-space z cull (in_frustum)
#define GIP_ALPHA_MIN (1.0f / 255.0f)
#define GIP_ALPHA_MAX 0.99f
#define GIP_T_MIN 0.0001f

// Per-(view, Gaussian) projected record, 64 bytes = four 16-byte quads so a lane fetches it with
// global_load_dwordx4.  Layout documented in include/gip_raster.h.
struct __attribute__((aligned(16))) GipRecord {
  float x, y, depth, opacity;           // q0
  float ca, cb, cc; uint32_t tiles;     // q1
  float r, g, b; int32_t radius;        // q2
  uint32_t rmin, rmax, clamped, tmask;  // q3  (rect in tiles: x | y << 16; max exclusive; tmask: see gip_rect_has)
};

// Instances of a Gaussian: the tiles of its rectangle whose bit is set in `tmask` (bit k = tile k of the rectangle in
// row-major order) when the rectangle has at most GIP_MASK_TILES tiles; every tile of a larger rectangle.
#define GIP_MASK_TILES 32
__host__ __device__ inline bool gip_rect_has(uint32_t tmask, int area, int k) { return area > GIP_MASK_TILES || ((tmask >> k) & 1u); }
// index of tile k among the Gaussian's instances (its row / slot number)
__host__ __device__ inline int gip_rect_rank(uint32_t tmask, int area, int k) {
  return area > GIP_MASK_TILES ? k : __builtin_popcount(tmask & ((1u << k) - 1u));
}

static_assert(sizeof(GipRecord) == GIP_RECORD_BYTES, "record size");

struct GipViewConst {           // per-view host scalars, passed by value in kernel args
  float tanfovx, tanfovy, focal_x, focal_y;
};

struct GipKernelParams {
  int P, V, H, W;
  int tiles_x, tiles_y, T;      // T = tiles_x * tiles_y
  int nblk;                     // ceil(P / 256)
  int D, M;
  float scale_modifier;
  uint32_t capacity;
  uint32_t seg_capacity;    // capacity / GIP_SEGMENT + V*T
  uint32_t ckpt_capacity;   // capacity / GIP_SEGMENT
  int exact_lists;          // GipRasterConfig::exact_lists
  int forward_only;         // GipRasterConfig::forward_only: nothing is kept for a backward
  int sh_mfma;              // SH colours / their backward on the matrix cores (sh_mfma.hip): shs given, degree >= 1, V >= 2, !sh_scalar
  GipViewConst view[GIP_MAX_VIEWS];
};

struct GipStatePtrs {
  GipRasterHeader* header;
  GipRecord* records;       // [V,P]
  uint32_t* inst_offset;    // [V,P]
  uint32_t* tile_count;     // [V*T]
  uint32_t* tile_start;     // [V*T+1]
  uint32_t* tile_cursor;    // [V*T]
  uint32_t* tile_count_b;   // [V*T]
  uint32_t* inst_slot;      // [V,P,GIP_SLOTS]
  uint32_t* block_sums;     // [V*nblk]
  uint32_t* block_offset;   // [V*nblk+1]
  unsigned long long* keys; // [capacity]
  uint32_t* n_contrib;      // [V,H,W]
  float* final_T;           // [V,H,W]
  uint32_t* tile_order;     // [V*T] heavy-first launch order
  uint32_t* seg_start;      // [V*T+1]
  uint32_t* ckpt_start;     // [V*T+1]
  uint32_t* seg_tile;       // [seg_capacity]
  float* checkpoints;       // [ckpt_capacity][5][256]
  uint32_t* host_header;    // optional pinned host mirror of the header's first words (GipRasterOutputs::host_header)
  float* sh_colors;         // [V,P,4] (matrix-core SH path only): forward colours before the clamp / backward dL/dcolour
};

// --- launchers implemented in the .hip translation units -----------------------------------------
void gip_launch_preprocess(const GipKernelParams& kp, const GipRasterInputs& in, int32_t* radii, GipStatePtrs st, hipStream_t s);
void gip_launch_scan(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s);
void gip_launch_scatter(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s);
void gip_launch_tile_sort(const GipKernelParams& kp, GipStatePtrs st, hipStream_t s);
void gip_launch_render_forward(const GipKernelParams& kp, const float* bg, GipStatePtrs st, float* color, float* depth, float* alpha, hipStream_t s);
void gip_launch_render_backward(const GipKernelParams& kp, const float* bg, GipStatePtrs st, const GipRasterGradsIn& gin, float* partial, hipStream_t s);
void gip_launch_gather_backward(const GipKernelParams& kp, const GipRasterInputs& in, GipStatePtrs st, const float* partial, const GipRasterGradsOut& gout, hipStream_t s);
void gip_launch_sh_forward_mfma(const GipKernelParams& kp, const GipRasterInputs& in, float* sh_colors, hipStream_t s);
void gip_launch_sh_backward_mfma(const GipKernelParams& kp, const GipRasterInputs& in, GipStatePtrs st, const float* sh_gcol, const GipRasterGradsOut& gout, hipStream_t s);
void gip_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);

// --- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__
// Wave64 inclusive sum via DPP row shifts + row broadcasts (no LDS traffic).
__device__ __forceinline__ uint32_t gip_wave_incl_scan_u32(uint32_t v) {
  // log-step scan with shuffles; hipcc lowers constant-offset __shfl_up to DPP / ds_bpermute
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Sum over the 64 lanes of a wave, result valid in every lane.
__device__ __forceinline__ float gip_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t gip_wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
  return v;
}
#endif
