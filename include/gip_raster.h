/*
 * gip_raster.h — C-ABI of the MI355X-native differentiable 3D-Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for the rasterizer half of GaussianIP's hot path.  Every entry point
 * replaces one binding of the (un-vendored) `diff_gaussian_rasterization._C` module that the
 * reference calls through its Python package:
 *
 *   gip_raster_forward   <->  _C.rasterize_gaussians            (called from GaussianRasterizer.forward;
 *                              reference call sites gaussiansplatting/gaussian_renderer/__init__.py:85-93,
 *                              :175-183, :240-248 and gs_renderer.py:992-1001)
 *   gip_raster_backward  <->  _C.rasterize_gaussians_backward   (autograd backward of the same call;
 *                              grads consumed at threestudio/systems/GaussianIP.py:452-457 and by Adam)
 *   gip_raster_mark_visible <-> _C.mark_visible                 (GaussianRasterizer.markVisible; unused by
 *                              the reference but part of the package surface)
 *
 * Contract:
 *   - plain C, no torch / pybind types; every pointer is a raw DEVICE pointer unless marked [host];
 *   - the caller owns every buffer (inputs, outputs, state, scratch); the library never allocates,
 *     frees, or keeps a pointer after return, and has no global mutable state (re-entrant, one
 *     process per GPU safe);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); the calls never synchronise
 *     with the host.  The reference's one blocking D2H per forward (`num_rendered`) is replaced by a
 *     caller-chosen `capacity` and a device-side status header the caller may read back whenever it
 *     likes (gip_raster_read_header);
 *   - integer status codes, no exceptions.
 *
 * Batched views: the reference renders its 4 cameras one after another (GaussianIP.py:154-173).
 * Here a call renders V >= 1 views of the SAME Gaussians in one launch set; V = 1 reproduces the
 * reference's per-camera call exactly.  Per-view arrays are laid out view-major: [V, ...].
 */
#ifndef GIP_RASTER_H
#define GIP_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIP_ABI_VERSION 4
#define GIP_TILE 16            /* tile edge in pixels (BLOCK_X = BLOCK_Y = 16 in the reference's rasterizer) */
#define GIP_MAX_VIEWS 16       /* views per call */
#define GIP_RECORD_BYTES 64    /* per-(view, Gaussian) projected record kept for backward */
#define GIP_PARTIAL_FLOATS 12  /* per-(tile, Gaussian) gradient partial row, 48 bytes (10 used) */
#ifndef GIP_SEGMENT
#define GIP_SEGMENT 64
#endif
//#define GIP_SEGMENT_DOC        /* list entries per backward work item; forward checkpoints every GIP_SEGMENT entries */
#define GIP_SLOTS 8            /* bucket slots remembered per (view, Gaussian): scatter needs no second atomic for these */
#define GIP_CKPT_FLOATS 5      /* per pixel per checkpoint: T, C.r, C.g, C.b, D */

/* status codes */
enum {
  GIP_OK = 0,
  GIP_ERR_BAD_ARGUMENT = 1,    /* null pointer / inconsistent optional inputs / bad sizes */
  GIP_ERR_BUFFER_TOO_SMALL = 2,/* state or scratch smaller than gip_raster_*_bytes() says */
  GIP_ERR_HIP = 3,             /* a HIP runtime call or launch failed */
  GIP_ERR_UNSUPPORTED = 4      /* e.g. sh_degree > 3, V > GIP_MAX_VIEWS */
};

/* Per-call constants: the fields of GaussianRasterizationSettings
 * (gaussian_renderer/__init__.py:36-49) plus sizes. */
typedef struct GipRasterConfig {
  int32_t P;              /* number of Gaussians */
  int32_t V;              /* number of views rendered by this call (1..GIP_MAX_VIEWS) */
  int32_t H, W;           /* image_height, image_width (shared by all views) */
  int32_t sh_degree;      /* active SH degree 0..3 */
  int32_t sh_coeffs;      /* M: SH coefficients per channel stored in `shs` ([P, M, 3]); 0 if colors_precomp */
  int32_t prefiltered;    /* settings.prefiltered (always False in the reference) */
  int32_t debug;          /* settings.debug: synchronise + check after every launch */
  float scale_modifier;   /* settings.scale_modifier */
  float tanfovx[GIP_MAX_VIEWS]; /* [host] per view */
  float tanfovy[GIP_MAX_VIEWS]; /* [host] per view */
  uint64_t capacity;      /* max number of (tile, Gaussian) instances (= num_rendered summed over views)
                             the state buffers can hold; overflow is flagged in the header */
  int32_t exact_lists;    /* 0 (default): instances are made only for the tiles a Gaussian's alpha >= 1/255 region can
                             reach (see the record layout below) — the tile / index buffers are then the fork's MINUS
                             entries that cannot contribute, every output is unchanged.  1: every tile of the fork's
                             3-sigma rectangle gets its instance: tiles_touched, num_rendered, the sorted key / value
                             lists and the tile ranges are bit-for-bit the fork's (parity tests run both). */
  int32_t forward_only;   /* 0 (default): the forward leaves in `state` what gip_raster_backward needs.  1: no backward will
                             follow (rendering under torch.no_grad(): orbit renders, the refine pass' inputs) — the render
                             kernel then skips the per-segment blend-state checkpoints and the n_contrib / final_T images
                             (~100 MB of writes per 4 x 1024^2 launch); outputs are bit-identical, and gip_raster_backward on
                             such a state returns GIP_ERR_BAD_ARGUMENT.  (This field was `reserved`; the layout is unchanged.) */
  int32_t sh_scalar;      /* ABI 4.  0 (default): with `shs`, sh_degree >= 1 and V >= 2 the SH colour contraction and its backward run
                             on the matrix cores, batched over the views of the launch set (csrc/sh_mfma.hip: v_mfma_f32_4x4x1, one
                             Gaussian's [4 views x K] . [K x 3] product per 4 x 4 block); colours then agree with the scalar sum
                             to a few ulp — no integer buffer depends on them.  1: always the scalar chain of eval_sh
                             (sh_utils.py:57-112), bit-exact against the oracle's colours.  V = 1 and degree 0 are always scalar. */
} GipRasterConfig;

/* Device inputs.  Exactly one of (shs, colors_precomp) and one of (scales+rotations, cov3D_precomp)
 * must be non-null — same rule as GaussianRasterizer.forward. */
typedef struct GipRasterInputs {
  const float* means3D;        /* [P,3] */
  const float* shs;            /* [P,M,3] or null */
  const float* colors_precomp; /* [P,3]  or null */
  const float* opacities;      /* [P,1]  (post-sigmoid) */
  const float* scales;         /* [P,3]  (post-exp) or null */
  const float* rotations;      /* [P,4]  (w,x,y,z) or null */
  const float* cov3D_precomp;  /* [P,6]  (xx,xy,xz,yy,yz,zz) or null */
  const float* viewmatrix;     /* [V,16] world_view_transform, row-vector convention as stored by cameras.py:48 */
  const float* projmatrix;     /* [V,16] full_proj_transform, cameras.py:50 */
  const float* campos;         /* [V,3]  camera_center */
  const float* bg;             /* [3]    background colour */
} GipRasterInputs;

typedef struct GipRasterOutputs {
  float*   color;  /* [V,3,H,W] */
  int32_t* radii;  /* [V,P]     */
  float*   depth;  /* [V,1,H,W] */
  float*   alpha;  /* [V,1,H,W] */
  uint32_t* host_header; /* optional (null = unused): 16 x u32 of PINNED, device-mapped HOST memory.  The binning stage
                            stores {abi_version, num_rendered, overflow, max_tile_count} there as soon as the totals
                            are known, so the caller learns the capacity verdict from an event on `stream` without
                            enqueueing a device-to-host copy (the reference instead blocks on a D2H read of
                            num_rendered inside every forward). */
} GipRasterOutputs;

/* Upstream gradients (any may be null = zero) and the forward outputs (all three required: the
 * segment-parallel backward derives the blend suffix of every entry from the totals). */
typedef struct GipRasterGradsIn {
  const float* dL_dcolor; /* [V,3,H,W] */
  const float* dL_ddepth; /* [V,1,H,W] */
  const float* dL_dalpha; /* [V,1,H,W] */
  const float* alpha;     /* [V,1,H,W] forward output `alpha` (final T = 1 - alpha) */
  const float* color;     /* [V,3,H,W] forward output `color` */
  const float* depth;     /* [V,1,H,W] forward output `depth` */
} GipRasterGradsIn;

/* Gradients w.r.t. the inputs, summed over the V views (means2D is per view).  Null = not wanted.
 * All are fully overwritten (no pre-zeroing needed). */
typedef struct GipRasterGradsOut {
  float* dL_dmeans3D;        /* [P,3] */
  float* dL_dmeans2D;        /* [V,P,3] screen-space grad carrier (xy in NDC units, z = 0) */
  float* dL_dshs;            /* [P,M,3] */
  float* dL_dcolors_precomp; /* [P,3] */
  float* dL_dopacities;      /* [P,1] */
  float* dL_dscales;         /* [P,3] */
  float* dL_drotations;      /* [P,4] */
  float* dL_dcov3D_precomp;  /* [P,6] */
} GipRasterGradsOut;

/* Status header written by forward at the start of the state buffer (16 x u32). */
typedef struct GipRasterHeader {
  uint32_t abi_version;
  uint32_t num_rendered;   /* total (tile, Gaussian) instances required, over all views (of the rectangles described
                              at the record layout below: <= the fork's num_rendered) */
  uint32_t overflow;       /* 1 if num_rendered > capacity: outputs are invalid, re-run with more */
  uint32_t max_tile_count; /* longest per-tile list */
  uint32_t num_visible;    /* Gaussians with radii > 0, over all views */
  uint32_t num_segments;   /* sum over tiles of ceil(count / GIP_SEGMENT): work items of the backward replay */
  uint32_t num_checkpoints;/* sum over tiles of max(segments - 1, 0) */
  uint32_t class_end[4];   /* positions in tile_order where the per-tile-sort size classes end: [1] lists >= 2048, [2] >= 1024, [0] >= 512, [3] >= 1 */
  uint32_t sort_cursor;    /* scratch: work counters of the per-tile sort kernel (lists >= 2048 entries, ... */
  uint32_t sort_cursor_m;  /* ... lists of 512..2047 entries) */
  uint32_t reserved[3];
} GipRasterHeader;

/* Byte offsets of the sub-buffers inside `state` (for tests, debugging and parity checks of the
 * tile / index buffers).  All offsets are 256-byte aligned. */
typedef struct GipRasterStateLayout {
  size_t header;       /* GipRasterHeader */
  size_t records;      /* [V,P] x GIP_RECORD_BYTES */
  size_t inst_offset;  /* [V,P] u32: exclusive prefix sum of tiles_touched (view-major, global) */
  size_t tile_count;   /* [V,T] u32: entries per tile drawn through the remembered slots (total = this + tile_count_b) */
  size_t tile_start;   /* [V*T+1] u32: exclusive scan of tile_count == ranges[tile].x, ranges[tile].y = next */
  size_t tile_cursor;  /* [V,T] u32 scratch for bucket fill (instances beyond GIP_SLOTS per Gaussian) */
  size_t tile_count_b; /* [V,T] u32: part of the per-tile count contributed by instances beyond GIP_SLOTS */
  size_t inst_slot;    /* [V,P,GIP_SLOTS] u32: bucket slot each of a Gaussian's first GIP_SLOTS instances drew */
  size_t block_sums;   /* [V,ceil(P/256)] u32 */
  size_t block_offset; /* [V*ceil(P/256)+1] u32 */
  size_t keys;         /* [capacity] u64 sorted per tile: (depth_bits << 32) | gaussian_index */
  size_t n_contrib;    /* [V,H,W] u32 */
  size_t final_T;      /* [V,H,W] f32: transmittance after the last blended entry (the fork's final_T) */
  size_t tile_order;   /* [V*T] u32: tile ids, longest lists first (launch order of the render kernels) */
  size_t seg_start;    /* [V*T+1] u32: exclusive scan of per-tile segment counts */
  size_t ckpt_start;   /* [V*T+1] u32: exclusive scan of per-tile checkpoint counts (segments - 1) */
  size_t seg_tile;     /* [capacity/GIP_SEGMENT + V*T] u32: tile of each segment */
  size_t checkpoints;  /* [capacity/GIP_SEGMENT][GIP_CKPT_FLOATS][256] f32: per-pixel blend state at segment starts */
  size_t sh_colors;    /* ABI 4: [V,P,4] f32, present (non-zero size) only on the matrix-core SH path: forward = colours before the
                          clamp, backward = dL/dcolour after it (csrc/sh_mfma.hip) */
  size_t total;        /* total bytes */
  uint32_t tiles_x, tiles_y, num_blocks, reserved;
} GipRasterStateLayout;

/* Per-(view,Gaussian) record (GIP_RECORD_BYTES = 64), written by preprocess:
 *   float  x, y        pixel-space centre (points_xy_image)
 *   float  depth       view-space z
 *   float  opacity
 *   float  conic_a, conic_b, conic_c
 *   uint32 tiles_touched   instances made for this Gaussian = tiles of the rectangle below that are set in tile_mask.  NOT the fork's count: the
 *                          fork's 3-sigma rectangle is intersected with the extent of the alpha >= 1/255 region
 *                          (|dx| <= sqrt(2 ln(255 opacity) cov_xx), same for y): tiles outside it cannot pass the
 *                          fork's alpha test at any pixel, so no output depends on them
 *   float  r, g, b     colour after SH / clamp
 *   int32  radius          the fork's ceil(3 sigma_max) (the `radii` output / visibility), independent of the above
 *   uint32 rect_min    (x | y << 16) in tiles
 *   uint32 rect_max    (x | y << 16) in tiles, exclusive
 *   uint32 clamped     bit0..2 = colour channel clamped at 0
 *   uint32 tile_mask   rectangles of at most 32 tiles: bit k set = tile k of the rectangle (row-major) is an instance
 *                      (the others cannot pass the alpha test either: the ellipse misses them); larger rectangles: all
 */

int gip_abi_version(void);
const char* gip_status_string(int status);

/* Sizes.  Return 0 on invalid config. */
size_t gip_raster_state_bytes(const GipRasterConfig* cfg);
size_t gip_raster_scratch_bytes(const GipRasterConfig* cfg); /* backward scratch: capacity x 64 B partial rows */
int    gip_raster_state_layout(const GipRasterConfig* cfg, GipRasterStateLayout* out);

/* Forward: preprocess -> tile binning -> per-tile depth sort -> front-to-back blend.
 * `state` (>= gip_raster_state_bytes) must be kept by the caller until backward has run. */
int gip_raster_forward(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterOutputs* out,
                       void* state, size_t state_bytes, void* stream);

/* Backward: per-pixel reverse-order replay with wave/LDS segmented reduction into per-(tile,Gaussian)
 * partial rows, then a deterministic per-Gaussian gather fused with the cov2D / projection / SH /
 * cov3D backward.  No float atomics: results are bitwise reproducible.
 * If the forward overflowed its capacity (GipRasterHeader.overflow != 0) the backward kernels read the flag on the
 * device: every gradient output is written as ZERO (a caller may enqueue backward, the optimizer and a multi-GPU
 * gradient exchange before it has looked at the header: the step degenerates to a zero-gradient step, identically on
 * every rank, and the caller raises the capacity for the next call after reading the header). */
int gip_raster_backward(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterGradsIn* gin,
                        const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                        const GipRasterGradsOut* gout, void* stream);

/* Measurement variants: identical work, but a hipEvent pair brackets every stage on `stream` and the call
 * synchronises at the end to fill `times_ms` ([host]).  Used by bench.py for the live per-kernel durations the
 * roofline figure is computed from; never used on the training path. */
enum { GIP_STAGE_CLEAR = 0, GIP_STAGE_PREPROCESS, GIP_STAGE_SCAN, GIP_STAGE_SCATTER, GIP_STAGE_TILE_SORT,
       GIP_STAGE_RENDER_FWD, GIP_STAGE_RENDER_BWD, GIP_STAGE_GATHER_BWD, GIP_NUM_STAGES };
int gip_raster_forward_profiled(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterOutputs* out,
                                void* state, size_t state_bytes, void* stream, float* times_ms /* [GIP_NUM_STAGES] */);
int gip_raster_backward_profiled(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterGradsIn* gin,
                                 const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                                 const GipRasterGradsOut* gout, void* stream, float* times_ms /* [GIP_NUM_STAGES] */);

/* Asynchronously copies the header to `host_header` ([host], ideally pinned) on `stream`. */
int gip_raster_read_header(const void* state, GipRasterHeader* host_header, void* stream);

/* mark_visible: present[i] = view-space z of means3D[i] > 0.2 (the rasterizer's frustum test). */
int gip_raster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                            uint8_t* present, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GIP_RASTER_H */
