// api.hip — extern "C" entry points of libgip_raster.so (see include/gip_raster.h).
#include <string.h>

#include "gip_internal.h"

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// matrix-core SH path (sh_mfma.hip): SH coefficients given (sh_coeffs > 0; 0 = colors_precomp), something to contract
// (degree >= 1), views to batch (V >= 2), not switched off
static bool sh_mfma_path(const GipRasterConfig* c) { return c->sh_coeffs > 0 && c->sh_degree >= 1 && c->V >= 2 && !c->sh_scalar; }

static bool valid_config(const GipRasterConfig* c) {
  if (!c) return false;
  if (c->P < 0 || c->V < 1 || c->V > GIP_MAX_VIEWS || c->H < 1 || c->W < 1) return false;
  if (c->sh_degree < 0 || c->sh_degree > 3) return false;
  if (c->capacity < 1 || c->capacity > 0xfffffff0ull) return false;
  if ((c->H + GIP_TILE - 1) / GIP_TILE > 0xffff || (c->W + GIP_TILE - 1) / GIP_TILE > 0xffff) return false;
  return true;
}

extern "C" int gip_abi_version(void) { return GIP_ABI_VERSION; }

extern "C" const char* gip_status_string(int status) {
  switch (status) {
    case GIP_OK: return "ok";
    case GIP_ERR_BAD_ARGUMENT: return "bad argument (null pointer, or not exactly one of shs/colors_precomp and scales+rotations/cov3D_precomp)";
    case GIP_ERR_BUFFER_TOO_SMALL: return "state or scratch buffer smaller than gip_raster_state_bytes()/gip_raster_scratch_bytes()";
    case GIP_ERR_HIP: return "HIP runtime error";
    case GIP_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

extern "C" int gip_raster_state_layout(const GipRasterConfig* c, GipRasterStateLayout* L) {
  if (!valid_config(c) || !L) return GIP_ERR_BAD_ARGUMENT;
  memset(L, 0, sizeof(*L));
  const size_t P = (size_t)c->P, V = (size_t)c->V;
  L->tiles_x = (uint32_t)((c->W + GIP_TILE - 1) / GIP_TILE);
  L->tiles_y = (uint32_t)((c->H + GIP_TILE - 1) / GIP_TILE);
  L->num_blocks = (uint32_t)((c->P + GIP_BLOCK - 1) / GIP_BLOCK);
  const size_t T = (size_t)L->tiles_x * L->tiles_y, nblk = L->num_blocks;
  size_t off = 0;
  L->header = off;       off = align256(off + sizeof(GipRasterHeader));
  // the three zero-initialised arrays are adjacent so that one memset clears header..block region
  L->tile_count = off;   off = align256(off + V * T * 4);
  L->tile_cursor = off;  off = align256(off + V * T * 4);
  L->tile_count_b = off; off = align256(off + V * T * 4);
  L->tile_start = off;   off = align256(off + (V * T + 1) * 4);
  L->block_sums = off;   off = align256(off + (V * nblk + 1) * 4);
  L->block_offset = off; off = align256(off + (V * nblk + 1) * 4);
  L->records = off;      off = align256(off + V * P * GIP_RECORD_BYTES);
  L->inst_offset = off;  off = align256(off + V * P * 4);
  L->inst_slot = off;    off = align256(off + V * P * GIP_SLOTS * 4);
  L->n_contrib = off;    off = align256(off + V * (size_t)c->H * c->W * 4);
  L->final_T = off;      off = align256(off + V * (size_t)c->H * c->W * 4);
  L->tile_order = off;   off = align256(off + V * T * 4);
  const size_t ckpt_cap = (size_t)c->capacity / GIP_SEGMENT + 1, seg_cap = ckpt_cap + V * T;
  L->seg_start = off;    off = align256(off + (V * T + 1) * 4);
  L->ckpt_start = off;   off = align256(off + (V * T + 1) * 4);
  L->seg_tile = off;     off = align256(off + seg_cap * 4);
  L->checkpoints = off;  off = align256(off + ckpt_cap * GIP_CKPT_FLOATS * 256 * sizeof(float));
  L->keys = off;         off = align256(off + (size_t)c->capacity * 8);
  L->sh_colors = off;    off = align256(off + (sh_mfma_path(c) ? V * P * 4 * sizeof(float) : 0));
  L->total = off;
  return GIP_OK;
}

extern "C" size_t gip_raster_state_bytes(const GipRasterConfig* c) {
  GipRasterStateLayout L;
  if (gip_raster_state_layout(c, &L) != GIP_OK) return 0;
  return L.total;
}

extern "C" size_t gip_raster_scratch_bytes(const GipRasterConfig* c) {
  if (!valid_config(c)) return 0;
  return align256((size_t)c->capacity * GIP_PARTIAL_FLOATS * sizeof(float));
}

static void fill_params(const GipRasterConfig* c, const GipRasterStateLayout& L, GipKernelParams* kp) {
  memset(kp, 0, sizeof(*kp));
  kp->P = c->P; kp->V = c->V; kp->H = c->H; kp->W = c->W;
  kp->tiles_x = (int)L.tiles_x; kp->tiles_y = (int)L.tiles_y; kp->T = kp->tiles_x * kp->tiles_y;
  kp->nblk = (int)L.num_blocks;
  kp->D = c->sh_degree; kp->M = c->sh_coeffs;
  kp->scale_modifier = c->scale_modifier;
  kp->capacity = (uint32_t)c->capacity;
  kp->exact_lists = c->exact_lists ? 1 : 0;
  kp->forward_only = c->forward_only ? 1 : 0;
  kp->sh_mfma = sh_mfma_path(c) ? 1 : 0;
  kp->ckpt_capacity = (uint32_t)(c->capacity / GIP_SEGMENT + 1);
  kp->seg_capacity = kp->ckpt_capacity + (uint32_t)(kp->V * kp->T);
  for (int v = 0; v < c->V; v++) {
    kp->view[v].tanfovx = c->tanfovx[v];
    kp->view[v].tanfovy = c->tanfovy[v];
    kp->view[v].focal_x = c->W / (2.0f * c->tanfovx[v]);
    kp->view[v].focal_y = c->H / (2.0f * c->tanfovy[v]);
  }
}

static GipStatePtrs state_ptrs(void* state, const GipRasterStateLayout& L) {
  char* b = (char*)state;
  GipStatePtrs p;
  p.header = (GipRasterHeader*)(b + L.header);
  p.records = (GipRecord*)(b + L.records);
  p.inst_offset = (uint32_t*)(b + L.inst_offset);
  p.tile_count = (uint32_t*)(b + L.tile_count);
  p.tile_start = (uint32_t*)(b + L.tile_start);
  p.tile_cursor = (uint32_t*)(b + L.tile_cursor);
  p.tile_count_b = (uint32_t*)(b + L.tile_count_b);
  p.inst_slot = (uint32_t*)(b + L.inst_slot);
  p.block_sums = (uint32_t*)(b + L.block_sums);
  p.block_offset = (uint32_t*)(b + L.block_offset);
  p.keys = (unsigned long long*)(b + L.keys);
  p.n_contrib = (uint32_t*)(b + L.n_contrib);
  p.final_T = (float*)(b + L.final_T);
  p.tile_order = (uint32_t*)(b + L.tile_order);
  p.seg_start = (uint32_t*)(b + L.seg_start);
  p.ckpt_start = (uint32_t*)(b + L.ckpt_start);
  p.seg_tile = (uint32_t*)(b + L.seg_tile);
  p.checkpoints = (float*)(b + L.checkpoints);
  p.sh_colors = (float*)(b + L.sh_colors);
  p.host_header = nullptr;
  return p;
}

static int check_inputs(const GipRasterConfig* c, const GipRasterInputs* in) {
  if (!in || !in->means3D || !in->opacities || !in->viewmatrix || !in->projmatrix || !in->campos || !in->bg)
    return GIP_ERR_BAD_ARGUMENT;
  if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) return GIP_ERR_BAD_ARGUMENT;
  const bool sr = in->scales != nullptr && in->rotations != nullptr;
  if ((in->scales != nullptr) != (in->rotations != nullptr)) return GIP_ERR_BAD_ARGUMENT;
  if (sr == (in->cov3D_precomp != nullptr)) return GIP_ERR_BAD_ARGUMENT;
  if (in->shs) {
    const int need = (c->sh_degree + 1) * (c->sh_degree + 1);
    if (c->sh_coeffs < need || c->sh_coeffs > 64) return GIP_ERR_BAD_ARGUMENT;
  }
  return GIP_OK;
}

#define HIP_TRY(expr) do { if ((expr) != hipSuccess) return GIP_ERR_HIP; } while (0)

static int after_launch(const GipRasterConfig* c, hipStream_t s) {
  if (hipGetLastError() != hipSuccess) return GIP_ERR_HIP;
  if (c->debug) { if (hipStreamSynchronize(s) != hipSuccess) return GIP_ERR_HIP; }
  return GIP_OK;
}
#define LAUNCH_CHECK() do { int rc_ = after_launch(cfg, s); if (rc_ != GIP_OK) return rc_; } while (0)

namespace {
struct StageTimer {
  hipStream_t s; float* out; hipEvent_t ev[2 * GIP_NUM_STAGES]; bool used[GIP_NUM_STAGES]; bool ok; int created;
  StageTimer(hipStream_t s_, float* out_) : s(s_), out(out_), ok(true), created(0) {
    for (int i = 0; i < GIP_NUM_STAGES; i++) used[i] = false;
    if (out)
      for (int i = 0; i < 2 * GIP_NUM_STAGES && ok; i++) {
        ok = hipEventCreate(&ev[i]) == hipSuccess;
        if (ok) created++;
      }
  }
  void begin(int st) { if (out && ok) { ok = hipEventRecord(ev[2 * st], s) == hipSuccess; used[st] = true; } }
  void end(int st) { if (out && ok) ok = hipEventRecord(ev[2 * st + 1], s) == hipSuccess; }
  int finish() {
    if (!out) return GIP_OK;
    int rc = GIP_OK;
    if (!ok || hipStreamSynchronize(s) != hipSuccess) rc = GIP_ERR_HIP;
    for (int i = 0; i < GIP_NUM_STAGES; i++) {
      if (rc == GIP_OK && used[i]) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = GIP_ERR_HIP;
        out[i] = ms;
      }
    }
    for (int i = 0; i < created; i++) (void)hipEventDestroy(ev[i]);
    return rc;
  }
};
}  // namespace

static int forward_impl(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterOutputs* out,
                        void* state, size_t state_bytes, void* stream, float* times) {
  if (!valid_config(cfg)) return cfg && (cfg->V > GIP_MAX_VIEWS || cfg->sh_degree > 3) ? GIP_ERR_UNSUPPORTED : GIP_ERR_BAD_ARGUMENT;
  int rc = check_inputs(cfg, in);
  if (rc != GIP_OK) return rc;
  if (!out || !out->color || !out->radii || !out->depth || !out->alpha || !state) return GIP_ERR_BAD_ARGUMENT;
  GipRasterStateLayout L;
  gip_raster_state_layout(cfg, &L);
  if (state_bytes < L.total) return GIP_ERR_BUFFER_TOO_SMALL;
  GipKernelParams kp;
  fill_params(cfg, L, &kp);
  if (!in->shs) kp.sh_mfma = 0;                      // colors_precomp with a stale sh_coeffs: nothing to contract
  GipStatePtrs st = state_ptrs(state, L);
  st.host_header = out->host_header;
  hipStream_t s = (hipStream_t)stream;
  StageTimer tm(s, times);

  // header + tile_count + tile_cursor + tile_count_b are contiguous: one clear
  tm.begin(GIP_STAGE_CLEAR);
  HIP_TRY(hipMemsetAsync((char*)state + L.header, 0, L.tile_start - L.header, s));
  tm.end(GIP_STAGE_CLEAR);
  if (cfg->P > 0) {
    tm.begin(GIP_STAGE_PREPROCESS);
    if (kp.sh_mfma) gip_launch_sh_forward_mfma(kp, *in, st.sh_colors, s);      // colours of all views on the matrix cores (same stage bracket)
    gip_launch_preprocess(kp, *in, out->radii, st, s);
    tm.end(GIP_STAGE_PREPROCESS);
    LAUNCH_CHECK();
  }
  tm.begin(GIP_STAGE_SCAN);
  gip_launch_scan(kp, st, s);
  tm.end(GIP_STAGE_SCAN);
  LAUNCH_CHECK();
  if (cfg->P > 0) {
    tm.begin(GIP_STAGE_SCATTER);
    gip_launch_scatter(kp, st, s);
    tm.end(GIP_STAGE_SCATTER);
    LAUNCH_CHECK();
    tm.begin(GIP_STAGE_TILE_SORT);
    gip_launch_tile_sort(kp, st, s);
    tm.end(GIP_STAGE_TILE_SORT);
    LAUNCH_CHECK();
  }
  tm.begin(GIP_STAGE_RENDER_FWD);
  gip_launch_render_forward(kp, in->bg, st, out->color, out->depth, out->alpha, s);
  tm.end(GIP_STAGE_RENDER_FWD);
  LAUNCH_CHECK();
  return tm.finish();
}

extern "C" int gip_raster_forward(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterOutputs* out,
                                  void* state, size_t state_bytes, void* stream) {
  return forward_impl(cfg, in, out, state, state_bytes, stream, nullptr);
}
extern "C" int gip_raster_forward_profiled(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterOutputs* out,
                                           void* state, size_t state_bytes, void* stream, float* times_ms) {
  if (!times_ms) return GIP_ERR_BAD_ARGUMENT;
  return forward_impl(cfg, in, out, state, state_bytes, stream, times_ms);
}

static int backward_impl(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterGradsIn* gin,
                         const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                         const GipRasterGradsOut* gout, void* stream, float* times) {
  if (!valid_config(cfg)) return GIP_ERR_BAD_ARGUMENT;
  int rc = check_inputs(cfg, in);
  if (rc != GIP_OK) return rc;
  if (!gin || !gin->alpha || !gin->color || !gin->depth || !gout || !state || !scratch) return GIP_ERR_BAD_ARGUMENT;
  if (cfg->forward_only) return GIP_ERR_BAD_ARGUMENT;      // that forward kept nothing for a backward
  GipRasterStateLayout L;
  gip_raster_state_layout(cfg, &L);
  if (state_bytes < L.total || scratch_bytes < gip_raster_scratch_bytes(cfg)) return GIP_ERR_BUFFER_TOO_SMALL;
  if (cfg->P == 0) return GIP_OK;
  GipKernelParams kp;
  fill_params(cfg, L, &kp);
  if (!in->shs) kp.sh_mfma = 0;
  GipStatePtrs st = state_ptrs(const_cast<void*>(state), L);
  hipStream_t s = (hipStream_t)stream;
  StageTimer tm(s, times);
  tm.begin(GIP_STAGE_RENDER_BWD);
  gip_launch_render_backward(kp, in->bg, st, *gin, (float*)scratch, s);
  tm.end(GIP_STAGE_RENDER_BWD);
  LAUNCH_CHECK();
  tm.begin(GIP_STAGE_GATHER_BWD);
  gip_launch_gather_backward(kp, *in, st, (const float*)scratch, *gout, s);
  // matrix-core SH path: the gather kernel left dL/dcolour per (view, Gaussian) where the forward's colours were; dL/dshs and
  // the direction part of dL/dmeans3D follow from it (same stage bracket)
  if (kp.sh_mfma && (gout->dL_dshs || gout->dL_dmeans3D)) gip_launch_sh_backward_mfma(kp, *in, st, st.sh_colors, *gout, s);
  tm.end(GIP_STAGE_GATHER_BWD);
  LAUNCH_CHECK();
  return tm.finish();
}

extern "C" int gip_raster_backward(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterGradsIn* gin,
                                   const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                                   const GipRasterGradsOut* gout, void* stream) {
  return backward_impl(cfg, in, gin, state, state_bytes, scratch, scratch_bytes, gout, stream, nullptr);
}
extern "C" int gip_raster_backward_profiled(const GipRasterConfig* cfg, const GipRasterInputs* in, const GipRasterGradsIn* gin,
                                            const void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                                            const GipRasterGradsOut* gout, void* stream, float* times_ms) {
  if (!times_ms) return GIP_ERR_BAD_ARGUMENT;
  return backward_impl(cfg, in, gin, state, state_bytes, scratch, scratch_bytes, gout, stream, times_ms);
}

extern "C" int gip_raster_read_header(const void* state, GipRasterHeader* host_header, void* stream) {
  if (!state || !host_header) return GIP_ERR_BAD_ARGUMENT;
  HIP_TRY(hipMemcpyAsync(host_header, state, sizeof(GipRasterHeader), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return GIP_OK;
}

extern "C" int gip_raster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                       uint8_t* present, void* stream) {
  (void)projmatrix;
  if (P < 0 || !means3D || !viewmatrix || !present) return GIP_ERR_BAD_ARGUMENT;
  if (P == 0) return GIP_OK;
  gip_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? GIP_OK : GIP_ERR_HIP;
}
