/*
 * raster_oracle.c — CPU restatement of the differentiable 3D-Gaussian-splatting rasterizer that
 * GaussianIP calls through `diff_gaussian_rasterization` (ashawkey fork: colour + depth + alpha).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (gaussianip_amd/csrc) shares no code
 * with this file.
 *
 * PARITY STATUS: "parity unpinned" against the CUDA fork itself.  The rasterizer's source is a
 * third-party dependency that is NOT vendored under /root/reference (README.md:22-24 clones
 * github.com/ashawkey/diff-gaussian-rasterization at unpinned HEAD; gaussiansplatting/.gitmodules:4-6
 * names the graphdeco upstream but the directory is absent), there is no nvcc / NVIDIA GPU here, and
 * the reference ships no tests or golden vectors for it.  This file therefore restates the fork's
 * published algorithm (constants listed in SURVEY.md §2.1 / §8c), anchored on the reference's own
 * call sites:
 *     gaussiansplatting/gaussian_renderer/__init__.py:36-51,85-93   (settings + 4-tuple call)
 *     gs_renderer.py:943-1001                                       (second witness)
 * What IS pinned (tests/test_oracle_*.py): the pre-stages against golden vectors generated from the
 * importable reference modules (sh_utils.eval_sh, general_utils.build_scaling_rotation/strip_symmetric,
 * cameras.Camera, graphics_utils.getProjectionMatrix), and the whole forward/backward against an
 * independent dense float64 PyTorch-autograd formulation of the same image-formation model.
 *
 * Arithmetic: float32, evaluated left to right, compiled with -ffp-contract=off so that the HIP
 * kernels (same flag) can reproduce radii / tile rectangles / sort keys bit for bit.  Per-Gaussian
 * gradient sums are accumulated in double (the CUDA code uses float atomicAdd in nondeterministic
 * order, so no float summation order is "the" reference).
 *
 * Stage map (names of the fork's kernels):
 *   preprocess()              <- preprocessCUDA (forward)
 *   emit_keys()/sort/ranges   <- duplicateWithKeys, cub::DeviceRadixSort::SortPairs, identifyTileRanges
 *   render_forward()          <- renderCUDA (forward)
 *   render_backward()         <- renderCUDA (backward)
 *   preprocess_backward()     <- computeCov2DCUDA + preprocessCUDA (backward)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
  int P, H, W, D, M;
  int tiles_x, tiles_y;
  /* geometry state */
  float* means2D;       /* [P,2] */
  float* depths;        /* [P]   */
  float* cov3D;         /* [P,6] */
  float* rgb;           /* [P,3] */
  float* conic_opacity; /* [P,4] */
  uint32_t* tiles_touched; /* [P] */
  uint32_t* offsets;       /* [P] inclusive scan */
  uint8_t* clamped;        /* [P,3] */
  int32_t* radii;          /* [P] */
  /* binning state */
  uint64_t num_rendered;
  uint64_t* keys;   /* sorted (tile << 32 | depth bits) */
  uint32_t* values; /* sorted gaussian indices */
  uint32_t* ranges; /* [tiles,2] */
  /* image state */
  uint32_t* n_contrib; /* [H*W] */
  float* final_T;      /* [H*W] (not used by backward, which uses 1 - alpha; kept for inspection) */
} oracle_ctx;

oracle_ctx* oracle_create(void) { return (oracle_ctx*)calloc(1, sizeof(oracle_ctx)); }

static void free_state(oracle_ctx* c) {
  free(c->means2D); free(c->depths); free(c->cov3D); free(c->rgb); free(c->conic_opacity);
  free(c->tiles_touched); free(c->offsets); free(c->clamped); free(c->radii);
  free(c->keys); free(c->values); free(c->ranges); free(c->n_contrib); free(c->final_T);
  memset(c, 0, sizeof(*c));
}
void oracle_destroy(oracle_ctx* c) { if (c) { free_state(c); free(c); } }

/* ---- small helpers (column-major 4x4 in memory == the transposed matrices cameras.py stores) ---- */
static inline void xform4x3(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static inline float ndc2pix(float v, int S) { return ((v + 1.0f) * S - 1.0f) * 0.5f; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static void get_rect(const float* p, int max_radius, int gx, int gy, int* rmin, int* rmax) {
  rmin[0] = imin(gx, imax(0, (int)((p[0] - max_radius) / TILE)));
  rmin[1] = imin(gy, imax(0, (int)((p[1] - max_radius) / TILE)));
  rmax[0] = imin(gx, imax(0, (int)((p[0] + max_radius + TILE - 1) / TILE)));
  rmax[1] = imin(gy, imax(0, (int)((p[1] + max_radius + TILE - 1) / TILE)));
}

/* Sigma = (R S)(R S)^T, packed xx,xy,xz,yy,yz,zz; quaternion (w,x,y,z) used as given (the Python side
 * normalises it: gaussian_model.py:29,88-89).  Python mirror: general_utils.py:78-110, gaussian_model.py:16-20. */
static void compute_cov3D(const float* scale, float mod, const float* q, float* cov) {
  float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
  float r = q[0], x = q[1], y = q[2], z = q[3];
  float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
  float L[9];
  for (int i = 0; i < 3; i++) { L[i * 3 + 0] = R[i * 3 + 0] * s0; L[i * 3 + 1] = R[i * 3 + 1] * s1; L[i * 3 + 2] = R[i * 3 + 2] * s2; }
  cov[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
  cov[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
  cov[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
  cov[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
  cov[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
  cov[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

/* Intermediate quantities of the EWA projection, shared by forward and backward. */
typedef struct {
  float t[3];       /* view-space mean, x/y clamped to 1.3 * tanfov * z */
  float xmul, ymul; /* 0 if clamped (gradient gate) */
  float M0[3], M1[3]; /* rows of J * Rview (2x3) */
  float v0[3], v1[3]; /* Sigma * M0^T, Sigma * M1^T */
  float a, b, c;    /* cov2D with +0.3 low-pass on the diagonal */
} ewa_t;

static void compute_cov2D(const float* mean, float fx, float fy, float tanx, float tany, const float* cov3D,
                          const float* view, ewa_t* e) {
  float t[3];
  xform4x3(mean, view, t);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t[0] / t[2], tytz = t[1] / t[2];
  e->xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  e->ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
  t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
  e->t[0] = t[0]; e->t[1] = t[1]; e->t[2] = t[2];
  const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
  const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
  /* Rview[r][c] = view[c*4 + r] */
  for (int c = 0; c < 3; c++) {
    e->M0[c] = J00 * view[c * 4 + 0] + J02 * view[c * 4 + 2];
    e->M1[c] = J11 * view[c * 4 + 1] + J12 * view[c * 4 + 2];
  }
  const float S[9] = {cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]};
  for (int r = 0; r < 3; r++) {
    e->v0[r] = S[r * 3 + 0] * e->M0[0] + S[r * 3 + 1] * e->M0[1] + S[r * 3 + 2] * e->M0[2];
    e->v1[r] = S[r * 3 + 0] * e->M1[0] + S[r * 3 + 1] * e->M1[1] + S[r * 3 + 2] * e->M1[2];
  }
  e->a = (e->M0[0] * e->v0[0] + e->M0[1] * e->v0[1] + e->M0[2] * e->v0[2]) + 0.3f;
  e->b = e->M0[0] * e->v1[0] + e->M0[1] * e->v1[1] + e->M0[2] * e->v1[2];
  e->c = (e->M1[0] * e->v1[0] + e->M1[1] * e->v1[1] + e->M1[2] * e->v1[2]) + 0.3f;
}

/* SH -> RGB (+0.5, clamp at 0); Python mirror sh_utils.py:57-112 + gaussian_renderer/__init__.py:77-78.
 * sh layout [M,3] per Gaussian (gaussian_model.py:97-100). */
static void color_from_sh(int deg, const float* pos, const float* campos, const float* sh, float* out, uint8_t* clamped) {
  float d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  float x = d[0] / len, y = d[1] / len, z = d[2] / len;
  for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
    float res = SH_C0 * SH(0);
    if (deg > 0) {
      res = res - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
              SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
        }
      }
    }
#undef SH
    res += 0.5f;
    clamped[ch] = (res < 0.f);
    out[ch] = fmaxf(res, 0.f);
  }
}

/* ------------------------------- forward ------------------------------- */
static void preprocess(oracle_ctx* c, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                       float scale_modifier, const float* view, const float* proj, const float* campos,
                       float tanx, float tany) {
  const int P = c->P, H = c->H, W = c->W;
  const float fx = W / (2.0f * tanx), fy = H / (2.0f * tany);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    c->radii[i] = 0;
    c->tiles_touched[i] = 0;
    const float* p = means3D + 3 * i;
    float pv[3];
    xform4x3(p, view, pv);
    if (pv[2] <= 0.2f) continue; /* near cull */
    float ph[4];
    xform4x4(p, proj, ph);
    float pw = 1.0f / (ph[3] + 0.0000001f);
    float pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
    float* cov = c->cov3D + 6 * i;
    if (cov3D_precomp) memcpy(cov, cov3D_precomp + 6 * i, 6 * sizeof(float));
    else compute_cov3D(scales + 3 * i, scale_modifier, rotations + 4 * i, cov);
    ewa_t e;
    compute_cov2D(p, fx, fy, tanx, tany, cov, view, &e);
    float det = e.a * e.c - e.b * e.b;
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    float conic[3] = {e.c * det_inv, -e.b * det_inv, e.a * det_inv};
    float mid = 0.5f * (e.a + e.c);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    float pix[2] = {ndc2pix(pp[0], W), ndc2pix(pp[1], H)};
    int rmin[2], rmax[2];
    get_rect(pix, (int)my_radius, c->tiles_x, c->tiles_y, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
    if (colors_precomp) {
      memcpy(c->rgb + 3 * i, colors_precomp + 3 * i, 3 * sizeof(float));
      c->clamped[3 * i] = c->clamped[3 * i + 1] = c->clamped[3 * i + 2] = 0;
    } else {
      color_from_sh(c->D, p, campos, shs + (size_t)i * c->M * 3, c->rgb + 3 * i, c->clamped + 3 * i);
    }
    c->depths[i] = pv[2];
    c->radii[i] = (int32_t)my_radius;
    c->means2D[2 * i] = pix[0];
    c->means2D[2 * i + 1] = pix[1];
    c->conic_opacity[4 * i + 0] = conic[0];
    c->conic_opacity[4 * i + 1] = conic[1];
    c->conic_opacity[4 * i + 2] = conic[2];
    c->conic_opacity[4 * i + 3] = opacities[i];
    c->tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
  }
}

/* Stable LSD radix sort on the low `bits` bits of 64-bit keys (what cub's SortPairs guarantees). */
static void radix_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t n, int bits) {
  uint64_t* k2 = (uint64_t*)malloc(n * sizeof(uint64_t) + 8);
  uint32_t* v2 = (uint32_t*)malloc(n * sizeof(uint32_t) + 8);
  uint64_t *ka = keys, *kb = k2;
  uint32_t *va = vals, *vb = v2;
  for (int shift = 0; shift < bits; shift += 11) {
    size_t hist[2049];
    memset(hist, 0, sizeof(hist));
    for (uint64_t i = 0; i < n; i++) hist[((ka[i] >> shift) & 2047) + 1]++;
    for (int i = 0; i < 2048; i++) hist[i + 1] += hist[i];
    for (uint64_t i = 0; i < n; i++) {
      size_t d = hist[(ka[i] >> shift) & 2047]++;
      kb[d] = ka[i];
      vb[d] = va[i];
    }
    uint64_t* tk = ka; ka = kb; kb = tk;
    uint32_t* tv = va; va = vb; vb = tv;
  }
  if (ka != keys) { memcpy(keys, ka, n * sizeof(uint64_t)); memcpy(vals, va, n * sizeof(uint32_t)); }
  free(k2); free(v2);
}

static void bin_and_sort(oracle_ctx* c) {
  const int P = c->P;
  uint64_t run = 0;
  for (int i = 0; i < P; i++) { run += c->tiles_touched[i]; c->offsets[i] = (uint32_t)run; }
  c->num_rendered = run;
  c->keys = (uint64_t*)malloc((run + 1) * sizeof(uint64_t));
  c->values = (uint32_t*)malloc((run + 1) * sizeof(uint32_t));
  /* duplicateWithKeys: Gaussian-major, y outer, x inner */
  for (int i = 0; i < P; i++) {
    if (c->radii[i] <= 0) continue;
    uint64_t off = (i == 0) ? 0 : c->offsets[i - 1];
    int rmin[2], rmax[2];
    get_rect(c->means2D + 2 * i, c->radii[i], c->tiles_x, c->tiles_y, rmin, rmax);
    uint32_t dbits;
    memcpy(&dbits, c->depths + i, 4);
    for (int y = rmin[1]; y < rmax[1]; y++)
      for (int x = rmin[0]; x < rmax[0]; x++) {
        uint64_t key = (uint64_t)(y * c->tiles_x + x);
        key <<= 32;
        key |= dbits;
        c->keys[off] = key;
        c->values[off] = (uint32_t)i;
        off++;
      }
  }
  int tiles = c->tiles_x * c->tiles_y, bit = 0;
  while ((1 << bit) < tiles + 1 && bit < 31) bit++; /* >= getHigherMsb(tiles) */
  radix_sort_pairs(c->keys, c->values, run, 32 + bit + 1);
  /* identifyTileRanges */
  memset(c->ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t));
  for (uint64_t i = 0; i < run; i++) {
    uint32_t t = (uint32_t)(c->keys[i] >> 32);
    if (i == 0) c->ranges[2 * t] = 0;
    else {
      uint32_t pt = (uint32_t)(c->keys[i - 1] >> 32);
      if (t != pt) { c->ranges[2 * pt + 1] = (uint32_t)i; c->ranges[2 * t] = (uint32_t)i; }
    }
    if (i == run - 1) c->ranges[2 * t + 1] = (uint32_t)run;
  }
}

static void render_forward(oracle_ctx* c, const float* bg, float* out_color, float* out_depth, float* out_alpha) {
  const int H = c->H, W = c->W;
  const int tiles = c->tiles_x * c->tiles_y;
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < tiles; tile++) {
    const int tx = tile % c->tiles_x, ty = tile / c->tiles_x;
    const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
    for (int ly = 0; ly < TILE; ly++)
      for (int lx = 0; lx < TILE; lx++) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px >= W || py >= H) continue;
        const int pix_id = W * py + px;
        const float pixf[2] = {(float)px, (float)py};
        float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, weight = 0.f, Dacc = 0.f;
        uint32_t contributor = 0, last_contributor = 0;
        for (uint32_t k = r0; k < r1; k++) {
          contributor++;
          const uint32_t g = c->values[k];
          const float dx = c->means2D[2 * g] - pixf[0], dy = c->means2D[2 * g + 1] - pixf[1];
          const float* co = c->conic_opacity + 4 * g;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float alpha = fminf(0.99f, co[3] * expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = T * (1 - alpha);
          if (test_T < 0.0001f) break; /* done */
          for (int ch = 0; ch < 3; ch++) C[ch] += c->rgb[3 * g + ch] * alpha * T;
          weight += alpha * T;
          Dacc += c->depths[g] * alpha * T;
          T = test_T;
          last_contributor = contributor;
        }
        c->final_T[pix_id] = T;
        c->n_contrib[pix_id] = last_contributor;
        for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
        out_depth[pix_id] = Dacc;
        out_alpha[pix_id] = weight;
      }
  }
}

int oracle_raster_forward(oracle_ctx* c, int P, int H, int W, int D, int M, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, float scale_modifier,
                          const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
                          float tanfovx, float tanfovy, float* out_color, int32_t* out_radii, float* out_depth,
                          float* out_alpha) {
  if ((shs == NULL) == (colors_precomp == NULL)) return 1;
  if (((scales == NULL) || (rotations == NULL)) == (cov3D_precomp == NULL)) return 1;
  free_state(c);
  c->P = P; c->H = H; c->W = W; c->D = D; c->M = M;
  c->tiles_x = (W + TILE - 1) / TILE;
  c->tiles_y = (H + TILE - 1) / TILE;
  size_t Pn = (size_t)(P > 0 ? P : 1);
  c->means2D = (float*)calloc(Pn * 2, sizeof(float));
  c->depths = (float*)calloc(Pn, sizeof(float));
  c->cov3D = (float*)calloc(Pn * 6, sizeof(float));
  c->rgb = (float*)calloc(Pn * 3, sizeof(float));
  c->conic_opacity = (float*)calloc(Pn * 4, sizeof(float));
  c->tiles_touched = (uint32_t*)calloc(Pn, sizeof(uint32_t));
  c->offsets = (uint32_t*)calloc(Pn, sizeof(uint32_t));
  c->clamped = (uint8_t*)calloc(Pn * 3, 1);
  c->radii = (int32_t*)calloc(Pn, sizeof(int32_t));
  c->ranges = (uint32_t*)calloc((size_t)c->tiles_x * c->tiles_y * 2 + 2, sizeof(uint32_t));
  c->n_contrib = (uint32_t*)calloc((size_t)H * W + 1, sizeof(uint32_t));
  c->final_T = (float*)calloc((size_t)H * W + 1, sizeof(float));
  preprocess(c, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_modifier, viewmatrix,
             projmatrix, campos, tanfovx, tanfovy);
  bin_and_sort(c);
  render_forward(c, bg, out_color, out_depth, out_alpha);
  memcpy(out_radii, c->radii, (size_t)P * sizeof(int32_t));
  return 0;
}

/* accessors for the tile / index buffers */
uint64_t oracle_num_rendered(const oracle_ctx* c) { return c->num_rendered; }
void oracle_copy_binning(const oracle_ctx* c, uint64_t* keys, uint32_t* values, uint32_t* ranges,
                         uint32_t* tiles_touched, uint32_t* n_contrib) {
  if (keys) memcpy(keys, c->keys, c->num_rendered * sizeof(uint64_t));
  if (values) memcpy(values, c->values, c->num_rendered * sizeof(uint32_t));
  if (ranges) memcpy(ranges, c->ranges, (size_t)c->tiles_x * c->tiles_y * 2 * sizeof(uint32_t));
  if (tiles_touched) memcpy(tiles_touched, c->tiles_touched, (size_t)c->P * sizeof(uint32_t));
  if (n_contrib) memcpy(n_contrib, c->n_contrib, (size_t)c->H * c->W * sizeof(uint32_t));
}
void oracle_copy_geom(const oracle_ctx* c, float* means2D, float* depths, float* cov3D, float* rgb,
                      float* conic_opacity, uint8_t* clamped) {
  size_t P = (size_t)c->P;
  if (means2D) memcpy(means2D, c->means2D, P * 2 * sizeof(float));
  if (depths) memcpy(depths, c->depths, P * sizeof(float));
  if (cov3D) memcpy(cov3D, c->cov3D, P * 6 * sizeof(float));
  if (rgb) memcpy(rgb, c->rgb, P * 3 * sizeof(float));
  if (conic_opacity) memcpy(conic_opacity, c->conic_opacity, P * 4 * sizeof(float));
  if (clamped) memcpy(clamped, c->clamped, P * 3);
}

/* Knife-edge margin of a pixel: the smallest relative distance of any of the forward walk's three threshold tests
 * from flipping, min over the entries the walk visits of |alpha - 1/255| / (1/255), |T (1 - alpha) - 1e-4| / 1e-4 and
 * |power| (the power > 0 rejection).  Two correct implementations whose exp differs in the last ulp (libm expf here,
 * v_exp_f32 on the GPU, __expf / expf in CUDA) can only disagree by more than rounding at pixels where this margin is
 * of the order of 1e-6; the parity tests use it to PROVE that an out-of-tolerance pixel is such a pixel instead of
 * allowing a blanket fraction of mismatches. */
void oracle_pixel_margins(const oracle_ctx* c, const int32_t* pix_ids, int n, float* margins) {
  for (int q = 0; q < n; q++) {
    const int pix_id = pix_ids[q], px = pix_id % c->W, py = pix_id / c->W;
    const int tile = (py / TILE) * c->tiles_x + px / TILE;
    const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
    float T = 1.0f, m = 1e30f;
    for (uint32_t k = r0; k < r1; k++) {
      const uint32_t g = c->values[k];
      const float dx = c->means2D[2 * g] - (float)px, dy = c->means2D[2 * g + 1] - (float)py;
      const float* co = c->conic_opacity + 4 * g;
      const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
      m = fminf(m, fabsf(power));
      if (power > 0.0f) continue;
      const float alpha = fminf(0.99f, co[3] * expf(power));
      m = fminf(m, fabsf(alpha - 1.0f / 255.0f) * 255.0f);
      if (alpha < 1.0f / 255.0f) continue;
      const float test_T = T * (1 - alpha);
      m = fminf(m, fabsf(test_T - 0.0001f) * 10000.0f);
      if (test_T < 0.0001f) break;
      T = test_T;
    }
    margins[q] = m;
  }
}

/* flags[g] = 1 for every Gaussian g that is the SUBJECT of a knife-edge threshold test at some pixel (its alpha within
 * `thresh` (relative) of 1/255, its power within `thresh` of 0, or the transmittance test T (1 - alpha) < 1e-4 decided
 * within `thresh` at its entry): the set of gradient rows that a legitimately different decision changes by one whole
 * pixel contribution.  Entries behind a flipped one see T change by at most 0.4 % at that single pixel (alpha flips)
 * or contribute with T < 1e-4 (termination flips) and are not flagged.  Returns the number of knife-edge pixels. */
int oracle_knife_edge_gaussians2(const oracle_ctx* c, float thresh, uint8_t* flags, uint8_t* downstream, uint8_t* sharing);
int oracle_knife_edge_gaussians(const oracle_ctx* c, float thresh, uint8_t* flags, uint8_t* downstream) {
  return oracle_knife_edge_gaussians2(c, thresh, flags, downstream, NULL);
}

/* sharing[g] (optional) = 1 for every Gaussian that CONTRIBUTES (alpha >= 1/255, before the walk ends) at a pixel that
 * has a knife-edge subject — in front of it or behind it.  Behind: its transmittance changes by (1 - alpha_subject) if
 * the subject flips.  In FRONT: the reverse-order backward hands every earlier entry the colour accumulated behind it
 * (accum_rec), which gains or loses the subject's alpha-weighted term — the same 0.4 % of one pixel's contribution.
 * Round 3 (tools/diag/view_outlier.py): the one entry of the 4-view headline test that sat at 0.92 x the element-wise
 * bar is such a row — in front of the subject of pixel (489, 730) of view 1, whose alpha test is 4.7e-7 from 1/255. */
int oracle_knife_edge_gaussians2(const oracle_ctx* c, float thresh, uint8_t* flags, uint8_t* downstream, uint8_t* sharing) {
  /* downstream[g] (optional) = 1 for every Gaussian that is blended BEHIND a knife-edge subject at some pixel: if the
   * subject's decision flips, the transmittance of everything behind it at that pixel changes by the factor
   * (1 - alpha_subject) (0.4 % for an alpha >= 1/255 flip) — a second-order effect that is visible only on gradient
   * entries that are small sums of cancelling per-view terms. */
  const int H = c->H, W = c->W;
  int count = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : count)
  for (int pix_id = 0; pix_id < H * W; pix_id++) {
    const int px = pix_id % W, py = pix_id / W;
    const int tile = (py / TILE) * c->tiles_x + px / TILE;
    const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
    float T = 1.0f;
    int hit = 0;
    for (uint32_t k = r0; k < r1; k++) {
      const uint32_t g = c->values[k];
      const float dx = c->means2D[2 * g] - (float)px, dy = c->means2D[2 * g + 1] - (float)py;
      const float* co = c->conic_opacity + 4 * g;
      const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
      int subject = 0;
      if (fabsf(power) < thresh) subject = 1;
      if (power <= 0.0f) {
        const float alpha = fminf(0.99f, co[3] * expf(power));
        if (fabsf(alpha - 1.0f / 255.0f) * 255.0f < thresh) subject = 1;
        if (alpha >= 1.0f / 255.0f) {
          const float test_T = T * (1 - alpha);
          if (fabsf(test_T - 0.0001f) * 10000.0f < thresh) subject = 1;
          if (hit && downstream) downstream[g] = 1;
          if (subject) { flags[g] = 1; hit = 1; } /* benign race: every writer stores 1 */
          if (test_T < 0.0001f) break;
          T = test_T;
          continue;
        }
      }
      if (subject) { flags[g] = 1; hit = 1; }
    }
    if (hit && sharing) {                      /* second walk of a knife-edge pixel: everything that blends there */
      T = 1.0f;
      for (uint32_t k = r0; k < r1; k++) {
        const uint32_t g = c->values[k];
        const float dx = c->means2D[2 * g] - (float)px, dy = c->means2D[2 * g + 1] - (float)py;
        const float* co = c->conic_opacity + 4 * g;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, co[3] * expf(power));
        if (alpha < 1.0f / 255.0f) continue;
        const float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) break;
        sharing[g] = 1;
        T = test_T;
      }
    }
    count += hit;
  }
  return count;
}

/* ------------------------------- backward ------------------------------- */
/* per-Gaussian accumulators: 0,1 mean2D.xy  2,3,4 conic (x, y, w slots of the fork's float4)  5 opacity
 * 6,7,8 colour  9 depth */
#define NACC 10

static void render_backward(const oracle_ctx* c, const float* bg, const float* alphas, const float* dL_dpixels,
                            const float* dL_ddepths, const float* dL_dalphas, double* acc /* [P,NACC] */) {
  const int H = c->H, W = c->W;
  const int tiles = c->tiles_x * c->tiles_y;
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  const int exact_final_T = getenv("ORACLE_EXACT_FINAL_T") != NULL;
  /* One row of NACC sums per (tile, Gaussian) instance, filled by the tile's own thread; the rows are then added to
   * the per-Gaussian accumulators in list order by one thread.  The result does not depend on the thread count and
   * the scratch is R x NACC doubles (it used to be nthreads x P x NACC, 2 GB at 256 threads). */
  double* rows = (double*)calloc((size_t)(c->num_rendered ? c->num_rendered : 1) * NACC, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < tiles; tile++) {
    const int tx = tile % c->tiles_x, ty = tile / c->tiles_x;
    const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
    for (int ly = 0; ly < TILE; ly++)
      for (int lx = 0; lx < TILE; lx++) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px >= W || py >= H) continue;
        const int pix_id = W * py + px;
        const float pixf[2] = {(float)px, (float)py};
        const float T_final = exact_final_T ? c->final_T[pix_id] : 1.f - alphas[pix_id];
        float T = T_final;
        const uint32_t last_contributor = c->n_contrib[pix_id];
        float accum_rec[3] = {0, 0, 0}, accum_depth_rec = 0.f, accum_alpha_rec = 0.f;
        float dL_dpixel[3];
        for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpixels ? dL_dpixels[(size_t)ch * H * W + pix_id] : 0.f;
        const float dL_dpixel_depth = dL_ddepths ? dL_ddepths[pix_id] : 0.f;
        const float dL_dpixel_alpha = dL_dalphas ? dL_dalphas[pix_id] : 0.f;
        float last_alpha = 0.f, last_color[3] = {0, 0, 0}, last_depth = 0.f;
        for (uint32_t k = r1; k-- > r0;) {
          const uint32_t contributor = k - r0; /* 0-based position in the tile list */
          if (contributor >= last_contributor) continue;
          const uint32_t g = c->values[k];
          double* A = rows + (size_t)k * NACC;
          const float dx = c->means2D[2 * g] - pixf[0], dy = c->means2D[2 * g + 1] - pixf[1];
          const float* co = c->conic_opacity + 4 * g;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float G = expf(power);
          const float alpha = fminf(0.99f, co[3] * G);
          if (alpha < 1.0f / 255.0f) continue;
          T = T / (1.f - alpha);
          const float dchannel_dcolor = alpha * T;
          float dL_dalpha = 0.0f;
          for (int ch = 0; ch < 3; ch++) {
            const float col = c->rgb[3 * g + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = col;
            dL_dalpha += (col - accum_rec[ch]) * dL_dpixel[ch];
            A[6 + ch] += (double)(dchannel_dcolor * dL_dpixel[ch]);
          }
          const float c_d = c->depths[g];
          accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
          last_depth = c_d;
          dL_dalpha += (c_d - accum_depth_rec) * dL_dpixel_depth;
          A[9] += (double)(dchannel_dcolor * dL_dpixel_depth);
          accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
          dL_dalpha += (1.f - accum_alpha_rec) * dL_dpixel_alpha;
          dL_dalpha *= T;
          last_alpha = alpha;
          float bg_dot = 0.f;
          for (int ch = 0; ch < 3; ch++) bg_dot += bg[ch] * dL_dpixel[ch];
          dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
          const float dL_dG = co[3] * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * co[0] - gdy * co[1];
          const float dG_ddely = -gdy * co[2] - gdx * co[1];
          A[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
          A[1] += (double)(dL_dG * dG_ddely * ddely_dy);
          A[2] += (double)(-0.5f * gdx * dx * dL_dG);
          A[3] += (double)(-0.5f * gdx * dy * dL_dG);
          A[4] += (double)(-0.5f * gdy * dy * dL_dG);
          A[5] += (double)(G * dL_dalpha);
        }
      }
  }
  for (uint64_t k = 0; k < c->num_rendered; k++) {
    double* dst = acc + (size_t)c->values[k] * NACC;
    const double* src = rows + (size_t)k * NACC;
    for (int j = 0; j < NACC; j++) dst[j] += src[j];
  }
  free(rows);
}


/* SH colour backward of one Gaussian (preprocessCUDA backward -> computeColorFromSH backward): gcol = dL/d(rgb after
 * the +0.5 / clamp), already zeroed on clamped channels by the caller's `clamped` flags.  Writes dL/dsh [M,3] and ADDS the
 * view-direction part to dmean.  Pinned against autograd through the reference's own Python statement of the same
 * forward (gaussian_renderer/__init__.py:74-78 + sh_utils.py:57-112) by tests/test_golden_host.py. */
static void sh_backward(int D, int M, const float* m, const float* campos, const float* sh, const uint8_t* clamped,
                        const float* gcol_in, float* dsh, float* dmean) {
  float gcol[3] = {gcol_in[0], gcol_in[1], gcol_in[2]};
  (void)M;
  float dir0[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
  const float len = sqrtf(dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2]);
  const float x = dir0[0] / len, y = dir0[1] / len, z = dir0[2] / len;
  for (int ch = 0; ch < 3; ch++) gcol[ch] *= clamped[ch] ? 0.f : 1.f;
  float ddir[3] = {0, 0, 0};
  for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
#define DSH(k, v) do { if (dsh) dsh[(k) * 3 + ch] = (v) * gcol[ch]; } while (0)
    float dx_ = 0, dy_ = 0, dz_ = 0;
    DSH(0, SH_C0);
    if (D > 0) {
      DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
      dx_ = -SH_C1 * SH(3); dy_ = -SH_C1 * SH(1); dz_ = SH_C1 * SH(2);
      if (D > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        DSH(4, SH_C2[0] * xy); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
        DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
        dx_ += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
        dy_ += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
        dz_ += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
        if (D > 2) {
          DSH(9, SH_C3[0] * y * (3.f * xx - yy)); DSH(10, SH_C3[1] * xy * z);
          DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy)); DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
          DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, SH_C3[5] * z * (xx - yy));
          DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
          dx_ += SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                 SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                 SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy);
          dy_ += SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz + SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                 SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                 SH_C3[6] * SH(15) * -3.f * 2.f * xy;
          dz_ += SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz + SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                 SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy);
        }
      }
    }
#undef SH
#undef DSH
    ddir[0] += dx_ * gcol[ch]; ddir[1] += dy_ * gcol[ch]; ddir[2] += dz_ * gcol[ch];
  }
  /* through the normalisation dir / |dir| */
  const float sum2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  dmean[0] += ((sum2 - dir0[0] * dir0[0]) * ddir[0] - dir0[1] * dir0[0] * ddir[1] - dir0[2] * dir0[0] * ddir[2]) * invsum32;
  dmean[1] += (-dir0[0] * dir0[1] * ddir[0] + (sum2 - dir0[1] * dir0[1]) * ddir[1] - dir0[2] * dir0[1] * ddir[2]) * invsum32;
  dmean[2] += (-dir0[0] * dir0[2] * ddir[0] - dir0[1] * dir0[2] * ddir[1] + (sum2 - dir0[2] * dir0[2]) * ddir[2]) * invsum32;
}

/* computeCov3D backward of one Gaussian: dcov = dL/d(packed xx,xy,xz,yy,yz,zz) -> dL/dscale, dL/dquaternion.
 * Reference quirks reproduced: the scale gradient is taken w.r.t. scale_modifier*scale (not multiplied by the modifier)
 * and the quaternion is differentiated as given (no normalisation inside; GaussianModel.get_rotation normalises
 * upstream).  Pinned against autograd through general_utils.build_scaling_rotation / strip_symmetric by
 * tests/test_golden_host.py (which applies exactly those two relations). */
static void cov3D_backward(const float* scale, float scale_modifier, const float* q, const float* dcov, float* ds_out,
                           float* dq_out) {
  const float s[3] = {scale_modifier * scale[0], scale_modifier * scale[1], scale_modifier * scale[2]};
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                      2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                      2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
  /* full symmetric dSigma (off-diagonals halved), dL/dL = 2 dSigma L with L = R diag(s) */
  const float dS[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                       0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
  float dLm[9]; /* dL/dL[i][k] */
  for (int ii = 0; ii < 3; ii++)
    for (int k = 0; k < 3; k++)
      dLm[ii * 3 + k] = 2.0f * (dS[ii * 3 + 0] * R[0 * 3 + k] * s[k] + dS[ii * 3 + 1] * R[1 * 3 + k] * s[k] + dS[ii * 3 + 2] * R[2 * 3 + k] * s[k]);
  /* NOTE (reference quirk, reproduced): the fork's dL_dscale is the gradient w.r.t. mod*scale, it is
   * not multiplied by scale_modifier.  Identical at scale_modifier = 1 (the only differentiated case). */
  if (ds_out)
    for (int k = 0; k < 3; k++)
      ds_out[k] = R[0 * 3 + k] * dLm[0 * 3 + k] + R[1 * 3 + k] * dLm[1 * 3 + k] + R[2 * 3 + k] * dLm[2 * 3 + k];
  if (dq_out) {
    float dR[9];
    for (int ii = 0; ii < 3; ii++)
      for (int k = 0; k < 3; k++) dR[ii * 3 + k] = dLm[ii * 3 + k] * s[k];
#define DR(i_, j_) dR[(i_) * 3 + (j_)]
    dq_out[0] = 2 * z * (DR(1, 0) - DR(0, 1)) + 2 * y * (DR(0, 2) - DR(2, 0)) + 2 * x * (DR(2, 1) - DR(1, 2));
    dq_out[1] = 2 * y * (DR(0, 1) + DR(1, 0)) + 2 * z * (DR(0, 2) + DR(2, 0)) + 2 * r * (DR(2, 1) - DR(1, 2)) - 4 * x * (DR(2, 2) + DR(1, 1));
    dq_out[2] = 2 * x * (DR(0, 1) + DR(1, 0)) + 2 * r * (DR(0, 2) - DR(2, 0)) + 2 * z * (DR(1, 2) + DR(2, 1)) - 4 * y * (DR(2, 2) + DR(0, 0));
    dq_out[3] = 2 * r * (DR(1, 0) - DR(0, 1)) + 2 * x * (DR(0, 2) + DR(2, 0)) + 2 * y * (DR(1, 2) + DR(2, 1)) - 4 * z * (DR(1, 1) + DR(0, 0));
#undef DR
  }
}

/* entry points for the reference-pinned checks of the two per-Gaussian backward stages */
void oracle_sh_backward(int P, int D, int M, const float* means3D, const float* campos, const float* shs,
                        const uint8_t* clamped, const float* gcol, float* dL_dshs, float* dL_dmeans3D) {
  for (int i = 0; i < P; i++) {
    float dmean[3] = {0, 0, 0};
    float g[3];
    for (int ch = 0; ch < 3; ch++) g[ch] = gcol[3 * i + ch] * (clamped[3 * i + ch] ? 0.f : 1.f);
    sh_backward(D, M, means3D + 3 * i, campos, shs + (size_t)i * M * 3, clamped + 3 * i, g, dL_dshs + (size_t)i * M * 3, dmean);
    for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = dmean[k];
  }
}
void oracle_cov3D_backward(int P, const float* scales, float scale_modifier, const float* rotations, const float* dcov,
                           float* dL_dscales, float* dL_drots) {
  for (int i = 0; i < P; i++)
    cov3D_backward(scales + 3 * i, scale_modifier, rotations + 4 * i, dcov + 6 * i, dL_dscales + 3 * i, dL_drots + 4 * i);
}

static void preprocess_backward(const oracle_ctx* c, const float* means3D, const float* shs, const float* colors_precomp,
                                const float* scales, const float* rotations, const float* cov3D_precomp,
                                float scale_modifier, const float* view, const float* proj, const float* campos,
                                float tanx, float tany, const double* acc, float* dL_dmeans3D, float* dL_dmeans2D,
                                float* dL_dshs, float* dL_dcolors, float* dL_dopacity, float* dL_dscales,
                                float* dL_drots, float* dL_dcov3D) {
  const int P = c->P, H = c->H, W = c->W, M = c->M, D = c->D;
  const float fx = W / (2.0f * tanx), fy = H / (2.0f * tany);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    float dmean[3] = {0, 0, 0};
    if (dL_dmeans2D) { dL_dmeans2D[3 * i] = 0; dL_dmeans2D[3 * i + 1] = 0; dL_dmeans2D[3 * i + 2] = 0; }
    if (dL_dopacity) dL_dopacity[i] = 0;
    if (dL_dcolors) { dL_dcolors[3 * i] = dL_dcolors[3 * i + 1] = dL_dcolors[3 * i + 2] = 0; }
    if (dL_dshs) memset(dL_dshs + (size_t)i * M * 3, 0, (size_t)M * 3 * sizeof(float));
    if (dL_dscales) { dL_dscales[3 * i] = dL_dscales[3 * i + 1] = dL_dscales[3 * i + 2] = 0; }
    if (dL_drots) { dL_drots[4 * i] = dL_drots[4 * i + 1] = dL_drots[4 * i + 2] = dL_drots[4 * i + 3] = 0; }
    if (dL_dcov3D) memset(dL_dcov3D + 6 * i, 0, 6 * sizeof(float));
    if (dL_dmeans3D) { dL_dmeans3D[3 * i] = dL_dmeans3D[3 * i + 1] = dL_dmeans3D[3 * i + 2] = 0; }
    if (!(c->radii[i] > 0)) continue;
    const double* a = acc + (size_t)i * NACC;
    const float g2x = (float)a[0], g2y = (float)a[1];
    const float gcx = (float)a[2], gcy = (float)a[3], gcw = (float)a[4];
    const float* m = means3D + 3 * i;
    const float* cov3D = c->cov3D + 6 * i;
    if (dL_dmeans2D) { dL_dmeans2D[3 * i] = g2x; dL_dmeans2D[3 * i + 1] = g2y; }
    if (dL_dopacity) dL_dopacity[i] = (float)a[5];

    /* ---- computeCov2DCUDA: conic -> cov2D -> (cov3D, view-space mean) ---- */
    ewa_t e;
    compute_cov2D(m, fx, fy, tanx, tany, cov3D, view, &e);
    const float ca = e.a, cb = e.b, cc = e.c;
    const float denom = ca * cc - cb * cb;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6] = {0, 0, 0, 0, 0, 0};
    if (denom2inv != 0) {
      dL_da = denom2inv * (-cc * cc * gcx + 2 * cb * cc * gcy + (denom - ca * cc) * gcw);
      dL_dc = denom2inv * (-ca * ca * gcw + 2 * ca * cb * gcy + (denom - ca * cc) * gcx);
      dL_db = denom2inv * 2 * (cb * cc * gcx - (denom + 2 * cb * cb) * gcy + ca * cb * gcw);
      const float* M0 = e.M0; const float* M1 = e.M1;
      dcov[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
      dcov[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
      dcov[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
      dcov[1] = 2 * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + 2 * M1[0] * M1[1] * dL_dc;
      dcov[2] = 2 * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + 2 * M1[0] * M1[2] * dL_dc;
      dcov[4] = 2 * M0[2] * M0[1] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + 2 * M1[1] * M1[2] * dL_dc;
    }
    /* dL/dM rows, then J, then t */
    float dM0[3], dM1[3];
    for (int k = 0; k < 3; k++) {
      dM0[k] = 2 * e.v0[k] * dL_da + e.v1[k] * dL_db;
      dM1[k] = 2 * e.v1[k] * dL_dc + e.v0[k] * dL_db;
    }
    const float dJ00 = view[0] * dM0[0] + view[4] * dM0[1] + view[8] * dM0[2];
    const float dJ02 = view[2] * dM0[0] + view[6] * dM0[1] + view[10] * dM0[2];
    const float dJ11 = view[1] * dM1[0] + view[5] * dM1[1] + view[9] * dM1[2];
    const float dJ12 = view[2] * dM1[0] + view[6] * dM1[1] + view[10] * dM1[2];
    const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = e.xmul * -fx * tz2 * dJ02;
    const float dty = e.ymul * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * e.t[0]) * tz3 * dJ02 + (2 * fy * e.t[1]) * tz3 * dJ12;
    dmean[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
    dmean[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
    dmean[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;

    /* ---- preprocessCUDA (backward): screen-space mean -> 3D mean ---- */
    float mh[4];
    xform4x4(m, proj, mh);
    const float m_w = 1.0f / (mh[3] + 0.0000001f);
    const float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
    dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    /* depth -> 3D mean (the fork's depth output is view-space z) */
    const float gd = (float)a[9];
    const float mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
    dmean[0] += (view[2] - view[3] * mul3) * gd;
    dmean[1] += (view[6] - view[7] * mul3) * gd;
    dmean[2] += (view[10] - view[11] * mul3) * gd;

    /* ---- colour: SH backward, incl. view-direction dependence ---- */
    float gcol[3] = {(float)a[6], (float)a[7], (float)a[8]};
    if (colors_precomp) {
      if (dL_dcolors) { dL_dcolors[3 * i] = gcol[0]; dL_dcolors[3 * i + 1] = gcol[1]; dL_dcolors[3 * i + 2] = gcol[2]; }
    } else if (shs) {
      sh_backward(D, M, m, campos, shs + (size_t)i * M * 3, c->clamped + 3 * i, gcol,
                  dL_dshs ? dL_dshs + (size_t)i * M * 3 : NULL, dmean);
    }
    if (dL_dmeans3D) { dL_dmeans3D[3 * i] = dmean[0]; dL_dmeans3D[3 * i + 1] = dmean[1]; dL_dmeans3D[3 * i + 2] = dmean[2]; }

    /* ---- cov3D -> scale / rotation (computeCov3D backward) ---- */
    if (cov3D_precomp) {
      if (dL_dcov3D) memcpy(dL_dcov3D + 6 * i, dcov, 6 * sizeof(float));
    } else if (scales) {
      cov3D_backward(scales + 3 * i, scale_modifier, rotations + 4 * i, dcov, dL_dscales ? dL_dscales + 3 * i : NULL,
                     dL_drots ? dL_drots + 4 * i : NULL);
    }
  }
}

int oracle_raster_backward(oracle_ctx* c, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* scales, const float* rotations, const float* cov3D_precomp,
                           float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* campos,
                           const float* bg, float tanfovx, float tanfovy, const float* alphas, const float* dL_dcolor,
                           const float* dL_ddepth, const float* dL_dalpha, float* dL_dmeans3D, float* dL_dmeans2D,
                           float* dL_dshs, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drots,
                           float* dL_dcov3D, double* acc_out /* optional [P,10] raw sums */) {
  if (!c || !c->means2D) return 1;
  double* acc = (double*)calloc((size_t)(c->P > 0 ? c->P : 1) * NACC, sizeof(double));
  render_backward(c, bg, alphas, dL_dcolor, dL_ddepth, dL_dalpha, acc);
  preprocess_backward(c, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, scale_modifier, viewmatrix,
                      projmatrix, campos, tanfovx, tanfovy, acc, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors,
                      dL_dopacity, dL_dscales, dL_drots, dL_dcov3D);
  if (acc_out) memcpy(acc_out, acc, (size_t)c->P * NACC * sizeof(double));
  free(acc);
  return 0;
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
