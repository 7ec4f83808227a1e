// preprocess.hip — per-Gaussian projection (forward stage 1) for gfx950.
//
// Replaces the fork's preprocessCUDA (SURVEY.md §2.1 "fwd 1"); reference call site
// gaussiansplatting/gaussian_renderer/__init__.py:85-93.  Python mirrors of the math:
// general_utils.py:78-110 (quaternion -> R, L = R S), gaussian_model.py:16-20 (Sigma = L L^T),
// sh_utils.py:57-112 (+0.5 / clamp at gaussian_renderer/__init__.py:77-78), cameras.py:48-50 (matrix layout).
//
// Compiled with -ffp-contract=off: radii, tile rectangles and depth keys are integers derived from
// float32 arithmetic and must be reproducible bit for bit (parity bar of BASELINE.md §2).
//
// One lane per (view, Gaussian).  Loads are SoA-strided ([P,3] / [P,4] float arrays: a wave reads one
// contiguous 768 B / 1 KiB span per array); the 64-byte record is written as four dwordx4 stores.
// The kernel also produces (a) the per-tile histogram (integer atomics, one per touched tile) and
// (b) per-workgroup sums of tiles_touched, which the scan kernel turns into instance offsets.
#include "gip_internal.h"

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.0f) * S - 1.0f) * 0.5f; }

__device__ __forceinline__ float sh_channel(int deg, const float* __restrict__ sh, int ch, float x, float y, float z) {
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
  return res;
}

// Workgroup = PRE_THREADS lanes = PRE_THREADS / GIP_BLOCK of the 256-Gaussian blocks the scan / scatter kernels index by
// (block_sums stays per 256).  The wider workgroup exists for the tile histogram below: four times the Gaussians share
// one LDS table, so neighbouring Gaussians' instances of a tile collapse into one global atomic four times as often.
#define PRE_THREADS 1024
__global__ void __launch_bounds__(PRE_THREADS)
gip_preprocess_kernel(GipKernelParams kp, const float* __restrict__ means3D, const float* __restrict__ shs,
                      const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ cov3D_precomp, const float* __restrict__ viewmatrix,
                      const float* __restrict__ projmatrix, const float* __restrict__ camposs,
                      int32_t* __restrict__ radii, GipRecord* __restrict__ records,
                      uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_count_b,
                      uint32_t* __restrict__ inst_slot, uint32_t* __restrict__ block_sums,
                      GipRasterHeader* __restrict__ header, const float* __restrict__ sh_colors) {
  const int v = blockIdx.y;
  const int idx = blockIdx.x * PRE_THREADS + threadIdx.x;
  const float* view = viewmatrix + 16 * v;   // wave-uniform -> scalar loads
  const float* proj = projmatrix + 16 * v;
  const float* campos = camposs + 3 * v;
  const float tanx = kp.view[v].tanfovx, tany = kp.view[v].tanfovy;
  const float fx = kp.view[v].focal_x, fy = kp.view[v].focal_y;
  const int W = kp.W, H = kp.H;

  GipRecord rec;
  rec.x = rec.y = rec.depth = rec.opacity = 0.f;
  rec.ca = rec.cb = rec.cc = 0.f; rec.tiles = 0;
  rec.r = rec.g = rec.b = 0.f; rec.radius = 0;
  rec.rmin = rec.rmax = rec.clamped = 0; rec.tmask = 0xffffffffu;

  if (idx < kp.P) {
    const float p0 = means3D[3 * idx], p1 = means3D[3 * idx + 1], p2 = means3D[3 * idx + 2];
    const float pvz = view[2] * p0 + view[6] * p1 + view[10] * p2 + view[14];
    if (pvz > GIP_NEAR) {
      const float ph0 = proj[0] * p0 + proj[4] * p1 + proj[8] * p2 + proj[12];
      const float ph1 = proj[1] * p0 + proj[5] * p1 + proj[9] * p2 + proj[13];
      const float ph3 = proj[3] * p0 + proj[7] * p1 + proj[11] * p2 + proj[15];
      const float pw = 1.0f / (ph3 + 0.0000001f);
      const float ppx = ph0 * pw, ppy = ph1 * pw;
      // --- 3D covariance ---
      float c0, c1, c2, c3, c4, c5;
      if (cov3D_precomp) {
        const float* c = cov3D_precomp + 6 * idx;
        c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
      } else {
        const float mod = kp.scale_modifier;
        const float s0 = mod * scales[3 * idx], s1 = mod * scales[3 * idx + 1], s2 = mod * scales[3 * idx + 2];
        const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        const float R0 = 1.f - 2.f * (y * y + z * z), R1 = 2.f * (x * y - r * z), R2 = 2.f * (x * z + r * y);
        const float R3 = 2.f * (x * y + r * z), R4 = 1.f - 2.f * (x * x + z * z), R5 = 2.f * (y * z - r * x);
        const float R6 = 2.f * (x * z - r * y), R7 = 2.f * (y * z + r * x), R8 = 1.f - 2.f * (x * x + y * y);
        const float L0 = R0 * s0, L1 = R1 * s1, L2 = R2 * s2;
        const float L3 = R3 * s0, L4 = R4 * s1, L5 = R5 * s2;
        const float L6 = R6 * s0, L7 = R7 * s1, L8 = R8 * s2;
        c0 = L0 * L0 + L1 * L1 + L2 * L2;
        c1 = L0 * L3 + L1 * L4 + L2 * L5;
        c2 = L0 * L6 + L1 * L7 + L2 * L8;
        c3 = L3 * L3 + L4 * L4 + L5 * L5;
        c4 = L3 * L6 + L4 * L7 + L5 * L8;
        c5 = L6 * L6 + L7 * L7 + L8 * L8;
      }
      // --- EWA projection (cov2D = (J Rv) Sigma (J Rv)^T, +0.3 low-pass) ---
      float t0 = view[0] * p0 + view[4] * p1 + view[8] * p2 + view[12];
      float t1 = view[1] * p0 + view[5] * p1 + view[9] * p2 + view[13];
      const float t2 = pvz;
      const float limx = 1.3f * tanx, limy = 1.3f * tany;
      const float txtz = t0 / t2, tytz = t1 / t2;
      t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
      t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
      const float J00 = fx / t2, J02 = -(fx * t0) / (t2 * t2);
      const float J11 = fy / t2, J12 = -(fy * t1) / (t2 * t2);
      const float M00 = J00 * view[0] + J02 * view[2], M01 = J00 * view[4] + J02 * view[6], M02 = J00 * view[8] + J02 * view[10];
      const float M10 = J11 * view[1] + J12 * view[2], M11 = J11 * view[5] + J12 * view[6], M12 = J11 * view[9] + J12 * view[10];
      const float v00 = c0 * M00 + c1 * M01 + c2 * M02;
      const float v01 = c1 * M00 + c3 * M01 + c4 * M02;
      const float v02 = c2 * M00 + c4 * M01 + c5 * M02;
      const float v10 = c0 * M10 + c1 * M11 + c2 * M12;
      const float v11 = c1 * M10 + c3 * M11 + c4 * M12;
      const float v12 = c2 * M10 + c4 * M11 + c5 * M12;
      const float a = (M00 * v00 + M01 * v01 + M02 * v02) + 0.3f;
      const float b = M00 * v10 + M01 * v11 + M02 * v12;
      const float c = (M10 * v10 + M11 * v11 + M12 * v12) + 0.3f;
      const float det = a * c - b * b;
      if (det != 0.0f) {
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (a + c);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pixx = ndc2pix(ppx, W), pixy = ndc2pix(ppy, H);
        const int rad = (int)my_radius;
        const int gx = kp.tiles_x, gy = kp.tiles_y;
        int rminx = min(gx, max(0, (int)((pixx - rad) / GIP_TILE)));
        int rminy = min(gy, max(0, (int)((pixy - rad) / GIP_TILE)));
        int rmaxx = min(gx, max(0, (int)((pixx + rad + GIP_TILE - 1) / GIP_TILE)));
        int rmaxy = min(gy, max(0, (int)((pixy + rad + GIP_TILE - 1) / GIP_TILE)));
        const int ntiles_ref = (rmaxx - rminx) * (rmaxy - rminy);      // the fork's tiles_touched: decides visibility
        // Instances are only made for the tiles the alpha >= 1/255 region can reach: o exp(-q / 2) >= 1/255 bounds the
        // pixel offset by |dx| <= sqrt(2 ln(255 o) cov_xx), |dy| likewise (conservative: +1 % / +0.05 px, as in the
        // render kernels' block masks).  A pair outside fails the fork's alpha test, so the dropped instances change no
        // output; radii / visibility keep the fork's 3-sigma definition.  (Bench scene: 14 % fewer instances.)
        int ntiles = ntiles_ref;
        if (ntiles_ref != 0 && !kp.exact_lists) {
          const float t2 = 2.0f * logf(255.0f * opacities[idx]) + 0.02f;
          if (t2 <= 0.f) {
            ntiles = 0;
          } else {
            const float hx = sqrtf(t2 * a) * 1.01f + 0.05f, hy = sqrtf(t2 * c) * 1.01f + 0.05f;
            rminx = max(rminx, (int)floorf((pixx - hx) / GIP_TILE));
            rminy = max(rminy, (int)floorf((pixy - hy) / GIP_TILE));
            rmaxx = min(rmaxx, (int)floorf((pixx + hx) / GIP_TILE) + 1);
            rmaxy = min(rmaxy, (int)floorf((pixy + hy) / GIP_TILE) + 1);
            ntiles = max(rmaxx - rminx, 0) * max(rmaxy - rminy, 0);
          }
        }
        if (ntiles_ref != 0) {
          uint32_t clamped = 0;
          float cr, cg, cb;
          if (colors_precomp) {
            cr = colors_precomp[3 * idx]; cg = colors_precomp[3 * idx + 1]; cb = colors_precomp[3 * idx + 2];
          } else if (sh_colors) {
            // matrix-core SH path (sh_mfma.hip): sum_k basis_k sh_k + 0.5 of every view came out of one view-batched launch
            const float4 c4 = reinterpret_cast<const float4*>(sh_colors)[(size_t)v * kp.P + idx];
            cr = c4.x; cg = c4.y; cb = c4.z;
            clamped = (cr < 0.f ? 1u : 0u) | (cg < 0.f ? 2u : 0u) | (cb < 0.f ? 4u : 0u);
            cr = fmaxf(cr, 0.f); cg = fmaxf(cg, 0.f); cb = fmaxf(cb, 0.f);
          } else {
            const float d0 = p0 - campos[0], d1 = p1 - campos[1], d2 = p2 - campos[2];
            const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            const float x = d0 / len, y = d1 / len, z = d2 / len;
            const float* sh = shs + (size_t)idx * kp.M * 3;
            cr = sh_channel(kp.D, sh, 0, x, y, z) + 0.5f;
            cg = sh_channel(kp.D, sh, 1, x, y, z) + 0.5f;
            cb = sh_channel(kp.D, sh, 2, x, y, z) + 0.5f;
            clamped = (cr < 0.f ? 1u : 0u) | (cg < 0.f ? 2u : 0u) | (cb < 0.f ? 4u : 0u);
            cr = fmaxf(cr, 0.f); cg = fmaxf(cg, 0.f); cb = fmaxf(cb, 0.f);
          }
          rec.x = pixx; rec.y = pixy; rec.depth = pvz; rec.opacity = opacities[idx];
          rec.ca = c * det_inv; rec.cb = -b * det_inv; rec.cc = a * det_inv; rec.tiles = (uint32_t)ntiles;
          rec.r = cr; rec.g = cg; rec.b = cb; rec.radius = rad;
          rec.rmin = (uint32_t)rminx | ((uint32_t)rminy << 16);
          rec.rmax = (uint32_t)rmaxx | ((uint32_t)rmaxy << 16);
          rec.clamped = clamped;
        }
      }
    }
    // ... and, for rectangles of up to GIP_MASK_TILES tiles, only the tiles the ellipse itself reaches (the rectangle's
    // corners usually lie outside): the minimum of q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 over the tile's pixel box (0 when
    // the centre is inside, else on an edge, at the clamped 1-D minimiser) against t2, with the box grown by 0.05 px and
    // t2 by 2 %.  (Bench scene: another 12 % fewer instances.)  Done last, from the record alone, so that nothing else is
    // live across the loop (the kernel must stay under 64 VGPRs: two 1024-thread workgroups per CU).
    float4* dst = reinterpret_cast<float4*>(records + (size_t)v * kp.P + idx);
    {
      const float4* src = reinterpret_cast<const float4*>(&rec);
      dst[0] = src[0]; dst[2] = src[2];                        // position / depth / opacity and colour / radius are final
      radii[(size_t)v * kp.P + idx] = rec.radius;
    }
    if (rec.tiles > 1u && rec.tiles <= (uint32_t)GIP_MASK_TILES && !kp.exact_lists) {
      const int rminx = (int)(rec.rmin & 0xffffu), rminy = (int)(rec.rmin >> 16);
      const int rmaxx = (int)(rec.rmax & 0xffffu), rmaxy = (int)(rec.rmax >> 16);
      const float qa = rec.ca, qb = rec.cb, qc = rec.cc, tq = (2.0f * __logf(255.0f * rec.opacity) + 0.02f) * 1.02f;
      // x* = ia y on a horizontal edge, y* = ic x on a vertical one (v_rcp_f32: the position of a minimum needs no more)
      const float ia = -qb * __builtin_amdgcn_rcpf(qa), ic = -qb * __builtin_amdgcn_rcpf(qc), qb2 = 2.f * qb;
      uint32_t m = 0;
      int k = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
      for (int ty = rminy; ty < rmaxy; ty++) {
        const float Y0 = (float)(ty * GIP_TILE) - 0.05f - rec.y, Y1 = Y0 + (float)(GIP_TILE - 1) + 0.1f;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int tx = rminx; tx < rmaxx; tx++, k++) {
          const float X0 = (float)(tx * GIP_TILE) - 0.05f - rec.x, X1 = X0 + (float)(GIP_TILE - 1) + 0.1f;
          float qmin = 0.f;
          if (!(X0 <= 0.f && X1 >= 0.f && Y0 <= 0.f && Y1 >= 0.f)) {
            float y = fminf(fmaxf(ic * X0, Y0), Y1);
            qmin = (qa * X0 + qb2 * y) * X0 + qc * y * y;
            y = fminf(fmaxf(ic * X1, Y0), Y1);
            qmin = fminf(qmin, (qa * X1 + qb2 * y) * X1 + qc * y * y);
            float x = fminf(fmaxf(ia * Y0, X0), X1);
            qmin = fminf(qmin, (qa * x + qb2 * Y0) * x + qc * Y0 * Y0);
            x = fminf(fmaxf(ia * Y1, X0), X1);
            qmin = fminf(qmin, (qa * x + qb2 * Y1) * x + qc * Y1 * Y1);
          }
          if (qmin <= tq) m |= 1u << k;
        }
      }
      rec.tmask = m;
      rec.tiles = (uint32_t)__builtin_popcount(m);
    }
    {
      const float4* src = reinterpret_cast<const float4*>(&rec);
      dst[1] = src[1]; dst[3] = src[3];                        // conic + tiles_touched, rectangle + tile mask
    }
  }

  // ---- per-tile histogram, aggregated per workgroup ----
  // The first GIP_SLOTS instances of a Gaussian remember their bucket slot, so the scatter pass places them without a
  // second atomic; the (rare) rest only count (tile_count_b).  Global returning atomics run at ~10 G/s chip-wide and are
  // this kernel's bound (measured with wall_clock64 stamps: a workgroup's own latency is half the kernel time, the rest
  // is atomic throughput), so the 1024 Gaussians of a workgroup first count their instances per tile in an LDS hash
  // table (open addressing, <= 8192 insertions into 8192 slots), then ONE global atomic per distinct tile reserves that
  // tile's range for the whole workgroup, and slot = range base + rank inside the workgroup.  Neighbouring Gaussians
  // share tiles: 256 per table cut the atomics 3x on the limb-ordered bench scene, 1024 per table 8x (5-15x more when the
  // model is Morton-ordered); it costs a few LDS operations per instance otherwise.
  {
    constexpr int HT = 8192;
    __shared__ uint32_t s_key[HT], s_cnt[HT];
    for (int i = threadIdx.x; i < HT; i += PRE_THREADS) { s_key[i] = 0u; s_cnt[i] = 0u; }
    __syncthreads();
    const int gx = kp.tiles_x;
    const int rminx = (int)(rec.rmin & 0xffffu), rminy = (int)(rec.rmin >> 16);
    const int rmaxx = (int)(rec.rmax & 0xffffu), rmaxy = (int)(rec.rmax >> 16);
    const int ntiles = (int)rec.tiles;
    uint32_t* tc = tile_count + (size_t)v * kp.T;
    uint32_t* tcb = tile_count_b + (size_t)v * kp.T;
    uint32_t slots[GIP_SLOTS];
#pragma unroll
    for (int k = 0; k < GIP_SLOTS; k++) slots[k] = 0;
    if (ntiles > 0) {
      const int area = (rmaxx - rminx) * (rmaxy - rminy);
      int k = 0, kt = 0;                                            // k: instance number, kt: tile of the rectangle
      for (int ty = rminy; ty < rmaxy; ty++)
        for (int tx = rminx; tx < rmaxx; tx++, kt++) {
          if (!gip_rect_has(rec.tmask, area, kt)) continue;
          const int k_cur = k++;
          const uint32_t tile = (uint32_t)(ty * gx + tx);
          if (k_cur < GIP_SLOTS) {
            uint32_t h = (tile * 2654435761u) >> 19;                 // 13 bits
            for (;;) {
              const uint32_t prev = atomicCAS(&s_key[h], 0u, tile + 1u);
              if (prev == 0u || prev == tile + 1u) break;
              h = (h + 1u) & (HT - 1);
            }
            const uint32_t packed = (h << 16) | (atomicAdd(&s_cnt[h], 1u) & 0xffffu);   // rank < 8192
#pragma unroll
            for (int kk = 0; kk < GIP_SLOTS; kk++) if (kk == k_cur) slots[kk] = packed;
          } else {
            // instances beyond the remembered slots only count.  When their tile is already in the table (a neighbour's
            // — or this Gaussian's own — remembered instance put it there) they count in the HIGH half of its word (a
            // workgroup adds at most 1024 to a tile) and ride in that tile's flush; else one global atomic.  They never
            // claim a table slot: the table is sized for the remembered instances alone (8 x 1024 insertions).
            uint32_t h = (tile * 2654435761u) >> 19;
            bool joined = false;
#pragma unroll 1
            for (int probe = 0; probe < 8; probe++) {
              const uint32_t key = reinterpret_cast<volatile uint32_t*>(s_key)[h];
              if (key == tile + 1u) { atomicAdd(&s_cnt[h], 1u << 16); joined = true; break; }
              if (key == 0u) break;
              h = (h + 1u) & (HT - 1);
            }
            if (!joined) atomicAdd(&tcb[tile], 1u);
          }
        }
    }
    __syncthreads();
    {
      // count -> base of this workgroup's range; all of a thread's returning atomics are issued before the first result
      // is consumed
      uint32_t keyv[HT / PRE_THREADS], val[HT / PRE_THREADS];
#pragma unroll
      for (int k = 0; k < HT / PRE_THREADS; k++) {
        keyv[k] = s_key[threadIdx.x + k * PRE_THREADS];
        val[k] = s_cnt[threadIdx.x + k * PRE_THREADS];
      }
#pragma unroll
      for (int k = 0; k < HT / PRE_THREADS; k++)
        if (keyv[k]) {
          if (val[k] >> 16) atomicAdd(&tcb[keyv[k] - 1u], val[k] >> 16);
          val[k] = (val[k] & 0xffffu) ? atomicAdd(&tc[keyv[k] - 1u], val[k] & 0xffffu) : 0u;
        }
#pragma unroll
      for (int k = 0; k < HT / PRE_THREADS; k++)
        if (keyv[k]) s_cnt[threadIdx.x + k * PRE_THREADS] = val[k];
    }
    __syncthreads();
    if (ntiles > 0) {
#pragma unroll
      for (int k = 0; k < GIP_SLOTS; k++)
        if (k < ntiles) slots[k] = s_cnt[slots[k] >> 16] + (slots[k] & 0xffffu);
      uint4* sp = reinterpret_cast<uint4*>(inst_slot + ((size_t)v * kp.P + idx) * GIP_SLOTS);
      sp[0] = make_uint4(slots[0], slots[1], slots[2], slots[3]);
      if (ntiles > 4) sp[1] = make_uint4(slots[4], slots[5], slots[6], slots[7]);
    }
  }

  // per-256-Gaussian-block sums of tiles_touched (feed the instance-offset scan) + visible count
  constexpr int PRE_WAVES = PRE_THREADS / 64, PRE_SUB = PRE_THREADS / GIP_BLOCK;
  __shared__ uint32_t s_sum[PRE_WAVES], s_vis[PRE_WAVES];
  uint32_t t = rec.tiles, vis = rec.radius > 0 ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { t += __shfl_xor(t, d, 64); vis += __shfl_xor(vis, d, 64); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_sum[wave] = t; s_vis[wave] = vis; }
  __syncthreads();
  if (threadIdx.x < PRE_SUB) {
    const int blk = blockIdx.x * PRE_SUB + threadIdx.x;              // the 256-Gaussian block the scan / scatter kernels index by
    const int w0 = threadIdx.x * (GIP_BLOCK / 64);
    if (blk < kp.nblk) block_sums[(size_t)v * kp.nblk + blk] = s_sum[w0] + s_sum[w0 + 1] + s_sum[w0 + 2] + s_sum[w0 + 3];
  }
  if (threadIdx.x == 0) {
    uint32_t nv = 0;
#pragma unroll
    for (int w = 0; w < PRE_WAVES; w++) nv += s_vis[w];
    if (nv) atomicAdd(&header->num_visible, nv);
  }
}

void gip_launch_preprocess(const GipKernelParams& kp, const GipRasterInputs& in, int32_t* radii, GipStatePtrs st, hipStream_t s) {
  dim3 grid((kp.nblk + PRE_THREADS / GIP_BLOCK - 1) / (PRE_THREADS / GIP_BLOCK), kp.V), block(PRE_THREADS);
  hipLaunchKernelGGL(gip_preprocess_kernel, grid, block, 0, s, kp, in.means3D, in.shs, in.colors_precomp, in.opacities,
                     in.scales, in.rotations, in.cov3D_precomp, in.viewmatrix, in.projmatrix, in.campos, radii,
                     st.records, st.tile_count, st.tile_count_b, st.inst_slot, st.block_sums, st.header,
                     kp.sh_mfma ? st.sh_colors : nullptr);
}

// mark_visible: the fork's checkFrustum (view-space z > 0.2).
__global__ void __launch_bounds__(GIP_BLOCK)
gip_mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * GIP_BLOCK + threadIdx.x;
  if (idx >= P) return;
  const float z = view[2] * means3D[3 * idx] + view[6] * means3D[3 * idx + 1] + view[10] * means3D[3 * idx + 2] + view[14];
  present[idx] = z > GIP_NEAR ? 1 : 0;
}
void gip_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
  hipLaunchKernelGGL(gip_mark_visible_kernel, dim3((P + GIP_BLOCK - 1) / GIP_BLOCK), dim3(GIP_BLOCK), 0, s, P, means3D, view, present);
}
