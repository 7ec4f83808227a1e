// gather_backward.hip — deterministic per-Gaussian gather of the partial rows fused with the
// cov2D / projection / depth / SH / cov3D backward (gfx950).
//
// Replaces the fork's computeCov2DCUDA + backward preprocessCUDA (SURVEY.md §2.1 "bwd 2-3") and the
// zero-initialised per-Gaussian atomic accumulators they read.  One lane per Gaussian; for every view
// the lane sums its contiguous run of 64-byte rows (written by render_backward.hip) in a fixed order,
// pushes the sums through the analytic backward, and accumulates the 3-D gradients over the views in
// registers, so each output element is written exactly once (no pre-zeroing, no atomics).
#include "gip_internal.h"

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__device__ static const float GSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float GSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};

template <int MAXM>
__global__ void __launch_bounds__(GIP_BLOCK)
gip_gather_backward_kernel(GipKernelParams kp, const float* __restrict__ means3D, const float* __restrict__ shs,
                           const float* __restrict__ colors_precomp, const float* __restrict__ scales,
                           const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                           const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
                           const float* __restrict__ camposs, const GipRecord* __restrict__ records,
                           const uint32_t* __restrict__ inst_offset, const float* __restrict__ partial,
                           GipRasterGradsOut gout, const GipRasterHeader* __restrict__ header, float* __restrict__ sh_gcol) {
  // sh_gcol != NULL (matrix-core SH path, sh_mfma.hip): this kernel stops at dL/dcolour — it leaves (dL/dr, dL/dg, dL/db, 0) of every
  // (view, Gaussian), zeroed where the forward clamped the channel or the view does not see the Gaussian, in sh_gcol [V,P,4];
  // dL/dshs and the direction part of dL/dmeans3D are gip_sh_backward_mfma_kernel's (launched behind this one)
  const int idx = blockIdx.x * GIP_BLOCK + threadIdx.x;
  if (idx >= kp.P) return;
  if (header->overflow) {
    // overflowed forward (tile lists truncated at the capacity): the step degenerates to a ZERO-gradient step, the same
    // on every rank of a multi-GPU job; the host learns it from the header one step late, raises the capacity and
    // reports it (rasterizer.py).  Every output is still written exactly once.
    if (gout.dL_dmeans3D) { gout.dL_dmeans3D[3 * idx] = 0.f; gout.dL_dmeans3D[3 * idx + 1] = 0.f; gout.dL_dmeans3D[3 * idx + 2] = 0.f; }
    if (gout.dL_dmeans2D)
      for (int v = 0; v < kp.V; v++) {
        float* d2 = gout.dL_dmeans2D + ((size_t)v * kp.P + idx) * 3;
        d2[0] = 0.f; d2[1] = 0.f; d2[2] = 0.f;
      }
    if (gout.dL_dopacities) gout.dL_dopacities[idx] = 0.f;
    if (gout.dL_dcolors_precomp) { gout.dL_dcolors_precomp[3 * idx] = 0.f; gout.dL_dcolors_precomp[3 * idx + 1] = 0.f; gout.dL_dcolors_precomp[3 * idx + 2] = 0.f; }
    if (gout.dL_dshs) for (int k = 0; k < kp.M * 3; k++) gout.dL_dshs[(size_t)idx * kp.M * 3 + k] = 0.f;
    if (gout.dL_dcov3D_precomp) for (int k = 0; k < 6; k++) gout.dL_dcov3D_precomp[6 * idx + k] = 0.f;
    if (gout.dL_dscales) { gout.dL_dscales[3 * idx] = 0.f; gout.dL_dscales[3 * idx + 1] = 0.f; gout.dL_dscales[3 * idx + 2] = 0.f; }
    if (gout.dL_drotations) reinterpret_cast<float4*>(gout.dL_drotations)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float m0 = means3D[3 * idx], m1 = means3D[3 * idx + 1], m2 = means3D[3 * idx + 2];

  // 3-D covariance (recomputed rather than stored: 6 floats of state per view saved)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, qr = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
  float R[9];
  float c0, c1, c2, c3, c4, c5;
  if (cov3D_precomp) {
    const float* c = cov3D_precomp + 6 * idx;
    c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
  } else {
    const float mod = kp.scale_modifier;
    s0 = mod * scales[3 * idx]; s1 = mod * scales[3 * idx + 1]; s2 = mod * scales[3 * idx + 2];
    const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
    qr = q.x; qx = q.y; qy = q.z; qz = q.w;
    R[0] = 1.f - 2.f * (qy * qy + qz * qz); R[1] = 2.f * (qx * qy - qr * qz); R[2] = 2.f * (qx * qz + qr * qy);
    R[3] = 2.f * (qx * qy + qr * qz); R[4] = 1.f - 2.f * (qx * qx + qz * qz); R[5] = 2.f * (qy * qz - qr * qx);
    R[6] = 2.f * (qx * qz - qr * qy); R[7] = 2.f * (qy * qz + qr * qx); R[8] = 1.f - 2.f * (qx * qx + qy * qy);
    const float L0 = R[0] * s0, L1 = R[1] * s1, L2 = R[2] * s2;
    const float L3 = R[3] * s0, L4 = R[4] * s1, L5 = R[5] * s2;
    const float L6 = R[6] * s0, L7 = R[7] * s1, L8 = R[8] * s2;
    c0 = L0 * L0 + L1 * L1 + L2 * L2; c1 = L0 * L3 + L1 * L4 + L2 * L5; c2 = L0 * L6 + L1 * L7 + L2 * L8;
    c3 = L3 * L3 + L4 * L4 + L5 * L5; c4 = L3 * L6 + L4 * L7 + L5 * L8; c5 = L6 * L6 + L7 * L7 + L8 * L8;
  }

  float dmean0 = 0.f, dmean1 = 0.f, dmean2 = 0.f, dopac = 0.f;
  float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dcol[3] = {0.f, 0.f, 0.f};
  float dsh[MAXM * 3];
#pragma unroll
  for (int k = 0; k < MAXM * 3; k++) dsh[k] = 0.f;

  for (int v = 0; v < kp.V; v++) {
    const GipRecord* rec = records + (size_t)v * kp.P + idx;
    const uint4 q1 = reinterpret_cast<const uint4*>(rec)[1];
    const uint4 q2 = reinterpret_cast<const uint4*>(rec)[2];
    const uint32_t ntiles = q1.w;
    const int radius = (int)q2.w;
    float* d2 = gout.dL_dmeans2D ? gout.dL_dmeans2D + ((size_t)v * kp.P + idx) * 3 : nullptr;
    if (!(radius > 0)) {
      if (d2) { d2[0] = 0.f; d2[1] = 0.f; d2[2] = 0.f; }
      if (sh_gcol) reinterpret_cast<float4*>(sh_gcol)[(size_t)v * kp.P + idx] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    // ---- fixed-order sum of this Gaussian's rows ----
    float a[10];
#pragma unroll
    for (int k = 0; k < 10; k++) a[k] = 0.f;
    const uint32_t off = inst_offset[(size_t)v * kp.P + idx];
    for (uint32_t t = 0; t < ntiles; t++) {
      const uint32_t row = off + t;
      if (row >= kp.capacity) break;
      const float4* rp = reinterpret_cast<const float4*>(partial + (size_t)row * GIP_PARTIAL_FLOATS);
      const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
      a[0] += r0.x; a[1] += r0.y; a[2] += r0.z; a[3] += r0.w;
      a[4] += r1.x; a[5] += r1.y; a[6] += r1.z; a[7] += r1.w;
      a[8] += r2.x; a[9] += r2.y;
    }
    // rows hold moments of Q = dL/dG * G (render_backward.hip); apply this (view, Gaussian)'s constants once
    const float4 q0r = reinterpret_cast<const float4*>(rec)[0];
    const float4 q1r = reinterpret_cast<const float4*>(rec)[1];
    const float con_a = q1r.x, con_b = q1r.y, con_c = q1r.z, opac = q0r.w;
    const float g2x = -(con_a * a[0] + con_b * a[1]) * (0.5f * kp.W);
    const float g2y = -(con_c * a[1] + con_b * a[0]) * (0.5f * kp.H);
    const float gcx = -0.5f * a[2], gcy = -0.5f * a[3], gcw = -0.5f * a[4];
    if (d2) { d2[0] = g2x; d2[1] = g2y; d2[2] = 0.f; }
    dopac += opac != 0.f ? a[5] / opac : 0.f;

    const float* view = viewmatrix + 16 * v;
    const float* proj = projmatrix + 16 * v;
    const float tanx = kp.view[v].tanfovx, tany = kp.view[v].tanfovy;
    const float fx = kp.view[v].focal_x, fy = kp.view[v].focal_y;

    // ---- conic -> cov2D -> (cov3D, view-space mean) ----
    float t0 = view[0] * m0 + view[4] * m1 + view[8] * m2 + view[12];
    float t1 = view[1] * m0 + view[5] * m1 + view[9] * m2 + view[13];
    const float t2 = view[2] * m0 + view[6] * m1 + view[10] * m2 + view[14];
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t0 / t2, tytz = t1 / t2;
    const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
    t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
    const float J00 = fx / t2, J02 = -(fx * t0) / (t2 * t2);
    const float J11 = fy / t2, J12 = -(fy * t1) / (t2 * t2);
    const float M00 = J00 * view[0] + J02 * view[2], M01 = J00 * view[4] + J02 * view[6], M02 = J00 * view[8] + J02 * view[10];
    const float M10 = J11 * view[1] + J12 * view[2], M11 = J11 * view[5] + J12 * view[6], M12 = J11 * view[9] + J12 * view[10];
    const float v00 = c0 * M00 + c1 * M01 + c2 * M02, v01 = c1 * M00 + c3 * M01 + c4 * M02, v02 = c2 * M00 + c4 * M01 + c5 * M02;
    const float v10 = c0 * M10 + c1 * M11 + c2 * M12, v11 = c1 * M10 + c3 * M11 + c4 * M12, v12 = c2 * M10 + c4 * M11 + c5 * M12;
    const float ca = (M00 * v00 + M01 * v01 + M02 * v02) + 0.3f;
    const float cb = M00 * v10 + M01 * v11 + M02 * v12;
    const float cc = (M10 * v10 + M11 * v11 + M12 * v12) + 0.3f;
    const float denom = ca * cc - cb * cb;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (denom2inv != 0.f) {
      dL_da = denom2inv * (-cc * cc * gcx + 2 * cb * cc * gcy + (denom - ca * cc) * gcw);
      dL_dc = denom2inv * (-ca * ca * gcw + 2 * ca * cb * gcy + (denom - ca * cc) * gcx);
      dL_db = denom2inv * 2 * (cb * cc * gcx - (denom + 2 * cb * cb) * gcy + ca * cb * gcw);
      dcov[0] += M00 * M00 * dL_da + M00 * M10 * dL_db + M10 * M10 * dL_dc;
      dcov[3] += M01 * M01 * dL_da + M01 * M11 * dL_db + M11 * M11 * dL_dc;
      dcov[5] += M02 * M02 * dL_da + M02 * M12 * dL_db + M12 * M12 * dL_dc;
      dcov[1] += 2 * M00 * M01 * dL_da + (M00 * M11 + M01 * M10) * dL_db + 2 * M10 * M11 * dL_dc;
      dcov[2] += 2 * M00 * M02 * dL_da + (M00 * M12 + M02 * M10) * dL_db + 2 * M10 * M12 * dL_dc;
      dcov[4] += 2 * M02 * M01 * dL_da + (M01 * M12 + M02 * M11) * dL_db + 2 * M11 * M12 * dL_dc;
    }
    const float dM00 = 2 * v00 * dL_da + v10 * dL_db, dM01 = 2 * v01 * dL_da + v11 * dL_db, dM02 = 2 * v02 * dL_da + v12 * dL_db;
    const float dM10 = 2 * v10 * dL_dc + v00 * dL_db, dM11 = 2 * v11 * dL_dc + v01 * dL_db, dM12 = 2 * v12 * dL_dc + v02 * dL_db;
    const float dJ00 = view[0] * dM00 + view[4] * dM01 + view[8] * dM02;
    const float dJ02 = view[2] * dM00 + view[6] * dM01 + view[10] * dM02;
    const float dJ11 = view[1] * dM10 + view[5] * dM11 + view[9] * dM12;
    const float dJ12 = view[2] * dM10 + view[6] * dM11 + view[10] * dM12;
    const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = xmul * -fx * tz2 * dJ02;
    const float dty = ymul * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t0) * tz3 * dJ02 + (2 * fy * t1) * tz3 * dJ12;
    dmean0 += view[0] * dtx + view[1] * dty + view[2] * dtz;
    dmean1 += view[4] * dtx + view[5] * dty + view[6] * dtz;
    dmean2 += view[8] * dtx + view[9] * dty + view[10] * dtz;

    // ---- screen-space mean -> 3-D mean ----
    const float mh3 = proj[3] * m0 + proj[7] * m1 + proj[11] * m2 + proj[15];
    const float m_w = 1.0f / (mh3 + 0.0000001f);
    const float mul1 = (proj[0] * m0 + proj[4] * m1 + proj[8] * m2 + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m0 + proj[5] * m1 + proj[9] * m2 + proj[13]) * m_w * m_w;
    dmean0 += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dmean1 += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dmean2 += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    // ---- depth (view-space z) -> 3-D mean ----
    const float gdep = a[9];
    const float mul3 = view[2] * m0 + view[6] * m1 + view[10] * m2 + view[14];
    dmean0 += (view[2] - view[3] * mul3) * gdep;
    dmean1 += (view[6] - view[7] * mul3) * gdep;
    dmean2 += (view[10] - view[11] * mul3) * gdep;

    // ---- colour ----
    if (colors_precomp) {
      dcol[0] += a[6]; dcol[1] += a[7]; dcol[2] += a[8];
    } else if (sh_gcol) {
      const uint32_t clamped = reinterpret_cast<const uint4*>(rec)[3].z;
      reinterpret_cast<float4*>(sh_gcol)[(size_t)v * kp.P + idx] =
          make_float4((clamped & 1u) ? 0.f : a[6], (clamped & 2u) ? 0.f : a[7], (clamped & 4u) ? 0.f : a[8], 0.f);
    } else {
      const uint32_t clamped = reinterpret_cast<const uint4*>(rec)[3].z;
      float gcol[3] = {(clamped & 1u) ? 0.f : a[6], (clamped & 2u) ? 0.f : a[7], (clamped & 4u) ? 0.f : a[8]};
      const float* sh = shs + (size_t)idx * kp.M * 3;
      const float* campos = camposs + 3 * v;
      const float d0 = m0 - campos[0], d1 = m1 - campos[1], d2_ = m2 - campos[2];
      const float len = sqrtf(d0 * d0 + d1 * d1 + d2_ * d2_);
      const float x = d0 / len, y = d1 / len, z = d2_ / len;
      float ddir0 = 0.f, ddir1 = 0.f, ddir2 = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        const float gch = gcol[ch];
#define SH(k) sh[(k) * 3 + ch]
#define DSH(k, val) dsh[(k) * 3 + ch] += (val) * gch
        float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
        DSH(0, SH_C0);
        if (MAXM > 1 && kp.D > 0) {
          DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
          dx_ = -SH_C1 * SH(3); dy_ = -SH_C1 * SH(1); dz_ = SH_C1 * SH(2);
          if (MAXM > 4 && kp.D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            DSH(4, GSH_C2[0] * xy); DSH(5, GSH_C2[1] * yz); DSH(6, GSH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, GSH_C2[3] * xz); DSH(8, GSH_C2[4] * (xx - yy));
            dx_ += GSH_C2[0] * y * SH(4) + GSH_C2[2] * 2.f * -x * SH(6) + GSH_C2[3] * z * SH(7) + GSH_C2[4] * 2.f * x * SH(8);
            dy_ += GSH_C2[0] * x * SH(4) + GSH_C2[1] * z * SH(5) + GSH_C2[2] * 2.f * -y * SH(6) + GSH_C2[4] * 2.f * -y * SH(8);
            dz_ += GSH_C2[1] * y * SH(5) + GSH_C2[2] * 2.f * 2.f * z * SH(6) + GSH_C2[3] * x * SH(7);
            if (MAXM > 9 && kp.D > 2) {
              DSH(9, GSH_C3[0] * y * (3.f * xx - yy)); DSH(10, GSH_C3[1] * xy * z);
              DSH(11, GSH_C3[2] * y * (4.f * zz - xx - yy)); DSH(12, GSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
              DSH(13, GSH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, GSH_C3[5] * z * (xx - yy));
              DSH(15, GSH_C3[6] * x * (xx - 3.f * yy));
              dx_ += GSH_C3[0] * SH(9) * 3.f * 2.f * xy + GSH_C3[1] * SH(10) * yz + GSH_C3[2] * SH(11) * -2.f * xy +
                     GSH_C3[3] * SH(12) * -3.f * 2.f * xz + GSH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                     GSH_C3[5] * SH(14) * 2.f * xz + GSH_C3[6] * SH(15) * 3.f * (xx - yy);
              dy_ += GSH_C3[0] * SH(9) * 3.f * (xx - yy) + GSH_C3[1] * SH(10) * xz + GSH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                     GSH_C3[3] * SH(12) * -3.f * 2.f * yz + GSH_C3[4] * SH(13) * -2.f * xy + GSH_C3[5] * SH(14) * -2.f * yz +
                     GSH_C3[6] * SH(15) * -3.f * 2.f * xy;
              dz_ += GSH_C3[1] * SH(10) * xy + GSH_C3[2] * SH(11) * 4.f * 2.f * yz + GSH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                     GSH_C3[4] * SH(13) * 4.f * 2.f * xz + GSH_C3[5] * SH(14) * (xx - yy);
            }
          }
        }
#undef SH
#undef DSH
        ddir0 += dx_ * gch; ddir1 += dy_ * gch; ddir2 += dz_ * gch;
      }
      if (kp.D > 0) {
        const float sum2 = d0 * d0 + d1 * d1 + d2_ * d2_;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean0 += ((sum2 - d0 * d0) * ddir0 - d1 * d0 * ddir1 - d2_ * d0 * ddir2) * invsum32;
        dmean1 += (-d0 * d1 * ddir0 + (sum2 - d1 * d1) * ddir1 - d2_ * d1 * ddir2) * invsum32;
        dmean2 += (-d0 * d2_ * ddir0 - d1 * d2_ * ddir1 + (sum2 - d2_ * d2_) * ddir2) * invsum32;
      }
    }
  }

  // ---- write-out (every element exactly once) ----
  if (gout.dL_dmeans3D) { gout.dL_dmeans3D[3 * idx] = dmean0; gout.dL_dmeans3D[3 * idx + 1] = dmean1; gout.dL_dmeans3D[3 * idx + 2] = dmean2; }
  if (gout.dL_dopacities) gout.dL_dopacities[idx] = dopac;
  if (gout.dL_dcolors_precomp) { gout.dL_dcolors_precomp[3 * idx] = dcol[0]; gout.dL_dcolors_precomp[3 * idx + 1] = dcol[1]; gout.dL_dcolors_precomp[3 * idx + 2] = dcol[2]; }
  if (gout.dL_dshs && !sh_gcol) {
    float* o = gout.dL_dshs + (size_t)idx * kp.M * 3;
#pragma unroll
    for (int k = 0; k < MAXM * 3; k++) if (k < kp.M * 3) o[k] = dsh[k];
    for (int k = MAXM * 3; k < kp.M * 3; k++) o[k] = 0.f;
  }
  if (cov3D_precomp) {
    if (gout.dL_dcov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; k++) gout.dL_dcov3D_precomp[6 * idx + k] = dcov[k];
    }
  } else {
    // cov3D -> scale / rotation: dL/dL = 2 dSigma L, L = R diag(s)
    const float dS[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                         0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
    const float s[3] = {s0, s1, s2};
    float dLm[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int k = 0; k < 3; k++)
        dLm[i * 3 + k] = 2.0f * (dS[i * 3 + 0] * R[0 * 3 + k] * s[k] + dS[i * 3 + 1] * R[1 * 3 + k] * s[k] + dS[i * 3 + 2] * R[2 * 3 + k] * s[k]);
    if (gout.dL_dscales) {
      // gradient w.r.t. (scale_modifier * scale), as in the fork (see oracle note)
#pragma unroll
      for (int k = 0; k < 3; k++)
        gout.dL_dscales[3 * idx + k] = R[0 * 3 + k] * dLm[0 * 3 + k] + R[1 * 3 + k] * dLm[1 * 3 + k] + R[2 * 3 + k] * dLm[2 * 3 + k];
    }
    if (gout.dL_drotations) {
      float dR[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) dR[i * 3 + k] = dLm[i * 3 + k] * s[k];
#define DR(i_, j_) dR[(i_) * 3 + (j_)]
      float4 dq;
      dq.x = 2 * qz * (DR(1, 0) - DR(0, 1)) + 2 * qy * (DR(0, 2) - DR(2, 0)) + 2 * qx * (DR(2, 1) - DR(1, 2));
      dq.y = 2 * qy * (DR(0, 1) + DR(1, 0)) + 2 * qz * (DR(0, 2) + DR(2, 0)) + 2 * qr * (DR(2, 1) - DR(1, 2)) - 4 * qx * (DR(2, 2) + DR(1, 1));
      dq.z = 2 * qx * (DR(0, 1) + DR(1, 0)) + 2 * qr * (DR(0, 2) - DR(2, 0)) + 2 * qz * (DR(1, 2) + DR(2, 1)) - 4 * qy * (DR(2, 2) + DR(0, 0));
      dq.w = 2 * qr * (DR(1, 0) - DR(0, 1)) + 2 * qx * (DR(0, 2) + DR(2, 0)) + 2 * qy * (DR(1, 2) + DR(2, 1)) - 4 * qz * (DR(1, 1) + DR(0, 0));
#undef DR
      reinterpret_cast<float4*>(gout.dL_drotations)[idx] = dq;
    }
  }
}

void gip_launch_gather_backward(const GipKernelParams& kp, const GipRasterInputs& in, GipStatePtrs st, const float* partial,
                                const GipRasterGradsOut& gout, hipStream_t s) {
  const dim3 grid(kp.nblk), block(GIP_BLOCK);
#define LAUNCH(MM) hipLaunchKernelGGL((gip_gather_backward_kernel<MM>), grid, block, 0, s, kp, in.means3D, in.shs, \
    in.colors_precomp, in.scales, in.rotations, in.cov3D_precomp, in.viewmatrix, in.projmatrix, in.campos, st.records, \
    st.inst_offset, partial, gout, st.header, kp.sh_mfma ? st.sh_colors : nullptr)
  const int needed = (in.shs && !kp.sh_mfma) ? (kp.D + 1) * (kp.D + 1) : 1;      // matrix-core SH path: no dsh registers here
  if (needed <= 1) LAUNCH(1);
  else if (needed <= 4) LAUNCH(4);
  else if (needed <= 9) LAUNCH(9);
  else LAUNCH(16);
#undef LAUNCH
}
