// sh_mfma.hip — the spherical-harmonics colour contraction on the gfx950 matrix cores, forward and backward.
//
// Reference: gaussiansplatting/utils/sh_utils.py:57-112 (eval_sh), called per camera at gaussian_renderer/__init__.py:73-78
// (+0.5, clamp at 0); in the fork the same arithmetic lives in computeColorFromSH / its backward.  `north_star` allots the matrix
// cores to exactly this contraction.
//
// colour[g][view][c] = sum_k basis_k(dir(g, view)) * sh[g][k][c] has no operand shared BETWEEN Gaussians (every Gaussian has its own
// K x 3 coefficient block and its own direction), so a big matrix instruction finds nothing to reuse.  What is shared is one
// Gaussian's coefficient block across the V views of a launch set: a [4 views x K] . [K x 3] product per Gaussian.  That is the
// shape of v_mfma_f32_4x4x1_16b_f32: sixteen independent 4 x 4 blocks per instruction, K = 1, block b = lanes 4b .. 4b + 3
// (A[i] in lane 4b + i, B[j] in lane 4b + j, D[i][j] in VGPR i of lane 4b + j).  One wave = sixteen Gaussians, four lanes each:
//   forward   lane (g, i) evaluates the K basis values of view i, lane (g, j) holds sh[g][.][c = j]; K chained instructions leave
//             colour[view 0..3][c = j] in the four accumulator registers of lane (g, j) — K matrix instructions per sixteen
//             Gaussians x four views, against 3 K v_fma per (view, Gaussian) lane of the scalar form;
//   backward  dL/dsh[k][c] = sum_view basis_k(view) dL/dcolour[view][c]: A = dL/dcolour[view][c = i], B = basis_{4q + j}(view), chained
//             over the views, four accumulators q = 0..3;  and the direction gradient through
//             w[view][k] = sum_c sh[k][c] dL/dcolour[view][c]: A = sh[4q + i][c], B = dL/dcolour[view = j][c], chained over c, which
//             leaves w[own view][.] in the lane that holds that view's direction — d(dir) = sum_k w_k dbasis_k/d(dir) is lane-local.
// Taken for sh_degree >= 1 and V >= 2 (GipRasterConfig::sh_scalar = 0); one view has nothing to batch.  The matrix core's
// summation order is not the scalar chain's: colours agree with the scalar kernel / the oracle to a few ulp (the integer buffers
// — radii, rectangles, keys, ranges — never depend on colour), and `sh_scalar = 1` keeps the bit-exact scalar path.
// Compiled with -ffp-contract=off like preprocess.hip: the direction and the basis are the scalar kernel's own operation sequence.
#include "gip_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SHM_C0 0.28209479177387814f
#define SHM_C1 0.4886025119029199f
__device__ static const float SHM_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SHM_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};

// the K = (DEG + 1)^2 basis values with the SH constants and signs folded in: colour_c = sum_k b[k] * sh[k][c]
template <int DEG>
__device__ __forceinline__ void shm_basis(float x, float y, float z, float* b) {
  b[0] = SHM_C0;
  if constexpr (DEG > 0) {
    b[1] = -SHM_C1 * y; b[2] = SHM_C1 * z; b[3] = -SHM_C1 * x;
  }
  if constexpr (DEG > 1) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SHM_C2[0] * xy; b[5] = SHM_C2[1] * yz; b[6] = SHM_C2[2] * (2.0f * zz - xx - yy);
    b[7] = SHM_C2[3] * xz; b[8] = SHM_C2[4] * (xx - yy);
    if constexpr (DEG > 2) {
      b[9] = SHM_C3[0] * y * (3.0f * xx - yy); b[10] = SHM_C3[1] * xy * z;
      b[11] = SHM_C3[2] * y * (4.0f * zz - xx - yy); b[12] = SHM_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
      b[13] = SHM_C3[4] * x * (4.0f * zz - xx - yy); b[14] = SHM_C3[5] * z * (xx - yy);
      b[15] = SHM_C3[6] * x * (xx - 3.0f * yy);
    }
  }
}

#define SHM_THREADS 256
#define SHM_GAUSSIANS (SHM_THREADS / 4)      // Gaussians per workgroup: sixteen per wave

// sh_colors[view][g] = (r, g, b, -) BEFORE the clamp at 0 (preprocess.hip clamps and records the flags, as for its own scalar sum)
template <int DEG>
__global__ void __launch_bounds__(SHM_THREADS)
gip_sh_forward_mfma_kernel(GipKernelParams kp, const float* __restrict__ means3D, const float* __restrict__ shs,
                           const float* __restrict__ camposs, float* __restrict__ sh_colors) {
  constexpr int N = (DEG + 1) * (DEG + 1);
  const int lane = threadIdx.x & 63, sub = lane & 3;
  const int g = blockIdx.x * SHM_GAUSSIANS + (threadIdx.x >> 2);
  const bool valid = g < kp.P;
  const int gi = valid ? g : kp.P - 1;
  // B operand: column c = sub of this Gaussian's coefficient block (lane sub = 3: a zero column)
  float shv[N];
  const float* sh = shs + (size_t)gi * kp.M * 3;
#pragma unroll
  for (int k = 0; k < N; k++) shv[k] = sub < 3 ? sh[k * 3 + sub] : 0.f;
  const float p0 = means3D[3 * gi], p1 = means3D[3 * gi + 1], p2 = means3D[3 * gi + 2];
  for (int v0 = 0; v0 < kp.V; v0 += 4) {
    // A operand: the basis of view v0 + sub (a view past V repeats the last one: its rows of the product are not stored)
    const int v = min(v0 + sub, kp.V - 1);
    const float* campos = camposs + 3 * v;
    const float d0 = p0 - campos[0], d1 = p1 - campos[1], d2 = p2 - campos[2];
    const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    float b[N];
    shm_basis<DEG>(d0 / len, d1 / len, d2 / len, b);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < N; k++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(b[k], shv[k], acc, 0, 0, 0);
    // acc[i] = colour[view v0 + i][c = sub]: four lanes write one 16-byte (r, g, b, -) element
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (valid && v0 + i < kp.V) sh_colors[((size_t)(v0 + i) * kp.P + g) * 4 + sub] = acc[i] + 0.5f;
  }
}

// sh_gcol[view][g] = (dL/dr, dL/dg, dL/db, -) of the CLAMPED colour (zero for a clamped channel / an invisible view), left by
// gather_backward.hip in the buffer the forward's colours occupied.  Writes dL/dshs [P, M, 3] and ADDS the gradient that reaches
// the mean through the view direction to dL/dmeans3D (which the gather kernel has written without it).
template <int DEG>
__global__ void __launch_bounds__(SHM_THREADS)
gip_sh_backward_mfma_kernel(GipKernelParams kp, const float* __restrict__ means3D, const float* __restrict__ shs,
                            const float* __restrict__ camposs, const float* __restrict__ sh_gcol, float* __restrict__ dL_dshs,
                            float* __restrict__ dL_dmeans3D, const GipRasterHeader* __restrict__ header) {
  constexpr int N = (DEG + 1) * (DEG + 1), NQ = (N + 3) / 4;
  if (header->overflow) return;      // the gather kernel has written zeros everywhere (gather_backward.hip)
  const int lane = threadIdx.x & 63, sub = lane & 3;
  const int g = blockIdx.x * SHM_GAUSSIANS + (threadIdx.x >> 2);
  const bool valid = g < kp.P;
  const int gi = valid ? g : kp.P - 1;
  const float p0 = means3D[3 * gi], p1 = means3D[3 * gi + 1], p2 = means3D[3 * gi + 2];
  // rows 4q + sub of the coefficient block (A operand of the direction product)
  float sha[NQ][3];
  const float* sh = shs + (size_t)gi * kp.M * 3;
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int c = 0; c < 3; c++) sha[q][c] = (4 * q + sub < N) ? sh[(4 * q + sub) * 3 + c] : 0.f;
  f32x4 dsh[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) dsh[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dm0 = 0.f, dm1 = 0.f, dm2 = 0.f;
  const float4* gc4 = reinterpret_cast<const float4*>(sh_gcol);
  for (int v0 = 0; v0 < kp.V; v0 += 4) {
    // ---- dL/dsh[4q + j][c] += basis_{4q + j}(view) * gcol[view][c], chained over the four views of the group ----
#pragma unroll
    for (int vv = 0; vv < 4; vv++) {
      const int view = v0 + vv;
      if (view >= kp.V) break;                                   // uniform
      const float ga = (valid && sub < 3) ? sh_gcol[((size_t)view * kp.P + gi) * 4 + sub] : 0.f;
      const float* campos = camposs + 3 * view;                  // uniform: scalar loads
      const float d0 = p0 - campos[0], d1 = p1 - campos[1], d2 = p2 - campos[2];
      const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
      float b[NQ * 4];
#pragma unroll
      for (int k = N; k < NQ * 4; k++) b[k] = 0.f;
      shm_basis<DEG>(d0 / len, d1 / len, d2 / len, b);
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const float bs = sub == 0 ? b[4 * q] : sub == 1 ? b[4 * q + 1] : sub == 2 ? b[4 * q + 2] : b[4 * q + 3];
        dsh[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(ga, bs, dsh[q], 0, 0, 0);
      }
    }
    // ---- w[view v0 + sub][4q + i] = sum_c sh[4q + i][c] * gcol[view][c]; then the direction gradient of this lane's view ----
    const int view = v0 + sub;
    const bool vok = valid && view < kp.V;
    const float4 gb = vok ? gc4[(size_t)view * kp.P + gi] : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 w[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      w[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      w[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(sha[q][0], gb.x, w[q], 0, 0, 0);
      w[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(sha[q][1], gb.y, w[q], 0, 0, 0);
      w[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(sha[q][2], gb.z, w[q], 0, 0, 0);
    }
    const int vc = min(view, kp.V - 1);
    const float* campos = camposs + 3 * vc;
    const float d0 = p0 - campos[0], d1 = p1 - campos[1], d2 = p2 - campos[2];
    const float sum2 = d0 * d0 + d1 * d1 + d2 * d2;
    const float len = sqrtf(sum2);
    const float x = d0 / len, y = d1 / len, z = d2 / len;
#define W(k) w[(k) >> 2][(k) & 3]
    // d colour / d(x, y, z) contracted with dL/dcolour: the scalar kernel's sums (gather_backward.hip) with SH(k) gch -> w_k
    float dx = -SHM_C1 * W(3), dy = -SHM_C1 * W(1), dz = SHM_C1 * W(2);
    if constexpr (DEG > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dx += SHM_C2[0] * y * W(4) + SHM_C2[2] * 2.f * -x * W(6) + SHM_C2[3] * z * W(7) + SHM_C2[4] * 2.f * x * W(8);
      dy += SHM_C2[0] * x * W(4) + SHM_C2[1] * z * W(5) + SHM_C2[2] * 2.f * -y * W(6) + SHM_C2[4] * 2.f * -y * W(8);
      dz += SHM_C2[1] * y * W(5) + SHM_C2[2] * 2.f * 2.f * z * W(6) + SHM_C2[3] * x * W(7);
      if constexpr (DEG > 2) {
        dx += SHM_C3[0] * W(9) * 3.f * 2.f * xy + SHM_C3[1] * W(10) * yz + SHM_C3[2] * W(11) * -2.f * xy +
              SHM_C3[3] * W(12) * -3.f * 2.f * xz + SHM_C3[4] * W(13) * (-3.f * xx + 4.f * zz - yy) +
              SHM_C3[5] * W(14) * 2.f * xz + SHM_C3[6] * W(15) * 3.f * (xx - yy);
        dy += SHM_C3[0] * W(9) * 3.f * (xx - yy) + SHM_C3[1] * W(10) * xz + SHM_C3[2] * W(11) * (-3.f * yy + 4.f * zz - xx) +
              SHM_C3[3] * W(12) * -3.f * 2.f * yz + SHM_C3[4] * W(13) * -2.f * xy + SHM_C3[5] * W(14) * -2.f * yz +
              SHM_C3[6] * W(15) * -3.f * 2.f * xy;
        dz += SHM_C3[1] * W(10) * xy + SHM_C3[2] * W(11) * 4.f * 2.f * yz + SHM_C3[3] * W(12) * 3.f * (2.f * zz - xx - yy) +
              SHM_C3[4] * W(13) * 4.f * 2.f * xz + SHM_C3[5] * W(14) * (xx - yy);
      }
    }
#undef W
    if (vok) {                                                   // through the normalisation of the direction
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      dm0 += ((sum2 - d0 * d0) * dx - d1 * d0 * dy - d2 * d0 * dz) * invsum32;
      dm1 += (-d0 * d1 * dx + (sum2 - d1 * d1) * dy - d2 * d1 * dz) * invsum32;
      dm2 += (-d0 * d2 * dx - d1 * d2 * dy + (sum2 - d2 * d2) * dz) * invsum32;
    }
  }
  // the four lanes of a Gaussian hold its views' shares: fixed-order fold inside the quad
  dm0 += __shfl_xor(dm0, 1, 64); dm1 += __shfl_xor(dm1, 1, 64); dm2 += __shfl_xor(dm2, 1, 64);
  dm0 += __shfl_xor(dm0, 2, 64); dm1 += __shfl_xor(dm1, 2, 64); dm2 += __shfl_xor(dm2, 2, 64);
  if (!valid) return;
  if (dL_dmeans3D && sub == 0) {
    dL_dmeans3D[3 * g] += dm0; dL_dmeans3D[3 * g + 1] += dm1; dL_dmeans3D[3 * g + 2] += dm2;
  }
  if (dL_dshs) {
    float* o = dL_dshs + (size_t)g * kp.M * 3;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int k = 4 * q + sub;                                 // D[i = c][j = sub]: VGPR c of this lane
      if (k < N) { o[k * 3] = dsh[q][0]; o[k * 3 + 1] = dsh[q][1]; o[k * 3 + 2] = dsh[q][2]; }
    }
    for (int k = N + sub; k < kp.M; k += 4) { o[k * 3] = 0.f; o[k * 3 + 1] = 0.f; o[k * 3 + 2] = 0.f; }   // inactive degrees
  }
}

void gip_launch_sh_forward_mfma(const GipKernelParams& kp, const GipRasterInputs& in, float* sh_colors, hipStream_t s) {
  const dim3 grid((kp.P + SHM_GAUSSIANS - 1) / SHM_GAUSSIANS), block(SHM_THREADS);
#define LAUNCH(DG) hipLaunchKernelGGL((gip_sh_forward_mfma_kernel<DG>), grid, block, 0, s, kp, in.means3D, in.shs, in.campos, sh_colors)
  if (kp.D == 1) LAUNCH(1);
  else if (kp.D == 2) LAUNCH(2);
  else LAUNCH(3);
#undef LAUNCH
}

void gip_launch_sh_backward_mfma(const GipKernelParams& kp, const GipRasterInputs& in, GipStatePtrs st, const float* sh_gcol,
                                 const GipRasterGradsOut& gout, hipStream_t s) {
  const dim3 grid((kp.P + SHM_GAUSSIANS - 1) / SHM_GAUSSIANS), block(SHM_THREADS);
#define LAUNCH(DG) hipLaunchKernelGGL((gip_sh_backward_mfma_kernel<DG>), grid, block, 0, s, kp, in.means3D, in.shs, in.campos, sh_gcol, \
                                      gout.dL_dshs, gout.dL_dmeans3D, st.header)
  if (kp.D == 1) LAUNCH(1);
  else if (kp.D == 2) LAUNCH(2);
  else LAUNCH(3);
#undef LAUNCH
}
