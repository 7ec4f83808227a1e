// valu_rate.hip — issue rate of wave64 fp32 vector instructions on gfx950 by waves per SIMD.
//
// Question: does one SIMD retire a wave64 v_fma_f32 every 4 cycles (what `roofline_valu` assumed in round 2, from
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.18 on the render kernels) or every 2 cycles once >= 2 waves share the SIMD
// (the guide's constants table: "v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4")?
//
// Method: every wave runs R x 64 INDEPENDENT instructions (16 accumulator chains, dependent distance 16) between two
// s_memtime stamps.  k waves per SIMD are made co-resident as workgroups of 256 * min(k, 4) threads, k / min(k, 4) of
// them per CU (grid = 256 CUs x that).  Per wave: cycles / instruction = dt / (R * 64); the SIMD's issue interval per
// wave-instruction is that divided by the k waves that share it (they overlap in time: start skew is printed).
// The clock is dt(s_memtime) / dt(s_memrealtime at 100 MHz).  Forms: v_fma_f32, v_pk_fma_f32 (two lanes' worth per
// instruction), v_exp_f32 (transcendental), and a render-like mix (3 fma : 1 mul : 1 cndmask : 1 exp per 6).
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip      run: ./valu_rate > profiles/r03_valu_rate.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { FMA = 0, PKFMA = 1, EXP = 2, MIX = 3 };

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(unsigned long long* __restrict__ stamps, float* __restrict__ sink, int R, float seed) {
  float a[16];
  v2f p[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    a[i] = seed + i + threadIdx.x * 1e-3f;
    if constexpr (KIND == PKFMA) p[i] = (v2f){a[i], a[i] + 0.5f}; else p[i] = (v2f){0.f, 0.f};
  }
  const float b = 0.999f + seed * 1e-6f, c = 1e-3f;
  const v2f pb = (v2f){b, b}, pc = (v2f){c, c};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < R; r++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if constexpr (KIND == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (KIND == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        REP16(X)
#undef X
      } else if constexpr (KIND == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      } else {
        // 16 instructions: per group of 6 -> fma fma fma mul cndmask exp (the blend's rough mix), independent registers
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[3]) : "v"(b));
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[4]) : "v"(b));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[5]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[6]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[7]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[8]) : "v"(b), "v"(c));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[9]) : "v"(b));
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[10]) : "v"(b));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[11]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[12]) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[13]) : "v"(b), "v"(c));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[14]) : "v"(b));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[15]) : "v"(c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += KIND == PKFMA ? p[i][0] + p[i][1] : a[i];
  if (s == 123.456f) sink[0] = s;                      // keeps the chains live; never true
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[4 * w + 0] = t0; stamps[4 * w + 1] = t1; stamps[4 * w + 2] = r0; stamps[4 * w + 3] = r1;
  }
}

template <int KIND>
static void run(const char* name, int k, unsigned long long* d_st, float* d_sink) {
  const int cus = 256, wg_waves_per_simd = std::min(k, 4), threads = 256 * wg_waves_per_simd, wgs_per_cu = k / wg_waves_per_simd;
  const int grid = cus * wgs_per_cu, waves = grid * threads / 64, R = 4096;
  std::vector<unsigned long long> h(4 * (size_t)waves);
  double best_cyc = 1e30, best_clock = 0, best_span = 0, best_ms = 0;
  int occ_blocks = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_blocks, rate_kernel<KIND>, threads, 0);
  const int resident_waves_per_simd = std::min(occ_blocks, wgs_per_cu) * threads / 256;
  for (int rep = 0; rep < 5; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(threads), 0, 0, d_st, d_sink, R, 1.0f + rep);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves);
    double clk = 0; unsigned long long tmin = ~0ull, tmax0 = 0;
    for (int w = 0; w < waves; w++) {
      cyc[w] = double(h[4 * w + 1] - h[4 * w]) / (double(R) * 64.0);
      clk += double(h[4 * w + 1] - h[4 * w]) / (double(h[4 * w + 3] - h[4 * w + 2]) * 10.0);   // cycles per ns -> GHz (realtime = 100 MHz)
      if (w < threads / 64) { tmin = std::min(tmin, h[4 * w]); tmax0 = std::max(tmax0, h[4 * w]); }   // start skew inside workgroup 0
    }
    std::sort(cyc.begin(), cyc.end());
    const double med = cyc[waves / 2];
    if (med < best_cyc) { best_cyc = med; best_clock = clk / waves; best_span = double(tmax0 - tmin); best_ms = ms; }
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  const double lanes_per_instr = KIND == PKFMA ? 128.0 : 64.0;
  // cross-check from the launch's wall time: total wave-instructions / (1024 SIMDs x cycles of the launch at the measured clock)
  const double wall_cyc = best_ms * 1e-3 * best_clock * 1e9;
  const double simd_cyc_wall = wall_cyc / (double(waves) * R * 64.0 / 1024.0);
  printf("{\"instr\": \"%s\", \"waves_per_simd_launched\": %d, \"waves_per_simd_resident\": %d, \"workgroup\": %d, \"grid\": %d, "
         "\"cycles_per_instr_per_wave\": %.3f, \"simd_cycles_per_wave_instr\": %.3f, \"simd_cycles_per_wave_instr_from_wall\": %.3f, "
         "\"lane_ops_per_simd_cycle\": %.2f, \"clock_ghz\": %.3f, \"start_skew_cycles_wg0\": %.0f, \"launch_ms\": %.3f}\n",
         name, k, resident_waves_per_simd, threads, grid, best_cyc, best_cyc / resident_waves_per_simd, simd_cyc_wall,
         lanes_per_instr * resident_waves_per_simd / best_cyc, best_clock, best_span, best_ms);
}

int main() {
  unsigned long long* d_st; float* d_sink;
  hipMalloc(&d_st, 4 * 8 * (size_t)256 * 32 * 2); hipMalloc(&d_sink, 64);
  for (int k : {1, 2, 4, 8}) run<FMA>("v_fma_f32", k, d_st, d_sink);
  for (int k : {1, 2, 4, 8}) run<PKFMA>("v_pk_fma_f32", k, d_st, d_sink);
  for (int k : {1, 2, 4, 8}) run<EXP>("v_exp_f32", k, d_st, d_sink);
  for (int k : {1, 2, 4, 8}) run<MIX>("render_mix", k, d_st, d_sink);
  return 0;
}
