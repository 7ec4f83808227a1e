#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float fold32(float a, float b) {
  // inline asm: the clang builtin's second result is mis-selected for float operands on ROCm 7.2 (both
  // extracts return the first register); the swap updates both registers in place.
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float fold16(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float row_reduce(float v) {
  v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
  return v;
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float v[10];
  for (int i = 0; i < 10; i++) v[i] = (float)((i + 1) * 1000 + l);   // sum over lanes = 64*(i+1)*1000 + 2016
  float p02 = fold32(v[0], v[2]), p13 = fold32(v[1], v[3]), p46 = fold32(v[4], v[6]), p57 = fold32(v[5], v[7]), p89 = fold32(v[8], v[9]);
  float qa = row_reduce(fold16(p02, p13)), qb = row_reduce(fold16(p46, p57));
  float qc = row_reduce(p89);
  qc = dpp_add<0x142, 0xa>(qc);
  out[l] = qa; out[64 + l] = qb; out[128 + l] = qc;
  out[192 + l] = p02; 
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 10; i++) printf("expect v%d = %d\n", i, 64 * (i + 1) * 1000 + 2016);
  for (int r = 0; r < 4; r++) printf("qa lane %d = %.0f  qb = %.0f  qc = %.0f\n", r * 16 + 15, h[r * 16 + 15], h[64 + r * 16 + 15], h[128 + r * 16 + 15]);
  printf("p02 lanes 0,31,32,63: %.0f %.0f %.0f %.0f\n", h[192], h[192+31], h[192+32], h[192+63]);
  return 0;
}
