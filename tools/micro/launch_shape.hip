// Empty-kernel cost by launch shape on gfx950: threads per workgroup x dynamic LDS x grid.  (Why: the 1024-thread /
// 128 KB long-list sort kernel cost 25 us per launch even when its class is empty.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k_empty(const unsigned* flag, unsigned* out) {
  extern __shared__ unsigned long long s[];
  if (flag[0] == 0) return;
  s[threadIdx.x] = threadIdx.x; __syncthreads(); out[blockIdx.x] = (unsigned)s[(threadIdx.x + 1) & 1023];
}
int main() {
  unsigned *flag, *out; hipMalloc(&flag, 4); hipMalloc(&out, 4 * 4096); hipMemset(flag, 0, 4);
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int shapes[][3] = {{256, 1024, 131072}, {256, 1024, 65536}, {256, 1024, 16384}, {256, 1024, 0}, {256, 256, 131072}, {256, 256, 0},
                           {1024, 256, 16384}, {128, 1024, 131072}, {64, 1024, 131072}, {256, 512, 131072}};
  for (auto& sh : shapes) {
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), sh[2], 0, flag, out);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), sh[2], 0, flag, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("grid %4d x %4d threads, %6d B LDS: %.2f us per launch (back to back)\n", sh[0], sh[1], sh[2], ms * 1000 / 200);
  }
  return 0;
}
