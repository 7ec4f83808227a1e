// Measures workgroup dispatch cost on MI355X: empty-ish kernels with N workgroups, by block size and static LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS>
__global__ void k(const unsigned* __restrict__ in, unsigned* __restrict__ out) {
  __shared__ unsigned s[LDS > 0 ? LDS / 4 : 1];
  if (LDS > 0) s[threadIdx.x % (LDS / 4)] = threadIdx.x;
  unsigned v = in[blockIdx.x & 1023];            // one global load per lane, like reading a tile range
  if (v == 0xdeadbeef) out[blockIdx.x] = v + (LDS > 0 ? s[0] : 0);
}
template <int LDS>
void run(int nwg, int bs, unsigned* in, unsigned* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) k<LDS><<<nwg, bs>>>(in, out);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; i++) k<LDS><<<nwg, bs>>>(in, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("nwg=%6d block=%4d lds=%6d : %8.2f us per launch  (%.1f ns / WG)\n", nwg, bs, LDS, ms * 1e3 / reps, ms * 1e6 / reps / nwg);
}
int main() {
  unsigned *in, *out; hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&out, 1 << 20);
  for (int bs : {64, 256}) for (int nwg : {2048, 16384, 65536}) {
    run<0>(nwg, bs, in, out); run<3072>(nwg, bs, in, out); run<11776>(nwg, bs, in, out); run<65536>(nwg, bs, in, out);
  }
  return 0;
}
