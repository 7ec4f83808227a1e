// What does a DEPENDENT stage boundary cost on MI355X — as a kernel boundary inside a replayed HIP graph, and as a grid-wide
// barrier inside one persistent kernel (one workgroup per CU)?  VERDICT r5 item 1 asks for persistent multi-stage block kernels
// for the 8^2 / 16^2 / 32^2 levels of the denoiser; this sizes what such a kernel could save per boundary before it is built.
//
// Every stage: workgroup b reads `per_wg` floats that a DIFFERENT workgroup (on another XCD) wrote in the previous stage, adds 1
// and writes its own `per_wg` floats.  After S stages every value must equal S (checks that the barrier really made the data
// visible across the eight L2s).
//   graph     : S launches of a plain kernel, captured once, replayed
//   barrier   : one launch, S stages, grid barrier = agent-scope release fetch_add + acquire spin (the compiler's L2 write-back
//               / invalidate around them), plain loads / stores
//   bypass    : one launch, data moved with system-scope relaxed atomics (sc0 sc1: past the L2), barrier with RELAXED atomics:
//               no cache maintenance at all
//   hier      : as bypass, but the barrier is hierarchical: one arrival counter per XCD (workgroups are dealt round-robin to the 8
//               XCDs), the last arriver of an XCD bumps a chip-wide counter, the last of those publishes the stage number in a
//               flag that everybody polls read-only — 32 + 8 contended atomics instead of 256 on one address
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int src_wg(int b, int nwg) { return (b * 37 + 11) % nwg; }      // another workgroup, another XCD

__global__ void __launch_bounds__(256) stage_kernel(const float* __restrict__ in, float* __restrict__ out, int per_wg, int nwg) {
  const int b = blockIdx.x, s = src_wg(b, nwg);
  for (int i = threadIdx.x; i < per_wg; i += 256) out[(size_t)b * per_wg + i] = in[(size_t)s * per_wg + i] + 1.f;
}

template <int MODE>      // 0: release / acquire fences, plain accesses; 1: cache-bypassing accesses, relaxed barrier; 2: + hierarchical barrier
__global__ void __launch_bounds__(256) persistent_kernel(float* buf0, float* buf1, unsigned* counter, int per_wg, int nwg, int stages) {
  const int b = blockIdx.x, s = src_wg(b, nwg);
  float* in = buf0;
  float* out = buf1;
  for (int st = 0; st < stages; st++) {
    for (int i = threadIdx.x; i < per_wg; i += 256) {
      if (MODE == 0) out[(size_t)b * per_wg + i] = in[(size_t)s * per_wg + i] + 1.f;
      else {      // MODE 1, 2
        const float v = __hip_atomic_load(in + (size_t)s * per_wg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out + (size_t)b * per_wg + i, v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // ---- grid barrier ----
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned target = (unsigned)(st + 1) * (unsigned)nwg;
      if (MODE == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      } else if (MODE == 2) {
        // counter[16 * (1 + xcd)]: arrivals of this XCD (its own 64-byte line); counter[0]: XCDs done; counter[16 * 9]: the stage flag
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned xcd = (unsigned)b & 7u, per_xcd = (unsigned)nwg >> 3;
        const unsigned mine = __hip_atomic_fetch_add(counter + 16 * (1 + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine == (unsigned)(st + 1) * per_xcd - 1u) {
          const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (done == (unsigned)(st + 1) * 8u - 1u) __hip_atomic_store(counter + 16 * 9, (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        while (__hip_atomic_load(counter + 16 * 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)(st + 1)) __builtin_amdgcn_s_sleep(1);
      } else {
        __builtin_amdgcn_s_waitcnt(0);      // this wave's stores have left (the other waves': s_waitcnt before their barrier arrival is the compiler's)
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    float* t = in; in = out; out = t;
  }
}

static float check(const float* dbuf, int n, float expect) {
  std::vector<float> h(n);
  CHECK(hipMemcpy(h.data(), dbuf, n * sizeof(float), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; i++) if (h[i] != expect) bad++;
  return (float)bad / n;
}

int main(int argc, char** argv) {
  const int stages = argc > 1 ? atoi(argv[1]) : 64;
  hipStream_t stream; CHECK(hipStreamCreate(&stream));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  unsigned* counter; CHECK(hipMalloc(&counter, 1024));
  for (int nwg : {256, 512}) for (int per_wg : {64, 512, 4096, 32768}) {      // 64 KB .. 32 MB per stage at 256 workgroups
    const size_t n = (size_t)nwg * per_wg;
    float *b0, *b1; CHECK(hipMalloc(&b0, n * 4)); CHECK(hipMalloc(&b1, n * 4));
    // --- graph of `stages` dependent launches ---
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (int st = 0; st < stages; st++) stage_kernel<<<nwg, 256, 0, stream>>>(st & 1 ? b1 : b0, st & 1 ? b0 : b1, per_wg, nwg);
    CHECK(hipStreamEndCapture(stream, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    float ms_graph = 0.f, bad_graph = 0.f;
    for (int rep = 0; rep < 4; rep++) {
      CHECK(hipMemsetAsync(b0, 0, n * 4, stream));
      CHECK(hipEventRecord(e0, stream));
      CHECK(hipGraphLaunch(exec, stream));
      CHECK(hipEventRecord(e1, stream));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_graph, e0, e1));
    }
    bad_graph = check(stages & 1 ? b1 : b0, (int)n, (float)stages);
    // --- eager launches of the same chain ---
    float ms_eager = 0.f;
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(e0, stream));
      for (int st = 0; st < stages; st++) stage_kernel<<<nwg, 256, 0, stream>>>(st & 1 ? b1 : b0, st & 1 ? b0 : b1, per_wg, nwg);
      CHECK(hipEventRecord(e1, stream));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms_eager, e0, e1));
    }
    float ms_p[3] = {0.f, 0.f, 0.f}, bad_p[3] = {0.f, 0.f, 0.f};
    if (nwg == 256) {      // persistent: one workgroup per CU is guaranteed co-resident
      for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 4; rep++) {
          CHECK(hipMemsetAsync(b0, 0, n * 4, stream));
          CHECK(hipMemsetAsync(counter, 0, 1024, stream));
          CHECK(hipEventRecord(e0, stream));
          if (mode == 0) persistent_kernel<0><<<nwg, 256, 0, stream>>>(b0, b1, counter, per_wg, nwg, stages);
          else if (mode == 1) persistent_kernel<1><<<nwg, 256, 0, stream>>>(b0, b1, counter, per_wg, nwg, stages);
          else persistent_kernel<2><<<nwg, 256, 0, stream>>>(b0, b1, counter, per_wg, nwg, stages);
          CHECK(hipEventRecord(e1, stream));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&ms_p[mode], e0, e1));
        }
        bad_p[mode] = check(stages & 1 ? b1 : b0, (int)n, (float)stages);
      }
    }
    printf("nwg %4d  bytes/stage %9zu : graph %6.2f us/stage (bad %.3f)  eager %6.2f  |  barrier(release/acquire) %6.2f us/stage (bad %.3f)  "
           "bypass(sc0 sc1 + relaxed) %6.2f us/stage (bad %.3f)  hierarchical %6.2f us/stage (bad %.3f)\n",
           nwg, n * 4, ms_graph * 1e3 / stages, bad_graph, ms_eager * 1e3 / stages, ms_p[0] * 1e3 / stages, bad_p[0], ms_p[1] * 1e3 / stages, bad_p[1],
           ms_p[2] * 1e3 / stages, bad_p[2]);
    CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    CHECK(hipFree(b0)); CHECK(hipFree(b1));
  }
  return 0;
}
