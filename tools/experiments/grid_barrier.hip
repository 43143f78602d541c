// Developer probe: what a device-wide barrier between dependent phases costs inside ONE persistent launch, against the
// ~4 us of a launch boundary (B = 1 latency is 62 dependent launches).  G workgroups (all resident), N rounds of
//   phase work (a short dependent chain per lane) -> store one float per workgroup with write-through -> barrier
//   (agent-scope ticket counter, relaxed polling through sc1 loads) -> read every workgroup's float.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/grid_barrier.hip -o tools/bin/grid_barrier && tools/bin/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rounds_kernel(unsigned* counter, float* xch, float* out, int rounds, unsigned base) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  float v = (float)(b + 1);
  for (int r = 0; r < rounds; ++r) {
    if (tid == 0) __hip_atomic_store(xch + (size_t)(r & 1) * G + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + (unsigned)(r + 1) * (unsigned)G;
      while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < G; i += 256) s += __hip_atomic_load(xch + (size_t)(r & 1) * G + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    v = s * 1e-3f + (float)b;
  }
  if (tid == 0) out[b] = v;
}
__global__ void empty_kernel(float* out) { if (threadIdx.x == 1024) out[0] = 1.f; }

int main() {
  unsigned* counter; float *xch, *out;
  CHECK(hipMalloc(&counter, 4)); CHECK(hipMemset(counter, 0, 4));
  CHECK(hipMalloc(&xch, 2 * 256 * 4)); CHECK(hipMalloc(&out, 256 * 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  unsigned base = 0;
  for (int G : {8, 32, 64, 128, 256}) {
    for (int rounds : {1, 101}) {
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(rounds_kernel, dim3(G), dim3(256), 0, 0, counter, xch, out, rounds, base);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        base += (unsigned)rounds * G;
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
      }
      printf("G=%3d rounds=%3d  %.2f us%s\n", G, rounds, best * 1e3f, rounds > 1 ? "" : "  (launch + one round)");
    }
  }
  {  // 100 dependent empty launches for comparison
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      CHECK(hipEventRecord(a));
      for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, 0, out);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
    }
    printf("100 dependent empty launches: %.2f us each\n", best * 10.f);
  }
  return 0;
}
