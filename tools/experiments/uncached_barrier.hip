// Developer probe for the decoder megakernel: how can dependent phases inside ONE persistent launch hand activations to
// each other across a grid barrier (s_waitcnt vmcnt(0), agent-scope ticket, sc1 polling)?  (hipDeviceMallocUncached /
// hipDeviceMallocFinegrained allocations behaved exactly like hipMalloc here: every cross-XCD read was stale.)
// Per round every workgroup writes 4 KB of round-tagged values, passes the barrier, and checks the 4 KB that a workgroup
// on another XCD wrote.  Reports mismatches (stale reads) and the time per round for normal and uncached memory.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/uncached_barrier.hip -o tools/bin/uncached_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: plain stores / loads; 1: sc1 (write-through) stores + sc1 loads, no fence; 2: plain + release / acquire fences at the barrier;
// 3: sc1 stores + plain loads behind an acquire fence
template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned* counter, float4* buf, unsigned* bad, int rounds, unsigned base) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  unsigned nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    const float tag = (float)(r * 1000 + b);
    const float4 sv = make_float4(tag, tag + 0.25f, (float)tid, tag - 1.f);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
    if (MODE == 1 || MODE == 3) __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(sv.x), __float_as_uint(sv.y), __float_as_uint(sv.z), __float_as_uint(sv.w)}, rs, (b * 256 + tid) * 16, 0, 16);
    else buf[(size_t)b * 256 + tid] = sv;       // plain 16-byte stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + (unsigned)(r + 1) * (unsigned)G;
      while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const int o = (b + 3) % G;                       // a workgroup on another XCD (round-robin placement)
    if (MODE == 2 || MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float4 v;
    if (MODE == 1) { const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (o * 256 + tid) * 16, 0, 16); v = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])); }
    else v = buf[(size_t)o * 256 + tid];     // plain 16-byte load
    const float want = (float)(r * 1000 + o);
    if (v.x != want || v.y != want + 0.25f || v.z != (float)tid || v.w != want - 1.f) ++nbad;
    // second barrier: nobody overwrites before everyone has read
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + (unsigned)(r + 1) * (unsigned)G;
      while ((int)(__hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned* counter; unsigned* bad;
  CHECK(hipMalloc(&counter, 8)); CHECK(hipMemset(counter, 0, 8));
  CHECK(hipMalloc(&bad, 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int mode = 0; mode < 4; ++mode) {
    float4* buf = nullptr;
    const size_t bytes = 256 * 256 * sizeof(float4);
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes));
    unsigned base = 0;
    CHECK(hipMemset(counter, 0, 8));
    for (int G : {32, 128, 256}) {
      CHECK(hipMemset(bad, 0, 4));
      const int rounds = 2000;
      CHECK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(rounds_kernel<0>, dim3(G), dim3(256), 0, 0, counter, buf, bad, rounds, base);
      if (mode == 1) hipLaunchKernelGGL(rounds_kernel<1>, dim3(G), dim3(256), 0, 0, counter, buf, bad, rounds, base);
      if (mode == 2) hipLaunchKernelGGL(rounds_kernel<2>, dim3(G), dim3(256), 0, 0, counter, buf, bad, rounds, base);
      if (mode == 3) hipLaunchKernelGGL(rounds_kernel<3>, dim3(G), dim3(256), 0, 0, counter, buf, bad, rounds, base);
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      base += (unsigned)rounds * G;
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      unsigned hb = 0; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("%-12s G=%3d: %u stale reads in %d rounds, %.2f us per round (two barriers)\n", mode == 0 ? "plain" : (mode == 1 ? "sc1 st + ld" : (mode == 2 ? "fences" : "sc1 st, inv")), G, hb, rounds, ms * 1e3f / rounds);
    }
    CHECK(hipFree(buf));
  }
  return 0;
}
