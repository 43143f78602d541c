// Per-CU streaming bandwidth probe: one (or a few) blocks read a buffer with wave-wide float4 loads.
// hipcc -O3 --offload-arch=gfx950 tools/cu_bw.hip -o tools/bin/cu_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int UNR>
__global__ void stream_kernel(const float4* __restrict__ p, size_t n4, int reps, float* out, unsigned long long* cyc) {
  float4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i + (UNR - 1) * blockDim.x < n4; i += (size_t)UNR * blockDim.x) {
      float4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  float* out; hipMalloc(&out, 1 << 20); unsigned long long* cyc; hipMalloc(&cyc, 4096);
  for (size_t mb : {1, 2, 8, 64}) {
    const size_t bytes = mb << 20; float4* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
    for (int threads : {256, 512, 1024}) {
      for (int blocks : {1, 4}) {
        const int reps = mb <= 2 ? 8 : 2;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(threads), 0, 0, p, bytes / 16, reps, out, cyc);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stream_kernel<8>, dim3(blocks), dim3(threads), 0, 0, p, bytes / 16, reps, out, cyc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("buf %3zu MiB threads %4d blocks %d: %.1f GB/s per block, %.1f B/clk\n", mb, threads, blocks, bytes * reps / (ms * 1e6), (double)bytes * reps / c);
      }
    }
    hipFree(p);
  }
  return 0;
}
