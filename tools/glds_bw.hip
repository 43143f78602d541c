// Per-CU throughput of the two ways to stage a tile: global_load_lds (direct) vs global_load + ds_write.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void stage_kernel(const float* __restrict__ p, size_t nfloats, int iters, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[UNR * 256 * 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t span = (size_t)UNR * 1024;      // floats per block-iteration
  size_t off = ((size_t)blockIdx.x * 7919 * span) % (nfloats - span * 2);
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* src = p + off + tid * 4;
    if (MODE == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int u = 0; u < UNR; ++u) __builtin_amdgcn_global_load_lds(src + u * 1024, lds + (u * 4 + wave) * 256, 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    } else {
      float4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = *reinterpret_cast<const float4*>(src + u * 1024);
#pragma unroll
      for (int u = 0; u < UNR; ++u) *reinterpret_cast<float4*>(lds + (u * 256 + tid) * 4) = v[u];
    }
    off += span; if (off + span * 2 > nfloats) off = 0;
    if ((it & 63) == 63) { __syncthreads(); acc += lds[(tid * 5) & (UNR * 1024 - 1)]; __syncthreads(); }
  }
  out[blockIdx.x * 256 + tid] = acc;
}
template <int MODE, int UNR>
void run(const char* name, const float* p, size_t nfloats, int blocks, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stage_kernel<MODE, UNR>), dim3(blocks), dim3(256), 0, 0, p, nfloats, 100, out);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((stage_kernel<MODE, UNR>), dim3(blocks), dim3(256), 0, 0, p, nfloats, iters, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * iters * UNR * 4096.0;
  printf("%-26s unroll %d blocks %4d: %8.1f GB/s total, %6.1f GB/s per block\n", name, UNR, blocks, bytes / ms / 1e6, bytes / ms / 1e6 / blocks);
}
int main() {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  for (size_t mb : {2, 16, 64}) {
    const size_t nfloats = (mb << 20) / 4;
    float* p; hipMalloc(&p, nfloats * 4); hipMemset(p, 0, nfloats * 4);
    printf("--- buffer %zu MiB\n", mb);
    for (int blocks : {1, 256, 512, 1024}) {
      run<0, 4>("global_load_lds b128", p, nfloats, blocks, out);
      run<0, 8>("global_load_lds b128", p, nfloats, blocks, out);
      run<1, 8>("global_load + ds_write", p, nfloats, blocks, out);
    }
    hipFree(p);
  }
  return 0;
}
