// Ground-truth f32 MFMA issue rates on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int nacc, int blocks_per_cu, double flop_per_mfma) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(256);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)grid.x * 4 * iters * 4 * nacc;
  printf("%-28s acc=%d blocks/CU=%d: %8.2f ms %8.2f TFLOP/s\n", name, nacc, blocks_per_cu, ms, mf * flop_per_mfma / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int b = 1; b <= 4; b *= 2) {
    run("mfma_f32_32x32x2", k32<1>, 1, b, 4096.0);
    run("mfma_f32_32x32x2", k32<2>, 2, b, 4096.0);
    run("mfma_f32_32x32x2", k32<4>, 4, b, 4096.0);
    run("mfma_f32_16x16x4", k16<1>, 1, b, 2048.0);
    run("mfma_f32_16x16x4", k16<4>, 4, b, 2048.0);
  }
  return 0;
}
