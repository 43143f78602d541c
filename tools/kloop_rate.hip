// What does one matrix wave's K-step cost?  (developer tool)  One wave per SIMD runs the conv_mfma inner loop shape:
// per step [barrier] + 4 k-groups x [2 ds_read_b128 (prefetched one group ahead) + LeakyReLU on A + 4 dependent MFMAs].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool BAR, bool LDS, bool XF, int NACC>
__global__ __launch_bounds__(256) void kloop(float* out, int iters, float slope) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i % 7) - 3.f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int lane = threadIdx.x & 63;
  const float* ap = lds + lane * 4;
  const float* bp = lds + 4096 + lane * 4;
  float4 fa[2], fb[2];
  fa[0] = *(const float4*)ap; fb[0] = *(const float4*)bp;
  for (int it = 0; it < iters; ++it) {
    if (BAR) asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      if (LDS) { fa[(kc + 1) & 1] = *(const float4*)(ap + ((kc + 1 + it) & 7) * 256); fb[(kc + 1) & 1] = *(const float4*)(bp + ((kc + 1 + it) & 7) * 256); }
      else { fa[(kc + 1) & 1] = fa[kc & 1]; fb[(kc + 1) & 1] = fb[kc & 1]; }
      __builtin_amdgcn_sched_barrier(0);
      float4 a = fa[kc & 1]; const float4 b = fb[kc & 1];
      if (XF) { a.x *= a.x > 0.f ? 1.f : slope; a.y *= a.y > 0.f ? 1.f : slope; a.z *= a.z > 0.f ? 1.f : slope; a.w *= a.w > 0.f ? 1.f : slope; }
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int nacc, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(256);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, 0.1f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 0.1f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)grid.x * 4 * iters * 16 * nacc;
  printf("%-44s acc=%d blocks/CU=%d: %7.3f ms %7.2f TFLOP/s  (%.0f ns per 16-MFMA step per wave)\n", name, nacc, blocks_per_cu, ms, mf * 4096 / ms / 1e9, ms * 1e6 / iters);
  hipFree(out);
}
int main() {
  for (int b = 1; b <= 2; ++b) {
    run("regs only", kloop<false, false, false, 1>, 1, b);
    run("+ barrier", kloop<true, false, false, 1>, 1, b);
    run("+ LDS reads", kloop<true, true, false, 1>, 1, b);
    run("+ LeakyReLU on A  (= conv_mfma step)", kloop<true, true, true, 1>, 1, b);
    run("same, 2 accumulators (RM=2 shape)", kloop<true, true, true, 2>, 2, b);
    run("LDS + LeakyReLU, no barrier", kloop<false, true, true, 1>, 1, b);
  }
  return 0;
}
