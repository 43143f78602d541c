// Micro-benchmark of emformer_fused_kernel with per-phase cycle stamps (-DEF_STAMPS): random weights, B streams.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DEF_STAMPS -I conan_amd/csrc tools/ef_bench.hip -o tools/bin/ef_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "emformer_fused.hip"

static float* dev(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, past0 = argc > 2 ? atoi(argv[2]) : 100;
  const int D = 80, F = 2048, L = 6, R = 2, U = 4, H = 8, LC = 50, K = 100, LR = 64;
  cnk::EmfFusedArgs a; memset(&a, 0, sizeof(a));
  for (int l = 0; l < L; ++l) {
    cnk::EmfLayerW& w = a.layers[l];
    w.wqkv = dev((size_t)3 * D * D, 0.2f); w.wo = dev((size_t)D * D, 0.2f); w.w1 = dev((size_t)D * F, 0.2f); w.w2 = dev((size_t)D * F, 0.2f);
    w.params = dev(11 * D + F, 0.5f);
    a.kring[l] = dev((size_t)B * LR * D, 1.f); a.vring[l] = dev((size_t)B * LR * D, 1.f);
  }
  a.ring_slot_stride = (long long)LR * D; a.lmask = LR - 1;
  a.wp = dev((size_t)112 * D, 0.2f); a.bp = dev(128, .1f);
  a.chunk = dev((size_t)B * (R + U) * D, 1.f);
  float* lg; hipMalloc(&lg, (size_t)B * U * K * 4); a.logits = lg;
  int* codes; hipMalloc(&codes, (size_t)B * U * 4); a.codes = codes;
  std::vector<int> hs(B), hp(B, past0);
  for (int i = 0; i < B; ++i) hs[i] = i;
  int *ds, *dp; hipMalloc(&ds, B * 4); hipMalloc(&dp, B * 4);
  hipMemcpy(ds, hs.data(), B * 4, hipMemcpyHostToDevice);
  a.slots = ds; a.past = dp;
  a.n = B; a.L = L; a.R = R; a.U = U; a.D = D; a.H = H; a.LC = LC; a.F = F; a.K = K; a.scaling = 1.f / sqrtf(10.f);
  { const unsigned long long per_g = (unsigned long long)LC * (D / 4); a.magic_per_g = (unsigned)(((1ull << 32) + per_g - 1) / per_g); }
#ifdef EF_STAMPS
  hipMalloc(&a.dbg, 64 * 8); hipMemset(a.dbg, 0, 64 * 8);
#endif
  a.cs = argc > 3 ? atoi(argv[3]) : 1;
  { const size_t xf = cnk::emformer_cluster_xch_floats(B, D), fw = cnk::emformer_cluster_flag_words(B);
    hipMalloc(&a.xch, xf * 4); unsigned* words; hipMalloc(&words, fw * 4); hipMemset(words, 0, fw * 4);
    a.xflag = words; a.xepoch = words + (size_t)B * cnk::EMF_MAX_LAYERS * cnk::EMF_MAX_CLUSTER; }
  if (!cnk::emformer_fused_supported(a)) { printf("unsupported\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) { hipMemcpy(dp, hp.data(), B * 4, hipMemcpyHostToDevice); cnk::launch_emformer_fused(a, 0); }
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int it = 0; it < 20; ++it) {
    hipMemcpy(dp, hp.data(), B * 4, hipMemcpyHostToDevice);
    hipEventRecord(e0, 0); cnk::launch_emformer_fused(a, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  { // reference run with one workgroup per group, same state
    std::vector<float> h1((size_t)B * U * K), h2(h1.size());
    const int cs_keep = a.cs;
    hipMemcpy(dp, hp.data(), B * 4, hipMemcpyHostToDevice); cnk::launch_emformer_fused(a, 0); hipMemcpy(h2.data(), lg, h2.size() * 4, hipMemcpyDeviceToHost);
    a.cs = 1; hipMemcpy(dp, hp.data(), B * 4, hipMemcpyHostToDevice); cnk::launch_emformer_fused(a, 0); hipMemcpy(h1.data(), lg, h1.size() * 4, hipMemcpyDeviceToHost);
    a.cs = cs_keep;
    double worst = 0, mx = 0; for (size_t i = 0; i < h1.size(); ++i) { worst = std::max(worst, (double)std::fabs(h1[i] - h2[i])); mx = std::max(mx, (double)std::fabs(h1[i])); }
    printf("cs=%d vs cs=1: max|diff| %.3e (max|ref| %.3f)\n", a.cs, worst, mx); }
  { std::vector<float> hl((size_t)B * U * K); hipMemcpy(hl.data(), lg, hl.size() * 4, hipMemcpyDeviceToHost);
    double cs_ = 0; for (float v : hl) cs_ += (double)v * v; printf("logits sum of squares %.9e (cs=%d)\n", cs_, a.cs); }
  printf("B=%d past=%d  kernel %.1f us (hipGetLastError=%d)\n", B, past0, best * 1e3f, (int)hipGetLastError());
#ifdef EF_STAMPS
  unsigned long long st[16]; hipMemcpy(st, a.dbg, sizeof(st), hipMemcpyDeviceToHost);
  const char* nm[] = {"top: prefetch+LN_in", "qkv gemm + tables", "attention", "out_proj + b1 issue", "LN_ff", "FFN chunks", "RED write", "combine"};
  for (int i = 0; i < 8; ++i) printf("  %-22s %8llu cycles\n", nm[i], st[i + 1] - st[i]);
  printf("  layer total            %8llu cycles\n", st[8] - st[0]);
  printf("  chunk 1: b2 issue %llu | FF1 %llu | barrier %llu | b1 issue %llu | FF2 %llu | barrier %llu\n", st[9] - st[15], st[10] - st[9], st[11] - st[10], st[12] - st[11], st[13] - st[12], st[14] - st[13]);
#endif
  return 0;
}
