// Micro-benchmark of cnk::launch_conv on synthetic shapes (developer tool; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I conan_amd/csrc tools/conv_bench.hip conan_amd/csrc/conv_mfma.hip -o gpurun_out/conv_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "kernels.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int n, T, Cin, Cout, k, dil, nprob, cfg; };

int main(int argc, char** argv) {
  int ablate = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<Shape> shapes = {
    {"stage1 rb (B64: M=2048,C=256,k7)", 64, 32, 256, 256, 7, 3, 3, cnk::CFG_64x64},
    {"stage1 mixed k=3/7/11 64x64", 64, 32, 256, 256, -1, 3, 3, cnk::CFG_64x64},
    {"stage1 mixed k 64x64 KS64", 64, 32, 256, 256, -1, 3, 3, cnk::CFG_64x64_KS64},
    {"stage1 mixed k 32x64 K2", 64, 32, 256, 256, -1, 3, 3, cnk::CFG_32x64_K2},
    {"stage2 mixed k 64x64", 64, 160, 128, 128, -1, 3, 3, cnk::CFG_64x64},
    {"stage3 mixed k 64x64", 64, 640, 64, 64, -1, 3, 3, cnk::CFG_64x64},
    {"stage3 mixed k 128x64", 64, 640, 64, 64, -1, 3, 3, cnk::CFG_128x64},
    {"stage4 mixed k 128x32", 64, 1280, 32, 32, -1, 3, 3, cnk::CFG_128x32},
    {"stage2 rb (M=10240,C=128,k7)", 64, 160, 128, 128, 7, 3, 3, cnk::CFG_128x64},
    {"stage2 rb 64x64", 64, 160, 128, 128, 7, 3, 3, cnk::CFG_64x64},
    {"stage2 rb 64x64 KS64", 64, 160, 128, 128, 7, 3, 3, cnk::CFG_64x64_KS64},
    {"stage2 rb 128x64 KS64", 64, 160, 128, 128, 7, 3, 3, cnk::CFG_128x64},
    {"stage3 rb (M=40960,C=64,k7)", 64, 640, 64, 64, 7, 3, 3, cnk::CFG_128x64},
    {"stage3 rb 128x64 KS64", 64, 640, 64, 64, 7, 3, 3, cnk::CFG_128x64},
    {"stage3 rb 64x64 KS64", 64, 640, 64, 64, 7, 3, 3, cnk::CFG_64x64_KS64},
    {"stage4 rb (M=81920,C=32,k7)", 64, 1280, 32, 32, 7, 3, 3, cnk::CFG_128x32},
    {"stage4 rb 128x32 KS64... (Cin=32: n/a)", 64, 1280, 32, 32, 7, 3, 3, cnk::CFG_128x32},
    {"ups0 (M=256,Cin=512,N=2048,k16)", 64, 4, 512, 2048, 16, 1, 1, cnk::CFG_64x64},
    {"ups1 (M=2048,256->640,k10) 64x64", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_64x64},
    {"ups1 64x64 KS64", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_64x64_KS64},
    {"ups1 32x64 K2", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_32x64_K2},
    {"ups2 (M=10240,128->256,k8) 64x64", 64, 160, 128, 256, 8, 1, 1, cnk::CFG_64x64},
    {"ups3 (M=40960,64->64,k4) 64x64", 64, 640, 64, 64, 4, 1, 1, cnk::CFG_64x64},
    {"stage4 mixed 64x32 K2", 64, 1280, 32, 32, -1, 3, 3, cnk::CFG_64x32_K2},
    {"ups0 32x64", 64, 4, 512, 2048, 16, 1, 1, cnk::CFG_32x64_K2},
    {"dec c1 (M=256,256->512,k5)", 64, 4, 256, 512, 5, 1, 1, cnk::CFG_32x32_K4},
    {"emf ff2 (M=384,2048->80)", 64, 6, 2048, 80, 1, 1, 1, cnk::CFG_32x32_K4},
    {"ups1 64x32 K2", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_64x32_K2},
    {"ups1 128x32", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_128x32},
    {"ups1 128x64", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_128x64},
    {"ups1 32x32 K4", 64, 32, 256, 640, 10, 1, 1, cnk::CFG_32x32_K4},
    {"ups3 128x64", 64, 640, 64, 64, 4, 1, 1, cnk::CFG_128x64},
    {"ups3 128x32", 64, 640, 64, 64, 4, 1, 1, cnk::CFG_128x32},
    {"ups3 64x64 KS64", 64, 640, 64, 64, 4, 1, 1, cnk::CFG_64x64_KS64},
    {"ups2 64x64 KS64", 64, 160, 128, 256, 8, 1, 1, cnk::CFG_64x64_KS64},
    {"ups2 128x64", 64, 160, 128, 256, 8, 1, 1, cnk::CFG_128x64},
    {"ups0 32x32 K4", 64, 4, 512, 2048, 16, 1, 1, cnk::CFG_32x32_K4},
    {"ups0 64x64 KS64", 64, 4, 512, 2048, 16, 1, 1, cnk::CFG_64x64_KS64},
    {"conv_pre (M=256,80->512,k7) K4", 64, 4, 80, 512, 7, 1, 1, cnk::CFG_32x32_K4},
  };
  int nslots = 64;
  // CB_N=<streams>: the small-batch shapes of one vocoder step with the inter-block split-K of streams.hip (S = CUs / tiles, <= K-steps / 2, <= 16)
  const int cbn = getenv("CB_N") ? atoi(getenv("CB_N")) : 0;
  if (cbn > 0) {
    shapes = {
      {"conv_pre (80->512,k7)", cbn, 4, 80, 512, 7, 1, 1, cnk::CFG_32x32_K4},
      {"ups0 (512->2048,k16)", cbn, 4, 512, 2048, 16, 1, 1, cnk::CFG_32x32_K4},
      {"stage1 c1 mixed k (C=256)", cbn, 32, 256, 256, -1, 3, 3, cnk::CFG_32x32_K4},
      {"stage1 c1 k3 only (C=256)", cbn, 32, 256, 256, 3, 1, 1, cnk::CFG_32x32_K4},
      {"ups1 (256->640,k10)", cbn, 32, 256, 640, 10, 1, 1, cnk::CFG_32x32_K4},
      {"stage2 c1 mixed k (C=128)", cbn, 160, 128, 128, -1, 3, 3, cnk::CFG_32x32_K4},
      {"ups2 (128->256,k8)", cbn, 160, 128, 256, 8, 1, 1, cnk::CFG_32x32_K4},
      {"stage3 c1 mixed k (C=64)", cbn, 640, 64, 64, -1, 3, 3, cnk::CFG_32x32_K4},
    };
  }
  float* slab = nullptr; int* counters = nullptr;
  const size_t slab_floats = 256 * 16 * 32 * 32;
  CHECK(hipMalloc(&slab, slab_floats * 4)); CHECK(hipMalloc(&counters, 4096 * 4)); CHECK(hipMemset(counters, 0, 4096 * 4));
  for (auto& s : shapes) {
    const int L = 4096;  // ring rows (pow2) >= T + halo
    int Lr = 1; while (Lr < s.T + 64) Lr <<= 1;
    size_t xfl = (size_t)nslots * Lr * s.Cin, yfl = (size_t)nslots * Lr * s.Cout;
    float *x, *y, *w, *b; int *slots, *pos;
#ifdef CK_STAMPS
    unsigned long long* dbg; CHECK(hipMalloc(&dbg, (1100 + 512 * 8) * 8)); CHECK(hipMemset(dbg, 0, (1100 + 512 * 8) * 8));
#endif
    CHECK(hipMalloc(&x, xfl * 4 * 3)); CHECK(hipMalloc(&y, yfl * 4 * 3));
    int Cin_pad = (s.Cin + 31) / 32 * 32, Cin_alloc = (s.Cin + 127) / 128 * 128, Cout_pad = (s.Cout + 63) / 64 * 64;
    const int kmax = s.k < 0 ? 11 : s.k;
    size_t wfl = (size_t)kmax * Cin_alloc * Cout_pad;
    CHECK(hipMalloc(&w, wfl * 4 * 3)); CHECK(hipMalloc(&b, Cout_pad * 4));
    CHECK(hipMemset(x, 0, xfl * 4 * 3)); CHECK(hipMemset(w, 0, wfl * 4 * 3)); CHECK(hipMemset(b, 0, Cout_pad * 4));
    std::vector<float> hx(xfl); for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (int p = 0; p < 3; ++p) CHECK(hipMemcpy(x + p * xfl, hx.data(), xfl * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(wfl); for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    for (int p = 0; p < 3; ++p) CHECK(hipMemcpy(w + p * wfl, hw.data(), wfl * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&slots, nslots * 4)); CHECK(hipMalloc(&pos, nslots * 4));
    std::vector<int> hs(nslots); for (int i = 0; i < nslots; ++i) hs[i] = i;
    CHECK(hipMemcpy(slots, hs.data(), nslots * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(pos, 0, nslots * 4));
    cnk::ConvGroup g; memset(&g, 0, sizeof(g));
    for (int p = 0; p < s.nprob; ++p) {
      cnk::ConvArgs& a = g.p[p];
      cnk::TRef xr; xr.base = x + p * xfl; xr.slot_stride = (long long)Lr * s.Cin; xr.C = s.Cin; xr.lmask = Lr - 1; xr.rate = 1; xr.off = 0; xr.mode = 0; xr.pad_ = 0;
      cnk::TRef yr = xr; yr.base = y + p * yfl; yr.slot_stride = (long long)Lr * s.Cout; yr.C = s.Cout;
      a.x = xr; a.y = yr; a.res = xr; a.has_res = (s.Cin == s.Cout);
      a.w = w + p * wfl; a.bias = b; a.slots = slots; a.pos = pos;
      a.Cin = s.Cin; a.Cin_pad = Cin_pad; a.Cin_alloc = Cin_alloc; a.Cout = s.Cout; a.Cout_pad = Cout_pad; const int kk = s.k < 0 ? (p == 0 ? 3 : (p == 1 ? 7 : 11)) : s.k;
      a.ktaps = kk; a.dil = s.dil; a.pad_left = (kk - 1) * s.dil;
      a.T = s.T; a.n = s.n; a.in_act = getenv("CB_INACT") ? cnk::ACT_LRELU : cnk::ACT_NONE; a.in_slope = 0.1f; a.out_scale = 1.f; a.shuffle_r = 1;
#ifdef CK_STAMPS
      a.dbg = dbg;
#endif
    }
    if (cbn > 0) {
      long long tiles = 0; int nks = 0;
      for (int p = 0; p < s.nprob; ++p) { tiles += (long long)((s.n * s.T + 31) / 32) * ((s.Cout + 31) / 32); nks = std::max(nks, g.p[p].ktaps * ((Cin_pad + 127) / 128)); }
      int S = (int)(256 / std::max(1LL, tiles)); if (S > nks / 2) S = nks / 2; if (S > 16) S = 16;
      if (getenv("CB_S")) S = atoi(getenv("CB_S"));
      g.slab = slab; g.counters = counters; g.ksplit = S >= 2 ? S : 1; g.split_from = 0;
      printf("  tiles %lld K-steps %d split %d\n", tiles, nks, g.ksplit);
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) cnk::launch_conv(g, s.nprob, s.cfg, 0);
    CHECK(hipDeviceSynchronize());
    const int iters = 20;
    CHECK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; ++it) cnk::launch_conv(g, s.nprob, s.cfg, 0);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    double fl = 2.0 * s.n * s.T * (double)s.Cout * (s.k < 0 ? 21.0 / 3.0 : (double)s.k) * s.Cin * s.nprob;
    printf("%-40s cfg=%d  %8.1f us  %7.2f TFLOP/s (%.1f%% of 157.3)\n", s.name, s.cfg, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
#ifdef CK_STAMPS
    { unsigned long long h[16]; CHECK(hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost)); double n = (double)h[2];
      printf("      [block 1] matrix wave: barrier-wait/step=%.0f mfma/step=%.0f kloop=%llu epilogue=%llu | loader: vmcnt-wait/step=%.0f barrier/step=%.0f issue/step=%.0f (cycles); in-kernel clock %.2f GHz\n", h[0] / n, h[1] / n, h[3], h[4], h[8] / n, h[9] / n, h[10] / n, (double)h[5] / (double)h[6] * 0.1); }
    { std::vector<unsigned long long> h(1100 + 512 * 8); CHECK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      // split-K blocks: K loop end, partial stores acknowledged, ticket known, (reducers) partials summed, epilogue stores acknowledged - us after the block's start
      double sum[5] = {0}, rsum[5] = {0}; int nb = 0, nr = 0;
      for (int b = 0; b < 512; ++b) { const unsigned long long* r = &h[1100 + b * 8]; const unsigned long long st0 = h[32 + 2 * b];
        if (!st0 || !r[0] || !r[2]) continue;
        if (r[3] && r[4]) { for (int i = 0; i < 5; ++i) rsum[i] += (r[i] - st0) * 0.01; ++nr; } else { for (int i = 0; i < 3; ++i) sum[i] += (r[i] - st0) * 0.01; ++nb; } }
      if (nb) printf("      %3d non-reducers: K loop end %.2f  stores acked %.2f  ticket known %.2f\n", nb, sum[0] / nb, sum[1] / nb, sum[2] / nb);
      if (nr) printf("      %3d reducers:     K loop end %.2f  stores acked %.2f  ticket known %.2f  partials summed %.2f  epilogue stored %.2f\n", nr, rsum[0] / nr, rsum[1] / nr, rsum[2] / nr, rsum[3] / nr, rsum[4] / nr); }
    { std::vector<unsigned long long> h(32 + 1024); CHECK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> st, en; unsigned long long t0 = ~0ull;
      for (int b = 0; b < 512; ++b) if (h[32 + 2 * b]) t0 = std::min(t0, h[32 + 2 * b]);
      for (int b = 0; b < 512; ++b) if (h[32 + 2 * b]) { st.push_back((h[32 + 2 * b] - t0) * 0.01); en.push_back((h[33 + 2 * b] - t0) * 0.01); }
      if (!st.empty()) { std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); const size_t n = st.size();
        { double byx[8] = {0}; int nx[8] = {0}; double byq[8] = {0}; int nq[8] = {0};
          for (int b = 0; b < 512; ++b) if (h[32 + 2 * b]) { const double e = (h[33 + 2 * b] - t0) * 0.01; byx[b % 8] += e; nx[b % 8]++; byq[b / 64] += e; nq[b / 64]++; }
          printf("      mean end by blockIdx%%8:"); for (int i = 0; i < 8; ++i) printf(" %.1f", nx[i] ? byx[i] / nx[i] : 0.0);
          printf(" | by blockIdx/64:"); for (int i = 0; i < 8; ++i) printf(" %.1f", nq[i] ? byq[i] / nq[i] : 0.0); printf("\n"); }
        printf("      block timeline (us, %zu blocks): start p50 %.1f p90 %.1f max %.1f | end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", n, st[n / 2], st[n * 9 / 10], st[n - 1], en[0], en[n / 10], en[n / 2], en[n * 9 / 10], en[n - 1]); } }
#endif
    hipFree(x); hipFree(y); hipFree(w); hipFree(b); hipFree(slots); hipFree(pos);
  }
  return 0;
}
