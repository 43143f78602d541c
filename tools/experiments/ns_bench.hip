// Developer tool: correctness (against a scalar CPU restatement on sampled outputs) and timing of cnk::conv_ns_kernel
// on the shapes of the first vocoder stage (three branches k = 3 / 7 / 11, C = 256, 32 rows per stream).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I conan_amd/csrc -I tools/experiments tools/experiments/ns_bench.hip tools/experiments/conv_ns.hip -o tools/bin/ns_bench
//   tools/bin/ns_bench [B=64] [T=32] [C=256] [iters=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "conv_ns.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct HostConv { std::vector<float> w, b; int Cin, Cout, k; };   // w[co][ci][j]

static std::vector<float> pack_frag(const HostConv& c) {
  const int KQ = c.Cin / 16, NCT = c.Cout / 16, k = c.k;
  std::vector<float> out((size_t)NCT * (k + 1) * KQ * 256, 0.f);
  for (int ct = 0; ct < NCT; ++ct)
    for (int j = 0; j < k; ++j)
      for (int q = 0; q < KQ; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int ci = q * 16 + 4 * (lane >> 4) + s, co = ct * 16 + (lane & 15);
            out[(((size_t)ct * (k + 1) + j) * KQ + q) * 256 + lane * 4 + s] = c.w[((size_t)co * c.Cin + ci) * k + j];
          }
  return out;
}
static float lrelu(float v, float s) { return v > 0.f ? v : v * s; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 256, iters = argc > 4 ? atoi(argv[4]) : 20;
  const int ks[3] = {3, 7, 11}, dils[3] = {1, 3, 5};
  const float slope = 0.1f;
  std::mt19937 rng(11);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  int num_cu = 256;
  { hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); num_cu = p.multiProcessorCount; }
  for (int mode = 0; mode < 2; ++mode)          // 0: c1-like (LeakyReLU in and out), 1: c2-like (residual, activated twin)
    for (int di = 0; di < 3; ++di) {
      const int d = mode == 0 ? dils[di] : 1;
      if (mode == 1 && di > 0) break;
      const int hist = 10 * d + 8;
      int L = 1; while (L < hist + T + 8) L <<= 1;
      const long long ss = (long long)L * C;
      const int rate = T;                         // rows per step
      std::vector<float> hx((size_t)3 * B * ss), hy((size_t)3 * B * ss, 0.f), hr((size_t)B * ss);
      for (auto& v : hx) v = U(rng);
      for (auto& v : hr) v = U(rng);
      std::vector<int> hslots(B), hpos(B);
      for (int i = 0; i < B; ++i) { hslots[i] = (i * 7 + 3) % B; hpos[i] = 2 + (i % 3); }
      float *dx, *dy, *dy2, *dr; int *dslots, *dpos;
      CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dy, hy.size() * 4)); CHECK(hipMalloc(&dy2, hy.size() * 4)); CHECK(hipMalloc(&dr, hr.size() * 4));
      CHECK(hipMalloc(&dslots, B * 4)); CHECK(hipMalloc(&dpos, B * 4));
      CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(dy, 0, hy.size() * 4)); CHECK(hipMemset(dy2, 0, hy.size() * 4));
      CHECK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dslots, hslots.data(), B * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpos, hpos.data(), B * 4, hipMemcpyHostToDevice));
      HostConv cv[3];
      cnk::NSArgs a; memset(&a, 0, sizeof(a));
      for (int b = 0; b < 3; ++b) {
        HostConv& c = cv[b];
        c.Cin = C; c.Cout = C; c.k = ks[b]; c.w.resize((size_t)C * C * ks[b]); c.b.resize(C);
        const float sc = 1.7f / std::sqrt((float)C * ks[b]);
        for (auto& v : c.w) v = U(rng) * sc;
        for (auto& v : c.b) v = U(rng) * 0.1f;
        auto up = [&](const std::vector<float>& v) { float* p; CHECK(hipMalloc(&p, v.size() * 4)); CHECK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return p; };
        a.p[b].w = up(pack_frag(c)); a.p[b].bias = up(c.b);
        cnk::TRef r; r.base = dx + (size_t)b * B * ss; r.slot_stride = ss; r.C = C; r.lmask = L - 1; r.rate = rate; r.off = 0; r.mode = 0; r.pad_ = 0;
        a.p[b].x = r; r.base = dy + (size_t)b * B * ss; a.p[b].y = r;
        r.base = dr; a.p[b].res = r; a.p[b].has_res = mode == 1;
        a.p[b].y2_base = mode == 1 ? dy2 + (size_t)b * B * ss : nullptr;
        a.p[b].k = ks[b]; a.p[b].dil = d;
      }
      a.slots = dslots; a.pos = dpos; a.nprob = 3; a.n = B; a.T = T; a.Cin = C; a.Cout = C; a.SR = 0; a.rows32 = getenv("NS_ROWS32") ? 1 : 0;
      a.in_slope = mode == 0 ? slope : 1.0f; a.out_act = mode == 0 ? cnk::ACT_LRELU : cnk::ACT_NONE; a.out_slope = slope; a.y2_slope = slope; a.shuffle_r = 1;
      int* dsched; CHECK(hipMalloc(&dsched, 8)); CHECK(hipMemset(dsched, 0, 8)); a.sched = dsched;
#ifdef NS_STAMPS
      unsigned long long* ddbg; CHECK(hipMalloc(&ddbg, 256 * 6 * 8)); CHECK(hipMemset(ddbg, 0, 256 * 6 * 8)); a.dbg = ddbg;
#endif
      if (!cnk::launch_conv_ns(a, num_cu, 0)) { printf("launch failed (unsupported shape)\n"); return 1; }
      CHECK(hipDeviceSynchronize());
      std::vector<float> hy2(hy.size());
      CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hy2.data(), dy2, hy2.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0.0, scale = 0.0, worst2 = 0.0;
      std::uniform_int_distribution<int> Ui(0, B - 1), Ut(0, T - 1), Uc(0, C - 1);
      for (int smp = 0; smp < 200; ++smp) {
        const int b = smp % 3, i = smp < 12 ? (smp * 5) % B : (smp < 24 ? B - 1 - smp % 3 : Ui(rng)), t = smp < 12 ? smp % 7 : (smp < 24 ? T - 1 - smp % 5 : Ut(rng)), co = Uc(rng);
        const int slot = hslots[i], pos = hpos[slot], k = ks[b];
        const float* xr = hx.data() + (size_t)b * B * ss + (size_t)slot * ss;
        auto row = [&](long long tt) { return (size_t)(((long long)pos * rate + tt) & (L - 1)) * C; };
        double s = 0.0;
        for (int j = 0; j < k; ++j) {
          const float* xx = xr + row(t - (long long)(k - 1 - j) * d);
          for (int ci = 0; ci < C; ++ci) s += (double)cv[b].w[((size_t)co * C + ci) * k + j] * (mode == 0 ? lrelu(xx[ci], slope) : xx[ci]);
        }
        double want = s + cv[b].b[co];
        if (mode == 0) want = want > 0 ? want : want * slope;
        else want += hr[(size_t)slot * ss + row(t) + co];
        const size_t oi = (size_t)b * B * ss + (size_t)slot * ss + row(t) + co;
        worst = std::max(worst, std::fabs(want - hy[oi])); scale = std::max(scale, std::fabs(want));
        if (mode == 1) worst2 = std::max(worst2, std::fabs((want > 0 ? want : want * slope) - hy2[oi]));
      }
      long long stray = 0;     // rows outside [0, T) of the step must stay untouched (zeros)
      for (int b = 0; b < 3; ++b) for (int i = 0; i < B; i += 5) {
        const int slot = hslots[i], pos = hpos[slot];
        for (int tt = T; tt < T + 6; ++tt) for (int c = 0; c < C; c += 7)
          if (hy[(size_t)b * B * ss + (size_t)slot * ss + (size_t)(((long long)pos * rate + tt) & (L - 1)) * C + c] != 0.f) ++stray;
      }
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      for (int w = 0; w < 3; ++w) cnk::launch_conv_ns(a, num_cu, 0);
      CHECK(hipEventRecord(e0, 0));
      for (int w = 0; w < iters; ++w) cnk::launch_conv_ns(a, num_cu, 0);
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      const double flops = 2.0 * 21.0 * C * C * (double)B * T;
      printf("%s C=%d B=%d T=%d dil=%d: max|err| %.2e twin %.2e (max|ref| %.2f) stray=%lld  %7.1f us  %6.1f TFLOP/s (%.1f%% of 157.3)\n", mode == 0 ? "c1-like" : "c2-like", C, B, T, d,
             worst, worst2, scale, stray, ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3 * 100);
#ifdef NS_STAMPS
      { std::vector<unsigned long long> hd(256 * 6); CHECK(hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost));
        double bw = 0, life = 0, rt = 0, first = 0, nb = 0, nt = 0, rmax = 0; int cnt = 0;
        for (int b = 0; b < 256; ++b) if (hd[b * 6 + 1]) { bw += hd[b * 6]; life += hd[b * 6 + 1]; rt += hd[b * 6 + 2]; first += hd[b * 6 + 3]; nb += hd[b * 6 + 4]; nt += hd[b * 6 + 5]; rmax = std::max(rmax, (double)hd[b * 6 + 2]); ++cnt; }
        printf("   stamps (%d blocks): life %.0f cyc = %.1f us avg / %.1f us max, clock %.2f GHz; first barrier %.0f cyc, other barriers %.1f%% of life (%.0f cyc per barrier), %.1f tiles, %.1f barriers per block\n",
               cnt, life / cnt, rt / cnt / 100.0, rmax / 100.0, (life / cnt) / (rt / cnt / 100.0) / 1e3, first / cnt, 100 * bw / life, bw / (nb - cnt), nt / cnt, nb / cnt); }
#endif
      CHECK(hipFree(dx)); CHECK(hipFree(dy)); CHECK(hipFree(dy2)); CHECK(hipFree(dr)); CHECK(hipFree(dslots)); CHECK(hipFree(dpos));
    }
  return 0;
}
