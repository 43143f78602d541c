// Developer tool: correctness (against a scalar CPU restatement on sampled outputs) and timing of
// cnk::resblock_fused_kernel on the vocoder's stage shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I conan_amd/csrc tools/rb_bench.hip conan_amd/csrc/resblock_fused.hip conan_amd/csrc/resblock_limb.hip -o tools/bin/rb_bench
//   tools/bin/rb_bench [B=64] [frames=4] [iters=20]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "kernels.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct HostConv { std::vector<float> w, b; int C, k; };   // w[co][ci][j] (PyTorch Conv1d layout)

static std::vector<float> pack_frag(const HostConv& c) {
  const int C = c.C, k = c.k, KQ = C / 16, NCT = C / 16;
  std::vector<float> out((size_t)NCT * (k + 1) * KQ * 256, 0.f);
  for (int ct = 0; ct < NCT; ++ct)
    for (int j = 0; j < k; ++j)
      for (int q = 0; q < KQ; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int ci = q * 16 + 4 * (lane >> 4) + s, co = ct * 16 + (lane & 15);
            out[(((size_t)ct * (k + 1) + j) * KQ + q) * 256 + lane * 4 + s] = c.w[((size_t)co * C + ci) * k + j];
          }
  return out;
}

// bf16 limbs of the same weights (resblock_limb.hip): [ct][k + 1 taps][C/32 K blocks][3 limbs][64 lanes][8]
static unsigned short bf16_rne(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static std::vector<unsigned short> pack_limb(const HostConv& c) {
  const int C = c.C, k = c.k, KB = C / 32, NCT = C / 16;
  std::vector<unsigned short> out((size_t)NCT * (k + 1) * KB * 3 * 512, 0);
  for (int ct = 0; ct < NCT; ++ct)
    for (int j = 0; j < k; ++j)
      for (int q = 0; q < KB; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int ci = q * 32 + 8 * (lane >> 4) + e, co = ct * 16 + (lane & 15);
            const float w = c.w[((size_t)co * C + ci) * k + j];
            const unsigned short h = bf16_rne(w); const float r1 = w - bf16_f(h);
            const unsigned short m = bf16_rne(r1); const float r2 = r1 - bf16_f(m);
            const unsigned short l = bf16_rne(r2);
            const size_t base = ((((size_t)ct * (k + 1) + j) * KB + q) * 3) * 512 + lane * 8 + e;
            out[base] = h; out[base + 512] = m; out[base + 1024] = l;
          }
  return out;
}

static float lrelu(float v, float s) { return v > 0.f ? v : v * s; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, frames = argc > 2 ? atoi(argv[2]) : 4, iters = argc > 3 ? atoi(argv[3]) : 20;
  struct Stage { int C, rate; } stages[] = {{128, 40}, {64, 160}, {32, 320}};
  const int ks[3] = {3, 7, 11}, dils[3] = {1, 3, 5};
  const float slope = 0.1f;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  int num_cu = 256;
  { hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); num_cu = p.multiProcessorCount; }
  for (const Stage& sg : stages) {
    const int C = sg.C, T = frames * sg.rate;
    for (int di = 0; di < 3; ++di) {
      const int d = dils[di];
      const int hist = 10 * (d + 1);
      int L = 1; while (L < hist + T + 64) L <<= 1;
      const long long ss = (long long)L * C + (getenv("RB_PAD") ? atoi(getenv("RB_PAD")) : 0);
      // rings: x per branch (distinct tensors, like xo[b][d-1]), y per branch
      std::vector<float> hx((size_t)3 * B * ss), hy((size_t)3 * B * ss, 0.f);
      for (auto& v : hx) v = U(rng);
      std::vector<int> hslots(B), hpos(B);
      for (int i = 0; i < B; ++i) { hslots[i] = (i * 7 + 3) % B; hpos[i] = (i % 5 == 0) ? 0 : 3 + i; }   // pos 0: rows before the stream start
      // note B coprime with 7 for the bench shapes (64, 128, 1, 4)
      float *dx, *dy; int *dslots, *dpos;
      CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dy, hy.size() * 4));
      CHECK(hipMalloc(&dslots, B * 4)); CHECK(hipMalloc(&dpos, B * 4));
      CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(dy, 0, hy.size() * 4));
      CHECK(hipMemcpy(dslots, hslots.data(), B * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpos, hpos.data(), B * 4, hipMemcpyHostToDevice));
      HostConv c1[3], c2[3];
      cnk::RBArgs a; memset(&a, 0, sizeof(a));
      for (int b = 0; b < 3; ++b) {
        for (HostConv* c : {&c1[b], &c2[b]}) {
          c->C = C; c->k = ks[b]; c->w.resize((size_t)C * C * ks[b]); c->b.resize(C);
          const float sc = 1.0f / std::sqrt((float)C * ks[b]);
          for (auto& v : c->w) v = U(rng) * sc * 1.7f;
          for (auto& v : c->b) v = U(rng) * 0.1f;
        }
        auto up = [&](const std::vector<float>& v) { float* p; CHECK(hipMalloc(&p, v.size() * 4)); CHECK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return p; };
        a.p[b].w1 = up(pack_frag(c1[b])); a.p[b].w2 = up(pack_frag(c2[b])); a.p[b].b1 = up(c1[b].b); a.p[b].b2 = up(c2[b].b);
        auto up16 = [&](const std::vector<unsigned short>& v) { unsigned short* p; CHECK(hipMalloc(&p, v.size() * 2 + 65536)); CHECK(hipMemcpy(p, v.data(), v.size() * 2, hipMemcpyHostToDevice)); return p; };
        a.p[b].w1l = up16(pack_limb(c1[b])); a.p[b].w2l = up16(pack_limb(c2[b]));
        cnk::TRef r; r.base = dx + (size_t)b * B * ss; r.slot_stride = ss; r.C = C; r.lmask = L - 1; r.rate = sg.rate; r.off = 0; r.mode = 0; r.pad_ = 0;
        a.p[b].x = r; r.base = dy + (size_t)b * B * ss; a.p[b].y = r;
        a.p[b].k = ks[b]; a.p[b].dil = d;
      }
      a.slots = dslots; a.pos = dpos; a.nprob = 3; a.n = B; a.T = T; a.slope = slope;
      // RB_MERGE=1: merged-branch build (the three branches of a (slot, row tile) in one workgroup, only leaky_relu(mean) stored)
      const bool limb = getenv("RB_LIMB") != nullptr && cnk::resblock_limb_supported(C, 11, 10 * d);
      const bool merge = getenv("RB_MERGE") != nullptr && (limb ? cnk::resblock_limb_can_merge(C, cnk::resblock_limb_rows(C, 10 * d)) : cnk::resblock_fused_can_merge(C, cnk::resblock_fused_rows(C, T, B, 21, 11, num_cu)));
      float* dmean = nullptr;
      if (merge) { CHECK(hipMalloc(&dmean, (size_t)B * ss * 4)); CHECK(hipMemset(dmean, 0, (size_t)B * ss * 4)); a.merge = 1; a.ymean = a.p[0].y; a.ymean.base = dmean; }
      unsigned long long* ddbg = nullptr;
      CHECK(hipMalloc(&ddbg, 264 * 4 * 8)); CHECK(hipMemset(ddbg, 0, 264 * 4 * 8));
      a.dbg = ddbg;
      int* dsched; CHECK(hipMalloc(&dsched, 8)); CHECK(hipMemset(dsched, 0, 8)); a.sched = dsched;
      const int rows = limb ? cnk::resblock_limb_rows(C, 10 * d) : getenv("RB_ROWS") && C == 64 ? atoi(getenv("RB_ROWS")) : cnk::resblock_fused_rows(C, T, B, 21, 11, num_cu);
      auto launch = [&] { return limb ? cnk::launch_resblock_limb(a, C, rows, num_cu, 0) : cnk::launch_resblock_fused(a, C, rows, num_cu, 0); };
      if (!launch()) { printf("launch failed\n"); return 1; }
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
      std::vector<float> hmean;
      if (merge) { hmean.resize((size_t)B * ss); CHECK(hipMemcpy(hmean.data(), dmean, hmean.size() * 4, hipMemcpyDeviceToHost)); }
      // ---- sampled check against a scalar restatement
      double worst = 0.0, scale = 0.0;
      std::uniform_int_distribution<int> Ui(0, B - 1), Ut(0, T - 1), Uc(0, C - 1);
      for (int smp = 0; smp < 60; ++smp) {
        double wsum = 0.0;
        const int i = (smp < 12) ? (smp / 3) * 5 % B : Ui(rng), t = (smp < 12) ? smp % 7 : (smp < 24 ? T - 1 - smp % 5 : Ut(rng)), co = Uc(rng);
       for (int bb = 0; bb < (merge ? 3 : 1); ++bb) {
        const int b = merge ? bb : smp % 3;
        const int slot = hslots[i], pos = hpos[slot], k = ks[b];
        const float* xr = hx.data() + (size_t)b * B * ss + (size_t)slot * ss;
        auto xrow = [&](long long tt) { return xr + (size_t)(((long long)pos * sg.rate + tt) & (L - 1)) * C; };
        // xt rows t-(k-1)..t, all channels
        std::vector<float> xt((size_t)k * C);
        for (int m = 0; m < k; ++m) {
          const long long tm = t - (k - 1) + m;
          for (int c = 0; c < C; ++c) {
            double s = 0.0;
            for (int j = 0; j < k; ++j) {
              const float* xx = xrow(tm - (long long)(k - 1 - j) * d);
              for (int ci = 0; ci < C; ++ci) s += (double)c1[b].w[((size_t)c * C + ci) * k + j] * lrelu(xx[ci], slope);
            }
            float v = lrelu((float)s + c1[b].b[c], slope);
            if ((long long)pos * sg.rate + tm < 0) v = 0.f;
            xt[(size_t)m * C + c] = v;
          }
        }
        double s = 0.0;
        for (int j = 0; j < k; ++j)
          for (int ci = 0; ci < C; ++ci) s += (double)c2[b].w[((size_t)co * C + ci) * k + j] * xt[(size_t)j * C + ci];
        const double want = s + c2[b].b[co] + xrow(t)[co];
        if (!merge) {
          const float got = hy[(size_t)b * B * ss + (size_t)slot * ss + (size_t)(((long long)pos * sg.rate + t) & (L - 1)) * C + co];
          worst = std::max(worst, std::fabs(want - got)); scale = std::max(scale, std::fabs(want));
        } else {
          wsum += want;
          if (bb == 2) {
            const double wm = lrelu((float)(wsum / 3.0), slope);
            const float got = hmean[(size_t)slot * ss + (size_t)(((long long)pos * sg.rate + t) & (L - 1)) * C + co];
            worst = std::max(worst, std::fabs(wm - got)); scale = std::max(scale, std::fabs(wm));
          }
        }
       }
      }
      // rows outside [0, T) of the step must be untouched (zeros)
      long long stray = 0;
      for (int b = 0; b < 3; ++b) for (int i = 0; i < B; i += 9) {
        const int slot = hslots[i], pos = hpos[slot];
        for (int tt = T; tt < T + 20; ++tt) for (int c = 0; c < C; c += 5)
          if (hy[(size_t)b * B * ss + (size_t)slot * ss + (size_t)(((long long)pos * sg.rate + tt) & (L - 1)) * C + c] != 0.f) ++stray;
      }
      // ---- timing
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      for (int w = 0; w < 3; ++w) launch();
      CHECK(hipEventRecord(e0, 0));
      for (int w = 0; w < iters; ++w) launch();
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      const double flops = 2.0 * 2.0 * 21.0 * C * C * (double)B * T;
      printf("%s C=%3d T=%5d dil=%d rows/tile=%3d: max|err| %.2e (max|ref| %.2f) stray=%lld  %8.1f us  %6.1f TFLOP/s (%.1f%% of 157.3)\n", limb ? "limb" : "f32 ", C, T, d, rows, worst, scale, stray,
             ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3 * 100);
      {
        std::vector<unsigned long long> hd(264 * 4);      // (256 blocks x 4 + block 0's 12 extra words)
        CHECK(hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost));
        if (hd[1]) {
          double g = 0, t = 0, r = 0, bw = 0; int nb = 0; double tmax = 0;
          for (int b = 0; b < 256; ++b) if (hd[b * 4 + 1]) { g += hd[b * 4]; t += hd[b * 4 + 1]; r += hd[b * 4 + 2]; bw += hd[b * 4 + 3]; tmax = std::max(tmax, (double)hd[b * 4 + 2]); ++nb; }
          printf("   stamps (%d blocks): gemm %.0f cyc = %.1f%% of block life, barriers %.1f%%, life %.0f cyc = %.1f us avg / %.1f us max, clock %.2f GHz\n", nb, g / nb, 100 * g / t,
                 100 * bw / t, t / nb, r / nb / 100.0, tmax / 100.0, (t / nb) / (r / nb / 100.0) / 1e3);
          printf("   block 0 barrier cycles: B3 %llu  B1 %llu  B4 %llu  B2 %llu;  %llu tiles: c1 epilogue %llu  c2 epilogue %llu  gemm %llu  life %llu  -> other %lld\n", hd[1024], hd[1025], hd[1026], hd[1027],
                 hd[1030], hd[1028], hd[1029], hd[1031], hd[1032], (long long)hd[1032] - (long long)(hd[1024] + hd[1025] + hd[1026] + hd[1027] + hd[1028] + hd[1029] + hd[1031]));
          printf("   block 0 helper wave: table + draw issued %llu, B(-1) passed %llu, window loads issued %llu, (deep draw) %llu, window written %llu, B0 passed %llu\n", hd[1036], hd[1037], hd[1038], hd[1039], hd[1040], hd[1041]);
          printf("   block 0: %llu cycles before the first tile, %llu between a tile's start and its c1 K loop, %llu between B1 and c2's K loop (sums over its tiles)\n", hd[1033], hd[1034], hd[1035]);

        }
      }
      CHECK(hipFree(ddbg));
      CHECK(hipFree(dx)); CHECK(hipFree(dy)); CHECK(hipFree(dslots)); CHECK(hipFree(dpos));
    }
  }
  return 0;
}
