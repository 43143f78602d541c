// Developer tool: timing (and a sampled check against a scalar restatement) of cnk::conv_limb_kernel on the C = 256 ResBlock
// convs of the vocoder's first stage: three problems (3 / 7 / 11 taps) per launch, 64 streams x 32 rows, ring inputs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I conan_amd/csrc tools/cl_bench.hip conan_amd/csrc/conv_limb.hip -o tools/bin/cl_bench
//   tools/bin/cl_bench [streams=64] [dil=5] [iters=20] [shape=-1: chosen] [residual=0: c1 form (LeakyReLU out) | 1: c2 form (+ x, raw out)]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "kernels.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short bf16_rne(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

// [Cout/16 column tiles][Cin/32 channel blocks][k taps][3 limbs][64 lanes][8] (ctx.hip pack_conv)
static std::vector<unsigned short> pack_limb(const std::vector<float>& W, int Cout, int Cin, int k) {
  const int NCT = Cout / 16, NCB = Cin / 32;
  std::vector<unsigned short> out((size_t)NCT * NCB * k * 3 * 512 + 4096, 0);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int j = 0; j < k; ++j) {
        const int ct = co / 16, ln = co % 16, cb = ci / 32, lane = ln + 16 * ((ci % 32) / 8), e = ci % 8;
        const float w = W[((size_t)co * Cin + ci) * k + j];
        const unsigned short h = bf16_rne(w); const float r1 = w - bf16_f(h);
        const unsigned short m = bf16_rne(r1); const float r2 = r1 - bf16_f(m);
        const unsigned short l = bf16_rne(r2);
        const size_t base = ((((size_t)ct * NCB + cb) * k + j) * 3) * 512 + (size_t)lane * 8 + e;
        out[base] = h; out[base + 512] = m; out[base + 1024] = l;
      }
  return out;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, dil = argc > 2 ? atoi(argv[2]) : 5, iters = argc > 3 ? atoi(argv[3]) : 20;
  const bool resid = argc > 5 && atoi(argv[5]) != 0;
  const int C = 256, T = 32, rate = 8, ks[3] = {3, 7, 11};
  const float slope = 0.1f;
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  int num_cu = 256;
  { hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); num_cu = p.multiProcessorCount; }
  int L = 1; while (L < 64 + T + 64) L <<= 1;
  const long long ss = (long long)L * C;
  std::vector<float> hx((size_t)3 * B * ss), hy((size_t)3 * B * ss, 0.f);
  for (auto& v : hx) v = U(rng);
  std::vector<int> hslots(B), hpos(B);
  for (int i = 0; i < B; ++i) { hslots[i] = (i * 7 + 3) % B; hpos[i] = 3 + i % 9; }
  float *dx, *dy; int *dslots, *dpos;
  CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dy, hy.size() * 4));
  CHECK(hipMalloc(&dslots, B * 4)); CHECK(hipMalloc(&dpos, B * 4));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(dy, 0, hy.size() * 4));
  CHECK(hipMemcpy(dslots, hslots.data(), B * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpos, hpos.data(), B * 4, hipMemcpyHostToDevice));
  cnk::ConvLimbGroup g; memset(&g, 0, sizeof(g));
  std::vector<float> W[3], bias[3];
  for (int b = 0; b < 3; ++b) {
    const int k = ks[b];
    W[b].resize((size_t)C * C * k); bias[b].resize(C);
    const float sc = 1.7f / std::sqrt((float)C * k);
    for (auto& v : W[b]) v = U(rng) * sc;
    for (auto& v : bias[b]) v = U(rng) * 0.1f;
    const auto wl = pack_limb(W[b], C, C, k);
    unsigned short* dwl; float* db;
    CHECK(hipMalloc(&dwl, wl.size() * 2)); CHECK(hipMemcpy(dwl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&db, C * 4)); CHECK(hipMemcpy(db, bias[b].data(), C * 4, hipMemcpyHostToDevice));
    cnk::ConvArgs& a = g.p[b];
    cnk::TRef r; r.base = dx + (size_t)b * B * ss; r.slot_stride = ss; r.C = C; r.lmask = L - 1; r.rate = rate; r.off = 0; r.mode = 0; r.pad_ = 0;
    a.x = r; r.base = dy + (size_t)b * B * ss; a.y = r;
    a.wl = dwl; a.bias = db; a.slots = dslots; a.pos = dpos;
    a.Cin = C; a.Cin_pad = C; a.Cin_alloc = C; a.Cout = C; a.Cout_pad = C; a.ktaps = k; a.dil = dil; a.pad_left = (k - 1) * dil;
    a.T = T; a.n = B; a.in_act = cnk::ACT_LRELU; a.in_slope = slope; a.out_act = resid ? cnk::ACT_NONE : cnk::ACT_LRELU; a.out_slope = slope; a.out_scale = 1.f; a.shuffle_r = 1;
    if (resid) { a.res = a.x; a.has_res = 1; }
  }
  g.nprob = 3;
  const int shape = (argc > 4 && atoi(argv[4]) >= 0) ? atoi(argv[4]) : cnk::conv_limb_shape(g.p, 3, num_cu);
  printf("shape %d (%s), %d streams, dil %d\n", shape, cnk::conv_limb_name(shape), B, dil);
  if (!cnk::launch_conv_limb(g, shape, num_cu, 0)) { printf("launch failed\n"); return 1; }
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  std::uniform_int_distribution<int> Ui(0, B - 1), Ut(0, T - 1), Uc(0, C - 1);
  for (int smp = 0; smp < 90; ++smp) {
    const int b = smp % 3, i = Ui(rng), t = Ut(rng), co = Uc(rng), k = ks[b], slot = hslots[i], pos = hpos[slot];
    const float* xr = hx.data() + (size_t)b * B * ss + (size_t)slot * ss;
    double s = 0;
    for (int j = 0; j < k; ++j) {
      const float* xx = xr + (size_t)(((long long)pos * rate + t + j * dil - (k - 1) * dil) & (L - 1)) * C;
      for (int ci = 0; ci < C; ++ci) { const float v = xx[ci] > 0.f ? xx[ci] : xx[ci] * slope; s += (double)W[b][((size_t)co * C + ci) * k + j] * v; }
    }
    float want = (float)s + bias[b][co];
    if (resid) want += xr[(size_t)(((long long)pos * rate + t) & (L - 1)) * C + co]; else want = want > 0.f ? want : want * slope;
    const float got = hy[(size_t)b * B * ss + (size_t)slot * ss + (size_t)(((long long)pos * rate + t) & (L - 1)) * C + co];
    worst = std::max(worst, (double)std::fabs(want - got)); scale = std::max(scale, (double)std::fabs(want));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) cnk::launch_conv_limb(g, shape, num_cu, 0);
  CHECK(hipEventRecord(e0, 0));
  for (int w = 0; w < iters; ++w) cnk::launch_conv_limb(g, shape, num_cu, 0);
  CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
  float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  const double flops = 2.0 * 21.0 * C * C * (double)B * T;
  printf("max|err| %.2e (max|ref| %.2f)  %7.1f us per launch  %6.1f TFLOP/s\n", worst, scale, ms * 1e3, flops / (ms * 1e-3) / 1e12);
  return 0;
}
