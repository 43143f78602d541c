// conv_mfma: channel-last causal / shifted 1-D convolution as an implicit GEMM on the gfx950
// f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
// GEMM view:  M = n*T output rows (slot, time), N = Cout, K = ktaps*Cin.
//   A[m][(j,ci)] = f(x[slot][t + j*dil - pad_left][ci])   gathered per tap from the activation rings
//   B[(j,ci)][co] = packed weights [tap][ci/4][co][4]
// Block = 256 threads (4 waves on the 4 SIMDs of a CU).  Per K-step (one tap, 32 input channels) the
// block stages an A tile [TM][32] and a W tile [32][TN] through LDS (register prefetch of step s+1
// while step s runs on the matrix pipe; two LDS buffers, one barrier per step).  Fragments are read
// with ds_read_b128: a K-chunk of 8 feeds 4 MFMAs, lanes 0-31 carrying k 0..3 and lanes 32-63 k 4..7
// (the MFMA's two k-slots), so one 16-byte LDS read per operand serves 4 matrix instructions.
// A rows are padded to 36 floats: bank-conflict-free for the b128 lane groups.
// Output tile D[time][co] keeps co on the lane -> 128-byte coalesced channel-last stores; the pixel
// shuffle of CausalUpsampleBlock3 is a pure address remap of that store (weights pre-permuted).
#include "kernels.h"

namespace ck {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ unsigned tref_row(const TRef& r, int slot, const int* pos, int t) {
  if (r.mode == 0) {
    unsigned p = pos ? (unsigned)pos[slot] : 0u;
    return (p * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask;
  }
  return (unsigned)(r.off + t);
}

template <int TM, int TN, int WM, int WN, int WK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvGroup g) {
  static_assert(WM * WN * WK == 4, "4 waves per block");
  constexpr int RM = TM / WM / 32;
  constexpr int RN = TN / WN / 32;
  constexpr int AQ = TM / 32;            // A rows staged per thread
  constexpr int WV = (8 * TN) / 256;     // W float4 staged per thread
  constexpr int LDA = 36;
  constexpr int A_FLOATS = TM * LDA;
  constexpr int W_FLOATS = 8 * TN * 4;
  constexpr int RED_FLOATS = (WK > 1) ? (WK - 1) * WM * WN * RM * RN * 16 * 64 : 0;
  constexpr int STAGE_FLOATS = 2 * (A_FLOATS + W_FLOATS);
  constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const ConvArgs& a = g.p[blockIdx.z];
  const int Mtot = a.n * a.T;
  const int m0 = blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;
  if (m0 >= Mtot || n0 >= a.Cout) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wm = (wave / WN) % WM;
  const int wk = wave / (WN * WM);
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  // ---- per-thread A staging geometry (fixed over the K loop)
  const int arow = tid >> 3;
  const int ac4 = tid & 7;
  const float* arowbase[AQ];
  int abrow[AQ];
  bool avalid[AQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    int m = m0 + arow + 32 * q;
    bool v = m < Mtot;
    int i = v ? m / a.T : 0;
    int t = v ? m - i * a.T : 0;
    int slot = (a.x[0].mode == 0) ? a.slots[i] : i;
    int p = (a.x[0].mode == 0 && a.pos) ? a.pos[slot] * a.x[0].rate : 0;
    abrow[q] = p + a.x[0].off + t - a.pad_left;
    arowbase[q] = a.x[0].base + (long long)slot * a.x[0].slot_stride;
    avalid[q] = v;
  }
  const long long d1 = (a.nsrc > 1) ? (a.x[1].base - a.x[0].base) : 0;
  const long long d2 = (a.nsrc > 2) ? (a.x[2].base - a.x[0].base) : 0;
  const int xC = a.x[0].C;
  const int xmask = a.x[0].lmask;
  const bool xring = a.x[0].mode == 0;
  const int ncb = a.Cin_pad >> 5;
  const int nks = a.ktaps * ncb;
  const int ci4n = a.Cin_pad >> 2;

  float4 pa[AQ];
  float4 pw0 = make_float4(0.f, 0.f, 0.f, 0.f), pw1 = pw0;
  static_assert(WV == 1 || WV == 2, "W staging vectors per thread");

  auto prefetch = [&](int ks) __attribute__((always_inline)) {
    const int cb = ks / a.ktaps;          // tap index fastest: consecutive steps re-touch the same rows
    const int j = ks - cb * a.ktaps;
    const int col = cb * 32 + ac4 * 4;
    const bool cok = col < a.Cin;
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (avalid[q] && cok) {
        int r = abrow[q] + j * a.dil;
        if (xring) r &= xmask;
        const float* p = arowbase[q] + (long long)r * xC + col;
        v = *reinterpret_cast<const float4*>(p);
        if (a.nsrc > 1) {
          float4 v1 = *reinterpret_cast<const float4*>(p + d1);
          v.x += v1.x; v.y += v1.y; v.z += v1.z; v.w += v1.w;
          if (a.nsrc > 2) {
            float4 v2 = *reinterpret_cast<const float4*>(p + d2);
            v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
          }
          const float dn = (float)a.nsrc;
          v.x /= dn; v.y /= dn; v.z /= dn; v.w /= dn;
        }
        if (a.in_act == ACT_LRELU) {
          v.x = v.x > 0.f ? v.x : v.x * a.in_slope;
          v.y = v.y > 0.f ? v.y : v.y * a.in_slope;
          v.z = v.z > 0.f ? v.z : v.z * a.in_slope;
          v.w = v.w > 0.f ? v.w : v.w * a.in_slope;
        }
      }
      pa[q] = v;
    }
    const float* wbase = a.w + ((long long)(j * ci4n + cb * 8) * a.Cout_pad + n0) * 4;
    {
      const int kq4 = tid / TN, co = tid - kq4 * TN;
      pw0 = *reinterpret_cast<const float4*>(wbase + ((long long)kq4 * a.Cout_pad + co) * 4);
      if constexpr (WV > 1) {
        const int idx1 = tid + 256, kq41 = idx1 / TN, co1 = idx1 - kq41 * TN;
        pw1 = *reinterpret_cast<const float4*>(wbase + ((long long)kq41 * a.Cout_pad + co1) * 4);
      }
    }
  };

  f32x16 acc[RM][RN];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int rn = 0; rn < RN; ++rn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rm][rn][e] = 0.f;

  prefetch(0);
  for (int ks = 0; ks < nks; ++ks) {
    float* As = lds + (ks & 1) * (A_FLOATS + W_FLOATS);
    float* Ws = As + A_FLOATS;
#pragma unroll
    for (int q = 0; q < AQ; ++q)
      *reinterpret_cast<float4*>(As + (arow + 32 * q) * LDA + ac4 * 4) = pa[q];
    *reinterpret_cast<float4*>(Ws + tid * 4) = pw0;
    if constexpr (WV > 1) *reinterpret_cast<float4*>(Ws + (tid + 256) * 4) = pw1;
    __syncthreads();
    if (ks + 1 < nks) prefetch(ks + 1);
#pragma unroll
    for (int kq = wk; kq < 4; kq += WK) {
      float4 af[RM], bf[RN];
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
        af[rm] = *reinterpret_cast<const float4*>(As + ((wm * RM + rm) * 32 + l31) * LDA + kq * 8 + lh * 4);
#pragma unroll
      for (int rn = 0; rn < RN; ++rn)
        bf[rn] = *reinterpret_cast<const float4*>(Ws + ((kq * 2 + lh) * TN + (wn * RN + rn) * 32 + l31) * 4);
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
          acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rm].x, bf[rn].x, acc[rm][rn], 0, 0, 0);
          acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rm].y, bf[rn].y, acc[rm][rn], 0, 0, 0);
          acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rm].z, bf[rn].z, acc[rm][rn], 0, 0, 0);
          acc[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rm].w, bf[rn].w, acc[rm][rn], 0, 0, 0);
        }
    }
  }

  // ---- intra-block split-K reduction through LDS
  if constexpr (WK > 1) {
    __syncthreads();
    constexpr int PER_WAVE = RM * RN * 16 * 64;
    if (wk > 0) {
      float* dst = lds + ((wk - 1) * WM * WN + wm * WN + wn) * PER_WAVE;
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((rm * RN + rn) * 16 + e) * 64 + lane] = acc[rm][rn][e];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < WK; ++k2) {
      const float* src = lds + ((k2 - 1) * WM * WN + wm * WN + wn) * PER_WAVE;
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[rm][rn][e] += src[((rm * RN + rn) * 16 + e) * 64 + lane];
    }
  }

  // ---- epilogue
  const int Cq = a.Cout / a.shuffle_r;
#pragma unroll
  for (int rm = 0; rm < RM; ++rm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ml = (wm * RM + rm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int m = m0 + ml;
      if (m >= Mtot) continue;
      const int i = m / a.T;
      const int t = m - i * a.T;
      if (a.lens && t >= a.lens[i]) continue;
      const int slot = a.slots ? a.slots[i] : i;
      float mk = 1.f;
      if (a.has_m1) { const TRef& r = a.m1; int s = r.mode == 0 ? slot : i; mk *= r.base[(long long)s * r.slot_stride + tref_row(r, s, a.pos, t)]; }
      if (a.has_m2) { const TRef& r = a.m2; int s = r.mode == 0 ? slot : i; mk *= r.base[(long long)s * r.slot_stride + tref_row(r, s, a.pos, t)]; }
      const float* resrow = nullptr;
      if (a.has_res) {
        const TRef& r = a.res; int s = r.mode == 0 ? slot : i;
        resrow = r.base + (long long)s * r.slot_stride + (long long)tref_row(r, s, a.pos, t) * r.C;
      }
      const int ys = a.y.mode == 0 ? slot : i;
      float* ybase = a.y.base + (long long)ys * a.y.slot_stride;
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const int co = n0 + (wn * RN + rn) * 32 + l31;
        if (co >= a.Cout) continue;
        float v = acc[rm][rn][e];
        if (a.bias) v += a.bias[co];
        v *= a.out_scale;
        v = apply_act(v, a.out_act, a.out_slope);
        if (a.bvec) v += a.bvec[(long long)slot * a.bvec_stride + co];
        if (resrow) v += resrow[co];
        if (a.has_m1 | a.has_m2) v *= mk;
        int orow_t, ocol;
        if (a.shuffle_r > 1) { int jj = co / Cq; ocol = co - jj * Cq; orow_t = t * a.shuffle_r + jj; }
        else { ocol = co; orow_t = t; }
        ybase[(long long)tref_row(a.y, ys, a.pos, orow_t) * a.y.C + ocol] = v;
      }
    }
  }
}

static const int kTM[NUM_CFG] = {128, 64, 128, 32, 32, 64};
static const int kTN[NUM_CFG] = {64, 64, 32, 64, 32, 32};
int conv_cfg_tm(int cfg) { return kTM[cfg]; }
int conv_cfg_tn(int cfg) { return kTN[cfg]; }

void launch_conv(const ConvGroup& g, int nprob, int cfg, hipStream_t st) {
  int maxM = 0, maxN = 0;
  for (int p = 0; p < nprob; ++p) {
    int M = g.p[p].n * g.p[p].T;
    if (M > maxM) maxM = M;
    if (g.p[p].Cout > maxN) maxN = g.p[p].Cout;
  }
  if (maxM == 0) return;
  const int TM = kTM[cfg], TN = kTN[cfg];
  dim3 grid((maxM + TM - 1) / TM, (maxN + TN - 1) / TN, nprob);
  dim3 block(256);
  switch (cfg) {
    case CFG_128x64: hipLaunchKernelGGL((conv_mfma_kernel<128, 64, 2, 2, 1>), grid, block, 0, st, g); break;
    case CFG_64x64: hipLaunchKernelGGL((conv_mfma_kernel<64, 64, 2, 2, 1>), grid, block, 0, st, g); break;
    case CFG_128x32: hipLaunchKernelGGL((conv_mfma_kernel<128, 32, 4, 1, 1>), grid, block, 0, st, g); break;
    case CFG_32x64_K2: hipLaunchKernelGGL((conv_mfma_kernel<32, 64, 1, 2, 2>), grid, block, 0, st, g); break;
    case CFG_32x32_K4: hipLaunchKernelGGL((conv_mfma_kernel<32, 32, 1, 1, 4>), grid, block, 0, st, g); break;
    case CFG_64x32_K2: hipLaunchKernelGGL((conv_mfma_kernel<64, 32, 2, 1, 2>), grid, block, 0, st, g); break;
    default: break;
  }
}

}  // namespace ck
