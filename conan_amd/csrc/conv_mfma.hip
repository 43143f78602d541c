// conv_mfma: channel-last causal / shifted 1-D convolution as an implicit GEMM on the gfx950
// f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
// GEMM view:  M = n*T output rows (slot, time), N = Cout, K = ktaps*Cin.
//   A[m][(j,ci)] = f(x[slot][t + j*dil - pad_left][ci])   gathered per tap from the activation rings
//   B[(j,ci)][co] = packed weights [tap][ci/4][co][4]
// Block = 256 threads (4 waves on the 4 SIMDs of a CU).  Per K-step (one tap, KS input channels) the
// block stages an A tile [TM][KS] and a W tile [KS][TN] through LDS.  The raw global loads of step
// s+1 are issued back-to-back right after the barrier of step s (no dependent instruction between
// them, so their latencies overlap each other and the MFMAs of step s); the input transform
// (branch mean, LeakyReLU, zero fill) runs on the registers when they are written to LDS at the top
// of step s+1.  Two LDS buffers, one barrier per step.
// Fragments are read with ds_read_b128: a K-chunk of 8 feeds 4 MFMAs, lanes 0-31 carrying k 0..3 and
// lanes 32-63 k 4..7 (the MFMA's two k-slots), so one 16-byte LDS read per operand serves 4 matrix
// instructions.  A rows are padded by 4 floats: bank-conflict-free for the b128 lane groups.
// Output tile D[time][co] keeps co on the lane -> 128-byte coalesced channel-last stores; the pixel
// shuffle of CausalUpsampleBlock3 is a pure address remap of that store (weights pre-permuted).
// KS = 32 for the large streaming tiles; KS = 128 for the small-M (latency-bound) tiles, where a
// longer K-step amortises the per-step load latency and barrier.
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

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int TM, int TN, int WM, int WN, int WK, int KS, int NSRC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvGroup g) {
  static_assert(WM * WN * WK == 4, "4 waves per block");
  static_assert(KS == 32 || KS == 64 || KS == 128, "K-step");
  constexpr int RM = TM / WM / 32;
  constexpr int RN = TN / WN / 32;
  constexpr int SB = KS / 32;                 // 32-channel sub-blocks per K-step
  constexpr int AQ = (TM / 32) * SB;          // A float4 staged per thread
  constexpr int WV = (KS / 4 * TN) / 256;     // W float4 staged per thread
  constexpr int NKQ = KS / 8;                 // 8-deep K chunks per step
  constexpr int LDA = KS + 4;
  constexpr int A_FLOATS = TM * LDA;
  constexpr int W_FLOATS = KS * TN;
  constexpr int RED_FLOATS = (WK > 1) ? (WK - 1) * WM * WN * RM * RN * 16 * 64 : 0;
  constexpr int STAGE_FLOATS = 2 * (A_FLOATS + W_FLOATS);
  constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const ConvArgs& a = g.p[blockIdx.z];
  // ---- launch-uniform scalars, read once (keeps the K loop free of kernarg re-loads)
  const int T = a.T, nslot = a.n, ktaps = a.ktaps, dil = a.dil, Cin = a.Cin, CoutP = a.Cout_pad, Cout = a.Cout;
  const int Mtot = nslot * T;
  const int m0 = blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;
  if (m0 >= Mtot || n0 >= Cout) return;
  const int* __restrict__ slots = a.slots;
  const int* __restrict__ posp = a.pos;
  const float* __restrict__ wbase = a.w;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wm = (wave / WN) % WM;
  const int wk = (WK == 1) ? 0 : wave / (WN * WM);
  const int l31 = lane & 31;
  const int lh = lane >> 5;

  // ---- tile row -> (batch index i, time t).  A tile that fits inside one stream's T rows touches at most two
  // streams: their slot / position are fetched once; short-T layers (frame-rate tensors) divide per row.
  const int i0 = m0 / T;
  const int t0 = m0 - i0 * T;
  const bool fast = T >= TM;
  const int i1 = (i0 + 1 < nslot) ? i0 + 1 : i0;
  const int slotA = slots ? slots[i0] : i0, slotB = slots ? slots[i1] : i1;
  const int posA = posp ? posp[slotA] : 0, posB = posp ? posp[slotB] : 0;
  auto rowmap = [&](int ml, int& i, int& t, int& slot, int& pv) __attribute__((always_inline)) {
    if (fast) {
      const int tt = t0 + ml;
      const bool w = tt >= T;
      i = w ? i1 : i0; t = w ? tt - T : tt; slot = w ? slotB : slotA; pv = w ? posB : posA;
    } else {
      const int m = m0 + ml;
      i = m / T; t = m - i * T;
      i = i < nslot ? i : nslot - 1;
      slot = slots ? slots[i] : i; pv = posp ? posp[slot] : 0;
    }
  };

  // ---- per-thread A staging geometry (fixed over the K loop): rows arow + 32*q, channel quad ac4
  const int arow = tid >> 3;
  const int ac4 = tid & 7;
  const float* arowbase[TM / 32];
  int abrow[TM / 32];
  unsigned avalid = 0;
  const bool xring = a.x[0].mode == 0;
  {
    const int xrate = a.x[0].rate, xoff = a.x[0].off - a.pad_left;
    const long long xss = a.x[0].slot_stride;
    const float* xb = a.x[0].base;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) {
      const int ml = arow + 32 * q;
      int i, t, slot, pv;
      rowmap(ml, i, t, slot, pv);
      abrow[q] = (xring ? pv * xrate : 0) + xoff + t;
      arowbase[q] = xb + (long long)(xring ? slot : i) * xss;
      avalid |= ((m0 + ml) < Mtot ? 1u : 0u) << q;
    }
  }
  const long long d1 = (NSRC > 1) ? (a.x[1].base - a.x[0].base) : 0;
  const long long d2 = (NSRC > 2) ? (a.x[2].base - a.x[0].base) : 0;
  const int xC = a.x[0].C;
  const int xmask = xring ? a.x[0].lmask : -1;
  const int ncb32 = a.Cin_pad >> 5;                 // 32-channel blocks per tap
  const int ncb = (ncb32 + SB - 1) / SB;            // K-steps per tap
  const int nks = ktaps * ncb;
  const int ci4n = a.Cin_pad >> 2;
  const float neg_mul = a.in_act == ACT_LRELU ? a.in_slope : 1.0f;
  // per-thread W staging offsets (floats, relative to the K-step's tile base)
  int woff[WV];
#pragma unroll
  for (int v = 0; v < WV; ++v) { const int idx = tid + 256 * v; const int kq4 = idx / TN, co = idx - kq4 * TN; woff[v] = (kq4 * CoutP + co) * 4; }

  float4 ra[AQ][NSRC];
  static_assert(WV >= 1 && WV <= 8, "W staging vectors per thread");
  float4 rw0 = f4zero(), rw1 = rw0, rw2 = rw0, rw3 = rw0, rw4 = rw0, rw5 = rw0, rw6 = rw0, rw7 = rw0;  // named (not an array): keeps them in VGPRs
  unsigned okmask = 0;     // bit (q*SB+sb): staged quad is inside the tile and inside Cin
  int jn = 0, cbn = 0;     // (tap, channel block) of the next K-step to issue; tap index fastest so that
                           // consecutive steps re-touch the same activation rows (L1/L2 hits)

  auto issue = [&]() __attribute__((always_inline)) {
    const int j = jn, cb = cbn;
    okmask = 0;
#pragma unroll
    for (int q = 0; q < TM / 32; ++q) {
      const int r = (abrow[q] + j * dil) & xmask;
      const float* rowp = arowbase[q] + r * xC;
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const int col = (cb * SB + sb) * 32 + ac4 * 4;
        const bool ok = ((avalid >> q) & 1u) && col < Cin;
        const float* p = ok ? rowp + col : arowbase[q];     // always a mapped address; value dropped if !ok
        ra[q * SB + sb][0] = *reinterpret_cast<const float4*>(p);
        if constexpr (NSRC > 1) ra[q * SB + sb][1] = *reinterpret_cast<const float4*>(p + d1);
        if constexpr (NSRC > 2) ra[q * SB + sb][2] = *reinterpret_cast<const float4*>(p + d2);
        okmask |= (ok ? 1u : 0u) << (q * SB + sb);
      }
    }
    const float* wstep = wbase + ((long long)(j * ci4n + cb * (KS / 4)) * CoutP + n0) * 4;
#define CK_W_ISSUE(V)                                                                                   \
    if constexpr (WV > V) {                                                                               \
      int o = woff[V];                                                                                    \
      if constexpr (KS == 128) { /* past Cin_pad: any mapped row; the A side is zero there */             \
        const int kq4 = (tid + 256 * V) / TN;                                                             \
        if (cb * (KS / 4) + kq4 >= ci4n) o -= kq4 * CoutP * 4;                                            \
      }                                                                                                   \
      rw##V = *reinterpret_cast<const float4*>(wstep + o);                                                \
    }
    CK_W_ISSUE(0) CK_W_ISSUE(1) CK_W_ISSUE(2) CK_W_ISSUE(3) CK_W_ISSUE(4) CK_W_ISSUE(5) CK_W_ISSUE(6) CK_W_ISSUE(7)
#undef CK_W_ISSUE
    if (++jn == ktaps) { jn = 0; ++cbn; }
  };

  f32x16 acc[RM][RN];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int rn = 0; rn < RN; ++rn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rm][rn][e] = 0.f;

  issue();
  for (int ks = 0; ks < nks; ++ks) {
    float* As = lds + (ks & 1) * (A_FLOATS + W_FLOATS);
    float* Ws = As + A_FLOATS;
    // ---- transform + store the staged registers
#pragma unroll
    for (int q = 0; q < TM / 32; ++q)
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        float4 v = ra[q * SB + sb][0];
        if constexpr (NSRC > 1) {
          const float4 v1 = ra[q * SB + sb][1];
          v.x += v1.x; v.y += v1.y; v.z += v1.z; v.w += v1.w;
          if constexpr (NSRC > 2) {
            const float4 v2 = ra[q * SB + sb][2];
            v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
          }
          const float dn = (float)NSRC;       // xs / num_resblocks (hifigan_causal.py:329): true division
          v.x /= dn; v.y /= dn; v.z /= dn; v.w /= dn;
        }
        // LeakyReLU as a select on the multiplier (neg_mul == 1 when no input activation)
        v.x *= v.x > 0.f ? 1.0f : neg_mul;
        v.y *= v.y > 0.f ? 1.0f : neg_mul;
        v.z *= v.z > 0.f ? 1.0f : neg_mul;
        v.w *= v.w > 0.f ? 1.0f : neg_mul;
        if (!((okmask >> (q * SB + sb)) & 1u)) v = f4zero();
        *reinterpret_cast<float4*>(As + (arow + 32 * q) * LDA + sb * 32 + ac4 * 4) = v;
      }
#define CK_W_STORE(V) if constexpr (WV > V) *reinterpret_cast<float4*>(Ws + (tid + 256 * V) * 4) = rw##V;
    CK_W_STORE(0) CK_W_STORE(1) CK_W_STORE(2) CK_W_STORE(3) CK_W_STORE(4) CK_W_STORE(5) CK_W_STORE(6) CK_W_STORE(7)
#undef CK_W_STORE
    __syncthreads();
#ifdef CK_ABLATE
    if (ks + 1 < nks && !(a.ksplit_unused & 1)) issue();
    if (a.ksplit_unused & 2) continue;
#else
    if (ks + 1 < nks) issue();
#endif
#pragma unroll
    for (int kc = 0; kc < NKQ / WK; ++kc) {
      const int kq = kc * WK + wk;      // fixed trip count: no divergent control flow around the MFMAs
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

  // ---- epilogue: per lane the column (co) is fixed per rn, the 16 accumulator registers walk the rows.
  // Pass 1 resolves every row's addresses and issues all residual / mask loads back-to-back (one latency,
  // not one per row); pass 2 applies bias/scale/activation/residual/mask and stores 128-byte row segments.
  constexpr int NR = RM * 16;
  const int shuf = a.shuffle_r;
  const int Cq = Cout / shuf;
  int ocol[RN], ojj[RN];
  float bco[RN];
  bool cok[RN];
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const int co = n0 + (wn * RN + rn) * 32 + l31;
    cok[rn] = co < Cout;
    bco[rn] = (a.bias && cok[rn]) ? a.bias[co] : 0.f;
    if (shuf > 1) { ojj[rn] = co / Cq; ocol[rn] = co - ojj[rn] * Cq; } else { ojj[rn] = 0; ocol[rn] = co; }
  }
  const float oscale = a.out_scale, oslope = a.out_slope;
  const int oact = a.out_act;
  const int* lens = a.lens;
  const bool yring = a.y.mode == 0;
  const int yC = a.y.C, ymask = yring ? a.y.lmask : -1, yrate = a.y.rate, yoff = a.y.off;
  const long long yss = a.y.slot_stride;
  float* const ybase0 = a.y.base;
  const bool has_res = a.has_res != 0, has_mask = (a.has_m1 | a.has_m2) != 0, has_bvec = a.bvec != nullptr;

  int yrow[NR];          // first output row (before the per-column shuffle offset), -1 = skip
  int ysel[NR];          // batch index / slot that owns the row (selects the y base)
  float rv[NR][RN];
  float mkv[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int rm = r >> 4, e = r & 15;
    const int ml = (wm * RM + rm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    int i, t, slot, pv;
    rowmap(ml, i, t, slot, pv);
    bool ok = (m0 + ml) < Mtot;
    if (lens) ok = ok && t < lens[i];
    yrow[r] = ok ? ((yring ? pv * yrate : 0) + yoff + t * shuf) : -1;
    ysel[r] = yring ? slot : i;
    mkv[r] = 1.f;
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) rv[r][rn] = 0.f;
    if (has_res) {
      const TRef& rr = a.res; const int sidx = rr.mode == 0 ? slot : i;
      const float* resrow = rr.base + (long long)sidx * rr.slot_stride + (long long)tref_row(rr, sidx, posp, t) * rr.C;
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) if (cok[rn]) rv[r][rn] = resrow[n0 + (wn * RN + rn) * 32 + l31];
    }
    if (has_bvec) {
      const float* bv = a.bvec + (long long)slot * a.bvec_stride;
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) if (cok[rn]) rv[r][rn] += bv[n0 + (wn * RN + rn) * 32 + l31];
    }
    if (has_mask) {
      float mk = 1.f;
      if (a.has_m1) { const TRef& q = a.m1; int sidx = q.mode == 0 ? slot : i; mk *= q.base[(long long)sidx * q.slot_stride + tref_row(q, sidx, posp, t)]; }
      if (a.has_m2) { const TRef& q = a.m2; int sidx = q.mode == 0 ? slot : i; mk *= q.base[(long long)sidx * q.slot_stride + tref_row(q, sidx, posp, t)]; }
      mkv[r] = mk;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int rm = r >> 4, e = r & 15;
    if (yrow[r] < 0) continue;
    float* ybase = ybase0 + (long long)ysel[r] * yss;
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) {
      if (!cok[rn]) continue;
      float v = acc[rm][rn][e] + bco[rn];
      v *= oscale;
      v = apply_act(v, oact, oslope);
      v += rv[r][rn];          // bvec (added after the activation) + residual
      v *= mkv[r];
      ybase[((yrow[r] + ojj[rn]) & ymask) * yC + ocol[rn]] = v;
    }
  }
}

static const int kTM[NUM_CFG] = {128, 64, 128, 32, 32, 64, 64, 128, 128};
static const int kTN[NUM_CFG] = {64, 64, 32, 64, 32, 32, 64, 64, 32};
int conv_cfg_tm(int cfg) { return kTM[cfg]; }
int conv_cfg_tn(int cfg) { return kTN[cfg]; }

template <int NSRC>
static void launch_conv_n(const ConvGroup& g, int cfg, dim3 grid, hipStream_t st) {
  dim3 block(256);
  switch (cfg) {
    case CFG_128x64: hipLaunchKernelGGL((conv_mfma_kernel<128, 64, 2, 2, 1, 32, NSRC>), grid, block, 0, st, g); break;
    case CFG_64x64: hipLaunchKernelGGL((conv_mfma_kernel<64, 64, 2, 2, 1, 32, NSRC>), grid, block, 0, st, g); break;
    case CFG_128x32: hipLaunchKernelGGL((conv_mfma_kernel<128, 32, 4, 1, 1, 32, NSRC>), grid, block, 0, st, g); break;
    case CFG_32x64_K2: hipLaunchKernelGGL((conv_mfma_kernel<32, 64, 1, 2, 2, 128, NSRC>), grid, block, 0, st, g); break;
    case CFG_32x32_K4: hipLaunchKernelGGL((conv_mfma_kernel<32, 32, 1, 1, 4, 128, NSRC>), grid, block, 0, st, g); break;
    case CFG_64x32_K2: hipLaunchKernelGGL((conv_mfma_kernel<64, 32, 2, 1, 2, 32, NSRC>), grid, block, 0, st, g); break;
    case CFG_64x64_KS64: hipLaunchKernelGGL((conv_mfma_kernel<64, 64, 2, 2, 1, 64, NSRC>), grid, block, 0, st, g); break;
    case CFG_128x64_KS64: hipLaunchKernelGGL((conv_mfma_kernel<128, 64, 2, 2, 1, 64, NSRC>), grid, block, 0, st, g); break;
    case CFG_128x32_KS64: hipLaunchKernelGGL((conv_mfma_kernel<128, 32, 4, 1, 1, 64, NSRC>), grid, block, 0, st, g); break;
    default: break;
  }
}

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
  const int nsrc = g.p[0].nsrc;
  if (nsrc == 1) launch_conv_n<1>(g, cfg, grid, st);
  else if (nsrc == 2) launch_conv_n<2>(g, cfg, grid, st);
  else launch_conv_n<3>(g, cfg, grid, st);
}

}  // namespace ck
