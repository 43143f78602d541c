// conv_tall: a causal convolution whose rows per slot and step are FEW and whose K is LONG - the first upsamplers of the vocoder
// (ups.0: 4 rows per stream and step, K = 16 taps x 512 channels, 2048 columns: 100 MB of limb weights) - as a GEMM on the bf16 MFMA in
// the limb arithmetic of resblock_limb.hip / conv_limb.hip (x = h + m + l per operand, six products, fp32 accumulation).
//
// conv_limb's tiles are rows of ONE slot's window (the taps of a channel block re-read the same LDS rows, row-shifted): with 4 rows per
// slot a 64-row tile is 16 slots x 19 window rows - it does not fit, and smaller tiles re-read the weights.  Here a tile is CT_TM = 128
// rows of ANY slots x CT_TN = 128 columns, and a K block is one (tap, 32-channel block) pair whose 128 rows are gathered from the rings
// by the helper waves (the row of tap j is the ring row t + j * dil - pad_left of its slot), split into limb planes and staged two
// blocks per barrier into one of two LDS buffers.  Every matrix wave owns two 16-column tiles for all 8 row tiles: per block 24 A
// fragment reads from LDS and 6 weight fragments from L2 for 96 MFMAs - a quarter of conv_limb<4,1,1,4>'s LDS reads per MFMA.
// Tiles are few (ups.0 at 64 streams: 2 x 16), so K is split over workgroups: an item = (tile, K slice); the slices' partial tiles
// meet in memory with conv_limb_sk's per-wave hand-over (write-through stores, a ticket per (tile, wave), the last arrival sums the
// slices IN SLICE ORDER and runs the epilogue: bit-reproducible, no fence).
#include <algorithm>
#include <atomic>
#include <cstring>

#include "kernels.h"

namespace cnk {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef const f32x4 __attribute__((address_space(1)))* gcf4;
typedef f32x4 __attribute__((address_space(1)))* gf4;
typedef const int __attribute__((address_space(1)))* gci;

__device__ __forceinline__ f32x4 ct_gload(const void* p) { return *(gcf4)(p); }
__device__ __forceinline__ void ct_gstore(float* p, const f32x4 v) { *(gf4)(p) = v; }
// (conv_limb.hip, cl_split2: two values at a time on the packed VALU forms, round-to-nearest-even limbs, exact remainders)
__device__ __forceinline__ void ct_split2(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  const f32x2 r1 = x - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  const f32x2 r2 = r1 - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

constexpr int CT_TM = 128, CT_TN = 128;
constexpr int CT_NRT = CT_TM / 16;        // row tiles per wave
constexpr int CT_NCW = 2;                 // column tiles per wave (4 matrix waves x 2 x 16 = CT_TN)
constexpr int CT_PO = 32;                 // a row's limb planes side by side: 3 x 32 bf16 ...
constexpr int CT_RS = 3 * CT_PO + 16;     // ... + 16: 224 bytes per row, 14 sixteen-byte slots = 2 mod 4 (conflict-free ds_read_b128, as conv_limb's 18)
constexpr int CT_BLK = 2;                 // K blocks staged per barrier
constexpr int CT_BUF = CT_BLK * CT_TM * CT_RS;      // u16 per buffer
constexpr int CT_LDS_BYTES = 2 * CT_BUF * 2;        // 114 688

}  // namespace

__global__ __launch_bounds__(512, 2) void conv_tall_kernel(const ConvTallArgs g) {
  extern __shared__ __attribute__((aligned(16))) u16 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto bar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const ConvArgs& a = g.p;
  // item -> (tile, K slice): the slices of a tile are adjacent (they finish together)
  const int item = (int)blockIdx.x, S = g.S;
  const int tile = item / S, ks = item - tile * S;
  const int mt = tile / g.ntiles, nt = tile - mt * g.ntiles;
  const int m0 = mt * CT_TM;
  const int k = a.ktaps, NB = (a.Cin / 32) * k;
  const int gb0 = ks * g.nbps, gb1 = gb0 + g.nbps < NB ? gb0 + g.nbps : NB;
  const int npairs = (gb1 - gb0 + CT_BLK - 1) / CT_BLK;
  const int T = a.T, n = a.n;

  if (wave >= 4) {
    // ============================================================ helper waves: gather + limb split, two blocks per barrier
    const int ht = tid - 256;
    __builtin_amdgcn_s_setprio(3);
    const float* xb = a.x.base;
    const int xC = a.x.C, xmask = a.x.lmask, xrate = a.x.rate, xoff = a.x.off - a.pad_left, dil = a.dil;
    const float act_slope = a.in_act == ACT_LRELU ? a.in_slope : 1.f;
    // this thread's four rows of a block (row = (ht >> 3) + 32 u, channels 4 (ht & 7) ..): slot offset and ring row of tap 0
    int soff[4], row0[4];
    const int c4 = ht & 7;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + (ht >> 3) + 32 * u;
      int i = m / T;
      const int t = m - i * T;
      i = i < n ? i : n - 1;                        // (rows past the launch's last slot: gathered from the last slot, never stored)
      const int slot = a.slots ? *(gci)(a.slots + i) : i, pv = a.pos ? *(gci)(a.pos + slot) : 0;
      soff[u] = (int)((long long)slot * a.x.slot_stride) + c4 * 4;
      row0[u] = pv * xrate + xoff + t;
    }
    // Loads run TWO pairs ahead of the conversion (two register sets, used alternately): a pair's 8 loads have the time of a whole
    // pair of MFMAs; one pair ahead they were issued just in front of the barrier and waited for right behind it (measured: the
    // staging cost the launch 10 of 64 us).
    f32x4 va[CT_BLK][4], vb[CT_BLK][4];
    auto issue = [&](const int pair, f32x4 (&v)[CT_BLK][4]) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < CT_BLK; ++b) {
        int gb = gb0 + pair * CT_BLK + b;
        gb = gb < gb1 ? gb : gb1 - 1;               // (past the slice's end: its last block again, staged and not consumed)
        const int cb = gb / k, j = gb - cb * k;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[b][u] = ct_gload(xb + soff[u] + ((row0[u] + j * dil) & xmask) * xC + cb * 32);
      }
    };
    auto convert = [&](const int pair, const f32x4 (&v)[CT_BLK][4]) __attribute__((always_inline)) {
      u16* const dstb = lds + (pair & 1) * CT_BUF;
#pragma unroll
      for (int b = 0; b < CT_BLK; ++b)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          unsigned h[2], m[2], l[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x2 s = {v[b][u][2 * e], v[b][u][2 * e + 1]};
            const f32x2 sx = s * act_slope;          // (LeakyReLU with 0 < slope <= 1 as max(x, slope x); slope 1 without it)
            s = (f32x2){__builtin_fmaxf(s[0], sx[0]), __builtin_fmaxf(s[1], sx[1])};
            ct_split2(s, h[e], m[e], l[e]);
          }
          u16* d = dstb + (b * CT_TM + (ht >> 3) + 32 * u) * CT_RS + c4 * 4;
          *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
          *reinterpret_cast<uint2*>(d + CT_PO) = make_uint2(m[0], m[1]);
          *reinterpret_cast<uint2*>(d + 2 * CT_PO) = make_uint2(l[0], l[1]);
        }
    };
#ifdef CT_ABL_STAGE      // developer ablation: the helpers only keep the barriers
    for (int pair = 0; pair < npairs; ++pair) bar();
    return;
#endif
    issue(0, va);
    issue(1, vb);
    for (int pair = 0; pair < npairs; pair += 2) {
      convert(pair, va);
      issue(pair + 2, va);
      bar();
      if (pair + 1 < npairs) {
        convert(pair + 1, vb);
        issue(pair + 3, vb);
        bar();
      }
    }
    return;
  }

  // ============================================================== matrix waves
  const int lr = lane & 15, lg = lane >> 4;
  const int ct0 = nt * (CT_TN / 16) + wave * CT_NCW;             // this wave's first 16-column tile
  const long long ct_stride = (long long)NB * 1536;              // elements per column tile: [cb][tap][limb][64 lanes][8]
  const u16* wl = a.wl + (long long)ct0 * ct_stride + lane * 8;
  // the epilogue's rows (slot, frame counter): requested now, long back when the K loop ends
  int eslot[CT_NRT], epos[CT_NRT];
#pragma unroll
  for (int r = 0; r < CT_NRT; ++r) {
    const int m = m0 + r * 16 + lr;
    int i = m / T;
    i = i < n ? i : n - 1;
    eslot[r] = a.slots ? *(gci)(a.slots + i) : i;
  }
#pragma unroll
  for (int r = 0; r < CT_NRT; ++r) epos[r] = a.pos ? *(gci)(a.pos + eslot[r]) : 0;
  f32x4 acc[CT_NRT][CT_NCW];
#pragma unroll
  for (int r = 0; r < CT_NRT; ++r)
#pragma unroll
    for (int c = 0; c < CT_NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // weight blocks: two in flight per wave (block b of a pair lives in ring slot b)
  f32x4 bw[CT_BLK][CT_NCW][3];
#pragma unroll
  for (int b = 0; b < CT_BLK; ++b) {
    const int gb = gb0 + b < gb1 ? gb0 + b : gb1 - 1;
#pragma unroll
    for (int c = 0; c < CT_NCW; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) bw[b][c][p] = ct_gload(wl + (long long)gb * 1536 + c * ct_stride + p * 512);
  }
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first: l*h, m*m, h*l, m*h, h*m, h*h
  for (int pair = 0; pair < npairs; ++pair) {
    bar();
    const u16* const buf = lds + (pair & 1) * CT_BUF;
#pragma unroll
    for (int b = 0; b < CT_BLK; ++b) {
      const int gb = gb0 + pair * CT_BLK + b;
      const bool live = gb < gb1;                   // (an odd slice's last pair has one block)
      if (live) {
        const u16* ab = buf + (b * CT_TM + lr) * CT_RS + 8 * lg;
        // (the NEXT row tile's fragments are requested in front of this one's MFMAs: two register sets)
        f32x4 af[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[0][p] = *reinterpret_cast<const f32x4*>(ab + p * CT_PO);
#pragma unroll
        for (int r = 0; r < CT_NRT; ++r) {
          if (r + 1 < CT_NRT) {
#pragma unroll
            for (int p = 0; p < 3; ++p) af[(r + 1) & 1][p] = *reinterpret_cast<const f32x4*>(ab + (r + 1) * 16 * CT_RS + p * CT_PO);
          }
#pragma unroll
          for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int c = 0; c < CT_NCW; ++c)
              acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[b][c][PB[s]]), __builtin_bit_cast(bf16x8, af[r & 1][PA[s]]), acc[r][c], 0, 0, 0);
        }
      }
      // this ring slot's next block (two blocks on): requested behind the MFMAs that read it
      int gn = gb + CT_BLK;
      gn = gn < gb1 ? gn : gb1 - 1;
#ifdef CT_ABL_W      // developer ablation: every weight block from the slice's start (cache hits)
      gn = gb0;
#endif
#pragma unroll
      for (int c = 0; c < CT_NCW; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[b][c][p] = ct_gload(wl + (long long)gn * 1536 + c * ct_stride + p * 512);
    }
  }
  // ---------------- split-K: the slices' partial tiles meet in memory, wave by wave (conv_limb_sk's hand-over)
  bool run_epi = true;
  if (S > 1) {
    float* const part = g.slab + (long long)tile * S * (CT_TM * CT_TN) + (long long)wave * (CT_NRT * CT_NCW * 256);
    float* const mine = part + (long long)ks * (CT_TM * CT_TN);
#pragma unroll
    for (int r = 0; r < CT_NRT; ++r)
#pragma unroll
      for (int c = 0; c < CT_NCW; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) __hip_atomic_store(mine + ((r * CT_NCW + c) * 4 + e) * 64 + lane, acc[r][c][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* const ctr = g.counters + tile * 4 + wave;
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    run_epi = ticket == S - 1;
    if (run_epi) {
      if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
#pragma unroll
      for (int r = 0; r < CT_NRT; ++r)
#pragma unroll
        for (int c = 0; c < CT_NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // One row tile of up to eight slices per round trip (64 loads per lane in flight: the register budget beside the decoder is 192),
      // added in slice order: a load-then-add loop over (slice, row tile) was 64 dependent round trips at the end of the launch's
      // critical path.
      for (int q0 = 0; q0 < S; q0 += 8) {
#pragma unroll
        for (int r = 0; r < CT_NRT; ++r) {
          float pv[8][CT_NCW * 4];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int s2 = q0 + q < S ? q0 + q : S - 1;          // (past the last slice: its partial again, dropped below)
            const float* src = part + (long long)s2 * (CT_TM * CT_TN) + (r * CT_NCW * 4) * 64 + lane;
#pragma unroll
            for (int f = 0; f < CT_NCW * 4; ++f) pv[q][f] = __hip_atomic_load(src + f * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const bool in = q0 + q < S;
#pragma unroll
            for (int c = 0; c < CT_NCW; ++c)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float t = acc[r][c][e] + pv[q][c * 4 + e];
                acc[r][c][e] = in ? t : acc[r][c][e];
              }
          }
        }
      }
    }
  }
  if (!run_epi) return;
  // ---------------- epilogue: bias -> activation -> (pixel-shuffled) store, 4 packed columns per lane
  {
    const int Cout = a.Cout, shuf = a.shuffle_r, Cq = Cout / shuf;
    const bool yring = a.y.mode == 0;
    float* const yb = a.y.base;
    const int yC = a.y.C, ymask = yring ? a.y.lmask : -1, yrate = a.y.rate, yoff = a.y.off;
    const long long yss = a.y.slot_stride;
    float* const y2b = a.y2_base;
    f32x4 bias[CT_NCW];
#pragma unroll
    for (int c = 0; c < CT_NCW; ++c) {
      const int cc = (ct0 + c) * 16 + 4 * lg;
      bias[c] = (a.bias && cc < Cout) ? ct_gload(a.bias + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < CT_NRT; ++r) {
      const int m = m0 + r * 16 + lr, i = m / T, t = m - i * T;
      if (i >= n) continue;
#pragma unroll
      for (int c = 0; c < CT_NCW; ++c) {
        const int cc = (ct0 + c) * 16 + 4 * lg;
        if (cc >= Cout) continue;
        f32x4 o = acc[r][c] + bias[c];
        if (a.out_act == ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * a.out_slope;
        }
        int jj = 0, oc = cc;
        if (shuf > 1) { jj = cc / Cq; oc = cc - jj * Cq; }
        const int yrow = ((yring ? epos[r] * yrate : 0) + yoff + t * shuf + jj) & ymask;
        const long long yo = (long long)(yring ? eslot[r] : i) * yss + (long long)yrow * yC + oc;
        ct_gstore(yb + yo, o);
        if (y2b) {
          f32x4 o2;
#pragma unroll
          for (int e = 0; e < 4; ++e) o2[e] = o[e] > 0.f ? o[e] : o[e] * a.y2_slope;
          ct_gstore(y2b + yo, o2);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side

// the launches this kernel is for: few rows per slot, a long K, limb weights, a plain epilogue
bool conv_tall_supported(const ConvArgs& a) {
  if (!a.wl || a.x.mode != 0 || a.Cin % 32 || a.has_m1 || a.has_m2 || a.bvec || a.lens || a.has_res || a.out_scale != 1.f) return false;
  if (a.in_act != ACT_NONE && a.in_act != ACT_LRELU) return false;
  if (a.in_act == ACT_LRELU && !(a.in_slope > 0.f && a.in_slope <= 1.f)) return false;
  if (a.out_act != ACT_NONE && a.out_act != ACT_LRELU) return false;
  if (a.Cout % CT_TN || a.Cout % a.shuffle_r || (a.Cout / a.shuffle_r) % 4 || a.x.C % 4 || a.y.C % 4) return false;
  return true;
}

// Plan of a launch, or false: this shape stays with the other kernels.  plan_n > 0 (fixed-plan stream-sets): the K split is sized as
// the full stream-set's launch would be - one factor for every tile, so a stream's sums do not depend on the active slots.
bool conv_tall_plan(const ConvArgs& a, int num_cu, int plan_n, ConvTallArgs* out, long long slab_floats, int max_counters, int max_T) {
  if (!conv_tall_supported(a)) return false;
  const int NB = (a.Cin / 32) * a.ktaps;
  const long long Mp = (long long)(plan_n > 0 ? plan_n : a.n) * a.T, M = (long long)a.n * a.T;
  // few rows per slot (a window per slot does not pay), at least one full tile of rows, a K loop worth splitting
  if (a.T > max_T || Mp < CT_TM || NB < 64) return false;
  const int ntiles = a.Cout / CT_TN;
  const long long ptiles = ((Mp + CT_TM - 1) / CT_TM) * ntiles;
  // (an item for every CU, with slices of at least 8 blocks: below that conv_mfma's small-M plans are the better fit - measured at 16
  // streams, where ups.1 had 200 items: 0.609 against 0.597 ms per vocoder step; from 24 streams on this kernel wins: 0.701 / 0.793 /
  // 0.981 / 1.169 against 0.712 / 0.804 / 1.056 / 1.202 at 24 / 32 / 48 / 64)
  if (ptiles * std::min<long long>(16, NB / 8) < num_cu) return false;
  int S = (int)std::max<long long>(1, std::min<long long>(16, num_cu / ptiles));
  int nbps = (NB + S - 1) / S;
  nbps += nbps & 1;                                  // whole pairs of blocks per slice
  if (nbps < 8) { nbps = 8; }
  S = (NB + nbps - 1) / nbps;
  const int mtiles = (int)((M + CT_TM - 1) / CT_TM);
  if ((long long)mtiles * ntiles * S * CT_TM * CT_TN > slab_floats || mtiles * ntiles * 4 > max_counters) return false;
  memset(out, 0, sizeof(*out));
  out->p = a; out->S = S; out->nbps = nbps; out->mtiles = mtiles; out->ntiles = ntiles;
  return true;
}

void launch_conv_tall(const ConvTallArgs& g, hipStream_t st) {
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)conv_tall_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(conv_tall_kernel, dim3(g.mtiles * g.ntiles * g.S), dim3(512), CT_LDS_BYTES, st, g);
}

}  // namespace cnk
