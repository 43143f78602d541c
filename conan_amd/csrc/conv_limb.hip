// conv_limb: causal / shifted 1-D convolution as an implicit GEMM with every fp32 product computed as SIX bf16 limb products on
// the bf16 MFMA (see resblock_limb.hip for the arithmetic: x = h + m + l per operand, hh + hm + mh + hl + mm + lh accumulated in
// fp32; error below the f32 MFMA's own) - the upsamplers and the wide first ResBlock stage of the vocoder, which conv_mfma /
// resblock_pair run on the f32 MFMA at 0.54 - 0.72 of its peak.
//
//   y[i][t][co] = epilogue( sum_{j<ktaps} sum_{ci<Cin} W[j][ci][co] * f(x[i][t + j*dil - pad_left][ci]) )      (ConvArgs)
//
// A workgroup (4 matrix waves + 4 helper waves, like the fused ResBlock passes) owns a TM x TN output tile, TM = 16*NRW*RW rows
// of one slot or of TM / T whole slots (T rows per slot and step), TN = 16*NCW*CW packed output columns:
//   * K runs over 32-channel blocks; for each the helper waves stage the block's WINDOW - per slot the tile's rows plus the
//     (ktaps-1)*dil rows of tap reach, LeakyReLU'd and split into three bf16 planes - into one of two LDS buffers while the
//     matrix waves compute on the other: all taps of a channel block read the same window, row-shifted (one block barrier per
//     channel block);
//   * weights are packed per limb at finalize ([16-column tile][channel block][tap][limb][64 lanes] x 16 bytes) and streamed from
//     L2 straight into registers, two blocks ahead, private per wave;
//   * MFMAs are issued transposed (weights first): a lane's accumulator holds 4 consecutive packed columns of one row, and the
//     epilogue (bias, activation, residual, pixel-shuffle row remap) stores 16 bytes per lane.
// Launches are persistent over a host-balanced tile list (up to 3 problems of different tap counts per launch).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace cnk {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef const f32x4 __attribute__((address_space(1)))* gcf4;
typedef f32x4 __attribute__((address_space(1)))* gf4;
typedef const int __attribute__((address_space(1)))* gci;

__device__ __forceinline__ f32x4 cl_gload(const void* p) { return *(gcf4)(p); }
__device__ __forceinline__ void cl_gstore(float* p, const f32x4 v) { *(gf4)(p) = v; }

// two values at a time on the packed VALU forms (resblock_limb.hip, rl_split2): the same roundings in half the instructions
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cl_split2(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  const f32x2 r1 = x - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  const f32x2 r2 = r1 - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

constexpr int CL_LDB = 48;          // bf16 elements per window row and limb plane: 32 channels + 16
constexpr int CL_RS = 3 * CL_LDB;   // a window row holds its three limb planes side by side: 288 bytes, a stride of 2 mod 4 16-byte slots (conflict-free
                                    // ds_read_b128), and a row's three A fragments are ONE address register + immediate offsets (planes of wr_max rows each
                                    // needed a v_add per read, into the register the MFMA in front of it was still reading: a write-after-read stall per read)
constexpr int CL_NIT = 12;          // float4 per helper thread and slice: windows of up to 384 rows
#ifndef CL_RING1
#define CL_RING1 4                  // weight blocks in flight per wave with one column tile per wave
#endif

template <int I, int N, class F>
__device__ __forceinline__ void cl_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); cl_static_for<I + 1, N>(f); }
}

}  // namespace

#ifdef CL_STAMPS
__device__ unsigned long long cl_dbg[256 * 4];
__device__ unsigned long long cl_dbg2[256 * 4];      // {cycles before the first K loop, epilogue cycles, first barrier wait of tile 1, tiles}      // developer build: per block {K-loop cycles, barrier-wait cycles, life cycles, life in 10 ns}
#endif

#define CL_SEL(q_, f) ((q_) == 0 ? g.p[0].f : ((q_) == 1 ? g.p[1].f : g.p[2].f))

template <int NRW, int NCW, int RW, int CW>
__global__ __launch_bounds__(512, 2) void conv_limb_kernel(const ConvLimbGroup g) {
  static_assert(RW * CW == 4, "four matrix waves");
  constexpr int TM = 16 * NRW * RW, TN = 16 * NCW * CW;
  extern __shared__ __attribute__((aligned(16))) u16 lds[];      // [2 buffers][wr_max][3 planes][CL_LDB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = g.wr_max * CL_LDB;                           // elements per buffer / 3
  constexpr int PO = CL_LDB;                                     // plane offset inside a row
  auto bar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto tile_word = [&](int idx, int w) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*(gci)(g.tiles + (long long)idx * 4 + w)); };
  // this block's tiles: assign[b * per + i] until -1 (host-balanced)
  const int* mine = g.assign + (long long)blockIdx.x * g.assign_per;
  int gslice = 0;                                                // slices staged / consumed so far (buffer = parity)

  if (wave >= 4) {
    // ============================================================ helper waves: window slices
    const int ht = tid - 256;
    __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < g.assign_per; ++it) {
      const int tile = __builtin_amdgcn_readfirstlane(*(gci)(mine + it));
      if (tile < 0) break;
      const int q = tile_word(tile, 0), mt = tile_word(tile, 1), tw3 = tile_word(tile, 3);
      const int T = CL_SEL(q, T), k = CL_SEL(q, ktaps), dil = CL_SEL(q, dil), Cin = CL_SEL(q, Cin);
      const int Tt = T < TM ? T : TM, wrs = Tt + (k - 1) * dil, S = TM / Tt, wr = S * wrs;
      const int ks = tw3 & 255, ns = (tw3 >> 8) & 255;            // K slice of a split tile: channel blocks [cb0, cb1)
      const int m0 = mt * TM, i0 = m0 / T, ta = m0 - i0 * T;
      const bool ring = CL_SEL(q, x.mode) == 0;
      const float* xb = CL_SEL(q, x.base);
      const int xC = CL_SEL(q, x.C), xmask = ring ? CL_SEL(q, x.lmask) : -1, xrate = CL_SEL(q, x.rate), xoff = CL_SEL(q, x.off) - CL_SEL(q, pad_left);
      const long long xss = CL_SEL(q, x.slot_stride);
      const int* slots = CL_SEL(q, slots);
      const int* pos = CL_SEL(q, pos);
      const bool act = CL_SEL(q, in_act) == ACT_LRELU;
      const float slope = CL_SEL(q, in_slope);
      const float act_slope = act ? slope : 1.f;
      // per thread: the rows it stages (window row w = idx / 8, 4-channel group idx % 8), as float offsets from xb
      int roff[CL_NIT], loff[CL_NIT];
      const int total = wr * 8;
#pragma unroll
      for (int u = 0; u < CL_NIT; ++u) {
        const int idx = ht + 256 * u;
        roff[u] = -1; loff[u] = 0;
        if (idx < total) {
          const int w = idx >> 3, c4 = idx & 7, s = w / wrs, o = w - s * wrs;
          // (a last tile with fewer slots than it has room for reads slot-table entries n .. n + TM / T - 2: the host keeps
          // kSlotTablePad copies of the last slot there, conan_streams::set_slots)
          const int i = i0 + s, slot = slots ? *(gci)(slots + i) : i, pv = (ring && pos) ? *(gci)(pos + slot) : 0;
          const int row = ((ring ? pv * xrate : 0) + xoff + ta + o) & xmask;
          roff[u] = (int)((long long)(ring ? slot : i) * xss) + row * xC + c4 * 4;
          loff[u] = w * CL_RS + c4 * 4;
        }
      }
      const int nblk_all = Cin / 32;
      const int cb0 = ns > 1 ? (nblk_all * ks) / ns : 0, cb1 = ns > 1 ? (nblk_all * (ks + 1)) / ns : nblk_all;
      // Software-pipelined: the loads of slice cb + 1 are issued BEFORE the barrier that publishes slice cb and land while the
      // matrix waves compute; behind the barrier the helper only converts and stores registers.  (Loading, converting and
      // storing a slice between two barriers put a global round trip - 2 us - on every channel block: 3-tap tiles ran as
      // long as 11-tap ones.)
      f32x4 v[CL_NIT];
#ifdef CL_ABL_STAGE      // developer ablation: the helpers only keep the barriers
      for (int cb = cb0; cb < cb1; ++cb) { bar(); ++gslice; }
      continue;
#endif
#pragma unroll
      for (int u = 0; u < CL_NIT; ++u) if (roff[u] >= 0) v[u] = cl_gload(xb + roff[u] + cb0 * 32);
      for (int cb = cb0; cb < cb1; ++cb) {
        u16* dstb = lds + (gslice & 1) * 3 * plane;
#pragma unroll
        for (int u = 0; u < CL_NIT; ++u) {
          if (roff[u] >= 0) {
            unsigned h[2], m[2], l[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              f32x2 s = {v[u][2 * e], v[u][2 * e + 1]};
              // (LeakyReLU with 0 < slope < 1 as max(x, slope * x): the same bits as x > 0 ? x : x * slope; act_slope is 1 without it)
              const f32x2 sx = s * act_slope;
              s = (f32x2){__builtin_fmaxf(s[0], sx[0]), __builtin_fmaxf(s[1], sx[1])};
              cl_split2(s, h[e], m[e], l[e]);
            }
            u16* d = dstb + loff[u];
            *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2*>(d + PO) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2*>(d + 2 * PO) = make_uint2(l[0], l[1]);
          }
        }
        if (cb + 1 < cb1) {
#pragma unroll
          for (int u = 0; u < CL_NIT; ++u) if (roff[u] >= 0) v[u] = cl_gload(xb + roff[u] + (cb + 1) * 32);
        }
        bar();                                                   // slice staged (and the matrix waves are done with the other buffer)
        ++gslice;
      }
    }
    return;
  }

  // ============================================================== matrix waves
#ifdef CL_STAMPS
  unsigned long long st_bar = 0, st_loop = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime(), st_pro = 0, st_epi = 0, st_fb = 0, st_tiles = 0;
#endif
  const int wr_ = wave / CW, wc = wave % CW;
  const int lr = lane & 15, lg = lane >> 4;
  for (int it = 0; it < g.assign_per; ++it) {
    const int tile = __builtin_amdgcn_readfirstlane(*(gci)(mine + it));
    if (tile < 0) break;
    const int q = tile_word(tile, 0), mt = tile_word(tile, 1), nt = tile_word(tile, 2), tw3 = tile_word(tile, 3);
    const int T = CL_SEL(q, T), k = CL_SEL(q, ktaps), dil = CL_SEL(q, dil), Cin = CL_SEL(q, Cin);
    const int Tt = T < TM ? T : TM, wrs = Tt + (k - 1) * dil;
    const int m0 = mt * TM, n0 = nt * TN;
    const int ks = tw3 & 255, ns = (tw3 >> 8) & 255, split_idx = tw3 >> 16;      // K slice of a split tile (ns <= 1: the whole K range)
    const int nblk_all = Cin / 32;
    const int cb0 = ns > 1 ? (nblk_all * ks) / ns : 0, cb1 = ns > 1 ? (nblk_all * (ks + 1)) / ns : nblk_all;
    const int NB = (cb1 - cb0) * k;
    const int ct0 = n0 / 16 + wc * NCW;
    const long long ct_stride = (long long)nblk_all * k * 1536;  // elements per column tile
    const u16* wl = CL_SEL(q, wl) + (long long)ct0 * ct_stride + (long long)cb0 * k * 1536 + lane * 8;
    // this lane's row of each of the wave's row tiles: window row of tap 0
    int abase[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) {
      const int row = (wr_ * NRW + r) * 16 + lr, s = row / Tt, tl = row - s * Tt;
      abase[r] = (s * wrs + tl) * CL_RS + 8 * lg;
    }
    f32x4 acc[NRW][NCW];
#pragma unroll
    for (int r = 0; r < NRW; ++r)
#pragma unroll
      for (int c = 0; c < NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the epilogue's rows: slot and frame counter of this lane's row of every row tile - two dependent loads, issued now so that
    // they are long back when the K loop ends (in the epilogue they were 2 us per tile with nothing to overlap them)
    int eslot[NRW], epos[NRW];
    {
      const int* slots = CL_SEL(q, slots);
      const int* pos = CL_SEL(q, pos);
#pragma unroll
      for (int r = 0; r < NRW; ++r) {
        const int m = m0 + (wr_ * NRW + r) * 16 + lr, i = m / T;      // (rows past the launch's last slot - slot-table entry n, a copy of the last one: computed, never stored)
        eslot[r] = slots ? *(gci)(slots + i) : i;
      }
#pragma unroll
      for (int r = 0; r < NRW; ++r) {
        // (a row past the launch's last slot - a ragged last tile - carries the sign bit in its frame counter: ring rows are masked, so
        // every address formed from it stays in bounds, and the epilogue stores nothing for it.  No register of its own: this build is
        // at its 192-register budget beside the decoder megakernel.)
        const int m = m0 + (wr_ * NRW + r) * 16 + lr;
        epos[r] = (pos ? *(gci)(pos + eslot[r]) : 0) | (m / T < CL_SEL(q, n) ? 0 : (int)0x80000000);
      }
    }
    // weight blocks in flight: four with one column tile per wave (a block is 6 * NRW MFMAs = 0.2 us of work there - two blocks
    // ahead is less than an L2 round trip under load), two with two
    constexpr int RING = NCW == 1 ? CL_RING1 : 2;
    f32x4 bw[RING][NCW][3];
#pragma unroll
    for (int s = 0; s < RING; ++s)
#pragma unroll
      for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[s][c][p] = cl_gload(wl + (s < NB ? s : 0) * 1536 + c * ct_stride + p * 512);
    f32x4 af[NRW][3];
    int j = 0;                                                   // tap of block gb
    const u16* buf = lds;
    const int dstep = dil * CL_RS;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    auto step = [&](const int gb, auto slot_c) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_c)::value;
      if (j == 0) {                                              // first tap of a channel block: its window slice
#ifdef CL_STAMPS
        { const unsigned long long q0 = __builtin_amdgcn_s_memtime(); bar(); st_bar += __builtin_amdgcn_s_memtime() - q0; }
#else
        bar();
#endif
        buf = lds + (gslice & 1) * 3 * plane;
        ++gslice;
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int p = 0; p < 3; ++p) af[r][p] = *reinterpret_cast<const f32x4*>(buf + abase[r] + p * PO);
      }
      const bool more = j + 1 < k;                               // the next tap reads the same slice, dil rows further
      const u16* anext = buf + (more ? (j + 1) * dstep : 0);
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
#pragma unroll
          for (int c = 0; c < NCW; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[SL][c][PB[s]]), __builtin_bit_cast(bf16x8, af[r][PA[s]]), acc[r][c], 0, 0, 0);
#ifndef CL_ABL_A      // (developer ablation: no A re-reads)
          if (s == 0) af[r][2] = *reinterpret_cast<const f32x4*>(anext + abase[r] + 2 * PO);
          if (s == 3) af[r][1] = *reinterpret_cast<const f32x4*>(anext + abase[r] + PO);
          if (s == 5) af[r][0] = *reinterpret_cast<const f32x4*>(anext + abase[r]);
#endif
        }
#ifdef CL_ABL_W
      const int gn = 0;                                          // developer ablation: every weight block from the stream's start (L1 hits)
#else
      const int gn = gb + RING < NB ? gb + RING : 0;             // (past the last block: block 0 again, unused)
#endif
#pragma unroll
      for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[SL][c][p] = cl_gload(wl + (long long)gn * 1536 + c * ct_stride + p * 512);
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        if (s == 0 || s == 3 || s == 5) {
#pragma unroll
          for (int r = 0; r < NRW; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, NCW, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, NRW * NCW, 0);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 3 * NCW, 0);
      j = more ? j + 1 : 0;
    };
#ifdef CL_STAMPS
    const unsigned long long st_l0 = __builtin_amdgcn_s_memtime();
    if (st_tiles == 0) st_pro = st_l0 - st_t0;
    const unsigned long long st_bar0 = st_bar;
    ++st_tiles;
#endif
    // The epilogue's operands - bias and, for c2, the residual rows - are requested here, a whole K loop before their use (loaded in
    // the epilogue, their round trip - 1-2 k cycles - stood at the end of every tile: 43.2 / 45.1 -> 40.5 / 42.2 us per launch of the
    // C = 256 c2 convs; issued behind the first blocks of the loop instead - vmcnt is in-order: the loop's counted waits for weight
    // blocks then also wait for these - 42.4 / 43.7).  Only in the builds whose register budget
    // has room for them (16 + 4 registers with four row tiles and one column tile per wave).
    constexpr bool PRE = NRW * NCW <= 4;      // the residual rows too; the bias in every build
    f32x4 pre_b[NCW], pre_r[PRE ? NRW : 1][PRE ? NCW : 1];
    auto pre_issue = [&]() __attribute__((always_inline)) {
     if constexpr (!PRE) {
      const float* bias = CL_SEL(q, bias);
      const int Cout = CL_SEL(q, Cout);
#pragma unroll
      for (int c = 0; c < NCW; ++c) {
        const int cc = (ct0 + c) * 16 + 4 * lg;
        pre_b[c] = (bias && cc < Cout) ? cl_gload(bias + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
     } else {
      const float* bias = CL_SEL(q, bias);
      const int Cout = CL_SEL(q, Cout);
      const bool hres = CL_SEL(q, has_res) != 0;
      const bool rring = CL_SEL(q, res.mode) == 0;
      const float* rb = CL_SEL(q, res.base);
      const int rC = CL_SEL(q, res.C), rmask = rring ? CL_SEL(q, res.lmask) : -1, rrate = CL_SEL(q, res.rate), roffs = CL_SEL(q, res.off);
      const long long rss = CL_SEL(q, res.slot_stride);
#pragma unroll
      for (int c = 0; c < NCW; ++c) {
        const int cc = (ct0 + c) * 16 + 4 * lg;
        pre_b[c] = (bias && cc < Cout) ? cl_gload(bias + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
          const int m = m0 + (wr_ * NRW + r) * 16 + lr, i = m / T, t = m - i * T;
          const int rrow = ((rring ? epos[r] * rrate : 0) + roffs + t) & rmask;
          pre_r[r][c] = (hres && cc < Cout) ? cl_gload(rb + (long long)(rring ? eslot[r] : i) * rss + (long long)rrow * rC + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    };
    int gb = 0;
    pre_issue();
    for (; gb + RING <= NB; gb += RING) cl_static_for<0, RING>([&](auto sl) __attribute__((always_inline)) { step(gb + decltype(sl)::value, sl); });
    cl_static_for<0, RING - 1>([&](auto sl) __attribute__((always_inline)) { if (gb + decltype(sl)::value < NB) step(gb + decltype(sl)::value, sl); });
#ifdef CL_STAMPS
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
    st_loop += __builtin_amdgcn_s_memtime() - st_l0;
    (void)st_bar0;
    const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
    // ---------------- split-K tail: the slices' partial tiles meet in memory.  No workgroup barrier (the helper waves are already
    // staging the next tile behind the K loop's barriers): every matrix wave hands over ITS sub-tile on its own - agent-scope
    // write-through stores, s_waitcnt, a ticket per (split tile, wave); the wave that draws the last ticket loads all slices' partials
    // of that sub-tile with sc1 loads, sums them IN SLICE ORDER (its own from memory too: bit-reproducible whoever is last) and runs the
    // epilogue.  The protocol of conv_mfma's split-K hand-off, per wave; no fence, nothing invalidates an L2.
    bool run_epi = true;
    if (ns > 1) {
      float* const part = g.slab + (long long)split_idx * ns * (TM * TN) + (long long)wave * (NRW * NCW * 256);
      float* const mine = part + (long long)ks * (TM * TN);
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) __hip_atomic_store(mine + ((r * NCW + c) * 4 + e) * 64 + lane, acc[r][c][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int* const ctr = g.counters + split_idx * 4 + wave;
      int ticket = 0;
      if (lane == 0) ticket = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      run_epi = ticket == ns - 1;
      if (run_epi) {
        if (lane == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int c = 0; c < NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < ns; ++s2) {
          const float* src = part + (long long)s2 * (TM * TN);
          float pv[NRW * NCW * 4];
#pragma unroll
          for (int f = 0; f < NRW * NCW * 4; ++f) pv[f] = __hip_atomic_load(src + f * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int r = 0; r < NRW; ++r)
#pragma unroll
            for (int c = 0; c < NCW; ++c)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[r][c][e] += pv[(r * NCW + c) * 4 + e];
        }
      }
    }
    // ---------------- epilogue: bias -> activation -> + residual -> (pixel-shuffled) store, 4 packed columns per lane
    if (run_epi) {

      const int oact = CL_SEL(q, out_act);
      const float oslope = CL_SEL(q, out_slope);
      const int shuf = CL_SEL(q, shuffle_r), Cout = CL_SEL(q, Cout), Cq = Cout / shuf;
      const bool yring = CL_SEL(q, y.mode) == 0;
      float* yb = CL_SEL(q, y.base);
      const int yC = CL_SEL(q, y.C), ymask = yring ? CL_SEL(q, y.lmask) : -1, yrate = CL_SEL(q, y.rate), yoff = CL_SEL(q, y.off);
      const long long yss = CL_SEL(q, y.slot_stride);
      const bool hres = CL_SEL(q, has_res) != 0;
      const bool rring = CL_SEL(q, res.mode) == 0;
      const float* rb = CL_SEL(q, res.base);
      const int rC = CL_SEL(q, res.C), rmask = rring ? CL_SEL(q, res.lmask) : -1, rrate = CL_SEL(q, res.rate), roffs = CL_SEL(q, res.off);
      const long long rss = CL_SEL(q, res.slot_stride);
      float* y2b = CL_SEL(q, y2_base);
      const float y2s = CL_SEL(q, y2_slope);
#pragma unroll
      for (int r = 0; r < NRW; ++r) {
        const int m = m0 + (wr_ * NRW + r) * 16 + lr, i = m / T, t = m - i * T;
        const int slot = eslot[r], pv = epos[r];
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
          const int cc = (ct0 + c) * 16 + 4 * lg;                // first of this lane's 4 packed columns
          if (cc < Cout && pv >= 0) {
            f32x4 o = acc[r][c] + pre_b[c];
            if (oact == ACT_LRELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * oslope;
            }
            if constexpr (PRE) { if (hres) o += pre_r[r][c]; }
            else if (hres) {
              const int rrow = ((rring ? pv * rrate : 0) + roffs + t) & rmask;
              o += cl_gload(rb + (long long)(rring ? slot : i) * rss + (long long)rrow * rC + cc);
            }
            int jj = 0, oc = cc;
            if (shuf > 1) { jj = cc / Cq; oc = cc - jj * Cq; }
            const int yrow = ((yring ? pv * yrate : 0) + yoff + t * shuf + jj) & ymask;
            const long long yo = (long long)(yring ? slot : i) * yss + (long long)yrow * yC + oc;
            cl_gstore(yb + yo, o);
            if (y2b) {
              f32x4 o2;
#pragma unroll
              for (int e = 0; e < 4; ++e) o2[e] = o[e] > 0.f ? o[e] : o[e] * y2s;
              cl_gstore(y2b + yo, o2);
            }
          }
        }
      }
    }
#ifdef CL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_epi += __builtin_amdgcn_s_memtime() - st_e0;
#endif
  }
#ifdef CL_STAMPS
  if (tid == 0 && blockIdx.x < 256) { cl_dbg2[blockIdx.x * 4] = st_pro; cl_dbg2[blockIdx.x * 4 + 1] = st_epi; cl_dbg2[blockIdx.x * 4 + 2] = st_fb; cl_dbg2[blockIdx.x * 4 + 3] = st_tiles; }
  if (tid == 0 && blockIdx.x < 256) {
    cl_dbg[blockIdx.x * 4] = st_loop; cl_dbg[blockIdx.x * 4 + 1] = st_bar;
    cl_dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - st_t0; cl_dbg[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
}

// ------------------------------------------------------------------------------------------------ host side

namespace {

struct CLShape { int NRW, NCW, RW, CW; };
// Tile shapes.  The weights come through the CU's vector memory path (64 bytes per clock) from L2: per 32-channel block a
// TM x TN tile needs 6 * TN * 32 bytes of them against 1.5 * TM * TN / 16 clocks of MFMA per SIMD - MFMA time / load time = TM / 32
// with every weight loaded once per workgroup, and chip-wide the L2 has to deliver tiles * K * TN * 6 bytes.  So: tall tiles,
// and the four matrix waves on DISJOINT column tiles (CW = 4; the A fragments, which every wave then reads, come from LDS).
// What this kernel is used for follows from that (measured at 64 streams, rocprofv3 kernel times):
//   * ups.2 / ups.3 (160 / 640 rows per slot, 80- and 160-row tiles): 33 us each against 57 / 28 us of conv_mfma's f32 passes;
//   * ups.1 (32 rows per slot, 2048 x 640 outputs) only tiles into 320 tiles of 64 x 64 - two rounds, 78 us against 72 - or
//     into 64 x 80 tiles with all four waves on the same five column tiles (4x the loads, 62 us): it stays with conv_mfma,
//     whose split-K tail evens out the 1.25 tiles per CU;
//   * ups.0 (4 rows per slot; K = 8192, 100 MB of limb weights) and the C = 256 ResBlock convs (32 rows per slot; 64-row
//     tiles because a 4-slot window of the dilation-5 conv does not fit in LDS twice) are bound by the L2 -> CU weight
//     traffic at M <= 64: 118 us against conv_mfma's 84, and 6 x 48 us against resblock_pair's 3 x 98.  (-DCL_STAMPS: the K
//     loops of the ResBlock convs run at 0.37 of the MFMA rate, 2.2 GHz - not power-limited -, about 1000 cycles per block of
//     24 MFMAs: 12 bytes per clock and CU of weights = 7 TB/s chip-wide, which is also what resblock_pair's f32 weights come to
//     - 705 MB per launch in 98 us; 128 x 32 tiles with two row waves per column tile reload the weights per wave and
//     measured 55 us.)  They keep the f32 kernels (CONAN_RB_NOPAIR=1 runs the ResBlock convs through this kernel: the group
//     path and its tile balancing are tested that way).
const CLShape kShapes[] = {{4, 1, 1, 4}, {5, 1, 1, 4}, {5, 2, 2, 2}};
constexpr int kNumShapes = (int)(sizeof(kShapes) / sizeof(kShapes[0]));

template <int NRW, int NCW, int RW, int CW>
void launch_cl(const ConvLimbGroup& g, int grid, size_t lds_bytes, hipStream_t st) {
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)conv_limb_kernel<NRW, NCW, RW, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((conv_limb_kernel<NRW, NCW, RW, CW>), dim3(grid), dim3(512), lds_bytes, st, g);
}

bool shape_fits(const CLShape& s, const ConvArgs& a, bool ragged_ok) {
  const int TM = 16 * s.NRW * s.RW, TN = 16 * s.NCW * s.CW;
  const long long M = (long long)a.n * a.T;
  // (tiles of whole slots - T < TM - may end in a tile with fewer slots than it has room for: the grouped launches of the C = 256
  // stage at an odd number of streams, which otherwise fell back to conv_mfma's small-M plan; a single problem - ups.1 - keeps the
  // rule that its rows divide into tiles: ragged 160-row tiles measured 104 us against 70 us of its f32 launch.  Tiles inside a slot
  // must divide it.)
  if (a.T < TM ? ((TM % a.T) != 0 || (!ragged_ok && (M % TM) != 0)) : ((a.T % TM) != 0 || (M % TM) != 0)) return false;
  // (a ragged last tile reads the slot table up to TM / T - 1 entries past n: the table carries kSlotTablePad copies of the last slot)
  if (a.T < TM && (M % TM) != 0 && TM / a.T - 1 > kSlotTablePad) return false;
  const int cols = ((a.Cout + 15) / 16) * 16;
  return cols % TN == 0;
}

}  // namespace

// the arguments this kernel covers (the rest of ConvArgs stays with conv_mfma)
bool conv_limb_supported(const ConvArgs& a) {
  if (!a.wl || a.x.mode != 0 || a.Cin % 32 || a.Cout % 4 || a.has_m1 || a.has_m2 || a.bvec || a.lens || a.out_scale != 1.f) return false;
  if (a.in_act != ACT_NONE && a.in_act != ACT_LRELU) return false;
  if (a.in_act == ACT_LRELU && !(a.in_slope > 0.f && a.in_slope <= 1.f)) return false;      // (the helpers form it as max(x, slope * x))
  if (a.out_act != ACT_NONE && a.out_act != ACT_LRELU) return false;
  if (a.shuffle_r > 1 && ((a.Cout / a.shuffle_r) % 4 || a.Cout % a.shuffle_r)) return false;
  if (a.x.C % 4 || a.y.C % 4 || (a.has_res && a.res.C % 4)) return false;
  return true;
}

// Split-K tail of a single problem: with tiles = q * CUs + rem (rem > 0, q >= 1) the last rem tiles are cut into S K slices over
// their channel blocks, rem * S <= CUs, so that every CU gets q + 1 / S tiles of MFMA work instead of rem CUs getting one tile more
// (ups.1 at 64 streams: 320 tiles of 64 x 64 on 256 CUs - two rounds for 1.25 rounds of work).  -> S (1: no split)
static int tail_slices(const ConvArgs& a, const CLShape& s, int num_cu) {
  const int TM = 16 * s.NRW * s.RW, TN = 16 * s.NCW * s.CW;
  const long long tiles = (((long long)a.n * a.T + TM - 1) / TM) * ((((a.Cout + 15) / 16) * 16) / TN);
  const long long rem = tiles % num_cu;
  if (tiles <= num_cu || rem == 0) return 1;
  int S = (int)std::min<long long>(8, num_cu / rem);
  S = std::min(S, a.Cin / 32);                                   // at least one channel block per slice
  return S < 2 ? 1 : S;
}
int conv_limb_tail_slices(const ConvArgs& a, int shape, int num_cu) { return shape >= 0 && shape < kNumShapes ? tail_slices(a, kShapes[shape], num_cu) : 1; }

// tile shape index for a group of problems (same n, T and column count), or -1: the shape with the smallest estimated
// makespan among those that fit and give every CU a tile
// plan_n > 0 (fixed-plan stream-sets): tile counts, the fill-the-chip thresholds and the cost model use plan_n slots instead of the
// launch's own, and the launch's own last tile may be ragged where the full set's is not - the choice must not depend on the active slots
int conv_limb_shape(const ConvArgs* p, int nprob, int num_cu, int plan_n, bool tail_split) {
  static const int forced = (dev_getenv("CONAN_CL_SHAPE") && *dev_getenv("CONAN_CL_SHAPE")) ? atoi(dev_getenv("CONAN_CL_SHAPE")) : -1;      // developer switch
  int best = -1; double best_cost = 1e30;
  for (int si = 0; si < kNumShapes; ++si) {
    const CLShape& s = kShapes[si];
    const int TM = 16 * s.NRW * s.RW, TN = 16 * s.NCW * s.CW;
    bool ok = true;
    double units = 0, umax = 0;
    long long tiles = 0;
    for (int q = 0; q < nprob && ok; ++q) {
      if (plan_n > 0) {      // the decision as the full stream-set would make it; the launch's own rows may then end in a ragged tile
        ConvArgs full = p[q]; full.n = plan_n;
        ok = shape_fits(s, full, nprob > 1) && shape_fits(s, p[q], true);
      } else ok = shape_fits(s, p[q], nprob > 1);
      if (!ok) break;
      const int Tt = std::min(p[q].T, TM), wr = (TM / Tt) * (Tt + (p[q].ktaps - 1) * p[q].dil);
      if (wr > 32 * CL_NIT || (size_t)2 * 3 * wr * CL_LDB * 2 > 126 * 1024) { ok = false; break; }
      const long long t = (((long long)(plan_n > 0 ? plan_n : p[q].n) * p[q].T + TM - 1) / TM) * ((((p[q].Cout + 15) / 16) * 16) / TN);
      const double u = (double)p[q].ktaps * p[q].Cin * TM * TN;
      tiles += t; units += u * t; umax = std::max(umax, u);
    }
    // (a single problem - ups.2 / ups.3 - from a tile for every second CU on: 16 / 24 streams measured 0.767 -> 0.762 / 0.827 -> 0.810 ms
    // per step against conv_mfma's f32 passes, nothing either way below that; a group of problems is what the C = 256 ResBlock stage
    // launches, whose alternative - the f32 pair kernel - takes the time of one whole tile at any size: from a third of the CUs on,
    // streams.hip build_vocoder)
    if (!ok || tiles * (nprob > 1 ? 3 : 2) < num_cu) continue;
    // (a single problem in 64-row tiles - ups.1: 320 tiles, two rounds - measured slower than conv_mfma's f32 pass with its
    // split-K tail, 78 against 72 us; groups of problems are list-scheduled and take them)
    // ... unless the launch may split the K range of its tail tiles (round 6): 1.25 rounds then cost 1 + 1 / S
    const int tailS = (nprob == 1 && tail_split && plan_n == 0) ? tail_slices(p[0], s, num_cu) : 1;
    if (nprob == 1 && TM < 80 && forced < 0 && tailS < 2) continue;
    if (forced >= 0) { if (si == forced) return si; continue; }
    // equal tiles run in rounds; tiles of several costs are list-scheduled (at least the largest one, at least the average)
    const double rounds = std::ceil((double)tiles / num_cu);
    const double makespan = nprob == 1 ? (tailS >= 2 ? (double)(tiles / num_cu) + 1.0 / tailS + 0.1 : rounds) * umax : std::max(units / num_cu, umax);
    const double cost = makespan * (s.CW == 4 ? 1.0 : 1.05);     // shared column tiles: redundant weight loads
    if (cost < best_cost) { best_cost = cost; best = si; }
  }
  return best;
}

size_t conv_limb_lds_bytes(const ConvArgs* p, int nprob, int shape, int* wr_max_out) {
  const CLShape& s = kShapes[shape];
  const int TM = 16 * s.NRW * s.RW;
  int wr_max = 0;
  for (int q = 0; q < nprob; ++q) {
    const int Tt = std::min(p[q].T, TM);
    wr_max = std::max(wr_max, (TM / Tt) * (Tt + (p[q].ktaps - 1) * p[q].dil));
  }
  *wr_max_out = wr_max;
  return (size_t)2 * 3 * wr_max * CL_LDB * 2;
}

// Balanced tile lists per launch shape, cached in device memory (a handful per model and device, never freed).
static bool cl_schedule(ConvLimbGroup& g, int shape, int num_cu, int* grid_out, int tailS) {
  struct Key { int v[17]; bool operator<(const Key& o) const { return memcmp(v, o.v, sizeof(v)) < 0; } };
  struct Val { const int* tiles; const int* assign; int per, grid; };
  static std::map<Key, Val> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Key key; memset(&key, 0, sizeof(key));
  key.v[0] = shape; key.v[1] = num_cu; key.v[2] = dev; key.v[3] = g.nprob; key.v[16] = tailS;
  for (int q = 0; q < g.nprob; ++q) { key.v[4 + 4 * q] = g.p[q].n; key.v[5 + 4 * q] = g.p[q].T; key.v[6 + 4 * q] = g.p[q].Cout; key.v[7 + 4 * q] = g.p[q].ktaps * 4096 + g.p[q].Cin; }
  auto it = cache.find(key);
  if (it == cache.end()) {
    const CLShape& s = kShapes[shape];
    const int TM = 16 * s.NRW * s.RW, TN = 16 * s.NCW * s.CW;
    struct Tl { int q, mt, nt, w3; double cost; };
    std::vector<Tl> tl;
    for (int q = 0; q < g.nprob; ++q) {
      const int mts = (int)(((long long)g.p[q].n * g.p[q].T + TM - 1) / TM), nts = (((g.p[q].Cout + 15) / 16) * 16) / TN;
      // n-tile outermost: workgroups that run at the same time then share their weight columns' rows ... the m tiles of one
      // n tile are adjacent in the list
      // (split tail - a single problem: the last tiles % CUs tiles of this order are items of tailS K slices each)
      const long long total = (long long)mts * nts, first_split = tailS >= 2 ? total - total % num_cu : total;
      long long idx = 0;
      for (int nt = 0; nt < nts; ++nt)
        for (int mt = 0; mt < mts; ++mt, ++idx) {
          const double cost = (double)g.p[q].ktaps * g.p[q].Cin;
          if (idx < first_split) tl.push_back({q, mt, nt, 0, cost});
          else for (int ks = 0; ks < tailS; ++ks) tl.push_back({q, mt, nt, ks | (tailS << 8) | ((int)(idx - first_split) << 16), cost / tailS + 0.02 * cost});
        }
    }
    // XCD of every item.  With a multiple of 8 n tiles (or fewer than 8) the rule below keeps an n tile's m tiles on one XCD (or on
    // 8 / nts of them).  Otherwise - ups.1: 10 n tiles - "n tile % 8" gives two XCDs twice the work of the others (two rounds for 1.25
    // rounds of tiles, and a split tail that lands on those two XCDs again): the items, in (n tile, m tile) order, are cut into eight
    // runs of equal cost instead - an n tile's weight columns are then shared by one or two XCDs.
    std::vector<int> xcd_of(tl.size(), -1);
    {
      bool by_cost = tailS >= 2;
      for (int q = 0; q < g.nprob; ++q) { const int nts = (((g.p[q].Cout + 15) / 16) * 16) / TN; by_cost = by_cost || (nts > 8 && nts % 8 != 0); }
      if (by_cost) {
        double total = 0, cum = 0;
        for (auto& t : tl) total += t.cost;
        for (size_t e = 0; e < tl.size(); ++e) { xcd_of[e] = std::min(7, (int)(cum * 8.0 / total)); cum += tl[e].cost; }
      }
    }
    {      // (sort items by cost, longest first, carrying their XCD along)
      std::vector<size_t> ord(tl.size());
      for (size_t e = 0; e < ord.size(); ++e) ord[e] = e;
      std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return tl[a].cost > tl[b].cost; });
      std::vector<Tl> tl2(tl.size()); std::vector<int> x2(tl.size());
      for (size_t e = 0; e < ord.size(); ++e) { tl2[e] = tl[ord[e]]; x2[e] = xcd_of[ord[e]]; }
      tl.swap(tl2); xcd_of.swap(x2);
    }
    const int grid = (int)std::min<size_t>(tl.size(), (size_t)num_cu);
    std::vector<std::vector<int>> per(grid);
    std::vector<double> load(grid, 0.0);
    // Longest first onto the least loaded block - among the blocks of ONE XCD (workgroups are dealt to the 8 XCDs round-robin:
    // block b runs on XCD b % 8): the tiles of an n tile, which stream the same weight columns, then share one L2 instead
    // of pulling those columns into all eight.  With fewer than 8 n tiles an n tile's m tiles are spread over 8 / nts XCDs.
    for (size_t e = 0; e < tl.size(); ++e) {
      const int nts = (((g.p[tl[e].q].Cout + 15) / 16) * 16) / TN;
      int x = tl[e].nt % 8;
      if (nts < 8) { const int share = 8 / nts; x = (tl[e].nt + nts * (tl[e].mt % share)) % 8; }
      if (xcd_of[e] >= 0) x = xcd_of[e];
      if (grid < 8) x = 0;
      int b = -1;
      for (int c = x; c < grid; c += (grid < 8 ? 1 : 8)) if (b < 0 || load[c] < load[b]) b = c;
      if (b < 0) b = 0;
      per[b].push_back((int)e); load[b] += tl[e].cost;
    }
    size_t mx = 0;
    for (auto& v : per) mx = std::max(mx, v.size());
    const int ap = (int)mx + 1;
    std::vector<int> flat(tl.size() * 4), asg((size_t)grid * ap, -1);
    for (size_t e = 0; e < tl.size(); ++e) { flat[e * 4] = tl[e].q; flat[e * 4 + 1] = tl[e].mt; flat[e * 4 + 2] = tl[e].nt; flat[e * 4 + 3] = tl[e].w3; }
    for (int b = 0; b < grid; ++b) for (size_t i2 = 0; i2 < per[b].size(); ++i2) asg[(size_t)b * ap + i2] = per[b][i2];
    int *dt = nullptr, *da = nullptr;
    if (hipMalloc(&dt, flat.size() * sizeof(int)) != hipSuccess || hipMalloc(&da, asg.size() * sizeof(int)) != hipSuccess) return false;
    (void)hipMemcpy(dt, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemcpy(da, asg.data(), asg.size() * sizeof(int), hipMemcpyHostToDevice);
    it = cache.emplace(key, Val{dt, da, ap, grid}).first;
  }
  g.tiles = it->second.tiles; g.assign = it->second.assign; g.assign_per = it->second.per;
  *grid_out = it->second.grid;
  return true;
}

bool launch_conv_limb(const ConvLimbGroup& gin, int shape, int num_cu, hipStream_t st) {
  ConvLimbGroup g = gin;
  if (shape < 0 || shape >= kNumShapes || g.nprob < 1 || g.nprob > 3) return false;
  const size_t lds = conv_limb_lds_bytes(g.p, g.nprob, shape, &g.wr_max);
  int grid = 0;
  // the split tail: where the caller gave the launch a slab and tickets, and they are large enough
  int tailS = (g.nprob == 1 && g.slab && g.counters) ? tail_slices(g.p[0], kShapes[shape], num_cu) : 1;
  if (tailS >= 2) {
    const int TM = 16 * kShapes[shape].NRW * kShapes[shape].RW, TN = 16 * kShapes[shape].NCW * kShapes[shape].CW;
    const long long tiles = (((long long)g.p[0].n * g.p[0].T + TM - 1) / TM) * ((((g.p[0].Cout + 15) / 16) * 16) / TN), rem = tiles % num_cu;
    if (rem * tailS * TM * TN > g.slab_floats || rem * 4 > g.max_counters || rem >= 65536) tailS = 1;
  }
  if (tailS < 2) { g.slab = nullptr; g.counters = nullptr; }
  if (!cl_schedule(g, shape, num_cu, &grid, tailS) || grid <= 0) return false;
  switch (shape) {
    case 0: launch_cl<4, 1, 1, 4>(g, grid, lds, st); break;
    case 1: launch_cl<5, 1, 1, 4>(g, grid, lds, st); break;
    case 2: launch_cl<5, 2, 2, 2>(g, grid, lds, st); break;
    default: return false;
  }
#ifdef CL_STAMPS
  {
    (void)hipStreamSynchronize(st);
    unsigned long long h[256 * 4];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(cl_dbg), sizeof(h));
    double lp = 0, br = 0, lf = 0, rt = 0, lfmax = 0; int nb = 0;
    for (int b = 0; b < grid && b < 256; ++b) { lp += h[b * 4]; br += h[b * 4 + 1]; lf += h[b * 4 + 2]; rt += h[b * 4 + 3]; lfmax = std::max(lfmax, (double)h[b * 4 + 3]); ++nb; }
    {
      unsigned long long h2[256 * 4];
      (void)hipMemcpyFromSymbol(h2, HIP_SYMBOL(cl_dbg2), sizeof(h2));
      double pro = 0, epi = 0, tl = 0; int nb2 = 0;
      for (int b2 = 0; b2 < grid && b2 < 256; ++b2) { pro += h2[b2 * 4]; epi += h2[b2 * 4 + 1]; tl += h2[b2 * 4 + 3]; ++nb2; }
      fprintf(stderr, "   before the first K loop %.0f cyc, epilogues %.0f cyc (%.2f tiles per block); block 0: %llu / %llu, block %d: %llu / %llu\n", pro / nb2, epi / nb2, tl / nb2,
              h2[0], h2[1], nb2 - 1, h2[(nb2 - 1) * 4], h2[(nb2 - 1) * 4 + 1]);
    }
    fprintf(stderr, "[conv_limb %s nprob %d k %d Cin %d T %d] K loops %.0f cyc = %.0f%% of life (barrier waits in them %.0f%%), life %.0f cyc = %.1f us avg / %.1f us max, clock %.2f GHz\n",
            conv_limb_name(shape), g.nprob, g.p[0].ktaps, g.p[0].Cin, g.p[0].T, lp / nb, 100 * lp / lf, 100 * br / lf, lf / nb, rt / nb / 100.0, lfmax / 100.0, (lf / nb) / (rt / nb / 100.0) / 1e3);
  }
#endif
  return true;
}

const char* conv_limb_name(int shape) {
  static const char* names[] = {"cnk::conv_limb_kernel<4, 1, 1, 4>", "cnk::conv_limb_kernel<5, 1, 1, 4>", "cnk::conv_limb_kernel<5, 2, 2, 2>"};
  return shape >= 0 && shape < kNumShapes ? names[shape] : "cnk::conv_limb_kernel<?>";
}

}  // namespace cnk
