// resblock_fused: one ResBlock1 unit of the causal HiFi-GAN MRF in ONE tile pass (hifigan_causal.py:230-238)
//     xt = c1(leaky_relu(x))            causal conv, k taps, dilation d
//     y  = c2(leaky_relu(xt)) + x       causal conv, k taps, dilation 1
// for the three kernel-size branches of a stage in one launch (gfx950, exact-f32 MFMA v_mfma_f32_16x16x4_f32).
//
// Why a tile pass: with c1 and c2 as separate launches every intermediate (xt, and the LeakyReLU'd twins the
// direct-to-LDS conv kernel needs) makes a round trip through HBM/L2 and every K-step of every tap re-reads its A tile
// from L2.  Here a workgroup owns RO = 16*NR2 output rows of one stream for ALL C channels:
//   * the input window (RO + halo rows x C) is loaded from the ring ONCE, LeakyReLU applied on the way, into LDS; the
//     k taps of c1 are row-shifted reads of that one window;
//   * c1 is computed for 16*(NR2+1) rows (the k-1 rows of left context c2 needs are recomputed, not fetched: 6 % extra
//     MFMAs at RO = 128), bias + LeakyReLU applied, and kept in LDS as c2's operand - xt never leaves the CU;
//   * c2 reads xt from LDS, its accumulators go through LDS to the helper waves, which add bias + residual and store
//     full 16-byte channel-last rows.
// Wave roles (512 threads): waves 0-3, one per SIMD, issue nothing but LDS fragment reads, weight-fragment loads and
// MFMAs; waves 4-7 load the next tile's window while c2 runs (the window is dead then) and write the previous tile's
// output while c1 runs.  A wave owns whole 16-column strips of the output, so its B operands (weights, fragment-major:
// 1 KiB per 16x16 operand, streamed from L2 straight into registers, prefetched one tap - C = 128: half a tap - ahead) are private and the K
// loops contain NO barrier: per 16-deep K group one ds_read_b128 per row tile + one 16-byte global load per column
// tile feed 4 x (row tiles x column tiles) MFMAs.  Four block barriers per tile in all.
// LDS rows are padded to C + 8 floats: the 16 lanes of every ds_read_b128 lane group hit 16 distinct 16-byte slots.
// Workgroups are persistent over a host-made longest-first tile list with an agent-scope draw counter (branch costs are
// k = 11 : 7 : 3).  MERGE build (last dilation of a stage): a workgroup owns (slot, row tile) groups, runs the branches of a
// group one after the other and stores only leaky_relu(mean of the three outputs) - see RBArgs::merge.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <vector>

#include "kernels.h"

#ifndef RB_ABLATE
#define RB_ABLATE 0     // developer builds (tools/rb_bench): bit 0 no window loads, bit 1 no output writes, bit 2 cycle stamps
#endif

namespace cnk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers that were themselves loaded from memory (RBProb lives in a device array) are "generic" to the compiler and
// would be accessed with flat_load / flat_store, which count on BOTH vmcnt and lgkmcnt and force full drains around
// every LDS read.  These helpers state the address space: global_load_dwordx4 / global_store_dwordx4.
// (native vector type: HIP's float4 is a class whose copy constructor takes a generic reference)
typedef const f32x4 __attribute__((address_space(1)))* gcf4;
typedef f32x4 __attribute__((address_space(1)))* gf4;
typedef const float __attribute__((address_space(1)))* gcf1;
__device__ __forceinline__ float4 gload4(const float* p) { const f32x4 v = *(gcf4)(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float gload1(const float* p) { return *(gcf1)(p); }
__device__ __forceinline__ void gstore4(float* p, const float4 v) { *(gf4)(p) = (f32x4){v.x, v.y, v.z, v.w}; }

template <int C, int NR2>
struct RBGeom {
  static constexpr int NR1 = NR2 + 1;                 // c1 row tiles: 16*NR1 >= 16*NR2 + (k-1) for k <= 17
  static constexpr int RO = 16 * NR2;                 // output rows per tile
  static constexpr int XT_ROWS = 16 * NR1;
  static constexpr int MAXSPAN = 50;                  // (k-1)*dil of c1 (k = 11, dil = 5)
  static constexpr int WR_MAX = XT_ROWS + MAXSPAN;    // window rows
  static constexpr int LDX = C + (C >= 128 ? 4 : 8);   // (C = 128: 4 floats of padding keep the block at 128 KB - with 8 a rowconv block no longer fits beside it)
  static constexpr int LDS_FLOATS = (WR_MAX + XT_ROWS) * LDX;
  static constexpr int NCT = C / 16;                  // 16-column tiles
  static constexpr int RSPLIT = NCT >= 4 ? 1 : 4 / NCT;   // C = 32: two waves share a column strip and split the rows
  static constexpr int NCW = NCT >= 4 ? NCT / 4 : 1;      // column tiles per matrix wave
  static constexpr int NRW1 = (NR1 + RSPLIT - 1) / RSPLIT;
  static constexpr int NRW2 = (NR2 + RSPLIT - 1) / RSPLIT;
  static constexpr int KQ = C / 16;                   // 16-deep K groups per tap
  static constexpr int RING = C >= 128 ? KQ / 2 : KQ; // weight-fragment groups in flight (C = 128: half a tap - 32 registers fewer)
  static constexpr int C4 = C / 4;
  static_assert(C % 32 == 0 && (KQ & (KQ - 1)) == 0 && KQ >= 2, "channel count");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

// Tap-0 weight fragments of a phase, issued in group order (the in-loop waits are counted vmcnt(N): hipcc merges the
// prologue's issue order with the loop's at the loop header, and a reversed prologue turns every tap's first wait into
// vmcnt(0)).  Called ahead of the barrier that precedes the phase, so the loads fly while the block synchronises.
template <int NCW, int RING>
__device__ __forceinline__ void rb_prefetch_w(float4 (&bw)[RING][NCW], const float* __restrict__ wl, const long long ct_stride) {
#pragma unroll
  for (int q = 0; q < RING; ++q) {
#pragma unroll
    for (int c = 0; c < NCW; ++c) bw[q][c] = gload4(wl + c * ct_stride + q * 256);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One GEMM phase of a matrix wave: acc[r][c] += sum over (tap j, K group q) of A(rows of tile r shifted by j*tap_stride)
// x W(j, q, column tile c).  `src` is the LDS operand image (row stride LDX), `wl` the wave's first weight fragment
// (+ lane*4), column tiles ct_stride floats apart, groups 256 floats apart; bw holds tap 0 on entry and tap 0 of
// (wl_next, ct_stride_next) on exit.
template <int NRW, int NCW, int LDX, int KQ, int RING>
__device__ __forceinline__ void rb_gemm(const float* __restrict__ src, const int row0, const int tap_stride, const int k,
                                        const float* __restrict__ wl, const long long ct_stride, const float* __restrict__ wl_next,
                                        const long long ct_stride_next, f32x4 (&acc)[NRW][NCW], float4 (&bw)[RING][NCW], const int lane) {
  static_assert(RING == KQ || 2 * RING == KQ, "weight ring: a tap or half a tap");
  const float* abase = src + (row0 + (lane & 15)) * LDX + 4 * (lane >> 4);
  float4 af[NRW];      // ONE fragment set: row tile r of the next group is read right after the last MFMA that uses af[r]
#pragma unroll
  for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const float4*>(abase + r * 16 * LDX);
  const int tstep = tap_stride * LDX;
  for (int j = 0; j < k; ++j) {
    const float* arow = abase + j * tstep;
    // the weight registers of a finished group are refilled with the same group of the next tap; behind the last tap
    // that is tap 0 of the NEXT phase (c2 of this tile / c1 of the block's next tile): no phase starts with a cold load
    const bool last = j + 1 == k;
    const float* wnext = last ? wl_next : wl + (long long)(j + 1) * KQ * 256;
    const long long cnext = last ? ct_stride_next : ct_stride;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      // next group's A rows: same tap, next 16 channels; after the last group of a tap the next tap's first group (past
      // the last tap: rows of the neighbouring LDS region, unused)
      const float* anext = (q + 1 < KQ) ? arow + (q + 1) * 16 : arow + tstep;
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].x, bw[q % RING][c].x, acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].y, bw[q % RING][c].y, acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].z, bw[q % RING][c].z, acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r) {
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].w, bw[q % RING][c].w, acc[r][c], 0, 0, 0);
        af[r] = *reinterpret_cast<const float4*>(anext + r * 16 * LDX);
      }
      // this group's weight registers are free: fetch the group RING ahead (the same group of the next tap, or - half-tap
      // ring - the second half of this tap / the first half of the next)
      if (q + RING < KQ) {
#pragma unroll
        for (int c = 0; c < NCW; ++c) bw[q % RING][c] = gload4(wl + (long long)j * KQ * 256 + c * ct_stride + (q + RING) * 256);
      } else {
#pragma unroll
        for (int c = 0; c < NCW; ++c) bw[q % RING][c] = gload4(wnext + c * cnext + (q + RING - KQ) * 256);
      }
      // pin the order: three plain passes, then one LDS read behind each row tile's last MFMA, then the weight loads
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * NRW * NCW, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, NCW, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, NCW, 0);
    }
  }
}

// per-branch kernel arguments are selected with scalar compares (a dynamically indexed kernarg array would be copied
// to scratch memory)
#define RB_SEL(br_, f) ((br_) == 0 ? a.p[0].f : ((br_) == 1 ? a.p[1].f : a.p[2].f))

template <int C, int NR2, bool MERGE>
__global__ __launch_bounds__(512, 2) void resblock_fused_kernel(const RBArgs a) {
  using G = RBGeom<C, NR2>;
  constexpr int LDX = G::LDX;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS + 4];      // + meta: {zero rows, next tile, its branch, draw generation}
  float* const win = lds;                              // [WR_MAX][LDX] leaky_relu(x) window
  float* const xt = lds + G::WR_MAX * LDX;             // [XT_ROWS][LDX] leaky_relu(c1 + b1); then c2's accumulators
  int* const meta = reinterpret_cast<int*>(lds + G::LDS_FLOATS);   // [0]: zero rows of the staged tile's xt (stream start); [1]: next tile index; [2]: its branch (-1: none)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work queue: a.tiles lists every tile {branch, slot index, first output row, -}, most expensive first.  Block b starts
  // with tile b (the launch has at most one block per CU and no more blocks than tiles); further tiles are drawn from
  // an agent-scope counter, so a block that was dispatched late (a CU held by another stream's kernel) simply takes
  // fewer tiles instead of holding the launch back.  One helper lane draws, the block learns the tile through LDS.
  typedef const int __attribute__((address_space(1)))* gci;
  auto tile_word = [&](int idx, int w) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*(gci)(a.tiles + (long long)idx * 4 + w)); };
  const int ntiles = a.ntiles;
  const float slope = a.slope;
  const int T = a.T;
  constexpr bool merge = MERGE;          // a template parameter: the separate-branch build carries none of the merge state
  const int np = a.nprob;
  const int first_tile = merge ? (int)blockIdx.x * np : (int)blockIdx.x;     // merge: a block owns whole (slot, row tile) groups

  if (wave >= 4) {
    // ============================================================ helper waves: window loader + output writer
    const int ht = tid - 256;
#ifndef RB_NOPRIO
    __builtin_amdgcn_s_setprio(3);   // few instructions, all latency: they must not queue behind the partner wave's MFMA stream
#endif
    // raw barriers behind an LDS-only wait: the output stores of write_out stay in flight across them
    auto hbar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto slot_of = [&](int i) __attribute__((always_inline)) { return a.slots ? __builtin_amdgcn_readfirstlane(*(gci)(a.slots + i)) : i; };
    auto pos_of = [&](int slot) __attribute__((always_inline)) { return a.pos ? __builtin_amdgcn_readfirstlane(*(gci)(a.pos + slot)) : 0; };
    auto load_window = [&](const int p, const int i, const int t0, const int slot, const int pos) __attribute__((always_inline)) {
      const int k = RB_SEL(p, k), d = RB_SEL(p, dil);
      const int wr = G::XT_ROWS + (k - 1) * d;                    // rows c1 reads
      const int tw0 = t0 - (k - 1) - (k - 1) * d;                 // time index (this step's row 0 = 0) of window row 0
      const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate;    // the branches' inputs share their geometry
      const float* xb = RB_SEL(p, x.base) + (long long)(xmode == 0 ? slot : i) * RB_SEL(p, x.slot_stride);
      const unsigned rbase = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)(RB_SEL(p, x.off) + tw0);
      const unsigned rmask = xmode == 0 ? (unsigned)RB_SEL(p, x.lmask) : 0xffffffffu;
      constexpr int NIT = (G::WR_MAX * G::C4 + 255) / 256;
      float4 v[NIT];
      const int total = wr * G::C4;
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int idx = ht + 256 * u;
        const int w = idx / G::C4, c4 = idx - w * G::C4;
        const unsigned row = (rbase + (unsigned)w) & rmask;
        v[u] = idx < total ? gload4(xb + (long long)row * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int idx = ht + 256 * u;
        const int w = idx / G::C4, c4 = idx - w * G::C4;
        float4 q = v[u];
        q.x = q.x > 0.f ? q.x : q.x * slope; q.y = q.y > 0.f ? q.y : q.y * slope;
        q.z = q.z > 0.f ? q.z : q.z * slope; q.w = q.w > 0.f ? q.w : q.w * slope;
        if (idx < total) *reinterpret_cast<float4*>(win + w * LDX + c4 * 4) = q;
      }
      // xt row m is time t0 - (k-1) + m; rows before the start of the stream (absolute time < 0) are c2's zero left
      // padding: m < zrows
      const long long abs0 = (xmode == 0 ? (long long)pos * xrate : 0ll) + t0 - (k - 1);
      if (ht == 0) meta[0] = abs0 >= 0 ? 0 : (abs0 < -(long long)G::XT_ROWS ? G::XT_ROWS : (int)-abs0);
    };
    // Output of a tile in two halves so that the matrix waves never wait for it: out_fetch() right after B2 moves c2's
    // accumulators from LDS to registers (B3 then needs nothing but those LDS reads); out_store() - during the next tile's
    // c2, after its window loads have been issued - reads the residual rows in two batches (fewer live registers: the
    // kernel's VGPR count is what decides whether another stream's blocks fit on the CU beside this one), adds bias +
    // residual and stores.
    constexpr int NOUT = (G::RO * G::C4) / 256;
    static_assert((G::RO * G::C4) % 256 == 0 && 256 % G::C4 == 0, "output tile / helper threads");
    constexpr int NOB = (NOUT + 1) / 2;                            // residual rows per batch
    const int oc4 = ht % G::C4;                                    // a thread keeps its channel quad
    float4 oacc[NOUT], osum[MERGE ? NOUT : 1];                    // osum: running sum of the branch outputs (merge)
    const float* oxb = nullptr; const float* ob2p = nullptr;
    float* oyb = nullptr;
    unsigned oyr0 = 0, oym = 0, oxr0 = 0, oxm = 0;
    int ot0 = 0, op = 0, oyC = C;
#pragma unroll
    for (int u = 0; u < (MERGE ? NOUT : 1); ++u) osum[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto out_fetch = [&](const int p, const int i, const int t0, const int slot, const int pos) __attribute__((always_inline)) {
      const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate, ymode = a.p[0].y.mode, yrate = a.p[0].y.rate;
      oxb = RB_SEL(p, x.base) + (long long)(xmode == 0 ? slot : i) * RB_SEL(p, x.slot_stride);
      oxr0 = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)(RB_SEL(p, x.off) + t0);
      oxm = xmode == 0 ? (unsigned)RB_SEL(p, x.lmask) : 0xffffffffu;
      if (!merge) {
        oyb = RB_SEL(p, y.base) + (long long)(ymode == 0 ? slot : i) * RB_SEL(p, y.slot_stride);
        oyr0 = (ymode == 0 ? (unsigned)pos * (unsigned)yrate : 0u) + (unsigned)(RB_SEL(p, y.off) + t0);
        oym = ymode == 0 ? (unsigned)RB_SEL(p, y.lmask) : 0xffffffffu;
      } else {      // the group's one output: the branch mean
        const int mmode = a.ymean.mode;
        oyb = a.ymean.base + (long long)(mmode == 0 ? slot : i) * a.ymean.slot_stride;
        oyr0 = (mmode == 0 ? (unsigned)pos * (unsigned)a.ymean.rate : 0u) + (unsigned)(a.ymean.off + t0);
        oym = mmode == 0 ? (unsigned)a.ymean.lmask : 0xffffffffu;
        oyC = a.ymean.C;
      }
      ot0 = t0; op = p;
      ob2p = RB_SEL(p, b2) + oc4 * 4;
#pragma unroll
      for (int u = 0; u < NOUT; ++u) oacc[u] = *reinterpret_cast<const float4*>(xt + ((ht + 256 * u) / G::C4) * LDX + oc4 * 4);
    };
    auto out_store = [&]() __attribute__((always_inline)) {
      const float4 ob2 = gload4(ob2p);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 ores[NOB];
#pragma unroll
        for (int q = 0; q < NOB; ++q) {
          const int u = h * NOB + q, r = (ht + 256 * u) / G::C4;
          ores[q] = (u < NOUT && ot0 + r < T) ? gload4(oxb + (long long)((oxr0 + (unsigned)r) & oxm) * C + oc4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NOB; ++q) {
          const int u = h * NOB + q, r = (ht + 256 * u) / G::C4;
          if (u < NOUT && ot0 + r < T) {
            const float4 v = make_float4((oacc[u].x + ob2.x) + ores[q].x, (oacc[u].y + ob2.y) + ores[q].y, (oacc[u].z + ob2.z) + ores[q].z, (oacc[u].w + ob2.w) + ores[q].w);
            if constexpr (!MERGE) gstore4(oyb + (long long)((oyr0 + (unsigned)r) & oym) * C + oc4 * 4, v);
            else {
              // mean_act_kernel's arithmetic on the values the branch launches would have stored: (v0 + v1) + v2, / n, LeakyReLU
              float4 sm = osum[MERGE ? u : 0];
              if (op == 0) sm = v; else { sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w; }
              osum[MERGE ? u : 0] = sm;
              if (op == np - 1) {
                const float dn = (float)np;
                if (np > 1) { sm.x /= dn; sm.y /= dn; sm.z /= dn; sm.w /= dn; }
                sm.x = sm.x > 0.f ? sm.x : sm.x * slope; sm.y = sm.y > 0.f ? sm.y : sm.y * slope;
                sm.z = sm.z > 0.f ? sm.z : sm.z * slope; sm.w = sm.w > 0.f ? sm.w : sm.w * slope;
                gstore4(oyb + (long long)((oyr0 + (unsigned)r) & oym) * oyC + oc4 * 4, sm);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int p = tile_word(first_tile, 0), i = tile_word(first_tile, 1), t0 = tile_word(first_tile, 2);
    int slot = slot_of(i), pos = pos_of(slot);
    int cur = first_tile;                                // index of the tile in flight (merge: its successor is cur + 1 inside a group)
    if (ht == 0) meta[3] = 0;
#if !(RB_ABLATE & 1)
    load_window(p, i, t0, slot, pos);
#endif
    hbar();                                              // B0: first window staged
    bool pending = false;                                // a fetched output tile waits for its stores
    int gen = 1;                                         // generation of the published draw (meta[3])
    for (;;) {
      hbar();                                            // B3: the previous tile's accumulators are in registers (out_fetch)
      hbar();                                            // B1: xt complete, window free
      // Draw the next tile now - half way through this one, not at its start: blocks on short tiles then draw before
      // blocks on long ones, which keeps the longest-first list scheduling balanced (drawn at the start, the blocks
      // still busy with an 11-tap tile would take the last 3-tap tiles from those that finish early).  Every helper
      // wave's lane 0 would be one draw too many: wave 4 draws and publishes {index, branch} through LDS; the other
      // helper waves poll the generation word (s_barrier is block wide, there is no helper-only rendezvous); the
      // matrix waves read it after B4.
      int nv = 0;
      if (wave == 4 && lane == 0) {
        if (merge && p != np - 1) nv = cur + 1;          // next branch of this group: no draw
        else {
          nv = (int)gridDim.x + __hip_atomic_fetch_add(a.sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (merge) nv *= np;                           // a drawn group's first tile
        }
      }
#if RB_ABLATE & 4
      const unsigned long long h0 = __builtin_amdgcn_s_memtime();
#endif
#if !(RB_ABLATE & 2)
      if (pending) out_store();                          // (its loads were issued a whole c1 phase ago); the draw returns meanwhile
#endif
#if RB_ABLATE & 4
      const unsigned long long h1 = __builtin_amdgcn_s_memtime();
#endif
      int nidx, pn;
      if (wave == 4) {
        int pv = -1;
        if (lane == 0) pv = nv < ntiles ? *(gci)(a.tiles + (long long)nv * 4) : -1;
        nidx = __builtin_amdgcn_readfirstlane(nv); pn = __builtin_amdgcn_readfirstlane(pv);
        if (lane == 0) {
          meta[1] = nidx; meta[2] = pn;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(&meta[3], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        while (__hip_atomic_load(&meta[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gen) __builtin_amdgcn_s_sleep(8);
        nidx = __builtin_amdgcn_readfirstlane(meta[1]); pn = __builtin_amdgcn_readfirstlane(meta[2]);
      }
      ++gen;
      int in = 0, t0n = 0, slotn = 0, posn = 0;
      if (pn >= 0) { in = tile_word(nidx, 1); t0n = tile_word(nidx, 2); slotn = slot_of(in); posn = pos_of(slotn); }
#if !(RB_ABLATE & 1)
      if (pn >= 0) load_window(pn, in, t0n, slotn, posn);
#endif
#if RB_ABLATE & 4
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (a.dbg && blockIdx.x == 0 && ht == 0) { unsigned long long* q = a.dbg + 256 * 4 + 4; q[0] = h1 - h0; q[1] = __builtin_amdgcn_s_memtime() - h1; }
#endif
      hbar();                                            // B4
      hbar();                                            // B2: c2 accumulators in LDS, next window staged
#if !(RB_ABLATE & 2)
      out_fetch(p, i, t0, slot, pos);
      pending = true;
#endif
      if (pn < 0) break;
      p = pn; i = in; t0 = t0n; slot = slotn; pos = posn; cur = nidx;
    }
#if !(RB_ABLATE & 2)
    if (pending) out_store();
#endif
    // the last block to leave re-arms the queue for the next launch
    if (ht == 0) {
      const int d = __hip_atomic_fetch_add(a.sched + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == (int)gridDim.x - 1) {
        __hip_atomic_store(a.sched, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sched + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  // ============================================================== matrix waves
#ifdef RB_MATRIX_PRIO
  __builtin_amdgcn_s_setprio(RB_MATRIX_PRIO);      // developer build: matrix waves above the default priority of co-resident kernels
#endif
  // Barriers are raw s_barrier behind an LDS-only wait: global loads (next phase's weights, biases) stay in flight
  // across them.
  auto bar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int wc = wave % (4 / G::RSPLIT);                 // column strip
  const int wr = wave / (4 / G::RSPLIT);                 // row share (C = 32 only)
  const int ct0 = wc * G::NCW;
  const int lr = lane & 15, lg = lane >> 4;
#if RB_ABLATE & 4
  unsigned long long st_gemm = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime(), st_bar = 0, st_b3 = 0, st_b1 = 0, st_b4 = 0, st_b2 = 0;
#define RB_T() __builtin_amdgcn_s_memtime()
#endif
  float4 bw[G::RING][G::NCW];
  int p = tile_word(first_tile, 0);
  {
    const long long cs = (long long)(RB_SEL(p, k) + 1) * G::KQ * 256;
    rb_prefetch_w<G::NCW, G::RING>(bw, RB_SEL(p, w1) + (long long)ct0 * cs + lane * 4, cs);
  }
  bar();                                                 // B0: first window staged
  while (p >= 0) {
    const int k = RB_SEL(p, k), d = RB_SEL(p, dil);
    const float* const w1 = RB_SEL(p, w1);
    const float* const w2 = RB_SEL(p, w2);
    const float* const b1 = RB_SEL(p, b1);
    const long long ct_stride = (long long)(k + 1) * G::KQ * 256;       // floats per column tile (k taps + one zero tap)
    const int zrows = __builtin_amdgcn_readfirstlane(meta[0]);
    float b1v[G::NCW];            // c1's bias: loaded now, needed behind the first GEMM
#pragma unroll
    for (int c = 0; c < G::NCW; ++c) b1v[c] = gload1(b1 + (ct0 + c) * 16 + lr);
    int pn;
    // ---------------- c1 over the halo-extended rows
    {
      f32x4 acc[G::NRW1][G::NCW];
#pragma unroll
      for (int r = 0; r < G::NRW1; ++r)
#pragma unroll
        for (int c = 0; c < G::NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int rt0 = wr * G::NRW1;
#if RB_ABLATE & 4
      unsigned long long s0 = RB_T();
#endif
      rb_gemm<G::NRW1, G::NCW, LDX, G::KQ, G::RING>(win, rt0 * 16, d, k, w1 + (long long)ct0 * ct_stride + lane * 4, ct_stride,
                                           w2 + (long long)ct0 * ct_stride + lane * 4, ct_stride, acc, bw, lane);
#if RB_ABLATE & 4
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
      unsigned long long s1 = RB_T(); st_gemm += s1 - s0;
#endif
#if RB_ABLATE & 4
      { unsigned long long q0 = RB_T(); bar(); unsigned long long q1 = RB_T(); st_b3 += q1 - q0; st_bar += q1 - s1; }
#else
      bar();                                             // B3: the helpers are done with the previous tile's accumulators
#endif
#pragma unroll
      for (int c = 0; c < G::NCW; ++c) {
        const int col = (ct0 + c) * 16 + lr;
#pragma unroll
        for (int r = 0; r < G::NRW1; ++r) {
          if (rt0 + r < G::NR1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int m = (rt0 + r) * 16 + 4 * lg + e;
              float v = acc[r][c][e] + b1v[c];
              v = v > 0.f ? v : v * slope;
              xt[m * LDX + col] = m < zrows ? 0.f : v;
            }
          }
        }
      }
    }
#if RB_ABLATE & 4
    { unsigned long long q0 = RB_T(); bar(); unsigned long long q1 = RB_T(); st_b1 += q1 - q0; st_bar += q1 - q0; }
#else
    bar();                                               // B1: xt complete, window free
#endif
    // ---------------- c2
    {
      f32x4 acc[G::NRW2][G::NCW];
#pragma unroll
      for (int r = 0; r < G::NRW2; ++r)
#pragma unroll
        for (int c = 0; c < G::NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int rt0 = wr * G::NRW2;
#if RB_ABLATE & 4
      unsigned long long s0 = RB_T();
#endif
      // (behind the last tap the weight registers are refilled with this phase's tap 0 again: harmless, the next tile
      // is only known after B4)
      rb_gemm<G::NRW2, G::NCW, LDX, G::KQ, G::RING>(xt, rt0 * 16, 1, k, w2 + (long long)ct0 * ct_stride + lane * 4, ct_stride,
                                           w2 + (long long)ct0 * ct_stride + lane * 4, ct_stride, acc, bw, lane);
#if RB_ABLATE & 4
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
      unsigned long long s1 = RB_T(); st_gemm += s1 - s0;
#endif
#if RB_ABLATE & 4
      { unsigned long long q0 = RB_T(); bar(); unsigned long long q1 = RB_T(); st_b4 += q1 - q0; st_bar += q1 - s1; }
#else
      bar();                                             // B4: every matrix wave is done reading xt
#endif
      // the next tile (drawn by the helpers after B1): its first c1 weight fragments fly across the accumulator hand-off
      pn = __builtin_amdgcn_readfirstlane(meta[2]);
      if (pn >= 0) {
        const long long csn = (long long)(RB_SEL(pn, k) + 1) * G::KQ * 256;
        rb_prefetch_w<G::NCW, G::RING>(bw, RB_SEL(pn, w1) + (long long)ct0 * csn + lane * 4, csn);
      }
#pragma unroll
      for (int c = 0; c < G::NCW; ++c) {
        const int col = (ct0 + c) * 16 + lr;
#pragma unroll
        for (int r = 0; r < G::NRW2; ++r) {
          if (rt0 + r < NR2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[((rt0 + r) * 16 + 4 * lg + e) * LDX + col] = acc[r][c][e];
          }
        }
      }
    }
#if RB_ABLATE & 4
    { unsigned long long s2 = RB_T(); bar(); unsigned long long s3 = RB_T(); st_bar += s3 - s2; st_b2 += s3 - s2; }
#else
    bar();                                               // B2: accumulators in LDS for the helpers, next window staged
#endif
    p = pn;
  }
#if RB_ABLATE & 4
  if (a.dbg && tid == 0) {
    unsigned long long* o = a.dbg + blockIdx.x * 4;
    if (blockIdx.x == 0) { unsigned long long* q = a.dbg + 256 * 4; q[0] = st_b3; q[1] = st_b1; q[2] = st_b4; q[3] = st_b2; }
    o[0] = st_gemm; o[1] = __builtin_amdgcn_s_memtime() - st_t0; o[2] = __builtin_amdgcn_s_memrealtime() - st_r0; o[3] = st_bar;
  }
#endif
}

// ------------------------------------------------------------------------------------------------ host side

struct RBCand { int C, NR2; };
static const RBCand kCands[] = {{32, 20}, {32, 4}, {64, 10}, {64, 8}, {64, 4}, {128, 5}, {128, 3}};

bool resblock_fused_supported(int C, int kmax, int span_max) {
  if (C != 32 && C != 64 && C != 128) return false;
  return kmax <= 16 && span_max <= 50;
}

// rows per tile: the candidate with the smallest estimated makespan (tiles are balanced over the CUs by cost k)
int resblock_fused_rows(int C, int T, int n, int ksum, int kmax, int num_cu) {
  int best = 0; double best_cost = 1e30;
  for (const RBCand& c : kCands) {
    if (c.C != C) continue;
    const int ro = 16 * c.NR2;
    const long long tps = (T + ro - 1) / ro;
    const double per_cu = std::max((double)n * tps * ksum / std::max(1, num_cu), (double)kmax);
    const double cost = per_cu * (2 * c.NR2 + 1) + 0.15 * per_cu / kmax * 40;      // MFMA row tiles + a per-tile constant
    if (cost < best_cost) { best_cost = cost; best = ro; }
  }
  return best;
}

// Tile list of a launch shape: {branch, slot index, first row, 0} per tile, most expensive branch first (the blocks draw
// from it in order: longest-processing-time-first list scheduling); cached per shape in device memory (a handful of
// shapes per model and device, never freed).
const int* resblock_tiles(const RBArgs& a, int ro, int* total_out) {
  struct Key { int v[10]; bool operator<(const Key& o) const { return memcmp(v, o.v, sizeof(v)) < 0; } };
  static std::map<Key, std::pair<const int*, int>> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Key key = {{a.nprob, a.n, a.tiles_per_slot, a.p[0].k, a.nprob > 1 ? a.p[1].k : 0, a.nprob > 2 ? a.p[2].k : 0, dev, ro, a.merge, 0}};
  auto it = cache.find(key);
  if (it != cache.end()) { *total_out = it->second.second; return it->second.first; }
  const int per_prob = a.n * a.tiles_per_slot, total = a.nprob * per_prob;
  std::vector<int> order(total);
  for (int i = 0; i < total; ++i) order[i] = i;
  if (!a.merge) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a.p[x / per_prob].k > a.p[y / per_prob].k; });
  else for (int e = 0; e < total; ++e) order[e] = (e % a.nprob) * per_prob + e / a.nprob;      // (slot, row tile)-major, branches adjacent
  std::vector<int> flat((size_t)(total + 1) * 4, -1);
  for (int e = 0; e < total; ++e) {
    const int id = order[e], p = id / per_prob, rem = id - p * per_prob, i = rem / a.tiles_per_slot;
    int* d = flat.data() + (size_t)e * 4;
    d[0] = p; d[1] = i; d[2] = (rem - i * a.tiles_per_slot) * ro; d[3] = 0;
  }
  int* d = nullptr;
  if (hipMalloc(&d, flat.size() * sizeof(int)) != hipSuccess) { *total_out = 0; return nullptr; }
  (void)hipMemcpy(d, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
  cache[key] = {d, total};
  *total_out = total;
  return d;
}

template <int C, int NR2, bool MERGE = false>
static bool launch_rb(const RBArgs& ain, int num_cu, hipStream_t st) {
  RBArgs a = ain;
  const int ro = RBGeom<C, NR2>::RO;
  a.tiles_per_slot = (a.T + ro - 1) / ro;
  const int total = a.nprob * a.n * a.tiles_per_slot;
  if (total <= 0) return true;
  if (!a.sched) return false;
  const int grid = std::min(a.merge ? total / a.nprob : total, num_cu);
  a.tiles = resblock_tiles(a, ro, &a.ntiles);
  if (!a.tiles) return false;
  hipLaunchKernelGGL((resblock_fused_kernel<C, NR2, MERGE>), dim3(grid), dim3(512), 0, st, a);
  return true;
}

bool launch_resblock_fused(const RBArgs& a, int C, int rows, int num_cu, hipStream_t st) {
  const int nr2 = rows / 16;
  if (a.merge) {     // instantiated where a stage can have a group per CU
    if (C == 32 && nr2 == 20) return launch_rb<32, 20, true>(a, num_cu, st);
    if (C == 64 && nr2 == 10) return launch_rb<64, 10, true>(a, num_cu, st);
    return false;
  }
  if (C == 32 && nr2 == 20) return launch_rb<32, 20>(a, num_cu, st);
  if (C == 32 && nr2 == 4) return launch_rb<32, 4>(a, num_cu, st);
  if (C == 64 && nr2 == 10) return launch_rb<64, 10>(a, num_cu, st);
  if (C == 64 && nr2 == 8) return launch_rb<64, 8>(a, num_cu, st);
  if (C == 64 && nr2 == 4) return launch_rb<64, 4>(a, num_cu, st);
  if (C == 128 && nr2 == 5) return launch_rb<128, 5>(a, num_cu, st);
  if (C == 128 && nr2 == 3) return launch_rb<128, 3>(a, num_cu, st);
  return false;
}

bool resblock_fused_can_merge(int C, int rows) { return (C == 32 && rows == 320) || (C == 64 && rows == 160); }

const char* resblock_fused_name(int C, int rows, bool merge) {
  static thread_local char buf[72];
  snprintf(buf, sizeof(buf), "cnk::resblock_fused_kernel<%d, %d, %s>", C, rows / 16, merge ? "true" : "false");
  return buf;
}

}  // namespace cnk
