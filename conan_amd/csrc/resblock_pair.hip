// resblock_pair: one ResBlock1 unit of the causal HiFi-GAN MRF (hifigan_causal.py:230-238)
//     xt = c1(leaky_relu(x));  y = c2(leaky_relu(xt)) + x
// for the WIDE first stage (C = 256), where a stream contributes only a few rows per step (32 at 8x upsampling of a
// 4-frame chunk) and a whole-width tile per workgroup (resblock_fused.hip) cannot be balanced over the chip: 64 streams x 3
// branches = 192 tiles of cost 11 : 7 : 3 on 256 CUs.
//
// Here TWO workgroups (a pair: blockIdx 2p, 2p+1) share a tile = (branch, stream), the way a tensor-parallel MLP is split:
//   * c1 is split over its OUTPUT columns: member h computes xt[:, 128h .. 128h+127] from the whole input window
//     (32 + (k-1)*dil rows x 256 channels in LDS, LeakyReLU applied on the way in) and keeps its half of xt in LDS;
//   * c2 is split over its INPUT channels: member h multiplies its half of xt with the rows 128h .. 128h+127 of c2's weights
//     and gets a partial sum for ALL 256 output columns;
//   * the only exchange is at the END of the tile: each member hands the partner the partial sums of the partner's 128
//     columns (16 KB) and finishes its own 128 columns: y = (P0 + P1) + bias + x, summed in member order, so both the result
//     and its bits are independent of timing.  The exchange is done by the helper waves while the matrix waves already
//     run the next tile: it is never on the MFMA critical path.
// No halo recompute: the k-1 rows of xt that c2 needs from before the tile are the last rows of the PREVIOUS step's xt,
// which every tile appends to a small per-unit history ring (activated values; zeros after a reset = c2's zero padding).
// A tile is all rows a stream gets in a step (T = 16 * NR), so there is no halo between tiles of one launch.
// Wave roles, weight streaming (fragment-major, private per wave, no barrier in the K loops), barriers and the tile queue
// follow resblock_fused.hip; per 16-deep K group a wave feeds 4 x NR x NCW MFMAs from NR ds_read_b128 + NCW 16-byte
// global loads (NCW = 2 in c1, 4 in c2).  Cross-workgroup data (partial sums, flags, the tile mailbox) travel as
// agent-scope write-through stores / sc1 loads (the per-XCD L2s are not coherent; pair members sit on different XCDs, which
// also halves every XCD's weight working set: even XCDs only ever read member 0's weight halves).
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace cnk {

typedef float rp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));
typedef const rp_f32x4 __attribute__((address_space(1)))* rp_gcf4;
typedef rp_f32x4 __attribute__((address_space(1)))* rp_gf4;
typedef const float __attribute__((address_space(1)))* rp_gcf1;
typedef const int __attribute__((address_space(1)))* rp_gci;
__device__ __forceinline__ float4 rp_gload4(const float* p) { const rp_f32x4 v = *(rp_gcf4)(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float rp_gload1(const float* p) { return *(rp_gcf1)(p); }
__device__ __forceinline__ void rp_gstore4(float* p, const float4 v) { *(rp_gf4)(p) = (rp_f32x4){v.x, v.y, v.z, v.w}; }

// 16-byte agent-scope accesses (buffer_load / buffer_store ... sc1): L2 write-through / L1 bypass, visible across XCDs
constexpr int RP_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rp_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void rp_xstore4(__amdgpu_buffer_rsrc_t r, int byte_off, const float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128((rp_u32x4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, byte_off, 0, RP_SC1);
}
__device__ __forceinline__ float4 rp_xload4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  const rp_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, RP_SC1);
  return make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
}

template <int NR>
struct RPGeom {
  static constexpr int C = 256, CH = 128;             // channels, channels per member
  static constexpr int RO = 16 * NR;                  // rows per tile = rows per stream and step
  static constexpr int MAXSPAN = 50;                  // (k-1)*dil of c1 (k = 11, dil = 5)
  static constexpr int MAXK = 16;
  static constexpr int WR_MAX = RO + MAXSPAN;         // window rows
  static constexpr int LDW = C + 4;                   // window row stride (floats): 65 16-byte slots = 1 mod 16, conflict-free b128 reads
  static constexpr int XT_ROWS = RO + MAXK;           // k-1 history rows + the tile's rows
  static constexpr int LDT = CH + 4;                  // xt row stride (33 slots = 1 mod 16)
  static constexpr int R2 = (RO * LDW > XT_ROWS * LDT) ? RO * LDW : XT_ROWS * LDT;   // xt, then c2's accumulators [RO][LDW]
  static constexpr int LDS_FLOATS = WR_MAX * LDW + R2;
  static constexpr int C4 = C / 4;
  static_assert(LDS_FLOATS * 4 + 64 <= 160 * 1024, "LDS budget");
};

// First D = 16 / NCW groups of a phase into the flat 16-slot weight ring: fragment (group g, column tile c) -> slot (g * NCW + c) % 16
template <int NCW>
__device__ __forceinline__ void rp_prefetch(float4 (&bw)[16], const float* __restrict__ wl, const long long cts) {
#pragma unroll
  for (int g = 0; g < 16 / NCW; ++g) {
#pragma unroll
    for (int c = 0; c < NCW; ++c) bw[g * NCW + c] = rp_gload4(wl + c * cts + g * 256);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One GEMM phase of a matrix wave: acc[r][c] += sum over (tap j < k, K group q < KQ) of A(rows of row tile r shifted by
// j * tap_stride, channels 16 q ..) x W(j, q, column tile c).  Fragment (j, q, c) is at wl + c * cts + (j * KT + q) * 256 (KT
// groups per tap in memory, of which this wave consumes KQ).  The weight ring is 16 fragments: fragment number
// f = (j * KQ + q) * NCW + c lives in slot f % 16 (KQ * NCW = 32, so the slot depends on (q, c) only); a consumed group is
// refilled with the group 16 / NCW further on, and behind the last tap with the first fragments of the NEXT phase
// (wn, ctsn; NCWN column tiles per group), whose fragment f' takes slot f' % 16 in turn.
template <int NRW, int NCW, int LDX, int KQ, int KT, int NCWN>
__device__ __forceinline__ void rp_gemm(const float* __restrict__ src, const int tap_stride, const int k, const float* __restrict__ wl, const long long cts,
                                        const float* __restrict__ wn, const long long ctsn, rp_f32x4 (&acc)[NRW][NCW], float4 (&bw)[16], const int lane) {
  constexpr int D = 16 / NCW;
  static_assert(KQ * NCW == 32 && 2 * D == KQ, "ring geometry");
  const float* abase = src + (lane & 15) * LDX + 4 * (lane >> 4);
  float4 af[NRW];
#pragma unroll
  for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const float4*>(abase + r * 16 * LDX);
  const int tstep = tap_stride * LDX;
  for (int j = 0; j < k; ++j) {
    const float* arow = abase + j * tstep;
    const bool last = j + 1 == k;
    const float* wj = wl + (long long)j * KT * 256;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float* anext = (q + 1 < KQ) ? arow + (q + 1) * 16 : arow + tstep;      // (past the last tap: rows of the neighbouring region, unused)
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].x, bw[(q * NCW + c) % 16].x, acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].y, bw[(q * NCW + c) % 16].y, acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r)
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].z, bw[(q * NCW + c) % 16].z, acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NRW; ++r) {
#pragma unroll
        for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].w, bw[(q * NCW + c) % 16].w, acc[r][c], 0, 0, 0);
        af[r] = *reinterpret_cast<const float4*>(anext + r * 16 * LDX);
      }
      if (q + D < KQ) {
#pragma unroll
        for (int c = 0; c < NCW; ++c) bw[(q * NCW + c) % 16] = rp_gload4(wj + c * cts + (q + D) * 256);
      } else {
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
          constexpr int dummy = 0; (void)dummy;
          const int fn = (q + D - KQ) * NCW + c;                       // fragment number in the next phase
          const float* pm = wj + (long long)KT * 256 + c * cts + (q + D - KQ) * 256;    // next tap of this phase
          const float* pn = wn + (fn % NCWN) * ctsn + (fn / NCWN) * 256;                // tap 0 of the next phase
          bw[(q * NCW + c) % 16] = rp_gload4(last ? pn : pm);
        }
      }
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

#define RP_SEL(br_, f) ((br_) == 0 ? a.p[0].f : ((br_) == 1 ? a.p[1].f : a.p[2].f))

template <int NR>
__global__ __launch_bounds__(512, 2) void resblock_pair_kernel(const RPArgs a) {
  using G = RPGeom<NR>;
  constexpr int LDW = G::LDW, LDT = G::LDT, C = G::C, CH = G::CH, RO = G::RO;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS + 8];
  float* const win = lds;                              // [WR_MAX][LDW] leaky_relu(x) window
  float* const xt = lds + G::WR_MAX * LDW;             // [k-1 + RO][LDT] leaky_relu(c1 + b1), own half; then c2's partial sums [RO][LDW]
  int* const meta = reinterpret_cast<int*>(lds + G::LDS_FLOATS);   // [1]: next tile index, [2]: its branch (-1: none), [3]: generation
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = (int)blockIdx.x >> 1, h = (int)blockIdx.x & 1;
  const int npairs = (int)gridDim.x >> 1;
  auto tile_word = [&](int idx, int w) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(*(rp_gci)(a.tiles + (long long)idx * 4 + w)); };
  const int ntiles = a.ntiles;
  const float slope = a.slope;
  const int T = a.T;                                   // rows of the step (<= RO; rows T .. RO-1 of a tile are computed and dropped)
  // tiles this pair has ever finished (both members keep the same count): the sequence numbers of flags and mailbox
  // entries go on from there, so nothing has to be cleared between launches
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(*(rp_gci)(reinterpret_cast<const int*>(a.xcount + pair * 2 + h)));

  if (wave >= 4) {
    // ============================================================ helper waves: window loader, exchange, output writer
    const int ht = tid - 256;
    const int hw = wave - 4;
    __builtin_amdgcn_s_setprio(3);
    auto hbar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto slot_of = [&](int i) __attribute__((always_inline)) { return a.slots ? __builtin_amdgcn_readfirstlane(*(rp_gci)(a.slots + i)) : i; };
    auto pos_of = [&](int slot) __attribute__((always_inline)) { return a.pos ? __builtin_amdgcn_readfirstlane(*(rp_gci)(a.pos + slot)) : 0; };
    const __amdgpu_buffer_rsrc_t xb_r = rp_rsrc(a.xb + (long long)pair * (4 * RO * CH));       // [parity][dest member][RO][CH]
    const __amdgpu_buffer_rsrc_t fl_r = rp_rsrc(a.xflag + (long long)pair * 8);                  // [member][helper wave]
    const __amdgpu_buffer_rsrc_t mb_r = rp_rsrc(a.mbox + (long long)pair * 4);                   // tile mailbox, 4 entries
    // k-1 history rows of xt for the tile being staged (own half): 2 x 16 bytes per thread cover 16 rows x 128 channels
    float4 hx[2];
    int hk = 0;                                          // k of the staged tile
    auto row_ptr = [&](const TRef& r, int i, int slot, int pos, int t) __attribute__((always_inline)) {
      return r.mode == 0 ? r.base + (long long)slot * r.slot_stride + (long long)(((unsigned)pos * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask) * r.C
                         : r.base + (long long)i * r.slot_stride + (long long)(r.off + t) * r.C;
    };
    auto load_window = [&](const int p, const int i, const int slot, const int pos) __attribute__((always_inline)) {
      const int k = RP_SEL(p, k), d = RP_SEL(p, dil);
      const int wr = T + (k - 1) * d;
      const int tw0 = -(k - 1) * d;
      const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate;
      const float* xb = RP_SEL(p, x.base) + (long long)(xmode == 0 ? slot : i) * RP_SEL(p, x.slot_stride);
      const unsigned rbase = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)(RP_SEL(p, x.off) + tw0);
      const unsigned rmask = xmode == 0 ? (unsigned)RP_SEL(p, x.lmask) : 0xffffffffu;
      // history rows of xt first: they stay in registers until the matrix waves are done with the previous tile's xt
      {
        const TRef xh = p == 0 ? a.p[0].xh : (p == 1 ? a.p[1].xh : a.p[2].xh);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = ht + 256 * u, m = e >> 5, c4 = e & 31;       // history row m = time m - (k-1)
          hx[u] = m < k - 1 ? rp_gload4(row_ptr(xh, i, slot, pos, m - (k - 1)) + h * CH + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        hk = k;
      }
      constexpr int NIT = (G::WR_MAX * G::C4 + 255) / 256;
      constexpr int NB = 6;                              // batches of 6 x 16 bytes per thread: the helpers' registers must stay below the matrix waves' (the kernel's
      constexpr int NBAT = (NIT + NB - 1) / NB;          // VGPR count decides whether another stream's blocks fit on the CU beside this one)
      const int total = wr * G::C4;
#pragma unroll
      for (int b = 0; b < NBAT; ++b) {
        float4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int idx = ht + 256 * (b * NB + u);
          const int w = idx / G::C4, c4 = idx - w * G::C4;
          const unsigned row = (rbase + (unsigned)w) & rmask;
          v[u] = (b * NB + u < NIT && idx < total) ? rp_gload4(xb + (long long)row * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int idx = ht + 256 * (b * NB + u);
          const int w = idx / G::C4, c4 = idx - w * G::C4;
          float4 q = v[u];
          q.x = q.x > 0.f ? q.x : q.x * slope; q.y = q.y > 0.f ? q.y : q.y * slope;
          q.z = q.z > 0.f ? q.z : q.z * slope; q.w = q.w > 0.f ? q.w : q.w * slope;
          if (b * NB + u < NIT && idx < total) *reinterpret_cast<float4*>(win + w * LDW + c4 * 4) = q;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // history rows registers -> LDS rows [0, k-1) of xt (after B3: the helpers have taken the previous tile's sums out)
    auto put_history = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = ht + 256 * u, m = e >> 5, c4 = e & 31;
        if (m < hk - 1) *reinterpret_cast<float4*>(xt + m * LDT + c4 * 4) = hx[u];
      }
    };
    // the tile's last k-1 rows of xt -> the unit's history ring (between B1 and B4: xt is complete and read-only)
    auto save_history = [&](const int p, const int i, const int slot, const int pos) __attribute__((always_inline)) {
      const int k = RP_SEL(p, k);
      const TRef xh = p == 0 ? a.p[0].xh : (p == 1 ? a.p[1].xh : a.p[2].xh);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = ht + 256 * u, m = e >> 5, c4 = e & 31;         // m-th of the last k-1 rows: time T - (k-1) + m = xt row T + m
        if (m < k - 1) {                                              // (T < k-1: the first of them are history rows again - rewritten as they are)
          const float4 v = *reinterpret_cast<const float4*>(xt + (T + m) * LDT + c4 * 4);
          rp_gstore4(const_cast<float*>(row_ptr(xh, i, slot, pos, T - (k - 1) + m)) + h * CH + c4 * 4, v);
        }
      }
    };
    // Output of a tile: out_fetch() right after B2 takes the pair's partial sums out of LDS - this member's 128 columns into
    // registers, the partner's 128 columns to the exchange buffer (+ flag); out_store() - during the next tile's c2 - waits
    // for the partner's flag, adds the two partial sums in member order, bias and residual, and stores.
    // thread -> (row r_u = ht / 32 + 8 u, 16-byte column group ht % 32 of a half)
    constexpr int NU = RO / 8;
    const int q32 = ht & 31, r0 = ht >> 5;
    float4 oacc[NU];
    const float* oxb = nullptr; const float* ob2p = nullptr; float* oyb = nullptr;
    unsigned oyr0 = 0, oym = 0, oxr0 = 0, oxm = 0, oseq = 0;
    auto out_fetch = [&](const int p, const int i, const int slot, const int pos, const unsigned seq) __attribute__((always_inline)) {
      const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate, ymode = a.p[0].y.mode, yrate = a.p[0].y.rate;
      oxb = RP_SEL(p, x.base) + (long long)(xmode == 0 ? slot : i) * RP_SEL(p, x.slot_stride);
      oxr0 = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)RP_SEL(p, x.off);
      oxm = xmode == 0 ? (unsigned)RP_SEL(p, x.lmask) : 0xffffffffu;
      oyb = RP_SEL(p, y.base) + (long long)(ymode == 0 ? slot : i) * RP_SEL(p, y.slot_stride);
      oyr0 = (ymode == 0 ? (unsigned)pos * (unsigned)yrate : 0u) + (unsigned)RP_SEL(p, y.off);
      oym = ymode == 0 ? (unsigned)RP_SEL(p, y.lmask) : 0xffffffffu;
      ob2p = RP_SEL(p, b2) + h * CH + q32 * 4;
      oseq = seq;
      const int par = (int)(seq & 1u);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int r = r0 + 8 * u;
        oacc[u] = *reinterpret_cast<const float4*>(xt + r * LDW + h * CH + q32 * 4);
        const float4 snd = *reinterpret_cast<const float4*>(xt + r * LDW + (1 - h) * CH + q32 * 4);
        rp_xstore4(xb_r, (((par * 2 + (1 - h)) * RO + r) * CH + q32 * 4) * 4, snd);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's partial sums are out ...
      if (lane == 0 && !(a.fault && h == 1)) __builtin_amdgcn_raw_buffer_store_b32(seq, fl_r, (h * 4 + hw) * 4, 0, RP_SC1);    // ... before its flag (the partner's wave hw reads exactly these rows)
    };
    auto out_store = [&]() __attribute__((always_inline)) {
      const float4 ob2 = rp_gload4(ob2p);
      float4 ores[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) ores[u] = r0 + 8 * u < T ? rp_gload4(oxb + (long long)((oxr0 + (unsigned)(r0 + 8 * u)) & oxm) * C + h * CH + q32 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      // the partner's wave hw has written the rows this wave finishes
      {
        SpinGuard sg;
        while ((int)(__builtin_amdgcn_raw_buffer_load_b32(fl_r, ((1 - h) * 4 + hw) * 4, 0, RP_SC1) - oseq) < 0) {
          __builtin_amdgcn_s_sleep(4);
          if (spin_expired(sg, a.guard, WAIT_PAIR_FLAG)) break;       // (bounded: kernels.h, SpinGuard)
        }
      }
      const int par = (int)(oseq & 1u);
      float4 opart[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) opart[u] = rp_xload4(xb_r, (((par * 2 + h) * RO + (r0 + 8 * u)) * CH + q32 * 4) * 4);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const float4 lo = h == 0 ? oacc[u] : opart[u], hi = h == 0 ? opart[u] : oacc[u];       // member order: the same bits whoever finishes
        const float4 v = make_float4(((lo.x + hi.x) + ob2.x) + ores[u].x, ((lo.y + hi.y) + ob2.y) + ores[u].y,
                                     ((lo.z + hi.z) + ob2.z) + ores[u].z, ((lo.w + hi.w) + ob2.w) + ores[u].w);
        if (r0 + 8 * u < T) rp_gstore4(oyb + (long long)((oyr0 + (unsigned)(r0 + 8 * u)) & oym) * C + h * CH + q32 * 4, v);
      }
    };
    unsigned seq = base;                                  // sequence number of the tile in flight is seq + 1
    int p = tile_word(pair, 0), i = tile_word(pair, 1);
    int slot = slot_of(i), pos = pos_of(slot);
    if (ht == 0) meta[3] = 0;
    load_window(p, i, slot, pos);
    hbar();                                              // B0: first window staged
    bool pending = false;
    int gen = 1;
    for (;;) {
      seq += 1;
      hbar();                                            // B3: the previous tile's sums are in registers / on their way to the partner
      put_history();
      // The pair's next tile: member 0 draws from the launch's queue and posts it to the pair's mailbox; member 1 reads it
      // there.  (The draw is issued here, a whole c1 epilogue before its result is needed.)
      int nv = 0;
      if (h == 0 && wave == 4 && lane == 0) nv = npairs + __hip_atomic_fetch_add(a.sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hbar();                                            // B1: xt complete, window free
      int nidx, pn;
      if (wave == 4) {
        int pv = -1;
        if (lane == 0) {
          if (h == 0) {
            if (nv >= ntiles) nv = -1;
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)(nv + 1) | ((seq & 0x7ffu) << 20), mb_r, (int)(seq & 3u) * 4, 0, RP_SC1);
          } else {
            unsigned w;
            SpinGuard sg;
            bool dead = false;
            while ((((w = __builtin_amdgcn_raw_buffer_load_b32(mb_r, (int)(seq & 3u) * 4, 0, RP_SC1)) >> 20) & 0x7ffu) != (seq & 0x7ffu)) {
              __builtin_amdgcn_s_sleep(4);
              if (spin_expired(sg, a.guard, WAIT_PAIR_MAILBOX)) { dead = true; break; }     // (bounded: kernels.h, SpinGuard)
            }
            nv = dead ? -1 : (int)(w & 0xfffffu) - 1;       // (a member that gave up takes no further tile)
          }
          pv = nv >= 0 ? *(rp_gci)(a.tiles + (long long)nv * 4) : -1;
        }
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
      save_history(p, i, slot, pos);
      int in = 0, slotn = 0, posn = 0;
      if (pn >= 0) { in = tile_word(nidx, 1); slotn = slot_of(in); posn = pos_of(slotn); }
      if (pn >= 0) load_window(pn, in, slotn, posn);
      if (pending) out_store();
      hbar();                                            // B4
      hbar();                                            // B2: c2's partial sums in LDS, next window staged
      out_fetch(p, i, slot, pos, seq);
      pending = true;
      if (pn < 0) break;
      p = pn; i = in; slot = slotn; pos = posn;
    }
    out_store();
    // the pair's tile count for the next launch; the last workgroup to leave re-arms the queue
    if (ht == 0) {
      a.xcount[pair * 2 + h] = seq;
      const int dn = __hip_atomic_fetch_add(a.sched + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (dn == (int)gridDim.x - 1) {
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
  auto bar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int lr = lane & 15, lg = lane >> 4;
  float4 bw[16];
  int p = tile_word(pair, 0);
  {
    const long long cts = (long long)(RP_SEL(p, k) + 1) * 16 * 256;
    rp_prefetch<2>(bw, RP_SEL(p, w1) + (long long)(8 * h + 2 * wave) * cts + lane * 4, cts);
  }
  bar();                                                 // B0
  while (p >= 0) {
    const int k = RP_SEL(p, k), d = RP_SEL(p, dil);
    const long long cts = (long long)(k + 1) * 16 * 256;               // floats per column tile (k taps + one zero tap), 16 K groups per tap
    const float* const w1p = RP_SEL(p, w1) + (long long)(8 * h + 2 * wave) * cts + lane * 4;     // c1: this member's columns 128h .., 2 tiles per wave
    const float* const w2p = RP_SEL(p, w2) + (long long)(4 * wave) * cts + (long long)(8 * h) * 256 + lane * 4;   // c2: all columns (4 tiles per wave), K groups 8h .. 8h+7 of every tap
    const float* const b1 = RP_SEL(p, b1);
    float b1v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) b1v[c] = rp_gload1(b1 + h * CH + (2 * wave + c) * 16 + lr);
    int pn;
    {
      rp_f32x4 acc[NR][2];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[r][c] = (rp_f32x4){0.f, 0.f, 0.f, 0.f};
      rp_gemm<NR, 2, LDW, 16, 16, 4>(win, d, k, w1p, cts, w2p, cts, acc, bw, lane);
      bar();                                             // B3: the helpers have taken the previous tile's sums out of this region
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int col = (2 * wave + c) * 16 + lr;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[r][c][e] + b1v[c];
            v = v > 0.f ? v : v * slope;
            xt[((k - 1) + r * 16 + 4 * lg + e) * LDT + col] = v;
          }
      }
    }
    bar();                                               // B1: xt complete (the helpers wrote its k-1 history rows), window free
    {
      rp_f32x4 acc[NR][4];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = (rp_f32x4){0.f, 0.f, 0.f, 0.f};
      // (behind the last tap the ring is refilled with this phase's tap 0 again: harmless, the next tile is only known after B4)
      rp_gemm<NR, 4, LDT, 8, 16, 4>(xt, 1, k, w2p, cts, w2p, cts, acc, bw, lane);
      bar();                                             // B4: every matrix wave is done reading xt
      pn = __builtin_amdgcn_readfirstlane(meta[2]);
      if (pn >= 0) {
        const long long csn = (long long)(RP_SEL(pn, k) + 1) * 16 * 256;
        rp_prefetch<2>(bw, RP_SEL(pn, w1) + (long long)(8 * h + 2 * wave) * csn + lane * 4, csn);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = (4 * wave + c) * 16 + lr;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) xt[(r * 16 + 4 * lg + e) * LDW + col] = acc[r][c][e];
      }
    }
    bar();                                               // B2: partial sums in LDS for the helpers, next window staged
    p = pn;
  }
}

// ------------------------------------------------------------------------------------------------ host side

bool resblock_pair_supported(int C, int kmax, int span_max, int T) {
  return C == 256 && kmax <= RPGeom<2>::MAXK && span_max <= RPGeom<2>::MAXSPAN && T >= 1 && T <= 32;
}

size_t resblock_pair_xb_floats(int num_cu) { return (size_t)(num_cu / 2) * 4 * 32 * 128; }

// tile list: {branch, slot index, 0, 0}, most expensive branch first; cached per shape in device memory
static const int* rp_tiles(const RPArgs& a, int* total_out) {
  struct Key { int v[6]; bool operator<(const Key& o) const { return memcmp(v, o.v, sizeof(v)) < 0; } };
  static std::map<Key, std::pair<const int*, int>> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Key key = {{a.nprob, a.n, a.p[0].k, a.nprob > 1 ? a.p[1].k : 0, a.nprob > 2 ? a.p[2].k : 0, dev}};
  auto it = cache.find(key);
  if (it != cache.end()) { *total_out = it->second.second; return it->second.first; }
  const int total = a.nprob * a.n;
  std::vector<int> order(total);
  for (int i = 0; i < total; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a.p[x / a.n].k > a.p[y / a.n].k; });
  std::vector<int> flat((size_t)(total + 1) * 4, -1);
  for (int e = 0; e < total; ++e) { int* d = flat.data() + (size_t)e * 4; d[0] = order[e] / a.n; d[1] = order[e] % a.n; d[2] = 0; d[3] = 0; }
  int* d = nullptr;
  if (hipMalloc(&d, flat.size() * sizeof(int)) != hipSuccess) { *total_out = 0; return nullptr; }
  (void)hipMemcpy(d, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice);
  cache[key] = {d, total};
  *total_out = total;
  return d;
}

bool launch_resblock_pair(const RPArgs& ain, int num_cu, hipStream_t st) {
  RPArgs a = ain;
  if (!a.sched || !a.xb || !a.xflag || !a.mbox || !a.xcount) return false;
  a.tiles = rp_tiles(a, &a.ntiles);
  if (!a.tiles) return false;
  if (a.ntiles <= 0) return true;
  const int pairs = std::min(a.ntiles, num_cu / 2);
  if (a.T < 1 || a.T > 32) return false;
  if (a.T > 16) hipLaunchKernelGGL(resblock_pair_kernel<2>, dim3(2 * pairs), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(resblock_pair_kernel<1>, dim3(2 * pairs), dim3(512), 0, st, a);
  return true;
}

const char* resblock_pair_name(int T) { return T > 16 ? "cnk::resblock_pair_kernel<2>" : "cnk::resblock_pair_kernel<1>"; }

}  // namespace cnk
