// resblock_limb: resblock_fused's tile pass (one ResBlock1 unit: xt = c1(lrelu(x)), y = c2(lrelu(xt)) + x, hifigan_causal.py:230-238)
// with every fp32 product computed as SIX bf16 limb products on the bf16 MFMA instead of one product on the f32 MFMA.
//
// Why: the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate and bounds the vocoder's ResBlock stages
// (0.7 of its peak is what resblock_fused reaches).  An fp32 value is the exact sum of three bf16 limbs,
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)      (8 + 8 + 8 significant bits and two sign bits),
// and a product of two bf16 values is exact in fp32, so
//     x * w  =  hh + (hm + mh) + (hl + mm + lh)  +  O(2^-25 |x w|)
// six products accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (smallest terms first) reproduce the fp32 product to the
// rounding of the accumulation itself: tools/experiments/bf16x3_gemm.hip measures the error against a float64 sum of the same
// fp32 inputs at 0.85 of the f32 MFMA's own (5.6e-7 against 6.7e-7 relative rms at K = 1408) and 1.8-2.0x its rate.  16 / 6 =
// 2.7x the f32 matrix rate is the ceiling; operands are 6 instead of 4 bytes per element (LDS images and weight streams).
//
// What changes against resblock_fused (same roles, barriers, work queue, merge build - see that file's header):
//   * LDS holds three bf16 PLANES (h, m, l) of the window and of xt, rows padded to C + 16 elements (conflict-free
//     ds_read_b128: a lane's A operand for one 16x16x32 MFMA is 8 consecutive channels of its row, 16 bytes per limb);
//     the helper waves split the window while staging it (LeakyReLU first), the matrix waves split xt in c1's epilogue;
//   * weights are packed per limb at finalize (ctx.hip pack_fragments): [column tile][k + 1 taps][C/32 K blocks][3 limbs][64 lanes]
//     x 16 bytes, streamed from L2 into registers as before, a ring of RING K blocks ahead;
//   * c2's accumulators go to the helpers through an f32 image that overlays the xt planes (dead behind B4);
//   * tiles are half as tall (6 bytes per element in LDS, and the block still has to share the CU with a decoder
//     megakernel workgroup: tests/test_kernel_resources.py);
//   * the MFMAs are issued transposed (weights as the first operand): a lane's accumulator holds 4 consecutive channels of one
//     row, so the epilogues store 8 / 16 bytes per LDS write;
//   * the helpers do everything for the NEXT tile while the matrix waves are in c1, the tile draw runs up to two tiles ahead,
//     every tile - the first one too - comes from the queue and is decoded arithmetically (no tile list, a {slot, pos} table
//     in LDS): at half the tile height the per-tile fixed costs are what decides (16-22 % of a block's life in barrier
//     waits before, 2-5 % now).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "kernels.h"

#define RL_MAX_SLOTS 256      // batch indices whose {slot, pos} the block keeps in LDS (launches with more take the f32 kernel)

#ifndef RL_HELPER_PRIO
#define RL_HELPER_PRIO 3      // (developer builds: the helper waves' s_setprio level)
#endif
#ifndef RL_MATRIX_PRIO
#define RL_MATRIX_PRIO 0
#endif
#ifndef RL_STAMPS
#define RL_STAMPS 0     // developer builds (tools/rb_bench -DRL_STAMPS=1): cycle stamps of the matrix waves per block
#endif

// developer build -DRL_ABLATE_W: every weight fragment from the tap-0 blocks (L1 hits, wrong results): what the L2 round trips cost
#ifdef RL_ABLATE_W
#define RL_WOFF(x) 0
#define RL_WOFF2(a_, b_) (b_)
#else
#define RL_WOFF(x) (x)
#define RL_WOFF2(a_, b_) (a_)
#endif

namespace cnk {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef const f32x4 __attribute__((address_space(1)))* gcf4;
typedef f32x4 __attribute__((address_space(1)))* gf4;
typedef const float __attribute__((address_space(1)))* gcf1;
typedef const int __attribute__((address_space(1)))* gci;

__device__ __forceinline__ f32x4 rl_gload(const void* p) { return *(gcf4)(p); }
__device__ __forceinline__ float4 rl_gload4(const float* p) { const f32x4 v = *(gcf4)(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void rl_gstore4(float* p, const float4 v) { *(gf4)(p) = (f32x4){v.x, v.y, v.z, v.w}; }

// x = h + m + l: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round-to-nearest-even at every step; the two subtractions are
// exact.  Two values at a time on the packed VALU forms (v_cvt_pk_bf16_f32, v_pk_add_f32): 6-7 instead of 11 instructions per
// element with the LeakyReLU in front - the matrix waves run c1's epilogue themselves, between two K loops
// (7-9 % of a block's life at C = 64 / 32), and the helpers' splits issue into the vector port the MFMAs use.
// h / m / l: two bf16 each, element 0 in the low half - the plane words as they are stored.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rl_split2(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  const f32x2 r1 = x - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  const f32x2 r2 = r1 - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}
// leaky_relu for 0 < slope < 1: max(x, slope * x) - the same bits as (x > 0 ? x : x * slope), signed zeros included
__device__ __forceinline__ f32x2 rl_lrelu2(const f32x2 x, const float slope) {
  const f32x2 sx = x * slope;
  return (f32x2){__builtin_fmaxf(x[0], sx[0]), __builtin_fmaxf(x[1], sx[1])};
}

template <int C_, int NR2_, int SPAN_>
struct RLGeom {
  static constexpr int C = C_, NR2 = NR2_;
  static constexpr int NR1 = NR2 + 1;                 // c1 row tiles: 16*NR1 >= 16*NR2 + (k-1) for k <= 17
  static constexpr int RO = 16 * NR2;                 // output rows per tile
  static constexpr int XT_ROWS = 16 * NR1;
  static constexpr int MAXSPAN = SPAN_;               // (k-1)*dil of c1
  static constexpr int WR_MAX = XT_ROWS + MAXSPAN;    // window rows
  // bf16 elements per LDS row: C + 16, i.e. a row stride of 2 (mod 4) 16-byte slots.  ds_read_b128 is serviced in the lane
  // groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with lane = (row & 15, 16-byte chunk
  // lane >> 4) a group mixes rows 0-3 / 12-15 at chunk c with rows 4-11 at chunk c + 1, and only these strides keep its 16
  // lanes on 16 distinct slots (C + 8 measured 2-way conflicts on most groups: the K loop ran LDS-bound at 0.65 of the MFMA rate)
  static constexpr int LDB = C + 16;
  static constexpr int PLW = WR_MAX * LDB;            // elements per window plane
  static constexpr int PLX = XT_ROWS * LDB;           // elements per xt plane
  static constexpr int LDA = C + 4;                   // floats per row of c2's accumulator image (overlays the xt planes)
  static constexpr int LDS_U16 = 3 * (PLW + PLX);
  static constexpr int NCT = C / 16;
  // Column tiles per matrix wave.  C = 128: two (four column pairs).  C = 32: two - one pair, the rows split four ways: per K
  // block a row tile's three A fragments then feed 12 MFMAs instead of 6 (the 16-cycle MFMAs leave the LDS reads little issue
  // room: 60 us against 65).  C = 64 keeps one (four strips, every wave all rows): two pairs x two row halves measured 92 us
  // against 86 - the row tiles of c2 (5) do not split evenly.
  static constexpr int NCW = C == 64 ? 1 : 2;
  static constexpr int RSPLIT = 4 / (NCT / NCW);
  static constexpr int NRW1 = (NR1 + RSPLIT - 1) / RSPLIT;
  static constexpr int NRW2 = (NR2 + RSPLIT - 1) / RSPLIT;
  static constexpr int KB = C / 32;                   // 32-deep K blocks per tap
  static constexpr int RING = 2;                      // weight K blocks in flight
  static constexpr int C4 = C / 4;
  static constexpr int BLK = 3 * 512;                 // u16 per (K block, column tile): 3 limbs x 1 KiB
  static_assert(C % 32 == 0 && NCT % NCW == 0 && 4 % (NCT / NCW) == 0 && (KB == 1 || KB % RING == 0), "channel count");
  static_assert(RO * LDA * 4 <= 3 * PLX * 2, "accumulator image inside the xt planes");
  static_assert(LDS_U16 * 2 + 32 + 8 * RL_MAX_SLOTS <= 160 * 1024, "LDS budget");
};

template <int NCW, int RING>
__device__ __forceinline__ void rl_prefetch_w(f32x4 (&bw)[RING][NCW][3], const u16* __restrict__ wl, const long long ct_stride) {
#pragma unroll
  for (int q = 0; q < RING; ++q) {
#pragma unroll
    for (int c = 0; c < NCW; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) bw[q][c][p] = rl_gload(wl + c * ct_stride + q * 1536 + p * 512);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One GEMM phase of a matrix wave: acc[r][c] += sum over (tap j, K block q) of A(rows of tile r shifted by j*tap_stride) x
// W(j, q, column tile c), each as six limb products.  The MFMA is issued TRANSPOSED (weights as its first operand, rows as
// its second - the register images are the same either way): acc[r][c][e] is output channel 16*c + 4*(lane>>4) + e of time
// row 16*r + (lane&15), four consecutive channels of one row per lane, so the epilogues store 8 / 16 bytes per LDS write.  `src` is plane 0 of the LDS operand image (planes `plane` elements
// apart, row stride LDB), `wl` the wave's first weight block (+ lane*8), column tiles ct_stride elements apart; bw holds the
// first RING blocks on entry and those of (wl_next, ct_stride_next) on exit.
template <int NRW, int NCW, int LDB, int KB, int RING>
__device__ __forceinline__ void rl_gemm(const u16* __restrict__ src, const int plane, const int row0, const int tap_stride, const int k,
                                        const u16* __restrict__ wl, const long long ct_stride, const u16* __restrict__ wl_next,
                                        const long long ct_stride_next, f32x4 (&acc)[NRW][NCW], f32x4 (&bw)[RING][NCW][3], const int lane) {
  const u16* abase = src + (row0 + (lane & 15)) * LDB + 8 * (lane >> 4);
  f32x4 af[NRW][3];
#pragma unroll
  for (int r = 0; r < NRW; ++r)
#pragma unroll
    for (int p = 0; p < 3; ++p) af[r][p] = *reinterpret_cast<const f32x4*>(abase + p * plane + r * 16 * LDB);
  const int tstep = tap_stride * LDB;
  if constexpr (KB == 1) {
    // one K block per tap (C = 32): the ring of two blocks spans two TAPS (one block ahead is 0.6 us of MFMAs, less than an L2
    // round trip under load).  Taps in pairs; an odd tap count ends on slot 0 and the slots are swapped, so that the next
    // phase again finds its first block in slot 0.
    static_assert(RING == 2, "ring");
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    auto block = [&](const int j, auto slot_c) __attribute__((always_inline)) {
      constexpr int SL = decltype(slot_c)::value;
      const u16* anext = abase + (j + 1) * tstep;
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
#pragma unroll
          for (int c = 0; c < NCW; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[SL][c][PB[s]]), __builtin_bit_cast(bf16x8, af[r][PA[s]]), acc[r][c], 0, 0, 0);
#ifndef RL_ABLATE_A      // (developer ablation: no A re-reads, wrong results: what the LDS fragment reads cost the K loops)
          if (s == 0) af[r][2] = *reinterpret_cast<const f32x4*>(anext + 2 * plane + r * 16 * LDB);
          if (s == 3) af[r][1] = *reinterpret_cast<const f32x4*>(anext + plane + r * 16 * LDB);
          if (s == 5) af[r][0] = *reinterpret_cast<const f32x4*>(anext + r * 16 * LDB);
#endif
        }
      const bool own = j + 2 < k;
      const u16* wsrc = own ? wl + (long long)(j + 2) * 1536 : wl_next + (long long)(j + 2 - k) * 1536;
      const long long cs = own ? ct_stride : ct_stride_next;
#pragma unroll
      for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[SL][c][p] = rl_gload(wsrc + c * cs + p * 512);
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
    };
    int j = 0;
    for (; j + 1 < k; j += 2) {
      block(j, std::integral_constant<int, 0>{});
      block(j + 1, std::integral_constant<int, 1>{});
    }
    if (j < k) {
      block(j, std::integral_constant<int, 0>{});
#pragma unroll
      for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) { const f32x4 t = bw[0][c][p]; bw[0][c][p] = bw[1][c][p]; bw[1][c][p] = t; }
    }
    return;
  }
  for (int j = 0; j < k; ++j) {
    const u16* arow = abase + j * tstep;
    const bool last = j + 1 == k;
    const u16* wnext = last ? wl_next : wl + (long long)(j + 1) * KB * 1536;
    const long long cnext = last ? ct_stride_next : ct_stride;
#pragma unroll
    for (int q = 0; q < KB; ++q) {
      const u16* anext = (q + 1 < KB) ? arow + (q + 1) * 32 : arow + tstep;
      // limb products, smallest first: l*h, m*m, h*l, m*h, h*m, h*h.  ONE fragment set: a row tile's l limb of the next block is
      // read right behind its last use (product 0), its m limb behind product 3, its h limb behind product 5 - needed again
      // at products 0, 1 and 2 of the next block, so every read has at least two product rounds to land.
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int r = 0; r < NRW; ++r) {
#pragma unroll
          for (int c = 0; c < NCW; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[q % RING][c][PB[s]]), __builtin_bit_cast(bf16x8, af[r][PA[s]]),
                                                                acc[r][c], 0, 0, 0);
#ifndef RL_ABLATE_A      // (developer ablation: no A re-reads, wrong results: what the LDS fragment reads cost the K loops)
          if (s == 0) af[r][2] = *reinterpret_cast<const f32x4*>(anext + 2 * plane + r * 16 * LDB);
          if (s == 3) af[r][1] = *reinterpret_cast<const f32x4*>(anext + plane + r * 16 * LDB);
          if (s == 5) af[r][0] = *reinterpret_cast<const f32x4*>(anext + r * 16 * LDB);
#endif
        }
      if (q + RING < KB) {
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) bw[q % RING][c][p] = rl_gload(wl + RL_WOFF(((long long)j * KB + q + RING) * 1536) + c * ct_stride + p * 512);
      } else {
#pragma unroll
        for (int c = 0; c < NCW; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) bw[q % RING][c][p] = rl_gload(RL_WOFF2(wnext, wl) + (q + RING - KB) * 1536 + c * cnext + p * 512);
      }
      // pin that order
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
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

}  // namespace

#define RL_SEL(br_, f) ((br_) == 0 ? a.p[0].f : ((br_) == 1 ? a.p[1].f : a.p[2].f))

template <int C, int NR2, int SPAN, bool MERGE>
__global__ __launch_bounds__(512, 2) void resblock_limb_kernel(const RBArgs a) {
  using G = RLGeom<C, NR2, SPAN>;
  constexpr int LDB = G::LDB;
  __shared__ __attribute__((aligned(16))) u16 lds[G::LDS_U16 + 16 + 4 * RL_MAX_SLOTS];   // + meta: {zero rows, drawn tile, next tile's branch, draw generation} + {slot, pos} per batch index
  u16* const win = lds;                                // [3][WR_MAX][LDB] limbs of leaky_relu(x)
  u16* const xt = lds + 3 * G::PLW;                    // [3][XT_ROWS][LDB] limbs of leaky_relu(c1 + b1)
  float* const accimg = reinterpret_cast<float*>(xt);  // [RO][LDA] c2's accumulators (behind B4)
  int* const meta = reinterpret_cast<int*>(lds + G::LDS_U16);
  int* const sptab = meta + 8;                          // [n][2]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = a.ntiles;
  const float slope = a.slope;
  const int T = a.T;
  constexpr bool merge = MERGE;
  const int np = a.nprob;

  if (wave >= 4) {
    // ============================================================ helper waves: window loader + output writer
    const int ht = tid - 256;
    __builtin_amdgcn_s_setprio(RL_HELPER_PRIO);
    auto hbar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // The next tile's window in two halves: win_issue() - behind B3, while the matrix waves are still in c1's epilogue - puts its
    // global loads in flight (registers); win_write() - behind B1, when the window is dead - applies LeakyReLU, splits into limbs
    // and stores the three planes.
    constexpr int NIT = (G::WR_MAX * G::C4 + 255) / 256;
    float4 wv[NIT];
    int wtotal = 0, wzr = 0;
    auto win_issue = [&](const int p, const int i, const int t0, const int slot, const int pos) __attribute__((always_inline)) {
      const int k = RL_SEL(p, k), d = RL_SEL(p, dil);
      const int wr = G::XT_ROWS + (k - 1) * d;
      const int tw0 = t0 - (k - 1) - (k - 1) * d;
      const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate;
      const float* xb = RL_SEL(p, x.base) + (long long)(xmode == 0 ? slot : i) * RL_SEL(p, x.slot_stride);
      const unsigned rbase = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)(RL_SEL(p, x.off) + tw0);
      const unsigned rmask = xmode == 0 ? (unsigned)RL_SEL(p, x.lmask) : 0xffffffffu;
      wtotal = wr * G::C4;
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int idx = ht + 256 * u;
        const int w = idx / G::C4, c4 = idx - w * G::C4;
        const unsigned row = (rbase + (unsigned)w) & rmask;
        wv[u] = idx < wtotal ? rl_gload4(xb + (long long)row * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // xt row m is time t0 - (k-1) + m; rows before the start of the stream (absolute time < 0) are c2's zero left padding
      const long long abs0 = (xmode == 0 ? (long long)pos * xrate : 0ll) + t0 - (k - 1);
      wzr = abs0 >= 0 ? 0 : (abs0 < -(long long)G::XT_ROWS ? G::XT_ROWS : (int)-abs0);
    };
    auto win_write = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int idx = ht + 256 * u;
        const int w = idx / G::C4, c4 = idx - w * G::C4;
        unsigned h[2], m[2], l[2];
        rl_split2(rl_lrelu2((f32x2){wv[u].x, wv[u].y}, slope), h[0], m[0], l[0]);
        rl_split2(rl_lrelu2((f32x2){wv[u].z, wv[u].w}, slope), h[1], m[1], l[1]);
        if (idx < wtotal) {
          u16* dst = win + w * LDB + c4 * 4;
          *reinterpret_cast<uint2*>(dst) = make_uint2(h[0], h[1]);
          *reinterpret_cast<uint2*>(dst + G::PLW) = make_uint2(m[0], m[1]);
          *reinterpret_cast<uint2*>(dst + 2 * G::PLW) = make_uint2(l[0], l[1]);
        }
      }
      if (ht == 0) meta[0] = wzr;
    };
    constexpr int NOUT = (G::RO * G::C4) / 256;
    static_assert((G::RO * G::C4) % 256 == 0 && 256 % G::C4 == 0, "output tile / helper threads");
    const int oc4 = ht % G::C4;
    float4 oacc[NOUT], ores[NOUT], osum[MERGE ? NOUT : 1], ob2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* oxb = nullptr; const float* ob2p = nullptr;
    float* oyb = nullptr;
    unsigned oyr0 = 0, oym = 0, oxr0 = 0, oxm = 0;
    int ot0 = 0, op = 0, oyC = C;
#pragma unroll
    for (int u = 0; u < (MERGE ? NOUT : 1); ++u) osum[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto out_fetch = [&](const int p, const int i, const int t0, const int slot, const int pos) __attribute__((always_inline)) {
      const int xmode = a.p[0].x.mode, xrate = a.p[0].x.rate, ymode = a.p[0].y.mode, yrate = a.p[0].y.rate;
      oxb = RL_SEL(p, x.base) + (long long)(xmode == 0 ? slot : i) * RL_SEL(p, x.slot_stride);
      oxr0 = (xmode == 0 ? (unsigned)pos * (unsigned)xrate : 0u) + (unsigned)(RL_SEL(p, x.off) + t0);
      oxm = xmode == 0 ? (unsigned)RL_SEL(p, x.lmask) : 0xffffffffu;
      if (!merge) {
        oyb = RL_SEL(p, y.base) + (long long)(ymode == 0 ? slot : i) * RL_SEL(p, y.slot_stride);
        oyr0 = (ymode == 0 ? (unsigned)pos * (unsigned)yrate : 0u) + (unsigned)(RL_SEL(p, y.off) + t0);
        oym = ymode == 0 ? (unsigned)RL_SEL(p, y.lmask) : 0xffffffffu;
      } else {
        const int mmode = a.ymean.mode;
        oyb = a.ymean.base + (long long)(mmode == 0 ? slot : i) * a.ymean.slot_stride;
        oyr0 = (mmode == 0 ? (unsigned)pos * (unsigned)a.ymean.rate : 0u) + (unsigned)(a.ymean.off + t0);
        oym = mmode == 0 ? (unsigned)a.ymean.lmask : 0xffffffffu;
        oyC = a.ymean.C;
      }
      ot0 = t0; op = p;
      ob2p = RL_SEL(p, b2) + oc4 * 4;
#pragma unroll
      for (int u = 0; u < NOUT; ++u) oacc[u] = *reinterpret_cast<const float4*>(accimg + ((ht + 256 * u) / G::C4) * G::LDA + oc4 * 4);
      // the residual rows (and c2's bias): in flight through the next tile's c1, consumed by out_store() behind B3 - which must
      // not wait for a load younger than the window loads issued just before it
      ob2 = rl_gload4(ob2p);
#pragma unroll
      for (int u = 0; u < NOUT; ++u) {
        const int r = (ht + 256 * u) / G::C4;
        ores[u] = ot0 + r < T ? rl_gload4(oxb + (long long)((oxr0 + (unsigned)r) & oxm) * C + oc4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto out_store = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < NOUT; ++u) {
        const int r = (ht + 256 * u) / G::C4;
        if (ot0 + r < T) {
          const float4 v = make_float4((oacc[u].x + ob2.x) + ores[u].x, (oacc[u].y + ob2.y) + ores[u].y, (oacc[u].z + ob2.z) + ores[u].z, (oacc[u].w + ob2.w) + ores[u].w);
          if constexpr (!MERGE) rl_gstore4(oyb + (long long)((oyr0 + (unsigned)r) & oym) * C + oc4 * 4, v);
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
              rl_gstore4(oyb + (long long)((oyr0 + (unsigned)r) & oym) * oyC + oc4 * 4, sm);
            }
          }
        }
      }
    };
    // EVERY tile is drawn from the queue, the first one too (issued here, its round trip hidden behind the table fill below): a
    // block that is dispatched late - its CU held by another stream's kernel for 100-200 us: the Emformer's workgroups need whole
    // CUs - then finds the queue empty and leaves, instead of starting a statically assigned first tile when everyone else is done
    // (pipelined steps: the launches' maxima were 60-80 us above their means).
#if RL_STAMPS
    const unsigned long long h_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long h_t[6] = {0, 0, 0, 0, 0, 0};
#define RL_HT(i_) h_t[i_] = __builtin_amdgcn_s_memtime() - h_t0
#else
#define RL_HT(i_)
#endif
    int dv0 = 0;
    if (wave == 4 && lane == 0) dv0 = __hip_atomic_fetch_add(a.sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // {slot, pos} of every batch index, once: a drawn tile is then decoded without a global load behind the draw itself
    for (int e = ht; e < a.n; e += 256) {
      const int sl = a.slots ? *(gci)(a.slots + e) : e;
      sptab[2 * e] = sl; sptab[2 * e + 1] = a.pos ? *(gci)(a.pos + sl) : 0;
    }
    const int tps = a.tiles_per_slot, per_prob = a.n * tps;
    // tile index -> {branch, batch index, first row} (the order of resblock_tiles(): separate branches by descending k, or
    // (slot, row tile)-major groups with the branches adjacent)
    auto decode = [&](const int e, int& p, int& i, int& t0) __attribute__((always_inline)) {
      if constexpr (!MERGE) {
        const int b = e / per_prob, rem = e - b * per_prob;
        p = b == 0 ? a.order[0] : (b == 1 ? a.order[1] : a.order[2]);
        i = rem / tps; t0 = (rem - i * tps) * G::RO;
      } else {
        const int g = e / np;
        p = e - g * np; i = g / tps; t0 = (g - i * tps) * G::RO;
      }
    };
    // The draw runs TWO tiles ahead of the matrix waves: `cur` is being computed, `nxt` (known since the previous tile) is
    // the window staged during cur, and the tile after it is drawn while cur's c1 runs - the atomic's round trip (and nothing
    // else: the decode is arithmetic + an LDS lookup) has that whole phase, so no barrier waits for it.  (One tile ahead, the
    // chain draw -> tile words -> slot -> pos -> window loads sat between B3 and B1 of every tile: 14-20 % of a block's life in
    // barrier waits at half the f32 pass's tile height.)
    int cur = -1, p = -1, i = 0, t0 = 0;
    if (wave == 4) {
      cur = __builtin_amdgcn_readfirstlane(dv0) * (merge ? np : 1);
      if (cur >= ntiles) cur = -1;
      if (cur >= 0) decode(cur, p, i, t0);
      if (lane == 0) { meta[1] = cur; meta[2] = p; meta[3] = 0; }
    }
    RL_HT(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // B(-1), block-wide: the table and the first tile are published
    RL_HT(1);
    if (wave != 4) {
      cur = __builtin_amdgcn_readfirstlane(meta[1]);
      if (cur >= 0) decode(cur, p, i, t0);
    }
    if (cur >= 0) {
    int slot = __builtin_amdgcn_readfirstlane(sptab[2 * i]), pos = __builtin_amdgcn_readfirstlane(sptab[2 * i + 1]);
    win_issue(p, i, t0, slot, pos);
    RL_HT(2);
    int gen = 1;                                         // generation of the published draw (meta[3])
    // draw(known_idx, known_p): issue; publish(): wave 4 lane 0 hands the index to the other helper waves through LDS
    int dv = 0, dnext = -1;                              // wave 4 lane 0: the counter value, in flight (NOT touched before draw_take: a
                                                         // use would put the atomic's round trip in front of B3), or the known successor
    auto draw_issue = [&](const int have, const int idx_prev, const int p_prev) __attribute__((always_inline)) {
      if (wave == 4 && lane == 0 && have) {
        if (merge && p_prev != np - 1) dnext = idx_prev + 1;       // next branch of this group: no draw
        else { dnext = -1; dv = __hip_atomic_fetch_add(a.sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
    };
    auto draw_take = [&](const int have) __attribute__((always_inline)) {
      int idx = -1;
      if (!have) return idx;
      if (wave == 4) {
        const int drawn = dv * (merge ? np : 1);                         // (merge: a drawn group's first tile)
        idx = __builtin_amdgcn_readfirstlane(dnext >= 0 ? dnext : drawn);
        if (idx >= ntiles) idx = -1;
        if (lane == 0) {
          meta[1] = idx;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __hip_atomic_store(&meta[3], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        while (__hip_atomic_load(&meta[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gen) __builtin_amdgcn_s_sleep(4);
        idx = __builtin_amdgcn_readfirstlane(meta[1]);
      }
      ++gen;
      return idx;
    };
    // (with few tiles per block - the C = 128 stage: 3.75 - committing two tiles ahead costs more balance than the barrier wait it
    // removes: 108 us against 97; such launches draw ONE tile ahead, the tile drawn during c1 is staged right behind B3)
    const bool deep = ntiles >= 5 * (int)gridDim.x * (merge ? np : 1);
    // (few scalars live across the loop: a tile is kept as its index and decoded where it is used)
    auto stage = [&](const int e) __attribute__((always_inline)) {      // the window loads of tile e
      int pe, ie, te;
      decode(e, pe, ie, te);
      win_issue(pe, ie, te, __builtin_amdgcn_readfirstlane(sptab[2 * ie]), __builtin_amdgcn_readfirstlane(sptab[2 * ie + 1]));
      return pe;
    };
    int nxt = -1, pn = -1;
    if (deep) {
      draw_issue(1, cur, p);
      nxt = draw_take(1);
    }
    RL_HT(3);
    win_write();
    RL_HT(4);
    hbar();                                              // B0: first window staged
    RL_HT(5);
#if RL_STAMPS
    if (a.dbg && blockIdx.x == 0 && wave == 4 && lane == 0) { unsigned long long* q = a.dbg + 256 * 4 + 12; for (int e = 0; e < 6; ++e) q[e] = h_t[e]; }
#endif
    bool pending = false;                                // a fetched output tile waits for its stores
    for (;;) {
      // cur's c1 is running and the helpers have nothing else to do: everything up to the next window's loads happens here, so
      // that B3 and B1 find the helper waves already waiting
      const bool have = deep ? nxt >= 0 : true;
      if (deep && nxt >= 0) pn = stage(nxt);
      draw_issue(have, deep ? nxt : cur, deep ? pn : p);   // deep: the tile after nxt
      if (pending) out_store();
      // deep: the drawn index is not needed before the next tile's loop top - it is TAKEN behind B2, a whole tile after the atomic
      // was issued (taken here, its L2 round trip stood in front of B3 whenever c1 was shorter than it: the 3-tap tiles of the
      // C = 32 stage spent 7 % of a block's life waiting there)
      int nn = -1;
      if (!deep) {
        nxt = draw_take(have); pn = -1;
        if (nxt >= 0) pn = stage(nxt);
      }
      if (ht == 0) meta[2] = nxt >= 0 ? pn : -1;         // the matrix waves read nxt's branch behind B1: c2's K loop ends by fetching nxt's first weight blocks
      hbar();                                            // B3: the previous tile's accumulators are in registers (out_fetch)
      hbar();                                            // B1: xt complete, window free
      if (nxt >= 0) win_write();
      hbar();                                            // B4
      hbar();                                            // B2: c2 accumulators in LDS, next window staged
      out_fetch(p, i, t0, slot, pos);
      pending = true;
      if (deep) nn = draw_take(have);
      if (nxt < 0) break;
      cur = nxt;
      decode(cur, p, i, t0);
      slot = __builtin_amdgcn_readfirstlane(sptab[2 * i]); pos = __builtin_amdgcn_readfirstlane(sptab[2 * i + 1]);
      nxt = nn;
    }
    if (pending) out_store();
    }      // (cur >= 0)
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
  if (RL_MATRIX_PRIO) __builtin_amdgcn_s_setprio(RL_MATRIX_PRIO);
  auto bar = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int wc = wave % (4 / G::RSPLIT);
  const int wr = wave / (4 / G::RSPLIT);
  const int ct0 = wc * G::NCW;
  const int lr = lane & 15, lg = lane >> 4;
#if RL_STAMPS
  unsigned long long st_gemm = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime(), st_bar = 0, st_b3 = 0, st_b1 = 0, st_b4 = 0, st_b2 = 0, st_e1 = 0, st_e2 = 0, st_tiles = 0, st_head = 0, st_pre = 0, st_mid = 0;
#define RL_T() __builtin_amdgcn_s_memtime()
#define RL_BAR(acc_) do { const unsigned long long q0_ = RL_T(); bar(); const unsigned long long q1_ = RL_T(); acc_ += q1_ - q0_; st_bar += q1_ - q0_; } while (0)
#define RL_GEMM(...) do { const unsigned long long s0_ = RL_T(); __VA_ARGS__; asm volatile("s_nop 0" ::"v"(acc[0][0][0])); st_gemm += RL_T() - s0_; } while (0)
#else
#define RL_BAR(acc_) bar()
#define RL_GEMM(...) __VA_ARGS__
#endif
  f32x4 bw[G::RING][G::NCW][3];
  bar();                                                 // B(-1): the helpers have drawn the block's first tile
  int p = __builtin_amdgcn_readfirstlane(meta[2]);
  if (p < 0) return;                                     // (dispatched after the queue ran empty)
  {
    const long long cs = (long long)(RL_SEL(p, k) + 1) * G::KB * G::BLK;
    rl_prefetch_w<G::NCW, G::RING>(bw, RL_SEL(p, w1l) + (long long)ct0 * cs + lane * 8, cs);
  }
  bar();                                                 // B0
#if RL_STAMPS
  st_head = RL_T() - st_t0;
#endif
  while (p >= 0) {
#if RL_STAMPS
    const unsigned long long top_ = RL_T();
#endif
    const int k = RL_SEL(p, k), d = RL_SEL(p, dil);
    const u16* const w1 = RL_SEL(p, w1l);
    const u16* const w2 = RL_SEL(p, w2l);
    const float* const b1 = RL_SEL(p, b1);
    const long long ct_stride = (long long)(k + 1) * G::KB * G::BLK;
    const int zrows = __builtin_amdgcn_readfirstlane(meta[0]);
    f32x4 b1v[G::NCW];
#pragma unroll
    for (int c = 0; c < G::NCW; ++c) b1v[c] = rl_gload(b1 + (ct0 + c) * 16 + 4 * lg);
    int pn;
    // ---------------- c1 over the halo-extended rows
    {
      f32x4 acc[G::NRW1][G::NCW];
#pragma unroll
      for (int r = 0; r < G::NRW1; ++r)
#pragma unroll
        for (int c = 0; c < G::NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int rt0 = wr * G::NRW1;
#if RL_STAMPS
      st_pre += RL_T() - top_;
#endif
      RL_GEMM(rl_gemm<G::NRW1, G::NCW, LDB, G::KB, G::RING>(win, G::PLW, rt0 * 16, d, k, w1 + (long long)ct0 * ct_stride + lane * 8, ct_stride,
                                                           w2 + (long long)ct0 * ct_stride + lane * 8, ct_stride, acc, bw, lane));
      RL_BAR(st_b3);                                     // B3
#if RL_STAMPS
      const unsigned long long e1_0 = RL_T();
#endif
      // (rows before the start of the stream - c2's zero padding - exist in a slot's first steps only: the selects that blank them
      // are compiled for those tiles alone)
      auto epilogue1 = [&](auto zr_c) __attribute__((always_inline)) {
        constexpr bool ZR = decltype(zr_c)::value;
#pragma unroll
        for (int c = 0; c < G::NCW; ++c) {
          const int col = (ct0 + c) * 16 + 4 * lg;
#pragma unroll
          for (int r = 0; r < G::NRW1; ++r) {
            if (rt0 + r < G::NR1) {
              const int m = (rt0 + r) * 16 + lr;
              unsigned h[2], mm[2], l[2];
              const f32x4 v = acc[r][c] + b1v[c];
              rl_split2(rl_lrelu2((f32x2){v[0], v[1]}, slope), h[0], mm[0], l[0]);
              rl_split2(rl_lrelu2((f32x2){v[2], v[3]}, slope), h[1], mm[1], l[1]);
              const bool zr = ZR && m < zrows;
              u16* dst = xt + m * LDB + col;
              *reinterpret_cast<uint2*>(dst) = zr ? make_uint2(0u, 0u) : make_uint2(h[0], h[1]);
              *reinterpret_cast<uint2*>(dst + G::PLX) = zr ? make_uint2(0u, 0u) : make_uint2(mm[0], mm[1]);
              *reinterpret_cast<uint2*>(dst + 2 * G::PLX) = zr ? make_uint2(0u, 0u) : make_uint2(l[0], l[1]);
            }
          }
        }
      };
      if (zrows > 0) epilogue1(std::true_type{}); else epilogue1(std::false_type{});
#if RL_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_e1 += RL_T() - e1_0; ++st_tiles;
#endif
    }
    RL_BAR(st_b1);                                       // B1: xt complete, window free
    // The next tile's branch is known by now (the helpers publish it before B3): c2's K loop refills its weight ring, behind its
    // last taps, with the first blocks of THAT tile's c1 - fetched behind B4 instead, their L2 round trip (2-3 k cycles) stood in
    // front of every tile's first MFMA (7 % of a C = 128 tile, a quarter of a C = 32 one).
#if RL_STAMPS
    const unsigned long long mid_ = RL_T();
#endif
    pn = __builtin_amdgcn_readfirstlane(meta[2]);
    const int pnx = pn >= 0 ? pn : p;
    const long long csn = (long long)(RL_SEL(pnx, k) + 1) * G::KB * G::BLK;
    const u16* const w1n = RL_SEL(pnx, w1l) + (long long)ct0 * csn + lane * 8;
    // ---------------- c2
    {
      f32x4 acc[G::NRW2][G::NCW];
#pragma unroll
      for (int r = 0; r < G::NRW2; ++r)
#pragma unroll
        for (int c = 0; c < G::NCW; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int rt0 = wr * G::NRW2;
#if RL_STAMPS
      st_mid += RL_T() - mid_;
#endif
      RL_GEMM(rl_gemm<G::NRW2, G::NCW, LDB, G::KB, G::RING>(xt, G::PLX, rt0 * 16, 1, k, w2 + (long long)ct0 * ct_stride + lane * 8, ct_stride,
                                                           w1n, csn, acc, bw, lane));
      RL_BAR(st_b4);                                     // B4: every matrix wave is done reading xt
#if RL_STAMPS
      const unsigned long long e2_0 = RL_T();
#endif
#pragma unroll
      for (int c = 0; c < G::NCW; ++c) {
        const int col = (ct0 + c) * 16 + 4 * lg;
#pragma unroll
        for (int r = 0; r < G::NRW2; ++r) {
          if (rt0 + r < NR2) *reinterpret_cast<f32x4*>(accimg + ((rt0 + r) * 16 + lr) * G::LDA + col) = acc[r][c];
        }
      }
#if RL_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_e2 += RL_T() - e2_0;
#endif
    }
    RL_BAR(st_b2);                                       // B2
    p = pn;
  }
#if RL_STAMPS
  if (a.dbg && tid == 0) {
    unsigned long long* o = a.dbg + blockIdx.x * 4;
    if (blockIdx.x == 0) { unsigned long long* q = a.dbg + 256 * 4; q[0] = st_b3; q[1] = st_b1; q[2] = st_b4; q[3] = st_b2; q[4] = st_e1; q[5] = st_e2; q[6] = st_tiles; q[7] = st_gemm; q[8] = __builtin_amdgcn_s_memtime() - st_t0; q[9] = st_head; q[10] = st_pre; q[11] = st_mid; }
    o[0] = st_gemm; o[1] = __builtin_amdgcn_s_memtime() - st_t0; o[2] = __builtin_amdgcn_s_memrealtime() - st_r0; o[3] = st_bar;
  }
#endif
}

// ------------------------------------------------------------------------------------------------ host side

namespace {
struct RLCand { int C, NR2, span; };
// tile heights that leave a decoder megakernel workgroup its 34 KB of LDS beside the block (6 bytes per element here)
const RLCand kLimbCands[] = {{32, 10, 50}, {64, 5, 50}, {128, 2, 50}};
}

bool resblock_limb_supported(int C, int kmax, int span_max) {
  for (const RLCand& c : kLimbCands)
    if (c.C == C && span_max <= c.span) return kmax <= 16;
  return false;
}

int resblock_limb_rows(int C, int span_max) {
  int best = 0;
  for (const RLCand& c : kLimbCands)
    if (c.C == C && span_max <= c.span) best = std::max(best, 16 * c.NR2);
  return best;
}

template <int C, int NR2, int SPAN, bool MERGE = false>
static bool launch_rl(const RBArgs& ain, int num_cu, hipStream_t st) {
  RBArgs a = ain;
  const int ro = 16 * NR2;
  a.tiles_per_slot = (a.T + ro - 1) / ro;
  const int total = a.nprob * a.n * a.tiles_per_slot;
  if (total <= 0) return true;
  if (!a.sched) return false;
  for (int p = 0; p < a.nprob; ++p) if (!a.p[p].w1l || !a.p[p].w2l) return false;
  if (a.n > RL_MAX_SLOTS) return false;
  if (!(a.slope > 0.f && a.slope <= 1.f)) return false;      // (LeakyReLU is formed as max(x, slope * x))
  for (int p = 0; p < 3; ++p) a.order[p] = p;
  std::stable_sort(a.order, a.order + a.nprob, [&](int x, int y) { return a.p[x].k > a.p[y].k; });
  const int grid = std::min(a.merge ? total / a.nprob : total, num_cu);
  a.tiles = nullptr; a.ntiles = total;      // (tile index -> {branch, batch index, row} is arithmetic in this kernel: no list)
  hipLaunchKernelGGL((resblock_limb_kernel<C, NR2, SPAN, MERGE>), dim3(grid), dim3(512), 0, st, a);
  return true;
}

bool launch_resblock_limb(const RBArgs& a, int C, int rows, int num_cu, hipStream_t st) {
  const int nr2 = rows / 16;
  if (a.merge) {
    if (C == 32 && nr2 == 10) return launch_rl<32, 10, 50, true>(a, num_cu, st);
    if (C == 64 && nr2 == 5) return launch_rl<64, 5, 50, true>(a, num_cu, st);
    if (C == 128 && nr2 == 2) return launch_rl<128, 2, 50, true>(a, num_cu, st);
    return false;
  }
  if (C == 32 && nr2 == 10) return launch_rl<32, 10, 50>(a, num_cu, st);
  if (C == 64 && nr2 == 5) return launch_rl<64, 5, 50>(a, num_cu, st);
  if (C == 128 && nr2 == 2) return launch_rl<128, 2, 50>(a, num_cu, st);
  return false;
}

bool resblock_limb_can_merge(int C, int rows) { return (C == 32 && rows == 160) || (C == 64 && rows == 80) || (C == 128 && rows == 32); }

const char* resblock_limb_name(int C, int rows, bool merge) {
  static thread_local char buf[72];
  const int span = 50;
  snprintf(buf, sizeof(buf), "cnk::resblock_limb_kernel<%d, %d, %d, %s>", C, rows / 16, span, merge ? "true" : "false");
  return buf;
}

}  // namespace cnk
