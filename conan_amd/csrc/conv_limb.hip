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

// SK: the build with the split-K tail (conv_limb_sk_kernel; the hand-over costs the <5,1,1,4> shape two registers over its 192-register
// budget beside the decoder, so the launches without a tail run the build without it)
template <int NRW, int NCW, int RW, int CW>
__global__ __launch_bounds__(512, 2) void conv_limb_kernel(const ConvLimbGroup g) {
  constexpr bool SK = false;
#include "conv_limb_body.inc"
}
template <int NRW, int NCW, int RW, int CW>
__global__ __launch_bounds__(512, 2) void conv_limb_sk_kernel(const ConvLimbGroup g) {
  constexpr bool SK = true;
#include "conv_limb_body.inc"
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
template <int NRW, int NCW, int RW, int CW>
void launch_cl_sk(const ConvLimbGroup& g, int grid, size_t lds_bytes, hipStream_t st) {
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)conv_limb_sk_kernel<NRW, NCW, RW, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((conv_limb_sk_kernel<NRW, NCW, RW, CW>), dim3(grid), dim3(512), lds_bytes, st, g);
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
    const int tailS = (nprob == 1 && tail_split && plan_n == 0 && si == 0) ? tail_slices(p[0], s, num_cu) : 1;
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
  int tailS = (g.nprob == 1 && g.slab && g.counters && shape == 0) ? tail_slices(g.p[0], kShapes[shape], num_cu) : 1;
  if (tailS >= 2) {
    const int TM = 16 * kShapes[shape].NRW * kShapes[shape].RW, TN = 16 * kShapes[shape].NCW * kShapes[shape].CW;
    const long long tiles = (((long long)g.p[0].n * g.p[0].T + TM - 1) / TM) * ((((g.p[0].Cout + 15) / 16) * 16) / TN), rem = tiles % num_cu;
    if (rem * tailS * TM * TN > g.slab_floats || rem * 4 > g.max_counters || rem >= 65536) tailS = 1;
  }
  if (tailS < 2) { g.slab = nullptr; g.counters = nullptr; }
  if (!cl_schedule(g, shape, num_cu, &grid, tailS) || grid <= 0) return false;
  if (tailS >= 2 && shape != 0) return false;       // (the split-tail build exists for the 64 x 64 shape only)
  if (tailS >= 2) launch_cl_sk<4, 1, 1, 4>(g, grid, lds, st);
  else switch (shape) {
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
