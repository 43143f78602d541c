// rowconv: causal conv / linear for the frame-rate layers of the Conan decoder step (a few hundred rows: streams x
// frames), with an optional LayerNorm fused in front (gfx950, exact-f32 MFMA v_mfma_f32_16x16x4_f32).
//
//   y[i][t][co] = epilogue( sum_j sum_ci W[j][ci][co] * f(x)[i][t - (k-1-j)*dil][ci] )
//   f = identity, or LayerNorm over the channels (gamma, beta, eps) of the NEW rows - rows of earlier steps come out of
//       the layer's ring already normalised, and the new normalised rows are appended to it (first column tile only);
//   epilogue(a) = ((act((a + bias) * scale) + bvec[slot]) + res) * m1 * m2
//
// Why not conv_mfma: at M = streams x frames = 256 rows its 32x32 tiles need inter-block split-K to fill the chip, a
// 96 KB LDS ring and 256 blocks per launch; beside the persistent vocoder launches of a pipelined step (one block per CU,
// 100+ KB of LDS each) such a launch waits for a vocoder kernel boundary.  Here a block is 4 waves with <= 36 KB of LDS
// and < 100 VGPRs - it fits on a CU NEXT to a resident vocoder block - and a launch has 16-128 blocks:
//   * the block owns 16 output rows (MFMA row tile) x 64*NCW output columns; the rows of up to a few streams with their
//     (k-1)*dil rows of left context are gathered ONCE into an LDS window (row stride Cin + 8 floats: conflict-free
//     ds_read_b128), LayerNorm is applied there, and the k taps are row-shifted fragment reads of that window;
//   * a wave owns whole 16-column strips, so its weights (fragment-major, 1 KiB per 16x16 operand, the layout of
//     resblock_fused.hip) stream from L2 straight into registers through a 4-deep ring; the K loop has no barrier;
//   * separate LayerNorm launches disappear (13 per decoder step).
#include <atomic>
#include <cstdlib>

#include "kernels.h"
#include "rowops.h"

namespace cnk {

// (device code: rowops.h - the tile functions are shared with the decoder megakernel)

// NCW column tiles x NRW row tiles per wave: <1,1> / <4,1> for the decoder's frame-rate layers.  (NRW = 2, 32-row tiles,
// was measured for the decoder at 64 streams and for the first vocoder stage: its 70-100 KB window no longer fits on a
// CU beside a vocoder block and the step got 14 % slower - see DESIGN.md; only NRW = 1 is instantiated.)
// KW = 4 (one row tile in the whole launch: a handful of streams): the four waves of a block share ONE 16-column strip
// and split its K groups, partial tiles meet in LDS - the serial MFMA chain and the weight stream per wave shrink 4x and
// the launch has 4x the blocks.
template <int NCW, int NRW, int KW = 1>
// (launch bound: 6 waves per SIMD = at most 80 VGPRs for the one-column-tile variants, which need no spill for it - a wave
// then fits beside two waves of a fused ResBlock pass (2 x 216 of a SIMD's 512 registers), so the decoder stream's
// launches run on CUs the vocoder's persistent blocks hold instead of waiting for the end of its kernel)
__global__ __launch_bounds__(256, NCW == 1 ? 6 : 4) void rowconv_kernel(const RowConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  ro::rowconv_tile<NCW, NRW, KW, false>(a, blockIdx.x, blockIdx.y, lds);
}

__global__ __launch_bounds__(256, 6) void rowlin_kernel(const RowConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float win[];     // [16][RL_LDX] (dynamic: a static 33 KB would make the compiler give up the 80-VGPR bound)
  ro::rowlin_tile<false>(a, blockIdx.x, blockIdx.y, win);
}

using ro::RC_TM; using ro::RC_MAXSEG; using ro::RL_CW; using ro::RL_LDX;

int rowconv_lds_bytes(const RowConvArgs& a, bool ksplit) {
  return (32 + a.wr_max * (a.Cin + 8) + (ksplit ? 3 * 64 * 4 : 0)) * 4;    // window (+ the K-split reduction patch)
}

static int rc_window_rows(int tm, int T, int halo) {
  const int segs = T >= tm ? (tm == T ? 1 : 2) : (tm + T - 1) / T + (tm % T ? 1 : 0);
  return tm + (segs > RC_MAXSEG ? RC_MAXSEG : segs) * halo;
}

bool rowconv_supported(int Cin, int ktaps, int dil, int T) {
  const int KQ = Cin / 16;
  if (Cin > 512) return ktaps == 1 && Cin % RL_CW == 0 && T >= 1;    // wide 1x1 layers: rowlin_kernel (no LayerNorm prologue)
  if (Cin % 64 || (KQ & (KQ - 1)) || (ktaps * KQ) % 8) return false;   // K groups: a power of two per tap, a multiple of the ring depth in all
  if (T < 2) return false;                                              // <= 8 streams per 16-row tile
  const int wr = rc_window_rows(RC_TM, T, (ktaps - 1) * dil);
  return (32 + wr * (Cin + 8)) * 4 <= 60 * 1024;
}

template <int NCW, int NRW, int KW = 1>
static void rc_launch(const RowConvArgs& a, int mt, int nt, int lds, hipStream_t st) {
  if (lds > 64 * 1024) {   // more dynamic LDS than the default cap: once per device
    static std::atomic<unsigned long long> devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(devs.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute((const void*)rowconv_kernel<NCW, NRW, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);   // + 240 B static
      devs.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((rowconv_kernel<NCW, NRW, KW>), dim3(nt, mt), dim3(256), lds, st, a);
}

// which kernel a launch uses: 0 <1,1,1>, 1 <4,1,1>, 2 <1,1,4>, 3 rowlin
// plan_rows: the row count the PLAN is made for (0: the launch's own n * T; a fixed-plan stream-set passes max_slots * T, so that the
// single-tile K-split form - a different summation order - is chosen by the stream-set's size, not by the active slots)
int rowconv_variant(const RowConvArgs& a, int plan_rows);
static int rc_variant(const RowConvArgs& a, int plan_rows) { return rowconv_variant(a, plan_rows); }
int rowconv_variant(const RowConvArgs& a, int plan_rows) {
  static const bool no_ksplit = dev_getenv("CONAN_RC_NOKSPLIT") != nullptr;   // developer switch
  if (a.Cin > 512) return 3;
  const int mt = ((plan_rows > 0 ? plan_rows : a.n * a.T) + RC_TM - 1) / RC_TM;
  // a single row tile (<= 16 rows in the launch): K split over the waves of a block, one 16-column strip per block
  // (the waves take runs of 4 K groups: rc_krange - uneven when the group count is not a multiple of 16, e.g. 128 channels x 5 taps)
  if (mt == 1 && ((a.ktaps * (a.Cin >> 4)) % 4) == 0 && !no_ksplit) return 2;
  // wide layers: 4 column tiles per wave (256 columns per block) keep the block count near the CU count
  static const int wide_min = dev_getenv("CONAN_RC_WIDE_MIN") ? atoi(dev_getenv("CONAN_RC_WIDE_MIN")) : 1024;    // developer switch
  return (a.Cout_pad >= wide_min && a.Cout_pad % 256 == 0) ? 1 : 0;
}
const char* rowconv_kernel_name(const RowConvArgs& a, int plan_rows) {
  static const char* names[4] = {"cnk::rowconv_kernel<1, 1, 1>", "cnk::rowconv_kernel<4, 1, 1>", "cnk::rowconv_kernel<1, 1, 4>", "cnk::rowlin_kernel"};
  return names[rc_variant(a, plan_rows)];
}

// Tile geometry of a launch for the decoder megakernel: fills wr_max, returns the variant it would run as ONE-column-tile
// tiles (0: <1,1,1>, 2: <1,1,4>, 3: rowlin; the 4-column-tile build needs more registers than the megakernel's bound) with
// the tile grid (nbx column strips x nby row tiles) and the LDS floats a tile needs.
int rowconv_plan(RowConvArgs& a, int* nbx, int* nby, int* lds_floats, int plan_rows) {
  const int M = a.n * a.T, halo = (a.ktaps - 1) * a.dil;
  int v = rc_variant(a, plan_rows);
  if (v == 1) v = 0;
  a.wr_max = rc_window_rows(RC_TM, a.T, halo);
  const int mt = (M + RC_TM - 1) / RC_TM;
  if (v == 3) { *nbx = (a.Cout_pad + 63) / 64; *nby = mt; *lds_floats = RC_TM * RL_LDX; return v; }
  *lds_floats = rowconv_lds_bytes(a, v == 2) / 4;
  if (v == 2) { *nbx = (a.Cout_pad + 15) / 16; *nby = 1; } else { *nbx = (a.Cout_pad + 63) / 64; *nby = mt; }
  return v;
}

void launch_rowconv(const RowConvArgs& ain, hipStream_t st, int plan_rows) {
  RowConvArgs a = ain;
  const int T = a.T, M = a.n * T;
  if (M <= 0) return;
  const int halo = (a.ktaps - 1) * a.dil;
  const int ncols = a.Cout_pad;
  const int v = rc_variant(a, plan_rows);
  a.wr_max = rc_window_rows(RC_TM, T, halo);
  const int lds = rowconv_lds_bytes(a, v == 2);
  const int mt = (M + RC_TM - 1) / RC_TM;
  if (v == 3) {      // (ln / in_lrelu / history are not this kernel's: rowconv_ok() callers pass plain 1x1 layers)
    hipLaunchKernelGGL(rowlin_kernel, dim3((ncols + 63) / 64, mt), dim3(256), RC_TM * RL_LDX * sizeof(float), st, a);
    return;
  }
  switch (v) {
    case 2: rc_launch<1, 1, 4>(a, 1, (ncols + 15) / 16, lds, st); break;
    case 1: rc_launch<4, 1>(a, mt, (ncols + 255) / 256, lds, st); break;
    default: rc_launch<1, 1>(a, mt, (ncols + 63) / 64, lds, st); break;
  }
}

}  // namespace cnk
