// Argument structs of the conv_ns experiment (tools/experiments/conv_ns.hip); not part of the library.
#pragma once
#include "kernels.h"

namespace cnk {

// conv_ns.hip: causal conv of up to three same-shape problems (the branches of the first vocoder stage) on 64-row x
// 64-column tiles with channel-chunked LDS windows; weights fragment-major (PackedConv::wf layout).
struct NSProb {
  const float* w;        // [Cout/16 column tiles][k + 1 taps (last zero)][Cin/16 K groups][64 lanes][4]
  const float* bias;     // [Cout] or nullptr
  TRef x, y, res;
  float* y2_base;        // optional activated twin of y (same geometry)
  int k, dil, has_res;
};
struct NSArgs {
  NSProb p[3];
  const int* slots; const int* pos;
  const int* tiles; int ntiles;        // filled by launch_conv_ns
  int* sched;                          // two zeroed ints: draw counter, finished-block counter (re-armed by the kernel)
  int nprob, n, T, Cin, Cout;
  int SR;                              // rows per stream segment (0: chosen from T)
  int rows32;                          // 32-row tiles (more, smaller tiles) instead of 64-row tiles
#ifdef NS_STAMPS
  unsigned long long* dbg;             // developer build (tools/ns_bench -DNS_STAMPS)
#endif
  float in_slope;                      // LeakyReLU on the way in (1.0f: none)
  int out_act; float out_slope;        // ACT_NONE or ACT_LRELU
  float y2_slope;
  int shuffle_r;
};
bool conv_ns_supported(const NSArgs& a);
int conv_ns_segment_rows(int T, int tile_rows);
bool launch_conv_ns(const NSArgs& a, int num_cu, hipStream_t st);

}  // namespace cnk
