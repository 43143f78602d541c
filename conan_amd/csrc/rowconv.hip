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

namespace cnk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* rc_gcf4;
typedef const int __attribute__((address_space(1)))* rc_gci;
__device__ __forceinline__ float4 rc_gload4(const float* p) { const f32x4 v = *(rc_gcf4)(p); return make_float4(v[0], v[1], v[2], v[3]); }

__device__ __forceinline__ float rc_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ const float* rc_row(const TRef& r, int i, int slot, int pos, int t) {
  if (r.mode == 0) return r.base + (long long)slot * r.slot_stride + (long long)(((unsigned)pos * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask) * r.C;
  return r.base + (long long)i * r.slot_stride + (long long)(r.off + t) * r.C;
}

constexpr int RC_TM = 16;          // output rows per MFMA row tile
constexpr int RC_MAXSEG = 8;       // streams a tile may touch (T >= 2)
// depth of the weight-fragment ring (K groups in flight): 8 x 4 MFMAs for one column tile per wave, 4 x 16 for four

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
  static_assert(KW == 1 || (NCW == 1 && NRW == 1), "K split: one tile per wave");
  constexpr int RC_D = (KW > 1) ? 4 : (NCW * NRW == 1) ? 8 : 4;
  constexpr int TMB = RC_TM * NRW;                     // output rows per block
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, n = a.n, Mtot = n * T;
  // blockIdx.x = column strip (fastest), blockIdx.y = row tile: workgroups are dealt round-robin over the 8 XCDs, so with
  // 8 (or 4) strips an XCD always draws the same strip(s) and its L2 fetches their weights once for all row tiles
  const int m0 = blockIdx.y * TMB;
  const int ntile = blockIdx.x;                       // 64*NCW output columns
  const int Cin = a.Cin, LDX = Cin + 8, C4 = Cin >> 2;
  const int k = a.ktaps, d = a.dil, halo = (k - 1) * d;
  // ---- window geometry: output row r of the tile is (stream i_r, time t_r); the rows of one stream are consecutive, each
  // stream segment is preceded by its `halo` rows of left context
  int* const tab = reinterpret_cast<int*>(lds);        // [0..32): window row of output row r at tap 0
  float* const win = lds + 32;                          // [wr_max][LDX]
  __shared__ int seg_i[RC_MAXSEG + 1], seg_t0[RC_MAXSEG + 1], seg_off[RC_MAXSEG + 1], seg_slot[RC_MAXSEG + 1], seg_pos[RC_MAXSEG + 1];
  __shared__ int s_wr;
  // one lane per output row (wave 0): rows of one stream are consecutive; a segment starts where the stream changes.
  // Window layout [halo_0 | rows_0 | halo_1 | rows_1 | ...]: segment s starting at tile row r0 begins at r0 + s*halo and
  // tap 0 of tile row r reads window row r + s*halo.  The slot / position loads of all segments fly together.
  if (tid < 64) {
    if (tid <= RC_MAXSEG) { seg_i[tid] = -1; seg_off[tid] = 0x7fffffff; }
    const int r = lane, m = m0 + r;
    const bool valid = r < TMB && m < Mtot;
    const int i = valid ? m / T : -1, t = valid ? m - i * T : 0;
    const int iprev = __shfl_up(i, 1);
    const bool start = valid && (r == 0 || i != iprev);
    const unsigned long long sb = __ballot(start);
    const int sidx = __popcll(sb & ((2ull << r) - 1ull)) - 1;
    const int slot = start ? (a.slots ? *(rc_gci)(a.slots + i) : i) : 0;
    const int pos = start ? (a.pos ? *(rc_gci)(a.pos + slot) : 0) : 0;
    if (r < TMB) tab[r] = valid ? r + sidx * halo : 0;
    if (start) { seg_i[sidx] = i; seg_t0[sidx] = t; seg_off[sidx] = r + sidx * halo; seg_slot[sidx] = slot; seg_pos[sidx] = pos; }
    if (lane == 0) { const int nvalid = Mtot - m0 < TMB ? Mtot - m0 : TMB; s_wr = nvalid + __popcll(sb) * halo; }
  }
  __syncthreads();
  const int WR = s_wr;
  // ---- gather the window (raw), 8 rows-of-16-bytes per thread in flight at a time
  auto wseg = [&](int w) __attribute__((always_inline)) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < RC_MAXSEG; ++q) s += (w >= seg_off[q]) ? 1 : 0;      // seg_off of unused segments is INT_MAX
    return s;
  };
  const int total = WR * C4;
  for (int e0 = 0; e0 < total; e0 += 256 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + tid + 256 * u;
      const int w = e < total ? e / C4 : 0, c4 = e < total ? e - w * C4 : 0;
      const int sg = wseg(w);
      const int tau = seg_t0[sg] - halo + (w - seg_off[sg]);   // time index within this step (negative: earlier steps)
      const float* src = (a.ln && tau >= 0) ? rc_row(a.x, seg_i[sg], seg_slot[sg], seg_pos[sg], tau)        // new rows: raw layer input
                                            : rc_row(a.ln ? a.hist : a.x, seg_i[sg], seg_slot[sg], seg_pos[sg], tau);   // ring (history, or plain input)
      v[u] = rc_gload4(src + c4 * 4);
    }
    const float isl = a.in_lrelu ? a.in_slope : 1.0f;      // LeakyReLU on the way in (HiFi-GAN resblock convs)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + tid + 256 * u;
      float4 q = v[u];
      q.x *= q.x > 0.f ? 1.0f : isl; q.y *= q.y > 0.f ? 1.0f : isl; q.z *= q.z > 0.f ? 1.0f : isl; q.w *= q.w > 0.f ? 1.0f : isl;
      if (e < total) { const int w = e / C4, c4 = e - w * C4; *reinterpret_cast<float4*>(win + w * LDX + c4 * 4) = q; }
    }
  }
  __syncthreads();
  // ---- LayerNorm of the new rows in place: 16 lanes per row, 4 rows per wave at a time; the first column tile appends
  // them to the layer's ring and writes the block mask (row has any non-zero input: nonpadding of a residual block)
  if (a.ln) {
    const int sub = lane >> 4, l16 = lane & 15;
    for (int w = wave * 4 + sub; w < ((WR + 15) & ~15); w += 16) {
      const bool inw = w < WR;
      const int sg = wseg(inw ? w : 0);
      const int tau = seg_t0[sg] - halo + ((inw ? w : 0) - seg_off[sg]);
      const bool live = inw && tau >= 0;
      float* row = win + (inw ? w : 0) * LDX;
      float4 v[8];                                          // Cin <= 512: 8 float4 per lane
      float sum = 0.f, sa = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = (l16 + 16 * q) * 4;
        v[q] = (live && c < Cin) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        sa += (fabsf(v[q].x) + fabsf(v[q].y)) + (fabsf(v[q].z) + fabsf(v[q].w));
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); sa += __shfl_xor(sa, o); }
      const float mean = sum / (float)Cin;
      float var = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = (l16 + 16 * q) * 4;
        if (c < Cin) { const float d0 = v[q].x - mean, d1 = v[q].y - mean, d2 = v[q].z - mean, d3 = v[q].w - mean; var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3); }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) var += __shfl_xor(var, o);
      const float rstd = 1.0f / sqrtf(var / (float)Cin + a.eps);
      if (live) {
        float mk = 1.f;
        if (a.has_lnmask) mk = *rc_row(a.lnmask, seg_i[sg], seg_slot[sg], seg_pos[sg], tau);
        float* hrow = const_cast<float*>(rc_row(a.hist, seg_i[sg], seg_slot[sg], seg_pos[sg], tau));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = (l16 + 16 * q) * 4;
          if (c < Cin) {
            const float4 g = *reinterpret_cast<const float4*>(a.gamma + c), bb = *reinterpret_cast<const float4*>(a.beta + c);
            const float4 o = make_float4(((v[q].x - mean) * rstd * g.x + bb.x) * mk, ((v[q].y - mean) * rstd * g.y + bb.y) * mk,
                                         ((v[q].z - mean) * rstd * g.z + bb.z) * mk, ((v[q].w - mean) * rstd * g.w + bb.w) * mk);
            *reinterpret_cast<float4*>(row + c) = o;
            if (ntile == 0) *reinterpret_cast<float4*>(hrow + c) = o;
          }
        }
        if (a.has_mask_out && ntile == 0 && l16 == 0) *const_cast<float*>(rc_row(a.mask_out, seg_i[sg], seg_slot[sg], seg_pos[sg], tau)) = sa > 0.f ? 1.f : 0.f;
      }
    }
    __syncthreads();
  }
  // ---- K loop: wave w owns column tiles ct0 .. ct0 + NCW - 1 of this block's strip
  const int KQ = Cin >> 4;                              // 16-deep K groups per tap (power of two, >= RC_D)
  const int NG = k * KQ;
  const int ct0 = KW > 1 ? ntile : (ntile * 4 + wave) * NCW;
  const int g_lo = KW > 1 ? (NG / KW) * wave : 0, g_hi = KW > 1 ? g_lo + NG / KW : NG;    // NG % (KW * RC_D) == 0 (host)
  const int lr = lane & 15, lg = lane >> 4;
  const float* abase[NRW];
#pragma unroll
  for (int r = 0; r < NRW; ++r) abase[r] = win + tab[r * RC_TM + lr] * LDX + 4 * lg;
  const long long ct_stride = (long long)(k + 1) * KQ * 256;      // floats per column tile (k taps + one zero tap)
  const float* wl = a.w + (long long)ct0 * ct_stride + lane * 4;
  const bool active = ct0 * 16 < a.Cout_pad;            // column tiles past the padded width have no weights
  // one row tile x one column tile per wave: its MFMAs would form ONE dependent chain (40-cycle latency against a 32-cycle
  // issue interval) - even and odd K groups accumulate separately and are summed at the end
  constexpr int NACC = (NCW * NRW == 1) ? 2 : 1;
  f32x4 accs[NACC][NRW][NCW];
#pragma unroll
  for (int s2 = 0; s2 < NACC; ++s2)
#pragma unroll
    for (int r = 0; r < NRW; ++r)
#pragma unroll
      for (int c = 0; c < NCW; ++c) accs[s2][r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (active) {
    float4 bw[RC_D][NCW];
#pragma unroll
    for (int u = 0; u < RC_D; ++u) {
#pragma unroll
      for (int c = 0; c < NCW; ++c) bw[u][c] = rc_gload4(wl + c * ct_stride + (long long)(g_lo + u) * 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int kqm = KQ - 1, kqs = 31 - __builtin_clz(KQ);
    const int tstep = d * LDX;
    float4 af[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const float4*>(abase[r] + (g_lo >> kqs) * tstep + (g_lo & kqm) * 16);
    for (int G0 = g_lo; G0 < g_hi; G0 += RC_D) {
#pragma unroll
      for (int u = 0; u < RC_D; ++u) {
        const int Gn = G0 + u + 1;                      // next group's A fragments (past the end: an in-bounds dummy)
        const int jn = Gn >> kqs, qn = Gn & kqm;
        const int aoff = Gn < g_hi ? jn * tstep + qn * 16 : 0;
        float4 afn[NRW];
#pragma unroll
        for (int r = 0; r < NRW; ++r) afn[r] = *reinterpret_cast<const float4*>(abase[r] + aoff);
        f32x4 (&acc)[NRW][NCW] = accs[NACC == 2 ? (u & 1) : 0];
        // (NACC == 2: the x/z products go to this group's set, the y/w products of the same group to the other one - two
        // interleaved chains; the sum of the two sets is the same K sum in a different association)
        f32x4 (&acb)[NRW][NCW] = accs[NACC == 2 ? ((u & 1) ^ 1) : 0];
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].x, bw[u][c].x, acc[r][c], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int c = 0; c < NCW; ++c) acb[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].y, bw[u][c].y, acb[r][c], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int c = 0; c < NCW; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].z, bw[u][c].z, acc[r][c], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRW; ++r)
#pragma unroll
          for (int c = 0; c < NCW; ++c) acb[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r].w, bw[u][c].w, acb[r][c], 0, 0, 0);
        // refill this ring slot with group G + RC_D (the packed weights end with a zero tap: reads past the last group stay in bounds)
#pragma unroll
        for (int c = 0; c < NCW; ++c) bw[u][c] = rc_gload4(wl + c * ct_stride + (long long)(G0 + u + RC_D) * 256);
#pragma unroll
        for (int r = 0; r < NRW; ++r) af[r] = afn[r];
      }
    }
  }
  if constexpr (KW > 1) {   // partial tiles of waves 1 .. KW-1 -> LDS (behind the window); wave 0 sums in wave order
    float* const red = win + a.wr_max * LDX;
    const f32x4 part = accs[0][0][0] + accs[NACC - 1][0][0];
    if (wave > 0) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = part;
    __syncthreads();
    if (wave > 0) return;
    f32x4 sum = part;
#pragma unroll
    for (int w = 1; w < KW; ++w) sum += *reinterpret_cast<const f32x4*>(red + ((w - 1) * 64 + lane) * 4);
    accs[0][0][0] = sum; accs[NACC - 1][0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // ---- epilogue: lane (g, n) holds rows 4g .. 4g+3 of column n of each of its column tiles
  const float scale = a.out_scale;
  const int act = a.out_act;
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int col = (ct0 + c) * 16 + lr;
    if (!active || col >= a.Cout) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int rr = 0; rr < NRW; ++rr)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = rr * RC_TM + 4 * lg + e, m = m0 + r;
      if (m >= Mtot) continue;
      const int i = m / T, t = m - i * T;
      int s = 0;
#pragma unroll
      for (int q = 1; q < RC_MAXSEG; ++q) if (seg_i[q] == i) s = q;
      const int slot = seg_slot[s], pos = seg_pos[s];
      float v = ((NACC == 2 && KW == 1 ? accs[0][rr][c][e] + accs[NACC - 1][rr][c][e] : accs[0][rr][c][e]) + bias) * scale;
      if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      else if (act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
      if (a.bvec) v += a.bvec[(long long)slot * a.bvec_stride + col];
      if (a.has_res) v += rc_row(a.res, i, slot, pos, t)[col];
      if (a.has_m1) v *= *rc_row(a.m1, i, slot, pos, t);
      if (a.has_m2) v *= *rc_row(a.m2, i, slot, pos, t);
      const_cast<float*>(rc_row(a.y, i, slot, pos, t))[col] = v;
    }
  }
}

// rowlin: the 1x1 layers whose input is wider than rowconv's window (aligner ff2: 2048 -> 256).  Same tile (16 rows x 64
// columns per block, one 16-column strip per wave, fragment-major weights through an 8-deep register ring, no barrier in
// the K loop), but the rows' channels pass through LDS in chunks of 512: gather chunk, barrier, 32 K groups, barrier.  No
// left context (k = 1), no LayerNorm prologue; the epilogue is rowconv's.  31 KB of LDS: the block shares a CU with a
// vocoder block, where the split-K conv_mfma build this layer used before (126 KB) needs CUs of its own.
constexpr int RL_CW = 512, RL_LDX = RL_CW + 8, RL_D = 8;

__global__ __launch_bounds__(256, 6) void rowlin_kernel(const RowConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float win[];     // [16][RL_LDX] (dynamic: a static 33 KB would make the compiler give up the 80-VGPR bound)
  __shared__ int r_i[RC_TM], r_t[RC_TM], r_slot[RC_TM], r_pos[RC_TM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, Mtot = a.n * T, Cin = a.Cin;
  const int m0 = blockIdx.y * RC_TM, ntile = blockIdx.x;
  if (tid < RC_TM) {       // one lane per row: stream, time, slot, position (the loads of all rows fly together)
    const int m = m0 + tid, mm = m < Mtot ? m : Mtot - 1;
    const int i = mm / T, t = mm - i * T;
    const int slot = a.slots ? *(rc_gci)(a.slots + i) : i;
    r_i[tid] = i; r_t[tid] = t; r_slot[tid] = slot; r_pos[tid] = a.pos ? *(rc_gci)(a.pos + slot) : 0;
  }
  __syncthreads();
  const int KQ = Cin >> 4;
  const int ct0 = ntile * 4 + wave;
  const int lr = lane & 15, lg = lane >> 4;
  const float* const abase = win + lr * RL_LDX + 4 * lg;
  const long long ct_stride = 2ll * KQ * 256;                    // one tap + the zero tap
  const float* wl = a.w + (long long)ct0 * ct_stride + lane * 4;
  const bool active = ct0 * 16 < a.Cout_pad;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two interleaved chains (rowconv_kernel)
  float4 bw[RL_D];
  if (active) {
#pragma unroll
    for (int u = 0; u < RL_D; ++u) { bw[u] = rc_gload4(wl + (long long)u * 256); __builtin_amdgcn_sched_barrier(0); }
  }
  // a thread's share of a chunk: 16 rows x 128 float4 = 2048 float4 -> 8 per thread, fetched four at a time (the kernel is
  // bounded to 80 VGPRs and the 8-deep weight ring lives across the chunks); row u of a thread is (tid >> 7) + 2 u
  const int gw = tid >> 7, gc4 = tid & 127;
  for (int c0 = 0; c0 < Cin; c0 += RL_CW) {
    if (c0 > 0) __syncthreads();                                  // every wave is done with the previous chunk
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int w = gw + 2 * (4 * h + u); v[u] = rc_gload4(rc_row(a.x, r_i[w], r_slot[w], r_pos[w], r_t[w]) + gc4 * 4 + c0); }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int w = gw + 2 * (4 * h + u); *reinterpret_cast<float4*>(win + w * RL_LDX + gc4 * 4) = v[u]; }
    }
    __syncthreads();
    if (active) {
      const int g0 = c0 >> 4;
      float4 af = *reinterpret_cast<const float4*>(abase);
      for (int G0 = 0; G0 < RL_CW / 16; G0 += RL_D) {
#pragma unroll
        for (int u = 0; u < RL_D; ++u) {
          const int Gn = G0 + u + 1;
          const float4 afn = *reinterpret_cast<const float4*>(abase + (Gn < RL_CW / 16 ? Gn * 16 : 0));
          f32x4& p = (u & 1) ? acc1 : acc0;
          f32x4& q = (u & 1) ? acc0 : acc1;
          p = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bw[u].x, p, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bw[u].y, q, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bw[u].z, p, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bw[u].w, q, 0, 0, 0);
          bw[u] = rc_gload4(wl + (long long)(g0 + G0 + u + RL_D) * 256);      // (past the last group: the zero tap, in bounds)
          af = afn;
        }
      }
    }
  }
  if (!active) return;
  const int col = ct0 * 16 + lr;
  if (col >= a.Cout) return;
  const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = 4 * lg + e, m = m0 + r;
    if (m >= Mtot) continue;
    const int i = r_i[r], t = r_t[r], slot = r_slot[r], pos = r_pos[r];
    float v = ((acc0[e] + acc1[e]) + bias) * a.out_scale;
    if (a.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (a.out_act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    else if (a.out_act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
    if (a.bvec) v += a.bvec[(long long)slot * a.bvec_stride + col];
    if (a.has_res) v += rc_row(a.res, i, slot, pos, t)[col];
    if (a.has_m1) v *= *rc_row(a.m1, i, slot, pos, t);
    if (a.has_m2) v *= *rc_row(a.m2, i, slot, pos, t);
    const_cast<float*>(rc_row(a.y, i, slot, pos, t))[col] = v;
  }
}

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
static int rc_variant(const RowConvArgs& a) {
  static const bool no_ksplit = getenv("CONAN_RC_NOKSPLIT") != nullptr;   // developer switch
  if (a.Cin > 512) return 3;
  const int mt = (a.n * a.T + RC_TM - 1) / RC_TM;
  // a single row tile (<= 16 rows in the launch): K split over the waves of a block, one 16-column strip per block
  if (mt == 1 && ((a.ktaps * (a.Cin >> 4)) % 16) == 0 && !no_ksplit) return 2;
  // wide layers: 4 column tiles per wave (256 columns per block) keep the block count near the CU count
  static const int wide_min = getenv("CONAN_RC_WIDE_MIN") ? atoi(getenv("CONAN_RC_WIDE_MIN")) : 1024;    // developer switch
  return (a.Cout_pad >= wide_min && a.Cout_pad % 256 == 0) ? 1 : 0;
}
const char* rowconv_kernel_name(const RowConvArgs& a) {
  static const char* names[4] = {"cnk::rowconv_kernel<1, 1, 1>", "cnk::rowconv_kernel<4, 1, 1>", "cnk::rowconv_kernel<1, 1, 4>", "cnk::rowlin_kernel"};
  return names[rc_variant(a)];
}

void launch_rowconv(const RowConvArgs& ain, hipStream_t st) {
  RowConvArgs a = ain;
  const int T = a.T, M = a.n * T;
  if (M <= 0) return;
  const int halo = (a.ktaps - 1) * a.dil;
  const int ncols = a.Cout_pad;
  const int v = rc_variant(a);
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
