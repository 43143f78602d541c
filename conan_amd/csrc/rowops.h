// Device-side tile functions of the frame-rate (row-wise) operators of the Conan decoder step: rowconv / rowlin (conv and
// linear layers with an optional LayerNorm prologue, rowconv.hip), LayerNorm, embedding gather, cross attention against the
// cached prosody K/V, the uv / f0 head.  Each operator is written once, as a function of (arguments, tile index, LDS base):
//   * the stand-alone kernels (rowconv.hip, misc_kernels.hip) are thin __global__ wrappers with COH = false;
//   * the decoder megakernel (decoder_mega.hip) runs the whole step in ONE persistent launch, walking a list of these
//     operators with a grid barrier between dependent ones.  Inside one launch the per-XCD L2s are not coherent, so with
//     COH = true every activation an operator writes goes out as an agent-scope write-through store and every activation it
//     reads comes in through an sc1 load (measured: 0 stale reads and 2.3 us per barrier round at 128 workgroups; plain
//     accesses are 100 % stale across XCDs, release / acquire fences at the barrier cost 6.6 us per round -
//     tools/experiments/uncached_barrier.hip).  Weights, per-utterance caches and the frame counters are read-only inside a
//     launch and stay plain.
#pragma once
#include <type_traits>

#include "kernels.h"

namespace cnk {
namespace ro {
// K groups of wave `wave` when a strip's K loop is split over the KW waves of a block (a single row tile in the step): runs of a
// multiple of RC_D = 4 groups, the last waves' runs shorter (or empty) when NG is not a multiple of 16 - NG % 16 == 0 gives the even split
template <int KW>
__device__ __forceinline__ void rc_krange(const int NG, const int wave, int& g_lo, int& g_hi) {
  if constexpr (KW == 1) { g_lo = 0; g_hi = NG; }
  else {
    const int per = ((NG + 4 * KW - 1) / (4 * KW)) * 4;
    g_lo = per * wave < NG ? per * wave : NG;
    g_hi = g_lo + per < NG ? g_lo + per : NG;
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gcf4;
typedef f32x4 __attribute__((address_space(1)))* gf4;
typedef const float __attribute__((address_space(1)))* gcf1;
typedef float __attribute__((address_space(1)))* gf1;
typedef const int __attribute__((address_space(1)))* gci;
typedef unsigned long long u64;

// activation accesses: COH = 0 plain (stand-alone kernels); 1 agent-scope (megakernel: write-through stores, L1-bypassing loads);
// 2 the megakernel with every workgroup on ONE XCD (decoder_mega.hip, xcd mode): that XCD's L2 is the point of coherence, so stores
// are plain (L1 writes through to L2, the line stays there) and only the loads bypass the reading CU's L1 (sc1: L2-served)
typedef const u64 __attribute__((address_space(1)))* gcu64;
typedef u64 __attribute__((address_space(1)))* gu64;
template <int COH> __device__ __forceinline__ float4 ld4(const float* p) {
  if constexpr (COH != 0) {
    const u64 a = __hip_atomic_load((gcu64)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 b = __hip_atomic_load((gcu64)(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
  } else {
    const f32x4 v = *(gcf4)(p);
    return make_float4(v[0], v[1], v[2], v[3]);
  }
}
template <int COH> __device__ __forceinline__ float ld1(const float* p) {
  if constexpr (COH != 0) return __hip_atomic_load((gcf1)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *(gcf1)(p);
}
// (COH = 3: the megakernel's groups decide at run time - l2 != 0: every member of the group sits on one XCD, stores as in mode 2)
template <int COH> __device__ __forceinline__ void st4(float* p, const float4 v, const int l2 = 0) {
  if constexpr (COH == 3) { if (l2) st4<2>(p, v); else st4<1>(p, v); }
  else if constexpr (COH == 1) {
    __hip_atomic_store((gu64)(p), (u64)__float_as_uint(v.x) | ((u64)__float_as_uint(v.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu64)(p) + 1, (u64)__float_as_uint(v.z) | ((u64)__float_as_uint(v.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *(gf4)(p) = (f32x4){v.x, v.y, v.z, v.w};
  }
}
template <int COH> __device__ __forceinline__ void st1(float* p, const float v, const int l2 = 0) {
  if constexpr (COH == 3) { if (l2) st1<2>(p, v); else st1<1>(p, v); }
  else if constexpr (COH == 1) __hip_atomic_store((gf1)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *(gf1)(p) = v;
}
// read-only data (weights, biases, per-utterance caches, slot tables, frame counters)
__device__ __forceinline__ float4 ldw4(const float* p) { const f32x4 v = *(gcf4)(p); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float ldw1(const float* p) { return *(gcf1)(p); }
__device__ __forceinline__ int ldi(const int* p) { return *(gci)(p); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// row t of (batch index i / slot) of a tensor; `pos` is the slot's frame counter (ring tensors)
// (TR: TRef, or TRef in the constant address space - the megakernel reads operator arguments where they are used, through
// the scalar cache, instead of holding ~110 words of them in registers)
template <class TR>
__device__ __forceinline__ float* row(const TR& r, int i, int slot, int pos, int t) {
  if (r.mode == 0) return r.base + (long long)slot * r.slot_stride + (long long)(((unsigned)pos * (unsigned)r.rate + (unsigned)(r.off + t)) & (unsigned)r.lmask) * r.C;
  return r.base + (long long)i * r.slot_stride + (long long)(r.off + t) * r.C;
}

// ------------------------------------------------------------------------------------------------ rowconv
// (see rowconv.hip for the design: 16 output rows x 64 * NCW columns per tile, LDS window with the LayerNorm applied in
// place, fragment-major weights through a register ring, no barrier in the K loop)
constexpr int RC_TM = 16;          // output rows per MFMA row tile
constexpr int RC_MAXSEG = 8;       // streams a tile may touch (T >= 2)

template <int NCW, int NRW, int KW, bool COH, class A>
__device__ __forceinline__ void rowconv_tile(const A& a, const int bx, const int by, float* __restrict__ lds) {
  static_assert(KW == 1 || (NCW == 1 && NRW == 1), "K split: one tile per wave");
  constexpr int RC_D = (KW > 1) ? 4 : (NCW * NRW == 1) ? 8 : 4;
  constexpr int TMB = RC_TM * NRW;                     // output rows per block
  int tid = threadIdx.x;
  // (megakernel: the tile loop around this function must not hoist per-lane address arithmetic out of it - the values would
  // stay live across the whole tile and the 80-register bound is tight)
  if constexpr (COH) asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, n = a.n, Mtot = n * T;
  // bx = column strip (fastest), by = row tile: workgroups are dealt round-robin over the 8 XCDs, so with 8 (or 4) strips
  // an XCD always draws the same strip(s) and its L2 fetches their weights once for all row tiles
  const int m0 = by * TMB;
  const int ntile = bx;                               // 64*NCW output columns
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
    const int slot = start ? (a.slots ? ldi(a.slots + i) : i) : 0;
    const int pos = start ? (a.pos ? ldi(a.pos + slot) : 0) : 0;
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
  constexpr int GB = COH ? 4 : 8;                       // 16-byte loads per thread in flight
  for (int e0 = 0; e0 < total; e0 += 256 * GB) {
    float4 v[GB];
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int e = e0 + tid + 256 * u;
      const int w = e < total ? e / C4 : 0, c4 = e < total ? e - w * C4 : 0;
      const int sg = wseg(w);
      const int tau = seg_t0[sg] - halo + (w - seg_off[sg]);   // time index within this step (negative: earlier steps)
      const float* src = (a.ln && tau >= 0) ? row(a.x, seg_i[sg], seg_slot[sg], seg_pos[sg], tau)        // new rows: raw layer input
                                            : row(a.ln ? a.hist : a.x, seg_i[sg], seg_slot[sg], seg_pos[sg], tau);   // ring (history, or plain input)
      v[u] = ld4<COH>(src + c4 * 4);
    }
    const float isl = a.in_lrelu ? a.in_slope : 1.0f;      // LeakyReLU on the way in (HiFi-GAN resblock convs)
#pragma unroll
    for (int u = 0; u < GB; ++u) {
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
      float* wrow = win + (inw ? w : 0) * LDX;
      float4 v[8];                                          // Cin <= 512: 8 float4 per lane
      float sum = 0.f, sa = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = (l16 + 16 * q) * 4;
        v[q] = (live && c < Cin) ? *reinterpret_cast<const float4*>(wrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        if (a.has_lnmask) mk = ld1<COH>(row(a.lnmask, seg_i[sg], seg_slot[sg], seg_pos[sg], tau));
        float* hrow = row(a.hist, seg_i[sg], seg_slot[sg], seg_pos[sg], tau);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = (l16 + 16 * q) * 4;
          if (c < Cin) {
            const float4 g = ldw4(a.gamma + c), bb = ldw4(a.beta + c);
            const float4 o = make_float4(((v[q].x - mean) * rstd * g.x + bb.x) * mk, ((v[q].y - mean) * rstd * g.y + bb.y) * mk,
                                         ((v[q].z - mean) * rstd * g.z + bb.z) * mk, ((v[q].w - mean) * rstd * g.w + bb.w) * mk);
            *reinterpret_cast<float4*>(wrow + c) = o;
            if (ntile == 0) st4<COH>(hrow + c, o);
          }
        }
        if (a.has_mask_out && ntile == 0 && l16 == 0) st1<COH>(row(a.mask_out, seg_i[sg], seg_slot[sg], seg_pos[sg], tau), sa > 0.f ? 1.f : 0.f);
      }
    }
    __syncthreads();
  }
  // ---- K loop: wave w owns column tiles ct0 .. ct0 + NCW - 1 of this block's strip
  const int KQ = Cin >> 4;                              // 16-deep K groups per tap (power of two)
  const int NG = k * KQ;
  const int ct0 = KW > 1 ? ntile : (ntile * 4 + wave) * NCW;
  int g_lo, g_hi;
  rc_krange<KW>(NG, wave, g_lo, g_hi);
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
      for (int c = 0; c < NCW; ++c) bw[u][c] = ldw4(wl + c * ct_stride + (long long)(g_lo + u) * 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int kqm = KQ - 1, kqs = 31 - __builtin_clz(KQ);
    const int tstep = d * LDX;
    float4 af[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) af[r] = *reinterpret_cast<const float4*>(abase[r] + (g_lo < g_hi ? (g_lo >> kqs) * tstep + (g_lo & kqm) * 16 : 0));
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
        for (int c = 0; c < NCW; ++c) bw[u][c] = ldw4(wl + c * ct_stride + (long long)(G0 + u + RC_D) * 256);
#pragma unroll
        for (int r = 0; r < NRW; ++r) af[r] = afn[r];
      }
    }
  }
  bool fin = true;          // this wave writes outputs
  if constexpr (KW > 1) {   // partial tiles of waves 1 .. KW-1 -> LDS (behind the window); wave 0 sums in wave order
    float* const red = win + a.wr_max * LDX;
    const f32x4 part = accs[0][0][0] + accs[NACC - 1][0][0];
    if (wave > 0) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = part;
    __syncthreads();
    fin = wave == 0;
    if (fin) {
      f32x4 sum = part;
#pragma unroll
      for (int w = 1; w < KW; ++w) sum += *reinterpret_cast<const f32x4*>(red + ((w - 1) * 64 + lane) * 4);
      accs[0][0][0] = sum; accs[NACC - 1][0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- epilogue: lane (g, n) holds rows 4g .. 4g+3 of column n of each of its column tiles
  const float scale = a.out_scale;
  const int act = a.out_act;
  if (fin) {
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int col = (ct0 + c) * 16 + lr;
    if (!active || col >= a.Cout) continue;
    const float bias = a.bias ? ldw1(a.bias + col) : 0.f;
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
      if (a.bvec) v += ldw1(a.bvec + (long long)slot * a.bvec_stride + col);
      if (a.has_res) v += ld1<COH>(row(a.res, i, slot, pos, t) + col);
      if (a.has_m1) v *= ld1<COH>(row(a.m1, i, slot, pos, t));
      if (a.has_m2) v *= ld1<COH>(row(a.m2, i, slot, pos, t));
      st1<COH>(row(a.y, i, slot, pos, t) + col, v);
    }
  }
  }
}

// rowlin: the 1x1 layers whose input is wider than rowconv's window (aligner ff2: 2048 -> 256).  Same tile (16 rows x 64
// columns per block, one 16-column strip per wave, fragment-major weights through an 8-deep register ring, no barrier in
// the K loop), but the rows' channels pass through LDS in chunks of 512: gather chunk, barrier, 32 K groups, barrier.  No
// left context (k = 1), no LayerNorm prologue; the epilogue is rowconv's.
constexpr int RL_CW = 512, RL_LDX = RL_CW + 8, RL_D = 8;

template <bool COH, class A>
__device__ __forceinline__ void rowlin_tile(const A& a, const int bx, const int by, float* __restrict__ win) {   // win: [16][RL_LDX]
  __shared__ int r_i[RC_TM], r_t[RC_TM], r_slot[RC_TM], r_pos[RC_TM];
  int tid = threadIdx.x;
  if constexpr (COH) asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, Mtot = a.n * T, Cin = a.Cin;
  const int m0 = by * RC_TM, ntile = bx;
  if (tid < RC_TM) {       // one lane per row: stream, time, slot, position (the loads of all rows fly together)
    const int m = m0 + tid, mm = m < Mtot ? m : Mtot - 1;
    const int i = mm / T, t = mm - i * T;
    const int slot = a.slots ? ldi(a.slots + i) : i;
    r_i[tid] = i; r_t[tid] = t; r_slot[tid] = slot; r_pos[tid] = a.pos ? ldi(a.pos + slot) : 0;
  }
  __syncthreads();
  const int KQ = Cin >> 4;
  const int ct0 = ntile * 4 + wave;
  const int lr = lane & 15, lg = lane >> 4;
  const float* const abase = win + lr * RL_LDX + 4 * lg;
  const long long ct_stride = 2ll * KQ * 256;                    // one tap + the zero tap
  const float* wl = a.w + (long long)ct0 * ct_stride + lane * 4;
  const bool active = ct0 * 16 < a.Cout_pad;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two interleaved chains (rowconv_tile)
  float4 bw[RL_D];
  if (active) {
#pragma unroll
    for (int u = 0; u < RL_D; ++u) { bw[u] = ldw4(wl + (long long)u * 256); __builtin_amdgcn_sched_barrier(0); }
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
      for (int u = 0; u < 4; ++u) { const int w = gw + 2 * (4 * h + u); v[u] = ld4<COH>(row(a.x, r_i[w], r_slot[w], r_pos[w], r_t[w]) + gc4 * 4 + c0); }
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
          bw[u] = ldw4(wl + (long long)(g0 + G0 + u + RL_D) * 256);      // (past the last group: the zero tap, in bounds)
          af = afn;
        }
      }
    }
  }
  const int col = ct0 * 16 + lr;
  if (active && col < a.Cout) {
    const float bias = a.bias ? ldw1(a.bias + col) : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * lg + e, m = m0 + r;
      if (m >= Mtot) continue;
      const int i = r_i[r], t = r_t[r], slot = r_slot[r], pos = r_pos[r];
      float v = ((acc0[e] + acc1[e]) + bias) * a.out_scale;
      if (a.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (a.out_act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      else if (a.out_act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
      if (a.bvec) v += ldw1(a.bvec + (long long)slot * a.bvec_stride + col);
      if (a.has_res) v += ld1<COH>(row(a.res, i, slot, pos, t) + col);
      if (a.has_m1) v *= ld1<COH>(row(a.m1, i, slot, pos, t));
      if (a.has_m2) v *= ld1<COH>(row(a.m2, i, slot, pos, t));
      st1<COH>(row(a.y, i, slot, pos, t) + col, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row, 4 rows per tile:  y = (LN(x (+ pre)) * gamma + beta) * m1 * m2 (+ post), optional mask_out = (sum |x| > 0)
constexpr int LN_MAXV = 8;  // C <= 512
template <bool COH, class A>
__device__ __forceinline__ void layernorm_tile(const A& a, const int bx) {
  const int lane = threadIdx.x & 63;
  const int m = bx * 4 + (threadIdx.x >> 6);
  if (m >= a.n * a.T) return;
  const int i = m / a.T, t = m - i * a.T;
  if (a.lens && t >= ldi(a.lens + i)) return;
  const int slot = a.slots ? ldi(a.slots + i) : i;
  const int pos = a.pos ? ldi(a.pos + slot) : 0;
  const float* x = row(a.x, i, slot, pos, t);
  const float* pre = a.has_pre ? row(a.pre, i, slot, pos, t) : nullptr;
  float v[LN_MAXV];
  float s = 0.f, sa = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int c = lane + 64 * k;
    float u = 0.f;
    if (c < a.C) { u = ld1<COH>(x + c); sa += fabsf(u); if (pre) u += ld1<COH>(pre + c); }
    v[k] = u; s += u;
  }
  s = wave_sum(s);
  const float mean = s / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) { int c = lane + 64 * k; if (c < a.C) { float d = v[k] - mean; q += d * d; } }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
  float mk = 1.f;
  if (a.has_m1) mk *= ld1<COH>(row(a.m1, i, slot, pos, t));
  if (a.has_m2) mk *= ld1<COH>(row(a.m2, i, slot, pos, t));
  if (a.has_mask_out) {
    sa = wave_sum(sa);
    if (lane == 0) st1<COH>(row(a.mask_out, i, slot, pos, t), sa > 0.f ? 1.f : 0.f);
  }
  float* y = row(a.y, i, slot, pos, t);
  const float* post = a.has_post ? row(a.post, i, slot, pos, t) : nullptr;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int c = lane + 64 * k;
    if (c < a.C) {
      float o = (v[k] - mean) * rstd * ldw1(a.gamma + c) + ldw1(a.beta + c);
      if (a.has_m1 | a.has_m2) o *= mk;
      if (post) o += ld1<COH>(post + c);
      st1<COH>(y + c, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------ embedding gather
// one row per tile: y[i][t][:] = table[idx[i][t]][:]
template <bool COH, class A>
__device__ __forceinline__ void embed_tile(const A& a, const int bx) {
  const int m = bx;
  const int i = m / a.T, t = m - i * a.T;
  const int slot = a.slots ? ldi(a.slots + i) : i;
  const int pos = a.pos ? ldi(a.pos + slot) : 0;
  int id = ldi(a.idx + m);
  id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
  float* y = row(a.y, i, slot, pos, t);
  const float* e = a.table + (long long)id * a.C;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) st1<COH>(y + c, ldw1(e + c));
}

// ------------------------------------------------------------------------------------------------ cross attention
// tile = one query row (slot, t); wave h = head h.  Scores: lane-per-key dot over dh dims with q broadcast from LDS;
// softmax by wave reductions; output: lane-per-dim sum over keys.  LDS: sq[1024] | sp[XA_MAX_H][XA_MAX_S] (12 KB).
constexpr int XA_MAX_S = 512;
constexpr int XA_MAX_H = 4;
constexpr int XA_LDS_FLOATS = 1024 + XA_MAX_H * XA_MAX_S;
template <bool COH, class A>
__device__ __forceinline__ void xattn_tile(const A& a, const int bx, float* __restrict__ lds) {
  float* const sq = lds;
  float (*sp)[XA_MAX_S] = reinterpret_cast<float (*)[XA_MAX_S]>(lds + 1024);
  const int m = bx;
  const int i = m / a.T, t = m - i * a.T;
  const int slot = ldi(a.slots + i);
  const int pos = a.pos ? ldi(a.pos + slot) : 0;
  const int S = ldi(a.slen + slot);
  const int dh = a.E / a.H;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* q = row(a.q, i, slot, pos, t);
  for (int c = threadIdx.x; c < a.E; c += blockDim.x) sq[c] = ld1<COH>(q + c);
  __syncthreads();
  const float* kv = a.kv + (long long)slot * a.kv_slot_stride;
  const float* km = a.kmask + (long long)slot * a.S_max;
  if (h < a.H) {
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) {
      const float* kp = kv + (long long)s * 2 * a.E + h * dh;
      const float4* qp = reinterpret_cast<const float4*>(sq + h * dh);
      float sc = 0.f;
      for (int d = 0; d < dh / 4; ++d) { float4 k4 = ldw4(kp + 4 * d), q4 = qp[d]; sc += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w; }
      sc += ldw1(km + s);
      sp[h][s] = sc;
      mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) { float p = expf(sp[h][s] - mx); sp[h][s] = p; sum += p; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int s = lane; s < S; s += 64) sp[h][s] *= inv;
  }
  __syncthreads();
  if (h < a.H) {
    float* o = row(a.out, i, slot, pos, t);
    for (int d = lane; d < dh; d += 64) {
      float acc = 0.f;
      const float* vp = kv + a.E + h * dh + d;
      for (int s = 0; s < S; ++s) acc += sp[h][s] * ldw1(vp + (long long)s * 2 * a.E);
      st1<COH>(o + h * dh + d, acc);
    }
  }
  if (a.attn_avg) {      // (a tap for the caller: read after the launch, plain stores)
    float* w = a.attn_avg + ((long long)i * a.T + t) * a.S_max;
    for (int s = threadIdx.x; s < a.S_max; s += blockDim.x) {
      float v = 0.f;
      if (s < S) { for (int hh = 0; hh < a.H; ++hh) v += sp[hh][s]; v /= (float)a.H; }
      w[s] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ uv / f0 head
// PitchPredictor tail (nar_tts_modules.py:141-146) + add_orig_pitch (Conan.py:330-340) + denorm_f0 /
// f0_to_coarse (pitch/utils.py:71-82, :17-28) + pitch_embed add (Conan.py:301, :181); fp32 op order kept.  4 rows per tile.
template <bool COH, class A>
__device__ __forceinline__ void pitch_head_tile(const A& a, const float mel_min, const float mel_den, const int bx) {
  const int lane = threadIdx.x & 63;
  const int m = bx * 4 + (threadIdx.x >> 6);
  if (m >= a.n * a.T) return;
  const int i = m / a.T, t = m - i * a.T;
  const int slot = ldi(a.slots + i);
  const int pos = a.pos ? ldi(a.pos + slot) : 0;
  const float* x = row(a.h, i, slot, pos, t);
  float v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { int c = lane + 64 * k; v[k] = c < a.Cp ? ld1<COH>(x + c) : 0.f; s += v[k]; }
  s = wave_sum(s);
  const float mean = s / (float)a.Cp;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { int c = lane + 64 * k; if (c < a.Cp) { float d = v[k] - mean; q += d * d; } }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)a.Cp + 1e-5f);
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = lane + 64 * k;
    if (c < a.Cp) { float y = (v[k] - mean) * rstd * ldw1(a.gamma + c) + ldw1(a.beta + c); d0 += y * ldw1(a.w + c); d1 += y * ldw1(a.w + a.Cp + c); }
  }
  d0 = wave_sum(d0) + ldw1(a.b);
  d1 = wave_sum(d1) + ldw1(a.b + 1);
  const int code = ldi(a.codes + m);
  const bool uv = (d0 > 0.f) || (code == a.silent_token);
  float f0 = exp2f(d1);
  f0 = fminf(fmaxf(f0, 50.f), 900.f);
  if (uv) f0 = 0.f;
  float fm = 1127.f * logf(1.f + f0 / 700.f);
  if (fm > 0.f) fm = (fm - mel_min) * 254.f / mel_den + 1.f;
  if (fm <= 1.f) fm = 1.f;
  if (fm > 255.f) fm = 255.f;
  const int bin = (int)(fm + 0.5f);
  if (lane == 0) {
    if (a.uv_pred) { a.uv_pred[(long long)m * 2] = d0; a.uv_pred[(long long)m * 2 + 1] = d1; }
    if (a.f0) a.f0[m] = f0;
    if (a.bins) a.bins[m] = bin;
  }
  const float* pi = row(a.pitch_inp, i, slot, pos, t);
  float* di = row(a.dec_inp, i, slot, pos, t);
  const float* pe = a.pitch_embed + (long long)bin * a.E;
  for (int c = lane; c < a.E; c += 64) st1<COH>(di + c, ld1<COH>(pi + c) + ldw1(pe + c));
}

// ================================================================================================ decoder megakernel
// The same operators restructured for decoder_mega.hip: a JOB is one 16-row tile taken through the whole operator list by a
// group of workgroups.  The tile's row table (stream, frame, slot, frame counter per row) is built once per job; a conv /
// linear operator is split into mg_stage (gather the input window, LayerNorm in place - once per member and operator) and
// mg_strip (K loop + epilogue of one output strip); row-wise operators take (row of the tile) instead of (tile).  All
// activation accesses are agent-scope (COH).
struct RowTab {
  int nvalid, nseg;                       // rows of the tile inside the step, stream segments among them
  int row_seg[RC_TM];                     // segment of tile row r
  int seg_i[RC_MAXSEG + 1], seg_t0[RC_MAXSEG + 1], seg_r0[RC_MAXSEG + 1], seg_slot[RC_MAXSEG + 1], seg_pos[RC_MAXSEG + 1];
  int tab[RC_TM];                         // per operator: window row of tile row r at tap 0
  int l2;                                 // the group's members sit on ONE XCD (decoder_mega.hip: activations stay in its L2 - plain stores, flag barriers)
};
constexpr int ROWTAB_FLOATS = 96;
static_assert(sizeof(RowTab) <= ROWTAB_FLOATS * 4, "row table");

// rows of one stream are consecutive; a segment starts where the stream changes (wave 0; callers synchronise around it)
__device__ __forceinline__ void rowtab_setup(RowTab& tb, const int* slots, const int* pos, const int n, const int T, const int m0) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int lane = tid;
    if (lane <= RC_MAXSEG) { tb.seg_i[lane] = -1; tb.seg_r0[lane] = 0x3fffffff; tb.seg_t0[lane] = 0; tb.seg_slot[lane] = 0; tb.seg_pos[lane] = 0; }
    const int Mtot = n * T;
    const int r = lane, m = m0 + r;
    const bool valid = r < RC_TM && m < Mtot;
    const int i = valid ? m / T : -1, t = valid ? m - i * T : 0;
    const int iprev = __shfl_up(i, 1);
    const bool start = valid && (r == 0 || i != iprev);
    const unsigned long long sbm = __ballot(start);
    const int sidx = __popcll(sbm & ((2ull << r) - 1ull)) - 1;
    const int slot = start ? (slots ? ldi(slots + i) : i) : 0;
    const int ps = start ? (pos ? ldi(pos + slot) : 0) : 0;
    if (r < RC_TM) tb.row_seg[r] = valid ? sidx : 0;
    if (start) { tb.seg_i[sidx] = i; tb.seg_t0[sidx] = t; tb.seg_r0[sidx] = r; tb.seg_slot[sidx] = slot; tb.seg_pos[sidx] = ps; }
    if (lane == 0) { tb.nvalid = Mtot - m0 < RC_TM ? (Mtot - m0 > 0 ? Mtot - m0 : 0) : RC_TM; tb.nseg = __popcll(sbm); }
  }
}
// stream index / frame / slot / frame counter of tile row r
struct RowId { int i, t, slot, pos; };
__device__ __forceinline__ RowId row_id(const RowTab& tb, const int r) {
  const int s = tb.row_seg[r];
  return RowId{tb.seg_i[s], tb.seg_t0[s] + (r - tb.seg_r0[s]), tb.seg_slot[s], tb.seg_pos[s]};
}

// A copy of an argument sub-struct out of the program (constant address space) - ONE clause of scalar loads.  The operators used to
// read every field where they used it: the compiler then emits one scalar load + s_waitcnt per use, in front of branches - the window
// gather of a 256 -> 256 1x1 conv was a chain of 21 dependent scalar / LDS round trips (2.5 us) in front of its single load per thread.
template <class T, class S>
__device__ __forceinline__ T as_copy(const S& src) {
  static_assert(sizeof(T) % 4 == 0 && sizeof(T) == sizeof(S), "argument structs are made of 32-bit words");
  union U { T v; int w[sizeof(T) / 4]; __device__ U() {} } u;
  const int __attribute__((address_space(4)))* p = (const int __attribute__((address_space(4)))*)(&src);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) u.w[i] = p[i];
  return u.v;
}

// gather the tile's input window of operator `a` into LDS (win = lds: [wr_max][Cin + 8]), LayerNorm of the new rows in place;
// `first` (the member that owns strip 0) appends the normalised rows to the layer's ring and writes the block mask
// (sb of nact: this member's index among the members that stage this operator - the normalised rows they all compute are
// appended to the layer's ring / the block mask is written by member row % nact, so that no single member carries the stores)
struct NoWarm { __device__ __forceinline__ void operator()() const {} };
// `warm`: called once, right BEHIND the first pass's loads and in front of their use - L2 warm-up loads of the operator's weights.
// (Vector-memory loads return in order: warm-up loads issued in front of the gather made its L2 hits wait for their misses.)
template <bool WIDE = false, int CM = 1, class A, class WF = NoWarm>
__device__ __forceinline__ void mg_stage(const A& a, RowTab& tb, float* __restrict__ win, const int sb, const int nact, WF&& warm = WF{}) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));             // (no hoisting of per-lane arithmetic out of this function: the 80-register bound is tight)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = a.Cin, LDX = Cin + 8, C4 = Cin >> 2;
  const int k = a.ktaps, d = a.dil, halo = (k - 1) * d;
  // (every argument the gather needs, in one batch of scalar loads)
  const TRef X = as_copy<TRef>(a.x), HS = as_copy<TRef>(a.hist);
  const int a_ln = a.ln, a_xparts = a.xparts, a_in_lrelu = a.in_lrelu;
  const float a_in_slope = a.in_slope;
  __syncthreads();                          // every wave is done with the previous operator's window
  if (tid < RC_TM) tb.tab[tid] = tid < tb.nvalid ? tid + tb.row_seg[tid] * halo : 0;
  const int WR = tb.nvalid + tb.nseg * halo;
  auto wseg = [&](int w) __attribute__((always_inline)) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < RC_MAXSEG; ++q) s += (w >= tb.seg_r0[q] + q * halo) ? 1 : 0;      // seg_r0 of unused segments is huge
    return s;
  };
  // Gather: 8 threads per window row, 32 rows per pass - a thread works out ITS row's source once (segment, time, ring row) and
  // then fetches the row's 16-byte columns j, j + 8, .. (8 threads = one 128-byte line per step); every load of a pass is in
  // flight before the first is consumed.  (Round 4 worked the source out per 16-byte element: ~8 x the address arithmetic, with
  // the segment table's LDS reads in front of every load - 3.7 us of an operator's 7 at one stream.)
  constexpr int TPR = 8;
  const int wsub = tid >> 3, j8 = tid & 7;
  const int nld_all = C4 >> 3;                           // loads per thread and row: 2, 4, 8 or 16 (Cin = 64 .. 512), at most 8 at a time
  // (the loads of a pass are UNCONDITIONAL, straight-line code: rows beyond the window / rows that come from partial tensors fetch a
  // valid row and drop it.  Behind a branch or an exec mask hipcc waits for every earlier load before it issues the next pair - 16
  // serialised L2 round trips, 1.8 us, for a 256-channel row)
  auto pass = [&](auto nld_c, const int w0, const int u0) __attribute__((always_inline)) {
    constexpr int NLD = decltype(nld_c)::value;
    const int w = w0 + wsub;
    const bool inw = w < WR;
    const int wc = inw ? w : 0;
    const int sg = wseg(wc);
    const int tau = tb.seg_t0[sg] - halo + (wc - (tb.seg_r0[sg] + sg * halo));   // time index within this step (negative: earlier steps)
    const bool part = a_xparts > 0 && tau >= 0;          // (summed from the partial tensors below)
    const int s_i = tb.seg_i[sg], s_slot = tb.seg_slot[sg], s_pos = tb.seg_pos[sg];
    const float* const src_x = row(X, s_i, s_slot, s_pos, tau);
    const float* const src_h = row(HS, s_i, s_slot, s_pos, tau);
    const float* src = ((a_ln && tau < 0) ? src_h : src_x) + (j8 + 8 * u0) * 4;      // (ln: rows of earlier steps come out of the layer's ring normalised)
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = ld4<CM>(src + 32 * u);
    if (w0 == 0 && u0 == 0) warm();
    const float isl = a_in_lrelu ? a_in_slope : 1.0f;
    float* const dst = win + wc * LDX + (j8 + 8 * u0) * 4;
    if (inw && !part) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        float4 q = v[u];
        q.x *= q.x > 0.f ? 1.0f : isl; q.y *= q.y > 0.f ? 1.0f : isl; q.z *= q.z > 0.f ? 1.0f : isl; q.w *= q.w > 0.f ? 1.0f : isl;
        *reinterpret_cast<float4*>(dst + 32 * u) = q;
      }
    }
  };
  for (int w0 = 0; w0 < WR; w0 += 256 / TPR) {
    if (nld_all >= 8) { for (int u0 = 0; u0 < nld_all; u0 += 8) pass(std::integral_constant<int, 8>{}, w0, u0); }
    else if (nld_all == 4) pass(std::integral_constant<int, 4>{}, w0, 0);
    else pass(std::integral_constant<int, 2>{}, w0, 0);
  }
  if (a_xparts > 0) {
    // the NEW rows of x are the sum of the group members' partial tensors (+ bias, + residual, x masks), summed in member order.
    // Every load of a 16-byte column group - its partials, the residual, the masks - is issued before the first is consumed: one
    // L2 round trip per group.
    constexpr int EB = 1;      // (two groups per thread in flight cost the 128-register build 15 spilled registers: one)
    const int nq = tb.nvalid * C4;
#pragma unroll 1
    for (int e0 = tid; e0 < nq; e0 += 256 * EB) {
      float4 pv[EB][8], xr[EB];
      float mk[EB];
      RowId id[EB];
      int rr[EB], cc[EB];
#pragma unroll
      for (int b = 0; b < EB; ++b) {
        const int e = e0 + 256 * b, ev = e < nq ? e : e0;
        rr[b] = ev / C4; cc[b] = ev - rr[b] * C4;
        id[b] = row_id(tb, rr[b]);
        const long long off = (long long)(id[b].i * a.T + id[b].t) * a.xp_ld + cc[b] * 4;
#pragma unroll
        for (int p = 0; p < 8; ++p) pv[b][p] = p < a.xparts ? ld4<CM>(a.xp + p * a.xp_stride + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        xr[b] = a.has_xres ? ld4<CM>(row(a.xres, id[b].i, id[b].slot, id[b].pos, id[b].t) + cc[b] * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        mk[b] = 1.f;
        if (a.has_xm1) mk[b] = ld1<CM>(row(a.xm1, id[b].i, id[b].slot, id[b].pos, id[b].t));
        if (a.has_xm2) mk[b] *= ld1<CM>(row(a.xm2, id[b].i, id[b].slot, id[b].pos, id[b].t));
      }
#pragma unroll
      for (int b = 0; b < EB; ++b) {
        const int e = e0 + 256 * b;
        float4 acc = pv[b][0];
#pragma unroll
        for (int p = 1; p < 8; ++p) if (p < a.xparts) { acc.x += pv[b][p].x; acc.y += pv[b][p].y; acc.z += pv[b][p].z; acc.w += pv[b][p].w; }
        if (a.xparts > 8) {      // (more than 8 members - 16-member groups, the 32 virtual members of xcd mode: eight at a time, in member order)
          const long long off = (long long)(id[b].i * a.T + id[b].t) * a.xp_ld + cc[b] * 4;
          for (int p0 = 8; p0 < a.xparts; p0 += 8) {
            float4 q8[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) q8[p] = p0 + p < a.xparts ? ld4<CM>(a.xp + (p0 + p) * a.xp_stride + off) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < 8; ++p) if (p0 + p < a.xparts) { acc.x += q8[p].x; acc.y += q8[p].y; acc.z += q8[p].z; acc.w += q8[p].w; }
          }
        }
        if (a.xbias) { const float4 q = ldw4(a.xbias + cc[b] * 4); acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
        if (a.has_xres) { acc.x += xr[b].x; acc.y += xr[b].y; acc.z += xr[b].z; acc.w += xr[b].w; }
        if (a.has_xm1 | a.has_xm2) { acc.x *= mk[b]; acc.y *= mk[b]; acc.z *= mk[b]; acc.w *= mk[b]; }
        if (e < nq) {
          *reinterpret_cast<float4*>(win + (rr[b] + (tb.row_seg[rr[b]] + 1) * halo) * LDX + cc[b] * 4) = acc;
          // (the tensor itself, for the operator behind this one that adds it as its residual)
          if (a.xstore && (rr[b] % nact) == sb) st4<CM>(row(a.x, id[b].i, id[b].slot, id[b].pos, id[b].t) + cc[b] * 4, acc, tb.l2);
        }
      }
    }
  }
  __syncthreads();
  if (a_ln) {
    // only the tile's own rows are new (rows of earlier steps come out of the ring normalised): tile row r = window row tab[r],
    // 16 lanes per row, the 16 rows in one pass over the 4 waves
    const float* const a_gamma = a.gamma; const float* const a_beta = a.beta;
    const float a_eps = a.eps;
    const int a_has_lnmask = a.has_lnmask, a_has_mask_out = a.has_mask_out;
    const TRef HL = as_copy<TRef>(a.hist);
    const int sub = lane >> 4, l16 = lane & 15;
    const int r = wave * 4 + sub;
    const bool live = r < tb.nvalid;
    const RowId id = row_id(tb, live ? r : 0);
    float* wrow = win + (live ? r + (tb.row_seg[r] + 1) * halo : 0) * LDX;      // (tab[r] is the row's OLDEST tap; its own row is halo rows on)
    // (NQ = Cin / 64 16-byte columns per lane; gamma / beta / the mask are fetched unconditionally and FIRST - inside `if (c < Cin)`
    // branches hipcc serialises them with full waits, 2.3 us of an operator's 10)
    auto norm = [&](auto nq_c) __attribute__((always_inline)) {
      constexpr int NQ = decltype(nq_c)::value;
      float4 g[NQ], bb[NQ], v[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { g[q] = ldw4(a_gamma + (l16 + 16 * q) * 4); bb[q] = ldw4(a_beta + (l16 + 16 * q) * 4); }
      float mk = 1.f;
      {
        const float* pm = a_gamma;
        if (a_has_lnmask) { const TRef LM = as_copy<TRef>(a.lnmask); pm = row(LM, id.i, id.slot, id.pos, id.t); }
        const float mv = ld1<CM>(pm);
        if (a_has_lnmask) mk = mv;
      }
      float sum = 0.f, sa = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v[q] = live ? *reinterpret_cast<const float4*>(wrow + (l16 + 16 * q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        sa += (fabsf(v[q].x) + fabsf(v[q].y)) + (fabsf(v[q].z) + fabsf(v[q].w));
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); sa += __shfl_xor(sa, o); }
      const float mean = sum / (float)Cin;
      float var = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const float d0 = v[q].x - mean, d1 = v[q].y - mean, d2 = v[q].z - mean, d3 = v[q].w - mean; var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3); }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) var += __shfl_xor(var, o);
      const float rstd = 1.0f / sqrtf(var / (float)Cin + a_eps);
      if (live) {
        const bool mine = (r % nact) == sb;
        float* hrow = row(HL, id.i, id.slot, id.pos, id.t);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int c = (l16 + 16 * q) * 4;
          const float4 o = make_float4(((v[q].x - mean) * rstd * g[q].x + bb[q].x) * mk, ((v[q].y - mean) * rstd * g[q].y + bb[q].y) * mk,
                                       ((v[q].z - mean) * rstd * g[q].z + bb[q].z) * mk, ((v[q].w - mean) * rstd * g[q].w + bb[q].w) * mk);
          *reinterpret_cast<float4*>(wrow + c) = o;
          if (mine) st4<CM>(hrow + c, o, tb.l2);
        }
        if (a_has_mask_out && mine && l16 == 0) { const TRef MO = as_copy<TRef>(a.mask_out); st1<CM>(row(MO, id.i, id.slot, id.pos, id.t), sa > 0.f ? 1.f : 0.f, tb.l2); }
      }
    };
    // (the generic form - any Cin up to 512, gamma / beta fetched column by column - where the registers of the form above are not
    // there: the 80-register build, and 512-channel inputs)
    auto norm_any = [&]() __attribute__((always_inline)) {
      float4 v[8];
      float sum = 0.f, sa = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = (l16 + 16 * q) * 4;
        v[q] = (live && c < Cin) ? *reinterpret_cast<const float4*>(wrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
      const float rstd = 1.0f / sqrtf(var / (float)Cin + a_eps);
      if (live) {
        const bool mine = (r % nact) == sb;
        float mk = 1.f;
        if (a_has_lnmask) { const TRef LM = as_copy<TRef>(a.lnmask); mk = ld1<CM>(row(LM, id.i, id.slot, id.pos, id.t)); }
        float* hrow = row(HL, id.i, id.slot, id.pos, id.t);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = (l16 + 16 * q) * 4;
          if (c < Cin) {
            const float4 g = ldw4(a_gamma + c), bb = ldw4(a_beta + c);
            const float4 o = make_float4(((v[q].x - mean) * rstd * g.x + bb.x) * mk, ((v[q].y - mean) * rstd * g.y + bb.y) * mk,
                                         ((v[q].z - mean) * rstd * g.z + bb.z) * mk, ((v[q].w - mean) * rstd * g.w + bb.w) * mk);
            *reinterpret_cast<float4*>(wrow + c) = o;
            if (mine) st4<CM>(hrow + c, o, tb.l2);
          }
        }
        if (a_has_mask_out && mine && l16 == 0) { const TRef MO = as_copy<TRef>(a.mask_out); st1<CM>(row(MO, id.i, id.slot, id.pos, id.t), sa > 0.f ? 1.f : 0.f, tb.l2); }
      }
    };
    const int nq = Cin >> 6;       // (Cin = 64 .. 512, a power of two: rowconv_supported)
    if (WIDE && nq == 4) norm(std::integral_constant<int, 4>{});
    else norm_any();
    __syncthreads();
  }
}

// K loop + epilogue of output strip bx of the staged operator: 64 columns (one 16-column tile per wave), or - KW = 4, a single
// row tile in the step - 16 columns with the K groups split over the 4 waves (partial tiles meet in LDS behind the window)
// Warm the L2 with the first 8 KB of this wave's weight stream for strip bx (one dword per 128-byte line and lane, result
// unused) BEFORE the window gather: the stream's cold start - an L2 miss to HBM - then overlaps the gather's round trip instead
// of following it.  (Holding the fragments themselves in registers across the gather + LayerNorm spilled 200 registers.)
// The loaded word is returned and must be kept alive (mg_keep) past the point where the weights are used: the register of a
// load the compiler believes dead is reused while the load is still in flight.
__device__ __forceinline__ void mg_keep(const float t) { asm volatile("" ::"v"(t)); }
// (ten warm-up words: kept apart - summing them would wait for all ten round trips right where they were issued)
struct Warm10 { float w[10]; };
__device__ __forceinline__ void mg_keep(const Warm10& t) {
  asm volatile("" ::"v"(t.w[0]), "v"(t.w[1]), "v"(t.w[2]), "v"(t.w[3]), "v"(t.w[4]), "v"(t.w[5]), "v"(t.w[6]), "v"(t.w[7]), "v"(t.w[8]), "v"(t.w[9]));
}
template <int KW, class A>
__device__ __forceinline__ float mg_wwarm(const A& a, const int bx) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KQ = a.Cin >> 4, NG = a.ktaps * KQ;
  const int ct0 = KW > 1 ? bx : bx * 4 + wave;
  int g_lo, g_hi;
  rc_krange<KW>(NG, wave, g_lo, g_hi);
  const float* wl = a.w + (long long)ct0 * ((long long)(a.ktaps + 1) * KQ * 256) + (long long)g_lo * 256 + lane * 32;
  return ct0 * 16 < a.Cout_pad ? ldw1(wl) : 0.f;
}
// the WHOLE stream of the wave's strip (<= 80 KB: ten loads of one dword per 128-byte line) - with a group on one XCD that XCD
// streams every layer's weights through its 4 MB L2 once per step, nothing is warm from the step before
template <int KW, class A>
__device__ __forceinline__ Warm10 mg_wwarm_all(const A& a, const int bx) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KQ = a.Cin >> 4, NG = a.ktaps * KQ;
  const int ct0 = KW > 1 ? bx : bx * 4 + wave;
  int g_lo, g_hi;
  rc_krange<KW>(NG, wave, g_lo, g_hi);
  const float* wl = a.w + (long long)(ct0 * 16 < a.Cout_pad ? ct0 : 0) * ((long long)(a.ktaps + 1) * KQ * 256) + (long long)g_lo * 256 + lane * 32;
  const int ng = g_hi - g_lo;
  Warm10 r;
#pragma unroll
  for (int u = 0; u < 10; ++u) r.w[u] = ldw1(wl + (long long)(8 * u < ng ? 8 * u : 0) * 256);
  return r;
}

// the wave's first weight fragments of strip bx, requested in front of the window gather (the 128-register build): every operator's
// weights are cold - in xcd mode one XCD streams the whole model through its 4 MB L2 - and the stream's first round trip to memory
// otherwise follows the gather's instead of overlapping it
template <int KW, class A>
__device__ __forceinline__ void mg_wpre(const A& a, const int bx, float4 (&bw)[(KW > 1) ? 4 : 8]) {
  constexpr int RC_D = (KW > 1) ? 4 : 8;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KQ = a.Cin >> 4, NG = a.ktaps * KQ;
  const int ct0 = KW > 1 ? bx : bx * 4 + wave;
  int g_lo, g_hi;
  rc_krange<KW>(NG, wave, g_lo, g_hi);
  // (column tiles past the padded width: the tensor's first tile - in bounds, unused)
  const float* wl = a.w + (long long)(ct0 * 16 < a.Cout_pad ? ct0 : 0) * ((long long)(a.ktaps + 1) * KQ * 256) + lane * 4;
#pragma unroll
  for (int u = 0; u < RC_D; ++u) bw[u] = ldw4(wl + (long long)(g_lo + u) * 256);
}

template <int KW, bool PRE, int CM = 1, bool PF = false, class A>
__device__ __forceinline__ void mg_strip(const A& a, const RowTab& tb, const int bx, float* __restrict__ win, float4 (&bw)[(KW > 1) ? 4 : 8]) {
  constexpr int RC_D = (KW > 1) ? 4 : 8;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = a.Cin, LDX = Cin + 8;
  const int k = a.ktaps, d = a.dil;
  const int KQ = Cin >> 4;
  const int NG = k * KQ;
  const int ct0 = KW > 1 ? bx : bx * 4 + wave;
  int g_lo, g_hi;
  rc_krange<KW>(NG, wave, g_lo, g_hi);
  const int lr = lane & 15, lg = lane >> 4;
  const float* const abase = win + tb.tab[lr] * LDX + 4 * lg;
  const long long ct_stride = (long long)(k + 1) * KQ * 256;
  const float* wl = a.w + (long long)ct0 * ct_stride + lane * 4;
  const bool active = ct0 * 16 < a.Cout_pad;
  // The MFMAs are issued transposed (weights as the first operand - the register images are the same either way): a lane's
  // accumulator then holds 4 consecutive output columns (4 lg ..) of ONE row (lr) instead of one column of 4 rows, and the epilogue's
  // bias / residual / output accesses are 16 bytes wide.  Every output element is the same sequence of multiply-adds as before.
  const int col0 = ct0 * 16 + 4 * lg;
  const bool fin = (KW == 1 || wave == 0) && active && col0 < a.Cout && lr < tb.nvalid;      // this lane stores an output
  const RowId id = row_id(tb, lr < tb.nvalid ? lr : 0);
  // (the epilogue's arguments in one batch of scalar loads)
  float* yptr;          // (the output row's address now - two registers across the K loop instead of the tensor's ten)
  { const TRef Y = as_copy<TRef>(a.y); yptr = row(Y, id.i, id.slot, id.pos, id.t) + col0; }
  const float* const a_bias = a.bias; const float* const a_bvec = a.bvec;
  const long long a_bvs = a.bvec_stride;
  const int a_has_res = a.has_res, a_has_m1 = a.has_m1, a_has_m2 = a.has_m2, a_act = a.out_act;
  const float a_scale = a.out_scale, a_oslope = a.out_slope;
  // PF (the 128-register build): the epilogue's operands are requested in front of the K loop
  float4 pbias = make_float4(0.f, 0.f, 0.f, 0.f), pbv = pbias, pres = pbias;
  float pm = 1.f;
  // (unconditional loads - an absent operand reads the head of the weight tensor and is dropped: behind branches hipcc puts a full
  // s_waitcnt between two loads, three serialised round trips in every epilogue)
  auto fetch = [&]() __attribute__((always_inline)) {
    const float* const dummy = a.w;
    const float* pb = a_bias ? a_bias + col0 : dummy;
    const float* pv = a_bvec ? a_bvec + (long long)id.slot * a_bvs + col0 : dummy;
    const float* pr = dummy; const float* p1 = dummy; const float* p2 = dummy;
    if (a_has_res) { const TRef RS = as_copy<TRef>(a.res); pr = row(RS, id.i, id.slot, id.pos, id.t) + col0; }
    if (a_has_m1) { const TRef M1 = as_copy<TRef>(a.m1); p1 = row(M1, id.i, id.slot, id.pos, id.t); }
    if (a_has_m2) { const TRef M2 = as_copy<TRef>(a.m2); p2 = row(M2, id.i, id.slot, id.pos, id.t); }
    pbias = ldw4(pb); pbv = ldw4(pv); pres = ld4<CM>(pr);
    const float m1v = ld1<CM>(p1), m2v = ld1<CM>(p2);
    if (!a_bias) pbias = make_float4(0.f, 0.f, 0.f, 0.f);
    pm = (a_has_m1 ? m1v : 1.f) * (a_has_m2 ? m2v : 1.f);
  };
  if constexpr (PF) { if (fin) fetch(); }
  f32x4 accs[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};     // two interleaved chains (rowconv_tile)
  if (active) {
    if constexpr (!PRE) {
#pragma unroll
      for (int u = 0; u < RC_D; ++u) { bw[u] = ldw4(wl + (long long)(g_lo + u) * 256); __builtin_amdgcn_sched_barrier(0); }
    }
    const int kqm = KQ - 1, kqs = 31 - __builtin_clz(KQ);
    const int tstep = d * LDX;
    float4 af = *reinterpret_cast<const float4*>(abase + (g_lo < g_hi ? (g_lo >> kqs) * tstep + (g_lo & kqm) * 16 : 0));
    for (int G0 = g_lo; G0 < g_hi; G0 += RC_D) {
#pragma unroll
      for (int u = 0; u < RC_D; ++u) {
        const int Gn = G0 + u + 1;
        const int jn = Gn >> kqs, qn = Gn & kqm;
        const float4 afn = *reinterpret_cast<const float4*>(abase + (Gn < g_hi ? jn * tstep + qn * 16 : 0));
        f32x4& p = accs[u & 1];
        f32x4& q = accs[(u & 1) ^ 1];
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u].x, af.x, p, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u].y, af.y, q, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u].z, af.z, p, 0, 0, 0);
        q = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[u].w, af.w, q, 0, 0, 0);
        bw[u] = ldw4(wl + (long long)(G0 + u + RC_D) * 256);      // (past the last group: the zero tap, in bounds)
        af = afn;
      }
    }
  }
  f32x4 res = accs[0] + accs[1];
  if constexpr (KW > 1) {
    float* const red = win + a.wr_max * LDX;
    __syncthreads();                       // wave 0 has read the previous strip's partial tiles
    if (wave > 0) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = res;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 1; w < KW; ++w) res += *reinterpret_cast<const f32x4*>(red + ((w - 1) * 64 + lane) * 4);
    }
  }
  if (fin) {
    if constexpr (!PF) fetch();
    const float scale = a_scale;
    const int act = a_act;
    float o[4];
    const float bb[4] = {pbias.x, pbias.y, pbias.z, pbias.w}, bv[4] = {pbv.x, pbv.y, pbv.z, pbv.w}, rs[4] = {pres.x, pres.y, pres.z, pres.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = (res[e] + bb[e]) * scale;
      if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      else if (act == ACT_LRELU) v = v > 0.f ? v : v * a_oslope;
      if (a_bvec) v += bv[e];
      if (a_has_res) v += rs[e];
      if (a_has_m1 | a_has_m2) v *= pm;
      o[e] = v;
    }
    st4<CM>(yptr, make_float4(o[0], o[1], o[2], o[3]), tb.l2);
  }
}

// MOP_FFN (the aligner's feed-forward, prosody_util.py:139-158, and the decoder's conv blocks [LN -> k5 conv -> GELU] ->
// [1x1 conv + residual], conv.py:127-264): after mg_stage (LayerNorm prologue) member `sb` computes ITS HC = Cout / GS hidden
// columns of the first conv (k taps; activation applied) into LDS and multiplies them straight away with its K range of the
// second, 1x1 conv: the hidden tensor never leaves the CU and the second conv's K loop is split over the group.
// The member's partial sums [rows][Cout2] go to part[sb]; bias, residual and the norm behind them are applied where the sum is
// consumed (RowConvArgs / LNArgs: xp ..).  LDS: window [wr_max][Cin + 8] | hidden [16][HC + 8].
template <bool WIDE = false, int CM = 1, class A>
__device__ __forceinline__ void mg_ffn(const A& a, const RowTab& tb, const int sb, const int GS, float* __restrict__ win) {
  constexpr int RC_D = 8;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int Cin = a.Cin, LDX = Cin + 8;
  const int HC = a.Cout / GS, LDH = HC + 8;           // hidden columns of this member (a multiple of 64)
  const bool overlay = a.hid_overlay != 0;            // one strip per member: the hidden tile takes the window's place once it is read out
  float* const hid = overlay ? win : win + a.wr_max * LDX;
  {   // ---- first conv (k taps, dilation d): strips sb * HC/64 .. of 64 columns, one 16-column tile per wave, into LDS
    const int k = a.ktaps;
    const int KQ = Cin >> 4, NG = k * KQ;             // NG % RC_D == 0 (host)
    const int kqm = KQ - 1, kqs = 31 - __builtin_clz(KQ);
    const int tstep = a.dil * LDX;
    const long long ct_stride = (long long)(k + 1) * KQ * 256;       // k taps + the zero tap
    const float* const abase = win + tb.tab[lr] * LDX + 4 * lg;
    for (int s = 0; s < HC / 64; ++s) {
      const int ct0 = (sb * (HC / 64) + s) * 4 + wave;
      const float* wl = a.w + (long long)ct0 * ct_stride + lane * 4;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      float4 bw[RC_D];
#pragma unroll
      for (int u = 0; u < RC_D; ++u) { bw[u] = ldw4(wl + (long long)u * 256); __builtin_amdgcn_sched_barrier(0); }
      float4 af = *reinterpret_cast<const float4*>(abase);
      for (int G0 = 0; G0 < NG; G0 += RC_D) {
#pragma unroll
        for (int u = 0; u < RC_D; ++u) {
          const int Gn = G0 + u + 1;
          const float4 afn = *reinterpret_cast<const float4*>(abase + (Gn < NG ? (Gn >> kqs) * tstep + (Gn & kqm) * 16 : 0));
          f32x4& p = (u & 1) ? acc1 : acc0;
          f32x4& q = (u & 1) ? acc0 : acc1;
          p = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bw[u].x, p, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bw[u].y, q, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bw[u].z, p, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bw[u].w, q, 0, 0, 0);
          bw[u] = ldw4(wl + (long long)(G0 + u + RC_D) * 256);      // (past the last group: the zero tap, in bounds)
          af = afn;
        }
      }
      const int col = ct0 * 16 + lr;
      const float bias = a.bias ? ldw1(a.bias + col) : 0.f;
      if (overlay) __syncthreads();                   // every wave has read its last window fragment
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = ((acc0[e] + acc1[e]) + bias) * a.out_scale;
        if (a.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (a.out_act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        else if (a.out_act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
        hid[(4 * lg + e) * LDH + (s * 4 + wave) * 16 + lr] = v;
      }
    }
  }
  const int KQ2 = a.Cout >> 4, KQm = HC >> 4;         // second conv: K groups per tap in memory / of this member (4, or a multiple of RC_D)
  const long long ct_stride2 = 2ll * KQ2 * 256;
  float4 bwn[4];                                        // KQm == 4: the weights of the wave's next column tile (the first one: fetched across the barrier)
  if (KQm < RC_D) {
    const float* wn = a.w2 + (long long)wave * ct_stride2 + (long long)(sb * KQm) * 256 + lane * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) bwn[u] = ldw4(wn + (long long)u * 256);
  }
  __syncthreads();
  {   // ---- second conv, this member's K range: hidden channels [sb * HC, (sb + 1) * HC) of every 16-column output tile
    const float* const abase = hid + lr * LDH + 4 * lg;
    float* const pbase = a.part + (long long)sb * a.part_stride;
    for (int ct = wave; ct * 16 < a.Cout2_pad; ct += 4) {
      const float* wl = a.w2 + (long long)ct * ct_stride2 + (long long)(sb * KQm) * 256 + lane * 4;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (KQm >= RC_D) {
        float4 bw[RC_D];
#pragma unroll
        for (int u = 0; u < RC_D; ++u) { bw[u] = ldw4(wl + (long long)u * 256); __builtin_amdgcn_sched_barrier(0); }
        float4 af = *reinterpret_cast<const float4*>(abase);
        for (int G0 = 0; G0 < KQm; G0 += RC_D) {
#pragma unroll
          for (int u = 0; u < RC_D; ++u) {
            const int Gn = G0 + u + 1;
            const float4 afn = *reinterpret_cast<const float4*>(abase + (Gn < KQm ? Gn * 16 : 0));
            f32x4& p = (u & 1) ? acc1 : acc0;
            f32x4& q = (u & 1) ? acc0 : acc1;
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bw[u].x, p, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bw[u].y, q, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bw[u].z, p, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bw[u].w, q, 0, 0, 0);
            // (past this member's range: the next member's groups / the zero tap - in bounds, never used)
            bw[u] = ldw4(wl + (long long)(G0 + u + RC_D) * 256);
            af = afn;
          }
        }
      } else {      // four K groups per member (a 64-column hidden strip): the next column tile's weights are in flight while this one is multiplied
        float4 bw[4], af[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { bw[u] = bwn[u]; af[u] = *reinterpret_cast<const float4*>(abase + u * 16); }
        if ((ct + 4) * 16 < a.Cout2_pad) {
          const float* wn = a.w2 + (long long)(ct + 4) * ct_stride2 + (long long)(sb * KQm) * 256 + lane * 4;
#pragma unroll
          for (int u = 0; u < 4; ++u) bwn[u] = ldw4(wn + (long long)u * 256);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f32x4& p = (u & 1) ? acc1 : acc0;
          f32x4& q = (u & 1) ? acc0 : acc1;
          p = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].x, bw[u].x, p, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].y, bw[u].y, q, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].z, bw[u].z, p, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u].w, bw[u].w, q, 0, 0, 0);
        }
      }
      const int col = ct * 16 + lr;
      if (col < a.Cout2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * lg + e;
          if (r >= tb.nvalid) continue;
          const RowId id = row_id(tb, r);
          st1<CM>(pbase + (long long)(id.i * a.T + id.t) * a.Cout2 + col, acc0[e] + acc1[e], tb.l2);
        }
      }
    }
  }
}

// one 64-column strip of a 1x1 layer wider than the window (aligner ff2): the rows' channels pass through LDS in chunks of 512
template <int CM = 1, class A>
__device__ __forceinline__ void mg_rowlin_strip(const A& a, const RowTab& tb, const int bx, float* __restrict__ win) {   // win: [16][RL_LDX]
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = a.Cin;
  const int KQ = Cin >> 4;
  const int ct0 = bx * 4 + wave;
  const int lr = lane & 15, lg = lane >> 4;
  const float* const abase = win + lr * RL_LDX + 4 * lg;
  const long long ct_stride = 2ll * KQ * 256;
  const float* wl = a.w + (long long)ct0 * ct_stride + lane * 4;
  const bool active = ct0 * 16 < a.Cout_pad;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float4 bw[RL_D];
  if (active) {
#pragma unroll
    for (int u = 0; u < RL_D; ++u) { bw[u] = ldw4(wl + (long long)u * 256); __builtin_amdgcn_sched_barrier(0); }
  }
  const int gw = tid >> 7, gc4 = tid & 127;
  for (int c0 = 0; c0 < Cin; c0 += RL_CW) {
    __syncthreads();                                              // every wave is done with the previous chunk / operator
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w = gw + 2 * (4 * h + u);
        const RowId id = row_id(tb, w < tb.nvalid ? w : 0);
        v[u] = ld4<CM>(row(a.x, id.i, id.slot, id.pos, id.t) + gc4 * 4 + c0);
      }
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
          bw[u] = ldw4(wl + (long long)(g0 + G0 + u + RL_D) * 256);
          af = afn;
        }
      }
    }
  }
  const int col = ct0 * 16 + lr;
  if (active && col < a.Cout) {
    const float bias = a.bias ? ldw1(a.bias + col) : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * lg + e;
      if (r >= tb.nvalid) continue;
      const RowId id = row_id(tb, r);
      float v = ((acc0[e] + acc1[e]) + bias) * a.out_scale;
      if (a.out_act == ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (a.out_act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      else if (a.out_act == ACT_LRELU) v = v > 0.f ? v : v * a.out_slope;
      if (a.bvec) v += ldw1(a.bvec + (long long)id.slot * a.bvec_stride + col);
      if (a.has_res) v += ld1<CM>(row(a.res, id.i, id.slot, id.pos, id.t) + col);
      if (a.has_m1) v *= ld1<CM>(row(a.m1, id.i, id.slot, id.pos, id.t));
      if (a.has_m2) v *= ld1<CM>(row(a.m2, id.i, id.slot, id.pos, id.t));
      st1<CM>(row(a.y, id.i, id.slot, id.pos, id.t) + col, v, tb.l2);
    }
  }
}

// LayerNorm of tile row r (one wave)
// C = 256 or 512: a lane holds 4 consecutive channels per 256 - every access 16 bytes wide, every load unconditional and in flight
// before the first is consumed (partial tensors: eight at a time, summed in member order)
template <int NV, int CM, class A>
__device__ __forceinline__ void mg_layernorm_row_v(const A& a, const RowTab& tb, const int r) {
  const int lane = threadIdx.x & 63;
  const RowId id = row_id(tb, r);
  const int C = NV * 256;
  const int xparts = a.xparts, has_pre = a.has_pre, has_post = a.has_post, has_m1 = a.has_m1, has_m2 = a.has_m2, has_mask_out = a.has_mask_out, has_xres = a.has_xres;
  const float* const gamma = a.gamma; const float* const beta = a.beta; const float* const xbias = a.xbias;
  const float eps = a.eps;
  const TRef Y = as_copy<TRef>(a.y);
  float4 g[NV], bb[NV], v[NV], pre[NV], post[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) { g[m] = ldw4(gamma + lane * 4 + 256 * m); bb[m] = ldw4(beta + lane * 4 + 256 * m); }
  const float* dummy = gamma;
  const float* ppre = dummy; const float* ppost = dummy; const float* pm1 = dummy; const float* pm2 = dummy;
  if (has_pre) { const TRef R = as_copy<TRef>(a.pre); ppre = row(R, id.i, id.slot, id.pos, id.t); }
  if (has_post) { const TRef R = as_copy<TRef>(a.post); ppost = row(R, id.i, id.slot, id.pos, id.t); }
  if (has_m1) { const TRef R = as_copy<TRef>(a.m1); pm1 = row(R, id.i, id.slot, id.pos, id.t); }
  if (has_m2) { const TRef R = as_copy<TRef>(a.m2); pm2 = row(R, id.i, id.slot, id.pos, id.t); }
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    pre[m] = ld4<CM>(has_pre ? ppre + lane * 4 + 256 * m : dummy);
    post[m] = ld4<CM>(has_post ? ppost + lane * 4 + 256 * m : dummy);
  }
  const float m1v = ld1<CM>(pm1), m2v = ld1<CM>(pm2);
  if (xparts > 0) {       // x = the sum of the group members' partial tensors (+ bias, + residual), in member order
    const float* const xp = a.xp; const long long xps = a.xp_stride;
    const long long off = (long long)(id.i * a.T + id.t) * a.xp_ld + lane * 4;
    const float* pres = dummy;
    if (has_xres) { const TRef R = as_copy<TRef>(a.xres); pres = row(R, id.i, id.slot, id.pos, id.t); }
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const float4 xr = ld4<CM>(has_xres ? pres + lane * 4 + 256 * m : dummy);
      const float4 xb = ldw4(xbias ? xbias + lane * 4 + 256 * m : dummy);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p0 = 0; p0 < xparts; p0 += 8) {
        float4 q8[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) q8[p] = ld4<CM>(xp + (long long)(p0 + p < xparts ? p0 + p : 0) * xps + off + 256 * m);
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p0 + p < xparts) {
          if (p0 + p == 0) acc = q8[p];
          else { acc.x += q8[p].x; acc.y += q8[p].y; acc.z += q8[p].z; acc.w += q8[p].w; }
        }
      }
      if (xbias) { acc.x += xb.x; acc.y += xb.y; acc.z += xb.z; acc.w += xb.w; }
      if (has_xres) { acc.x += xr.x; acc.y += xr.y; acc.z += xr.z; acc.w += xr.w; }
      v[m] = acc;
    }
  } else {
    const TRef X = as_copy<TRef>(a.x);
    const float* x = row(X, id.i, id.slot, id.pos, id.t);
#pragma unroll
    for (int m = 0; m < NV; ++m) v[m] = ld4<CM>(x + lane * 4 + 256 * m);
  }
  float s = 0.f, sa = 0.f;
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    sa += (fabsf(v[m].x) + fabsf(v[m].y)) + (fabsf(v[m].z) + fabsf(v[m].w));
    if (has_pre) { v[m].x += pre[m].x; v[m].y += pre[m].y; v[m].z += pre[m].z; v[m].w += pre[m].w; }
    s += (v[m].x + v[m].y) + (v[m].z + v[m].w);
  }
  s = wave_sum(s);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int m = 0; m < NV; ++m) { const float d0 = v[m].x - mean, d1 = v[m].y - mean, d2 = v[m].z - mean, d3 = v[m].w - mean; q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3); }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  const float mk = (has_m1 ? m1v : 1.f) * (has_m2 ? m2v : 1.f);
  if (has_mask_out) {
    sa = wave_sum(sa);
    if (lane == 0) { const TRef MO = as_copy<TRef>(a.mask_out); st1<CM>(row(MO, id.i, id.slot, id.pos, id.t), sa > 0.f ? 1.f : 0.f, tb.l2); }
  }
  float* y = row(Y, id.i, id.slot, id.pos, id.t);
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    float4 o = make_float4((v[m].x - mean) * rstd * g[m].x + bb[m].x, (v[m].y - mean) * rstd * g[m].y + bb[m].y,
                           (v[m].z - mean) * rstd * g[m].z + bb[m].z, (v[m].w - mean) * rstd * g[m].w + bb[m].w);
    if (has_m1 | has_m2) { o.x *= mk; o.y *= mk; o.z *= mk; o.w *= mk; }
    if (has_post) { o.x += post[m].x; o.y += post[m].y; o.z += post[m].z; o.w += post[m].w; }
    st4<CM>(y + lane * 4 + 256 * m, o, tb.l2);
  }
}

template <int CM = 1, bool WIDE = false, class A>
__device__ __forceinline__ void mg_layernorm_row(const A& a, const RowTab& tb, const int r) {
  if (r >= tb.nvalid) return;
  if constexpr (WIDE) { if (a.C == 256) { mg_layernorm_row_v<1, CM>(a, tb, r); return; } }      // (the 128-register build)
  const int lane = threadIdx.x & 63;
  const RowId id = row_id(tb, r);
  const float* x = row(a.x, id.i, id.slot, id.pos, id.t);
  const float* pre = a.has_pre ? row(a.pre, id.i, id.slot, id.pos, id.t) : nullptr;
  float v[LN_MAXV];
  float s = 0.f, sa = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int c = lane + 64 * k;
    float u = 0.f;
    if (c < a.C) {
      if (a.xparts > 0) {       // x = the sum of the group members' partial tensors (+ bias, + residual), in member order
        const long long off = (long long)(id.i * a.T + id.t) * a.xp_ld + c;
        float pv[8];
        for (int p0 = 0; p0 < a.xparts; p0 += 8) {
#pragma unroll
          for (int p = 0; p < 8; ++p) pv[p] = p0 + p < a.xparts ? ld1<CM>(a.xp + (p0 + p) * a.xp_stride + off) : 0.f;
#pragma unroll
          for (int p = 0; p < 8; ++p) if (p0 + p < a.xparts) u = p0 + p == 0 ? pv[p] : u + pv[p];
        }
        if (a.xbias) u += ldw1(a.xbias + c);
        if (a.has_xres) u += ld1<CM>(row(a.xres, id.i, id.slot, id.pos, id.t) + c);
      } else u = ld1<CM>(x + c);
      sa += fabsf(u); if (pre) u += ld1<CM>(pre + c);
    }
    v[k] = u; s += u;
  }
  s = wave_sum(s);
  const float mean = s / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) { int c = lane + 64 * k; if (c < a.C) { float d = v[k] - mean; q += d * d; } }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
  float mk = 1.f;
  if (a.has_m1) mk *= ld1<CM>(row(a.m1, id.i, id.slot, id.pos, id.t));
  if (a.has_m2) mk *= ld1<CM>(row(a.m2, id.i, id.slot, id.pos, id.t));
  if (a.has_mask_out) {
    sa = wave_sum(sa);
    if (lane == 0) st1<CM>(row(a.mask_out, id.i, id.slot, id.pos, id.t), sa > 0.f ? 1.f : 0.f, tb.l2);
  }
  float* y = row(a.y, id.i, id.slot, id.pos, id.t);
  const float* post = a.has_post ? row(a.post, id.i, id.slot, id.pos, id.t) : nullptr;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int c = lane + 64 * k;
    if (c < a.C) {
      float o = (v[k] - mean) * rstd * ldw1(a.gamma + c) + ldw1(a.beta + c);
      if (a.has_m1 | a.has_m2) o *= mk;
      if (post) o += ld1<CM>(post + c);
      st1<CM>(y + c, o, tb.l2);
    }
  }
}

// embedding row r of the tile (one wave)
template <int CM = 1, class A>
__device__ __forceinline__ void mg_embed_row(const A& a, const RowTab& tb, const int r) {
  if (r >= tb.nvalid) return;
  const int lane = threadIdx.x & 63;
  const RowId id = row_id(tb, r);
  int idx = ldi(a.idx + id.i * a.T + id.t);
  idx = idx < 0 ? 0 : (idx >= a.vocab ? a.vocab - 1 : idx);
  float* y = row(a.y, id.i, id.slot, id.pos, id.t);
  const float* e = a.table + (long long)idx * a.C;
  for (int c = lane; c < a.C; c += 64) st1<CM>(y + c, ldw1(e + c), tb.l2);
}

// cross attention of tile rows r0, r0 + 1 (two heads: waves 0-1 take row r0, waves 2-3 row r0 + 1); LDS: 2 x [sq 1024 | sp 2 x 512]
// The K / V rows of a slot are read-only for the launch: their loads are batched 8 deep (the serial form - one load, one
// multiply-add - made this operator 26 us of L2 latency for 38 keys).
constexpr int XA2_LDS_FLOATS = 1024 + 2 * XA_MAX_S;
template <int CM = 1, bool WIDE = false, class A>
__device__ __forceinline__ void mg_xattn_rows(const A& a, const RowTab& tb, const int r0, float* __restrict__ lds) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = wv >> 1, h = wv & 1, ht = threadIdx.x & 127;
  const int r = r0 + half;
  const bool live = r < tb.nvalid;
  float* const sq = lds + half * XA2_LDS_FLOATS;
  float (*sp)[XA_MAX_S] = reinterpret_cast<float (*)[XA_MAX_S]>(sq + 1024);
  const RowId id = row_id(tb, live ? r : 0);
  const int S = ldi(a.slen + id.slot);
  const int dh = a.E / a.H;
  if (live) { const float* q = row(a.q, id.i, id.slot, id.pos, id.t); for (int c = ht; c < a.E; c += 128) sq[c] = ld1<CM>(q + c); }
  __syncthreads();
  const float* kv = a.kv + (long long)id.slot * a.kv_slot_stride;
  const float* km = a.kmask + (long long)id.slot * a.S_max;
  if (live && h < a.H) {
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) {
      const float* kp = kv + (long long)s * 2 * a.E + h * dh;
      const float4* qp = reinterpret_cast<const float4*>(sq + h * dh);
      float sc = 0.f;
      constexpr int KB = 8;      // 16-byte key loads per round trip (16 in the 128-register build: measured the same, one more spilled register)
      for (int d0 = 0; d0 < dh / 4; d0 += KB) {
        float4 k4[KB];
        // (unconditional: past the head's last column the first one once more, dropped below - a load inside a branch waits for every load before it)
#pragma unroll
        for (int u = 0; u < KB; ++u) k4[u] = ldw4(kp + 4 * (d0 + u < dh / 4 ? d0 + u : 0));
#pragma unroll
        for (int u = 0; u < KB; ++u) if (d0 + u < dh / 4) { const float4 q4 = qp[d0 + u]; sc += q4.x * k4[u].x + q4.y * k4[u].y + q4.z * k4[u].z + q4.w * k4[u].w; }
      }
      sc += ldw1(km + s);
      sp[h][s] = sc;
      mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) { float p = expf(sp[h][s] - mx); sp[h][s] = p; sum += p; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int s = lane; s < S; s += 64) sp[h][s] *= inv;
  }
  __syncthreads();
  if (live && h < a.H) {
    float* o = row(a.out, id.i, id.slot, id.pos, id.t);
    // (a lane's two output columns d, d + 64 together and 16 keys per round trip: the loop used to be one column at a time in batches
    // of 8 - 2 x S / 8 = 38 dependent L2 round trips for the 151 keys of a 3 s reference, 17 us of the operator's 18.  Every
    // column's additions are the same, in the same order.)
    for (int d = lane; d < dh; d += 128) {
      const bool two = d + 64 < dh;
      float acc0 = 0.f, acc1 = 0.f;
      const float* vp0 = kv + a.E + h * dh + d;
      const float* vp1 = two ? vp0 + 64 : vp0;
      for (int s0 = 0; s0 < S; s0 += 16) {
        float v0[16], v1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const long long off = (long long)(s0 + u < S ? s0 + u : 0) * 2 * a.E;      // (unconditional, see above)
          v0[u] = ldw1(vp0 + off); v1[u] = ldw1(vp1 + off);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) if (s0 + u < S) { const float pw = sp[h][s0 + u]; acc0 += pw * v0[u]; acc1 += pw * v1[u]; }
      }
      st1<CM>(o + h * dh + d, acc0, tb.l2);
      if (two) st1<CM>(o + h * dh + d + 64, acc1, tb.l2);
    }
  }
}

// uv / f0 head of tile row r (one wave)
template <int CM = 1, class A>
__device__ __forceinline__ void mg_pitch_row(const A& a, const RowTab& tb, const float mel_min, const float mel_den, const int r) {
  if (r >= tb.nvalid) return;
  const int lane = threadIdx.x & 63;
  const RowId id = row_id(tb, r);
  const int m = id.i * a.T + id.t;
  const float* x = row(a.h, id.i, id.slot, id.pos, id.t);
  float v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { int c = lane + 64 * k; v[k] = c < a.Cp ? ld1<CM>(x + c) : 0.f; s += v[k]; }
  s = wave_sum(s);
  const float mean = s / (float)a.Cp;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { int c = lane + 64 * k; if (c < a.Cp) { float d = v[k] - mean; q += d * d; } }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)a.Cp + 1e-5f);
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = lane + 64 * k;
    if (c < a.Cp) { float y = (v[k] - mean) * rstd * ldw1(a.gamma + c) + ldw1(a.beta + c); d0 += y * ldw1(a.w + c); d1 += y * ldw1(a.w + a.Cp + c); }
  }
  d0 = wave_sum(d0) + ldw1(a.b);
  d1 = wave_sum(d1) + ldw1(a.b + 1);
  const int code = ldi(a.codes + m);
  const bool uv = (d0 > 0.f) || (code == a.silent_token);
  float f0 = exp2f(d1);
  f0 = fminf(fmaxf(f0, 50.f), 900.f);
  if (uv) f0 = 0.f;
  float fm = 1127.f * logf(1.f + f0 / 700.f);
  if (fm > 0.f) fm = (fm - mel_min) * 254.f / mel_den + 1.f;
  if (fm <= 1.f) fm = 1.f;
  if (fm > 255.f) fm = 255.f;
  const int bin = (int)(fm + 0.5f);
  const float* pi = row(a.pitch_inp, id.i, id.slot, id.pos, id.t);
  float* di = row(a.dec_inp, id.i, id.slot, id.pos, id.t);
  const float* pe = a.pitch_embed + (long long)bin * a.E;
  for (int c = lane; c < a.E; c += 64) st1<CM>(di + c, ld1<CM>(pi + c) + ldw1(pe + c), tb.l2);
}

}  // namespace ro
}  // namespace cnk
